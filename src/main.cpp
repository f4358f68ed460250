// src/main.cpp -- `simulation.out`: the reference's process contract on top of libfluidsim.so.
//
// `make && ./simulation.out` with no arguments does what the reference's main() does
// (simulation.cpp:429-451): a 128x64x64 tunnel, 100 steps, inlet speed 30, the constructor
// defaults of simulation.h:60-64, one STL obstacle (scale 2, rot_x 90 deg, translate -16,0,0),
// then Simulation::run(), which appends every frame to ./data/{data,obs,v_x,v_y,v_z}.bin so
// that GUI/main.py, gui.py and make_pngs.py of the reference read the output unchanged.
//
// The reference hard-codes an absolute STL path from its author's machine; when that file
// does not exist the reference prints an error and simulates an empty tunnel
// (object_loader.cpp:282-285).  Same here; FS_STL or --stl names another mesh.
//
// The reference needs a recompile for every change of configuration.  Optional overrides
// (all default to the reference's values):
//   --grid WxHxD  --steps N  --acc N  --speed N  --dt F  --diff F
//   --stl PATH[,scale,rot_x,rot_y,rot_z,tx,ty,tz]   (repeatable; "none" = no obstacle)
//   --dump-every N (0 = never, -1 = last frame only)  --dump-dir DIR
//   --precision fp32|fp64   --solver jacobi|gs_lex|rbsor|mg   --omega W (rbsor)   --mg-cycles N (mg)   --seed N   --quiet
//   --resume DIR   start from the last frame of DIR/{data,obs,v_x,v_y,v_z}.bin (a dumped frame is
//                  a complete state: everything else is rebuilt every step)
//   --json         append one machine-readable timing line to stdout
// Each flag can also be given as an environment variable FS_GRID, FS_STEPS, ...
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/fluidsim.h"

namespace {

struct Stl {
    std::string path;
    float v[7] = { 2.0f, 90.0f, 0.0f, 0.0f, -16.0f, 0.0f, 0.0f };   // simulation.cpp:442-446
};

Stl parse_stl(const std::string& spec)
{
    Stl s;
    size_t pos = spec.find(',');
    s.path = spec.substr(0, pos);
    int k = 0;
    while (pos != std::string::npos && k < 7) {
        size_t next = spec.find(',', pos + 1);
        s.v[k++] = (float)atof(spec.substr(pos + 1, next == std::string::npos ? std::string::npos : next - pos - 1).c_str());
        pos = next;
    }
    return s;
}

const char* opt(int argc, char** argv, int& i, const char* flag, const char* env)
{
    (void)env;
    if (strcmp(argv[i], flag) == 0 && i + 1 < argc) return argv[++i];
    return nullptr;
}

// last frame of one dump file -> field `which` (simulation.cpp:143-147 layout)
int resume_field(fs_sim* sim, const std::string& dir, const char* name, int which)
{
    const size_t n = fs_padded_size(sim);
    std::string path = dir + "/" + name;
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) { fprintf(stderr, "simulation.out: cannot open %s\n", path.c_str()); return 1; }
    fseek(fp, 0, SEEK_END);
    const long bytes = ftell(fp);
    if (bytes <= 0 || (size_t)bytes % (n * sizeof(float)) != 0) {
        fprintf(stderr, "simulation.out: %s is not a whole number of %zu-cell frames\n", path.c_str(), n);
        fclose(fp);
        return 1;
    }
    std::vector<float> frame(n);
    fseek(fp, bytes - (long)(n * sizeof(float)), SEEK_SET);
    const bool ok = fread(frame.data(), sizeof(float), n, fp) == n;
    fclose(fp);
    if (!ok) { fprintf(stderr, "simulation.out: short read from %s\n", path.c_str()); return 1; }
    if (fs_set_field(sim, which, frame.data(), n, 4)) { fprintf(stderr, "simulation.out: %s\n", fs_last_error()); return 1; }
    return 0;
}

int die(const char* what)
{
    fprintf(stderr, "simulation.out: %s: %s\n", what, fs_last_error());
    return 1;
}

}  // namespace

int main(int argc, char** argv)
{
    // simulation.cpp:431-436
    int scale = 1;
    int width = 128 * scale, height = 64 * scale, depth = 64 * scale;
    int iter = 100, speed = FS_DEFAULT_SPEED, acc = FS_DEFAULT_ACC;
    float dt = FS_DEFAULT_DT, diff = FS_DEFAULT_DIFF, visc = FS_DEFAULT_VISC;
    std::vector<Stl> stls;
    bool stl_given = false, json = false;
    std::string resume_dir;
    std::vector<std::pair<std::string, std::string>> options;

    auto apply = [&](const std::string& key, const char* val) -> bool {
        if (key == "grid") return sscanf(val, "%dx%dx%d", &width, &height, &depth) == 3;
        if (key == "steps") { iter = atoi(val); return true; }
        if (key == "acc") { acc = atoi(val); return true; }
        if (key == "speed") { speed = atoi(val); return true; }
        if (key == "dt") { dt = (float)atof(val); return true; }
        if (key == "diff") { diff = (float)atof(val); return true; }
        if (key == "stl") { stl_given = true; if (strcmp(val, "none") != 0) stls.push_back(parse_stl(val)); return true; }
        if (key == "dump-every") { options.push_back({ "dump_every", val }); return true; }
        if (key == "dump-dir") { options.push_back({ "dump_dir", val }); return true; }
        if (key == "precision") { options.push_back({ "precision", val }); return true; }
        if (key == "solver") { options.push_back({ "solver", val }); return true; }
        if (key == "omega") { options.push_back({ "sor_omega", val }); return true; }
        if (key == "mg-cycles") { options.push_back({ "mg_cycles", val }); return true; }
        if (key == "seed") { options.push_back({ "voxel_seed", val }); return true; }
        if (key == "resume") { resume_dir = val; return true; }
        return false;
    };
    static const char* const keys[] = { "grid", "steps", "acc", "speed", "dt", "diff", "stl", "dump-every", "dump-dir",
                                        "precision", "solver", "omega", "mg-cycles", "seed", "resume" };
    for (const char* k : keys) {
        std::string env = "FS_";
        for (const char* p = k; *p; ++p) env += (*p == '-') ? '_' : (char)toupper(*p);
        if (const char* v = getenv(env.c_str()))
            if (!apply(k, v)) { fprintf(stderr, "simulation.out: bad value for %s\n", env.c_str()); return 2; }
    }
    if (getenv("FS_QUIET")) options.push_back({ "quiet", "1" });
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--quiet") { options.push_back({ "quiet", "1" }); continue; }
        if (a == "--json") { json = true; continue; }
        if (a.rfind("--", 0) != 0 || i + 1 >= argc || !apply(a.substr(2), argv[i + 1])) {
            fprintf(stderr, "simulation.out: unknown or malformed argument '%s' (see src/main.cpp)\n", argv[i]);
            return 2;
        }
        ++i;
    }
    if (!stl_given) {
        Stl s;
        s.path = "/media/raoul/Speed/Data/3D-Printing/Models/Cars/F1Car-basic.stl";   // simulation.cpp:441
        stls.push_back(s);
    }

    fs_sim* sim = fs_create(width, height, depth, iter, speed, dt, diff, visc, acc);   // simulation.cpp:438
    if (!sim) return die("fs_create");
    for (auto& kv : options)
        if (fs_set_option(sim, kv.first.c_str(), kv.second.c_str())) return die(kv.first.c_str());
    for (const Stl& s : stls) {
        int rc = fs_load_stl(sim, s.path.c_str(), s.v[0], s.v[1], s.v[2], s.v[3], s.v[4], s.v[5], s.v[6], nullptr);
        if (rc != FS_OK && rc != FS_EIO) return die("fs_load_stl");   // unreadable STL: carry on with an empty tunnel
    }
    if (!resume_dir.empty()) {
        static const char* const names[5] = { "data.bin", "obs.bin", "v_x.bin", "v_y.bin", "v_z.bin" };
        static const int which[5] = { FS_DENS, FS_OBS, FS_VX, FS_VY, FS_VZ };
        for (int k = 0; k < 5; ++k)
            if (resume_field(sim, resume_dir, names[k], which[k])) return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (fs_run(sim)) return die("fs_run");                              // simulation.cpp:448
    if (fs_sync(sim)) return die("fs_sync");
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (json)
        printf("{\"grid\": [%d, %d, %d], \"steps\": %d, \"acc\": %d, \"seconds\": %.6f, \"cells_steps_per_sec\": %.6g}\n",
               width, height, depth, iter, acc, secs, (double)width * height * depth * iter / secs);
    fs_destroy(sim);
    return 0;
}
