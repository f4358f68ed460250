// dump.h -- asynchronous frame writer for the reference's dump contract
// (simulation.cpp:56-60, 140-148: five raw float32 arrays appended per step).
//
// The reference writes 20 bytes per cell per step from inside step() and stalls on it
// (2.7 GB per step at 512^3).  Here a frame is packed on the device into one of
// two staging buffers, copied to pinned host memory on a separate stream, and written by a
// host thread, so the next step's kernels run while the previous frame is still on its way
// to disk.  The bytes and their order in the files are unchanged.
#pragma once
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdio>
#include <deque>
#include <mutex>
#include <string>
#include <thread>

namespace fs {

struct FrameWriter {
    static constexpr int NSLOT = 2, NFILE = 5;
    struct Job {
        int slot;
        long nloc;        // floats per field in this frame
        long offset;      // byte offset of this rank's planes in each file, or -1 to append
    };

    int device = 0;
    hipStream_t copy_stream = nullptr;
    float* dev[NSLOT] = {nullptr, nullptr};
    float* host[NSLOT] = {nullptr, nullptr};
    hipEvent_t packed[NSLOT] = {nullptr, nullptr}, copied[NSLOT] = {nullptr, nullptr};
    long capacity = 0;    // floats per slot
    bool busy[NSLOT] = {false, false};
    FILE** fp = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> queue;
    bool stop = false, started = false;
    std::string error;

    // floats_per_slot = 5 fields x the planes this rank writes
    int init(int dev_id, long floats_per_slot, FILE** files, std::string* err)
    {
        if (started && floats_per_slot <= capacity) { fp = files; return 0; }
        shutdown();
        device = dev_id;
        fp = files;
        capacity = floats_per_slot;
        if (hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess) { *err = "hipStreamCreate"; return -1; }
        for (int k = 0; k < NSLOT; ++k) {
            if (hipMalloc((void**)&dev[k], capacity * sizeof(float)) != hipSuccess ||
                hipHostMalloc((void**)&host[k], capacity * sizeof(float), hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&packed[k], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&copied[k], hipEventDisableTiming) != hipSuccess) {
                *err = "allocating dump staging buffers";
                return -1;
            }
            busy[k] = false;
        }
        stop = false;
        error.clear();
        th = std::thread([this] { run(); });
        started = true;
        return 0;
    }

    // blocks until staging slot `k` is free again; returns -1 if the writer failed earlier
    int acquire(int k, std::string* err)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !busy[k] || !error.empty(); });
        if (!error.empty()) { *err = error; return -1; }
        return 0;
    }

    // the caller has queued the pack kernels of slot k on `compute`; ship the frame
    int submit(int k, hipStream_t compute, long nloc, long offset, std::string* err)
    {
        if (hipEventRecord(packed[k], compute) != hipSuccess || hipStreamWaitEvent(copy_stream, packed[k], 0) != hipSuccess ||
            hipMemcpyAsync(host[k], dev[k], (size_t)nloc * NFILE * sizeof(float), hipMemcpyDeviceToHost, copy_stream) != hipSuccess ||
            hipEventRecord(copied[k], copy_stream) != hipSuccess) {
            *err = "queueing the device-to-host copy of a frame";
            return -1;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            busy[k] = true;
            queue.push_back(Job{ k, nloc, offset });
        }
        cv.notify_all();
        return 0;
    }

    int flush(std::string* err)
    {
        if (!started) return 0;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return (queue.empty() && !busy[0] && !busy[1]) || !error.empty(); });
        if (!error.empty()) { *err = error; return -1; }
        for (int k = 0; k < NFILE; ++k)
            if (fp && fp[k]) fflush(fp[k]);
        return 0;
    }

    void shutdown()
    {
        if (started) {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            if (th.joinable()) th.join();
            started = false;
        }
        for (int k = 0; k < NSLOT; ++k) {
            if (dev[k]) hipFree(dev[k]);
            if (host[k]) hipHostFree(host[k]);
            if (packed[k]) hipEventDestroy(packed[k]);
            if (copied[k]) hipEventDestroy(copied[k]);
            dev[k] = nullptr; host[k] = nullptr; packed[k] = nullptr; copied[k] = nullptr;
        }
        if (copy_stream) hipStreamDestroy(copy_stream);
        copy_stream = nullptr;
        queue.clear();
    }

    void run()
    {
        hipSetDevice(device);
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !queue.empty(); });
                if (queue.empty()) return;        // stop requested and nothing left to write
                job = queue.front();
                queue.pop_front();
            }
            std::string fail;
            if (hipEventSynchronize(copied[job.slot]) != hipSuccess) fail = "frame copy failed";
            if (fail.empty()) {
                // the five files are independent: one short-lived thread each (a single fwrite
                // stream into the page cache tops out near 6 GB/s, far below the PCIe copy)
                const char* errs[NFILE] = {nullptr, nullptr, nullptr, nullptr, nullptr};
                std::thread workers[NFILE];
                for (int k = 0; k < NFILE; ++k)
                    workers[k] = std::thread([&, k] {
                        if (job.offset >= 0 && fseek(fp[k], job.offset, SEEK_SET) != 0) errs[k] = "fseek failed";
                        else if (fwrite(host[job.slot] + (size_t)k * job.nloc, sizeof(float), (size_t)job.nloc, fp[k]) !=
                                 (size_t)job.nloc)
                            errs[k] = "short write to a frame dump file";
                    });
                for (int k = 0; k < NFILE; ++k) {
                    workers[k].join();
                    if (errs[k] && fail.empty()) fail = errs[k];
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                busy[job.slot] = false;
                if (!fail.empty()) error = fail;
            }
            cv.notify_all();
        }
    }

    ~FrameWriter() { shutdown(); }
};

}  // namespace fs
