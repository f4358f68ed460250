#!/bin/bash
# Regenerates the evidence set under profiles/ for one round tag.  Run ON THE GPU BOX through gpurun:
#   gpurun --timeout 1200 -- 'tools/make_profiles.sh r02a'
# and then, back in the container, copy what it left in gpurun_out/profiles_<tag>/ into profiles/ and run
#   python3 tools/pmc_to_traffic.py gpurun_out/profiles_<tag>/pmc_fetch gpurun_out/profiles_<tag>/pmc_write c3 gpurun_out/profiles_<tag>/pmc_fetch_bench.json
# (rewrites profiles/sweep_traffic.json, which bench.py reads for roofline.traffic); likewise pmc_fetch_c4 / pmc_write_c4 with
# workload c4 and pmc_fetch_c5 / pmc_write_c5 with workload c3_fp64.
#
# Rules of the pool this script respects: the profiled program follows `--` directly (python3, no env /
# bash -c hops); --pmc runs are separate from --kernel-trace/--stats runs; FETCH_SIZE and WRITE_SIZE
# are collected in separate passes (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2).
set -e
TAG=${1:-rXX}
OUT=gpurun_out/profiles_$TAG
rm -rf $OUT && mkdir -p $OUT   # (gpurun merges into an existing local copy: clear that one by hand before re-running a tag)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-extra"
# 1. the bench line itself (with cpu_baseline and the extras)
timeout -k 10 400 python3 bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err
# the launch plans that line ran: every profiled pass below replays them (under counter collection the clock is a
# different one, and the plan is part of what roofline.traffic is stamped with)
plans() { python3 -c "import json,sys; r=json.load(open(sys.argv[1]))['roofline']; print('%d,%d' % (r['workgroup_shape_id'], r['launch_plan_three_sweeps']))" "$1"; }
P3=$(plans $OUT/bench_c3.json)
# 2. kernel durations under rocprofv3; the same command's own HIP-event number lands next to it
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kernel_stats --output-format csv -- $B --steps 5 --warmup 2 --launch-plans $P3 \
    > $OUT/bench_c3_under_rocprofv3.json 2> $OUT/rocprof.err
# 3. HBM traffic: two PMC passes
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch --output-format csv -- $B --steps 2 --warmup 1 --launch-plans $P3 > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_f.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write --output-format csv -- $B --steps 2 --warmup 1 --launch-plans $P3 > /dev/null 2> $OUT/pmc_w.err
# 4. issue-side counters of the solver kernels (VALU share, waits)
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    -d $OUT/pmc_sq --output-format csv -- $B --steps 1 --warmup 1 --launch-plans $P3 > /dev/null 2> $OUT/pmc_sq.err
python3 tools/pmc_summary.py $OUT/pmc_sq $OUT/pmc_sq_summary.json > /dev/null
# 5. the other configurations the metric names
timeout -k 10 300 $B --workload c2 --steps 10 > $OUT/bench_c2.json 2> /dev/null
timeout -k 10 300 $B --workload c4 --steps 5 > $OUT/bench_c4_n1.json 2> /dev/null
timeout -k 10 400 $B --precision fp64 --steps 3 --warmup 1 > $OUT/bench_c5_fp64.json 2> /dev/null
# 6. HBM traffic of the dominant kernel of configs 4 and 5 (the two-sweep kernel): the same two PMC passes each
P4=$(plans $OUT/bench_c4_n1.json)
P5=$(plans $OUT/bench_c5_fp64.json)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_c4 --output-format csv -- $B --workload c4 --steps 1 --warmup 1 --launch-plans $P4 > $OUT/pmc_fetch_c4_bench.json 2> /dev/null
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_c4 --output-format csv -- $B --workload c4 --steps 1 --warmup 1 --launch-plans $P4 > /dev/null 2> /dev/null
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_c5 --output-format csv -- $B --precision fp64 --steps 1 --warmup 1 --launch-plans $P5 > $OUT/pmc_fetch_c5_bench.json 2> /dev/null
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_c5 --output-format csv -- $B --precision fp64 --steps 1 --warmup 1 --launch-plans $P5 > /dev/null 2> /dev/null
ls -la $OUT
