#!/bin/bash
# Development rehearsal of bench.py's N>1 path on a one-GPU box: ranks share the GPU, halo planes
# travel through the FSIPC transport (device-to-device between the rank processes; default) or host shared memory
# (never a result).  Usage: tools/rehearse_bench.sh N [workload] [ipc|shm]
set -e
N=${1:-2}; WL=${2:-c2}; TR=${3:-ipc}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
    --master-port $((29500 + N)) bench.py --gpus $N --steps 2 --warmup 1 --workload $WL --transport $TR \
    > gpurun_out/rehearse_n$N.json 2> gpurun_out/rehearse_n$N.err || { tail -20 gpurun_out/rehearse_n$N.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/rehearse_n$N.json").read().strip().splitlines()[-1])
print("N=$N", d["slab_parity"], "value", d["value"], "comm", d["comm"])
assert d["slab_parity"]["ok"]
PY
