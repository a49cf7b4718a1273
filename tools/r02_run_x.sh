#!/bin/bash
# round-2 GPU session X: the driver's launch line for N > 1 (torch.distributed.run), two ranks on this one GPU (gloo: RCCL refuses two
# ranks per device) -- exercises the WORLD_SIZE path of bench.py, which the self-spawn tests do not.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
export NVDR_BENCH_OVERSUBSCRIBE=1 NVDR_BENCH_BACKEND=gloo
timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --no-pmc --no-cpu-baseline --no-extended > gpurun_out/r02x_torchrun_2rank.json 2> gpurun_out/r02x_torchrun.err
echo "rc=$?"; python tools/bench_brief.py gpurun_out/r02x_torchrun_2rank.json | head -1; grep -v "amdgpu.ids\|hostname of the client" gpurun_out/r02x_torchrun.err | tail -5 | cut -c1-300
