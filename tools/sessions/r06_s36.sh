#!/bin/bash
# round 6, GPU session 36: the driver's multi-rank launch line as a dry run at the final HEAD: 2 and 4 ranks on ONE GPU over gloo (NVDR_BENCH_OVERSUBSCRIBE), HIP graphs on and off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s36; O=$R/gpurun_out/r6s36
export NVDR_BENCH_BACKEND=gloo NVDR_BENCH_OVERSUBSCRIBE=1
for n in 2 4; do for graph in auto on; do
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29520 + n)) bench.py --gpus $n --steps 20 --warmup 5 --graph $graph --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation 2> $O/err_${n}_$graph.txt | tail -1 > $O/out_${n}_$graph.json
  python -c "
import json; d=json.load(open('$O/out_${n}_$graph.json')); e=d['config'].get('exchange') or {}
print('ranks $n graph $graph:', d['n_gpus'], d['hip_graph'], round(d['ms_per_step'],3), d['scaling'], d['config'].get('views_rank0'), {k: e.get(k) for k in ('mode','policy','exposed_ms')})" || { echo "ranks $n graph $graph FAILED"; grep -v "Warning\|warn\|amdgpu.ids" $O/err_${n}_$graph.txt | tail -12 | cut -c1-250; }
done; done
