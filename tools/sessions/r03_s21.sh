#!/bin/bash
# round 3, GPU session 21: diffuse + specular filtered in one pass
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== filter tests"; timeout 600 python -m pytest tests/test_gpu_denoiser.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -8
for v in 8 1; do echo "== $v views"; PROBE_VIEWS=$v timeout 300 python tools/dn_probe.py 5 2>&1 | tail -4; done
for pf in 1 0 1 0; do
NVDR_PAIR_FILTER=$pf timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-large-mesh --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pair $pf: 8 views', d['ms_per_step'], d['median_ms_per_step'])"
done
for pf in 1 0; do
NVDR_PAIR_FILTER=$pf timeout 300 python bench.py --batch 1 --no-cpu-baseline --no-pmc --no-large-mesh --steps 50 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pair $pf: 1 view', d['ms_per_step'], d['median_ms_per_step'])"
done
