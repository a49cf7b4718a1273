#!/bin/bash
# round-2 GPU session E: the whole -m gpu suite in ONE process (as the driver runs it), backward-shading variants, gather LDS sizes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
echo "== full suite, one process"
( time timeout 1200 python -m pytest tests -x -q -m gpu --tb=short -p no:cacheprovider --durations=8 2>&1 | grep -v "amdgpu.ids" | tail -30 | cut -c1-220 ) 2>&1 | tee $O/r02e_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== backward shading variants (8 views)"
PROBE_VIEWS=8 AB_WITH_CURRENT=1 bash tools/ab_run.sh 2>&1 | grep -E "==|fwd gen" | tee $O/r02e_ab_bwd.txt
echo "== env-shade tests on the roll1occ3 variant"
B=nvdiffrecmc_amd/csrc/build; cp $B/libnvdr_hip.so /tmp/lib.orig; cp $B/variants/libnvdr_hip.so.roll1occ3 $B/libnvdr_hip.so
timeout 300 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-200
cp /tmp/lib.orig $B/libnvdr_hip.so
echo "== gather LDS budget (8 views): 64 KB, 156 KB"
for kb in 64 156; do NVDR_LG_LDS_KB=$kb PROBE_VIEWS=8 timeout 200 python tools/stage_probe.py 8,6,6 2>&1 | grep -E "fwd gen" | sed "s/^/lds=$kb /"; done | tee $O/r02e_lds.txt
echo "== blocks per CU of the backward shading kernel (current build)"
PROBE_VIEWS=8 timeout 300 python tools/stage_probe.py 8,6,4 8,6,8 8,6,12 2>&1 | grep -E "fwd gen" | tee $O/r02e_pblocks.txt
