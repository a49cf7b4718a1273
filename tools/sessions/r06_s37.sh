#!/bin/bash
# round 6, GPU session 37: the group sums of the shading kernels as DPP adds instead of ds_bpermute butterflies: bit-equality with the butterfly build, parity suites, A/B, the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s37; O=$R/gpurun_out/r6s37
bash tools/build_variants.sh shfl:"-DNVDR_DPP_SUM=0" 2>&1 | tail -1
timeout 600 python tools/variant_equal.py shfl 8 4 16 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/variant_equal.txt
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py tests/test_gpu_geometry.py tests/test_gpu_distributed.py -q -m gpu 2>&1 | grep -v Warning | tail -3 | tee $O/tests.txt
ab() { out=$1; shift; env "$@" timeout 900 python tools/ab_inproc.py ${ROUNDS:-5} 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -5 | tee $O/$out.txt; }
ab ab_bob8 PROBE_VIEWS=8
ab ab_bob1 PROBE_VIEWS=1
ROUNDS=3 ab ab_bob8_n4 PROBE_VIEWS=8 PROBE_N=4
ROUNDS=3 ab ab_bob2_n16 PROBE_VIEWS=2 PROBE_N=16
rm -f nvdiffrecmc_amd/csrc/build/variants/*
echo "== bench"; timeout 600 python bench.py --gpus 1 2> $O/bench_err.log | tail -1 > $O/bench_bob512_n1.json
python -c "
import json; d=json.load(open('$O/bench_bob512_n1.json')); print(d['value'], d['ms_per_step'], d['median_ms_per_step'], d['roofline']['frac'], d['roofline']['forward_pass']['shade_ms'], d['roofline']['backward_pass']['shade_and_light_gradient_ms'])"
