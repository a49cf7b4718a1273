#!/bin/bash
# round 5, GPU session 25: the bilateral filter with two taps per packed instruction (guide planes read as register pairs, radius 11 as a constant):
# parity suite, then A/B against the filter of commit 528f4f7 (`prevfilter`; dn_probe asserts bit-identical outputs between the variants)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s25; O=$R/gpurun_out/r5s25
timeout 900 python -m pytest tests/test_gpu_denoiser.py -q -x 2>&1 | grep -v Warning | tail -5 | tee $O/pytest.txt
PROBE_VIEWS=8 timeout 600 python tools/dn_probe.py 5 2>&1 | grep -v Warning | tee $O/dn_bob8.txt
PROBE_VIEWS=1 timeout 600 python tools/dn_probe.py 5 2>&1 | grep -v Warning | tee $O/dn_bob1.txt
