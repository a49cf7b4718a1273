#!/bin/bash
# round 5, GPU session 24: light-gradient gather with the workgroups dealt to the bands by their records (mode 2) against the two fixed splits
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s24; O=$R/gpurun_out/r5s24
timeout 900 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -v Warning | tail -5 | tee $O/pytest.txt
AB_LG=1 PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
AB_LG=1 PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob1.txt
