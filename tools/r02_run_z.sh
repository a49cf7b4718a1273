#!/bin/bash
# round-2 GPU session Z: the whole GPU suite + smoke on the final state of the round
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 70 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300 | tee gpurun_out/r02z_pytest.txt
timeout 20 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
