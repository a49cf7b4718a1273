#!/bin/bash
# round-2 GPU session D: re-run of the test files that failed or did not run, A/B of chunk dealing on large meshes, slow-mode survey
# with clock sampling.  Small text outputs only.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
: > $O/r02d_tests.txt
run_tests() { echo "== $*" | tee -a $O/r02d_tests.txt; timeout 420 python -m pytest "$@" -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -22 | cut -c1-250 | tee -a $O/r02d_tests.txt; }
run_tests tests/test_gpu_bvh.py -k "overflow or canonical"
run_tests tests/test_gpu_gbuffer.py
run_tests tests/test_gpu_env_shade.py -k "dead_samples or regenerates"
run_tests tests/test_gpu_fullsize.py -k "training_step or hip_graph"
run_tests tests/test_gpu_distributed.py -k "bench_spawns"
echo "== chunk dealing on a 171k-triangle mesh, 800x800, 1 view (rr = round robin, cur = 64 queues)"
for rep in 1 2; do PROBE_SUBDIV=2 PROBE_RES=800 PROBE_VIEWS=1 AB_WITH_CURRENT=1 bash tools/ab_run.sh 2>&1 | grep -E "==|fwd gen"; done | tee $O/r02d_ab_171k.txt
echo "== chunk dealing on the 684k-triangle mesh, 800x800, 8 views"
PROBE_SUBDIV=3 PROBE_RES=800 PROBE_VIEWS=8 AB_WITH_CURRENT=1 bash tools/ab_run.sh 2>&1 | grep -E "==|fwd gen" | tee $O/r02d_ab_684k.txt
echo "== slow-mode survey with clock sampling"
for i in 1 2 3 4 5 6 7 8 9 10; do
  ( while true; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|socclk" | sed 's/.*clock level: *//' | tr '\n' ' '; echo; sleep 0.25; done ) > /tmp/clk_$i.txt &
  SP=$!
  PROBE_VIEWS=1 timeout 100 python tools/stage_probe.py 8,6,6 2>&1 | grep -E "fwd gen|shader clock" | sed 's/.*fwd gen [0-9.]* trace \([0-9.]*\) shade.*/trace \1 ms/; s/shader clock during the counting launch: //' | tr '\n' ' '
  kill $SP 2>/dev/null; wait $SP 2>/dev/null
  echo " | clock samples: $(sort /tmp/clk_$i.txt | uniq -c | sort -rn | head -3 | tr '\n' ';' | cut -c1-400)"
done | tee $O/r02d_mode.txt
