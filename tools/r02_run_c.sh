#!/bin/bash
# round-2 GPU session C: the test files one pytest process each (a hang or crash costs one file, not the run), slow-mode survey with
# queue claiming, bench (default) + kernel trace + large-mesh preset.  Small text outputs only.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
: > $O/r02c_tests.txt
for f in "tests/test_gpu_bvh.py -k overflow_or_canonical" tests/test_gpu_gbuffer.py tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py tests/test_gpu_denoiser.py tests/test_gpu_renderutils.py; do
  f=${f//overflow_or_canonical/overflow or canonical}
  echo "== $f" | tee -a $O/r02c_tests.txt
  timeout ${TEST_TIMEOUT:-420} python -m pytest $f -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -${TEST_TAIL:-25} | cut -c1-250 | tee -a $O/r02c_tests.txt
done
echo "== slow-mode survey (1 view, fresh processes, queue claiming)"
for i in 1 2 3 4 5 6 7 8; do PROBE_VIEWS=1 timeout 100 python tools/stage_probe.py 8,6,6 2>&1 | grep -E "fwd gen|shader clock" | sed 's/.*fwd gen [0-9.]* trace \([0-9.]*\) shade.*/trace \1 ms/; s/shader clock during the counting launch: //' | tr '\n' ' '; echo; done | tee $O/r02c_mode.txt
echo "== bench default"
timeout 500 python bench.py --steps 20 --warmup 5 > $O/r02c_bench_bob512.json 2> $O/r02c_bench.err
python tools/bench_brief.py $O/r02c_bench_bob512.json
echo "== kernel trace"
cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --no-extended --steps 20 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*_results.db" | head -1) $R/$O/r02c_kernel_trace.md | cut -c1-150 | head -22
cd $R
echo "== bench dmtet800"
timeout 500 python bench.py --config dmtet800 --steps 10 --warmup 3 --no-extended > $O/r02c_bench_dmtet800.json 2>> $O/r02c_bench.err
python tools/bench_brief.py $O/r02c_bench_dmtet800.json
tail -3 $O/r02c_bench.err | cut -c1-300
