#!/bin/bash
# round-2 GPU session A: full test suite, smoke, VALU micro-benchmark, bench (default + dmtet800) with in-run PMC passes,
# kernel trace, A/B of the light-gradient band gather vs atomics, slow-mode survey.  Everything lands in gpurun_out/r02a_*.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/r02a_build.log 2>&1; tail -2 $O/r02a_build.log
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > $O/r02a_tests.log 2>&1; tail -25 $O/r02a_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/r02a_smoke.log
timeout 120 tools/ubench/valu_rate > $O/r02a_valu_rate.txt 2>&1; cat $O/r02a_valu_rate.txt
echo "== stage probe (8 views): band gather vs atomics"
PROBE_VIEWS=8 timeout 300 python tools/stage_probe.py 8,6,6 2>&1 | grep -E "fwd gen|counting|clock" | tee $O/r02a_stage_gather.txt
NVDR_DEBUG=16 PROBE_VIEWS=8 timeout 300 python tools/stage_probe.py 8,6,6 2>&1 | grep -E "fwd gen" | tee $O/r02a_stage_atomics.txt
echo "== bench default"
timeout 900 python bench.py --pmc-keep $O/r02a_pmc_bob512 > $O/r02a_bench_bob512.json 2> $O/r02a_bench_bob512.err; tail -c 3000 $O/r02a_bench_bob512.json; tail -5 $O/r02a_bench_bob512.err
echo "== bench dmtet800"
timeout 900 python bench.py --config dmtet800 --steps 20 --warmup 5 --pmc-keep $O/r02a_pmc_dmtet800 > $O/r02a_bench_dmtet800.json 2> $O/r02a_bench_dmtet800.err; tail -c 2500 $O/r02a_bench_dmtet800.json; tail -5 $O/r02a_bench_dmtet800.err
echo "== kernel trace"
cd /tmp; rm -rf /tmp/kt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --steps 20 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*_results.db" | head -1) $R/$O/r02a_kernel_trace.md | cut -c1-170 | head -24
cd $R
echo "== slow-mode survey (one view per launch, fresh processes)"
for i in 1 2 3 4 5 6; do PROBE_VIEWS=1 timeout 120 python tools/stage_probe.py 8,6,6 2>&1 | grep -E "fwd gen|shader clock" | sed 's/.*fwd gen [0-9.]* trace \([0-9.]*\) shade.*/trace \1 ms/; s/shader clock during the counting launch: //' | tr '\n' ' '; echo; done | tee $O/r02a_mode.txt
