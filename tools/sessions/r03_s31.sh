#!/bin/bash
# round 3, GPU session 31: configs[4] stand-in (parity test + bench line), roctx ranges under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/s31; O=gpurun_out/s31
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "config5 or config3" 2>&1 | tail -3
echo "== bench hotdog512x256"; timeout 600 python bench.py --config hotdog512x256 --no-cpu-baseline --steps 8 --warmup 3 --pmc-keep $O 2> $O/bench_err.log | tail -1 > $O/bench_hotdog512x256_n1.json
python - <<P
import json; d=json.load(open('$O/bench_hotdog512x256_n1.json'))
print({k: d[k] for k in ('value','ms_per_step','median_ms_per_step','iters_per_sec')}, d['roofline'])
P
echo "== roctx"; cd /tmp; export TMPDIR=/tmp
NVDR_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --stats -d $R/$O/roctx -o roctx -- python $R/bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-large-mesh --no-extended > $R/$O/roctx.log 2>&1
cd $R; ls $O/roctx | head; f=$(ls $O/roctx/*marker_api_stats* 2>/dev/null | head -1); [ -n "$f" ] && head -20 $f
rm -f $O/roctx/*.db
