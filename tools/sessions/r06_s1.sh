#!/bin/bash
# round 6, GPU session 1: the marching-tets meshes -- new parity tests, bench lines of the new presets, treelet gain and rebuild time next to the subdivided stand-ins
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s1; O=$R/gpurun_out/r6s1
timeout 900 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py -q -m gpu -x -k "dmtet or frozen or oct_tree" -s 2>&1 | grep -v Warning | tail -25 | tee $O/tests.txt
for cfg in dmtet64_800 dmtet64_init512; do
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-one-view --pmc-keep $O 2>$O/bench_$cfg.err | tail -1 > $O/bench_$cfg.json
  python -c "
import json; d=json.load(open('$O/bench_$cfg.json')); r=d['roofline']; a=r['algorithmic']
print('$cfg', 'ms', round(d['ms_per_step'],2), 'med', round(d['median_ms_per_step'],2), 'covered', d['config']['covered_pixels_rank0'], 'rays', r['rays_per_launch'], 'trace ms', round(r['kernel_ms_hip_events'],3), 'Grays/s', round(r['kernel_rays_per_sec']/1e9,2), 'frac', r['frac'], 'hbm', (r.get('hbm') or {}).get('hbm_frac'), 'l2hit', (r.get('l2') or {}).get('l2_hit'))
print('   steps/ray', round(a['oct_walk_node_steps_per_ray'],2), 'box', round(a['wide_walk_box_tests_per_ray'],1), 'tri', round(a['wide_walk_tri_tests_per_ray'],2), 'bvh2 nodes', round(a['bvh2_node_visits_per_ray'],1), 'batch fill', round(a['triangle_test_batch_fill'],3))
print('   fwd', r['forward_pass']['gen_ms'], r['forward_pass']['shade_ms'], 'bwd', r['backward_pass'])
"
done
for m in dmtet64_mid dmtet64_init; do timeout 200 python tools/bvh_probe.py $m 0 40 2>&1 | grep triangles; done | tee $O/bvh_build_alone.txt
SRC=bvh.hip bash tools/build_variants.sh karras:"-DNVDR_TREELET_W=0" 2>&1 | tail -1
PROBE_MESH=dmtet64_mid PROBE_RES=800 PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v Warning | tee $O/ab_treelets_dmtet64_mid.txt
PROBE_MESH=dmtet64_init PROBE_RES=512 PROBE_VIEWS=4 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v Warning | tee $O/ab_treelets_dmtet64_init.txt
