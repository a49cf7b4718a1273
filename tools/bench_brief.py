"""Print the essentials of a bench.py JSON line (the full line stays in the file).  usage: bench_brief.py file.json"""
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print('no JSON line in', sys.argv[1], e)
    sys.exit(0)
r = j['roofline']
print({k: j.get(k) for k in ('value', 'ms_per_step', 'median_ms_per_step', 'iters_per_sec', 'iters_per_sec_cached_visibility', 'n_gpus', 'hip_graph', 'scaling')})
print({k: r.get(k) for k in ('bound', 'achieved', 'peak', 'frac', 'traffic', 'kernel_ms_hip_events', 'rays_per_launch', 'kernel_rays_per_sec')})
for k in ('valu', 'hbm', 'l2', 'forward_pass', 'backward_pass', 'pmc_note'):
    print(k, r.get(k))
a = r.get('algorithmic', {})
print('algorithmic', {k: a.get(k) for k in ('bvh2_node_visits_per_ray', 'bvh2_tri_tests_per_ray', 'wide_walk_box_tests_per_ray', 'GBs', 'frac_of_hbm_peak_if_it_were_hbm_traffic')})
print('other', json.dumps(r.get('other_kernels'))[:900])
print('cpu', j.get('cpu_baseline'))
