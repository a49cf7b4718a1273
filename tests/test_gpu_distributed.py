"""The data-parallel leg on real hardware: RCCL (torch.distributed backend "nccl") initialised on the GPU, the flat gradient
bucket reduced on device tensors, and bench.py starting its own ranks.  A one-GPU box can only run world_size 1 through
RCCL (it refuses two ranks on one device); the two-rank run below therefore uses gloo for the collective -- the sharding,
seeding and bucket code is the same."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RCCL_SNIPPET = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from nvdiffrecmc_amd.parallel import allreduce_gradients
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
a = torch.nn.Parameter(torch.zeros(256, 256, 3, device='cuda'))
b = torch.nn.Parameter(torch.zeros(3, device='cuda'))
c = torch.nn.Parameter(torch.zeros(5, device='cuda'))          # no gradient: contributes zeros
a.grad = torch.arange(256 * 256 * 3, dtype=torch.float32, device='cuda').view(256, 256, 3) * 1e-3
b.grad = torch.tensor([1.0, 2.0, 3.0], device='cuda')
want_a, want_b = a.grad.clone(), b.grad.clone()
n = allreduce_gradients([a, b, c], skip_single=False, local_weight=3)   # ONE RCCL all-reduce over the flat bucket
torch.cuda.synchronize()
assert n == (256 * 256 * 3 + 3 + 5) * 4, n
assert torch.allclose(a.grad, want_a) and torch.allclose(b.grad, want_b) and torch.equal(c.grad, torch.zeros(5, device='cuda'))
t = torch.ones(1 << 20, device='cuda')
dist.all_reduce(t)
dist.barrier()
assert float(t.sum()) == float(1 << 20)
dist.destroy_process_group()
print('RCCL_OK')
'''


def test_rccl_world1_allreduce_of_the_gradient_bucket(dev):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', _RCCL_SNIPPET % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _bench(extra, env_extra=None, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--res', '128', '--n-samples-x', '4',
           '--no-pmc', '--no-cpu-baseline', '--no-extended'] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    return r


def test_bench_spawns_its_own_ranks_and_shards_one_batch(dev):
    """bench.py --gpus 2 without a launcher starts two ranks itself, deals ONE batch of views over them (strong scaling,
    north_star's split) and reports the initialised world size.  (Both ranks share the one GPU of this box.)"""
    one = _bench(['--gpus', '1', '--batch', '4'])
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    two = _bench(['--gpus', '2', '--batch', '4'], {'NVDR_BENCH_OVERSUBSCRIBE': '1', 'NVDR_BENCH_BACKEND': 'gloo'})
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    j2 = json.loads(two.stdout.strip().splitlines()[-1])
    assert j1['n_gpus'] == 1 and j2['n_gpus'] == 2 and j2['scaling'] == 'strong'
    assert j1['config']['views_per_iteration'] == j2['config']['views_per_iteration'] == 4
    assert j1['config']['views_rank0'] == 4 and j2['config']['views_rank0'] == 2
    assert j2['config']['allreduce_bytes_per_step'] > 0
    # the same batch either way: the same covered pixels, hence the same shadow-ray queries per step (the traversed share is
    # counted at whatever the seed counter is when the run ends, which differs between the eager and the graph-captured run)
    q1 = j1['shadow_ray_queries_per_sec'] * j1['ms_per_step']
    q2 = j2['shadow_ray_queries_per_sec'] * j2['ms_per_step']
    assert abs(q1 - q2) < 1e-6 * q1
    r1, r2 = j1['value'] * j1['ms_per_step'], j2['value'] * j2['ms_per_step']
    assert abs(r1 - r2) < 0.01 * r1
    assert j2['hip_graph'] is True and j1['hip_graph'] is False      # 2 views per rank: the launch-bound regime is graph-captured


def test_bench_refuses_more_ranks_than_gpus(dev):
    if torch.cuda.device_count() >= 64:
        pytest.skip('needs fewer than 64 GPUs')
    r = _bench(['--gpus', '64', '--batch', '64'])
    assert r.returncode != 0 and 'visible' in (r.stdout + r.stderr)


_TRAIN_SNIPPET = r'''
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from nvdiffrecmc_amd.trainer import DirectLightingStep
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
if world > 1:
    dist.init_process_group('gloo', rank=rank, world_size=world)
res, n, views = 96, 4, [0, 1, 2, 3]
per = len(views) // world
mine = views[rank * per:(rank + 1) * per]
st = DirectLightingStep('bob', res, n, view=mine, n_views=len(views), device='cuda:0', lr=0.03, tex_res=512,
                        pixel_index_offset=mine[0] * res * res, use_graph=(os.environ.get('USE_GRAPH') == '1'),
                        exchange_mode=os.environ.get('EXCHANGE', 'auto'), pipeline=(os.environ.get('PIPELINE', '1') == '1'),
                        optimize_geometry=(os.environ.get('GEOM', '0') != '0'), lr_pos=1e-5, perturb_pos=0.002,
                        rebuild_every=(3 if os.environ.get('GEOM') == '3' else 1))
losses = []
for it in range(8):
    losses.append(float(st.step(world).item()))
st.finish()                 # the pipelined texture update of the last step
torch.cuda.synchronize()
kd = st.params[0].detach().double()
rep = st._ex.report() if world > 1 else {}
out = {'rank': rank, 'losses': losses, 'kd_sum': float(kd.sum()), 'kd_abs': float(kd.abs().sum()), 'light_sum': float(st.params[3].detach().double().sum()),
       'nrm_sum': float(st.params[2].detach().double().sum()), 'ks_sum': float(st.params[1].detach().double().sum()),
       'vpos_sum': float(st.params[-1].detach().double().abs().sum()), 'n_params': len(st.params),
       'resident': bool(getattr(st, '_tex_grad_resident', False)), 'graph': st._graphs is not None, 'exchange': rep}
print('RESULT ' + json.dumps(out))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


@pytest.mark.parametrize('exchange,pipeline,geom', [('dense', '0', '0'), ('dense', '1', '0'), ('sparse', '1', '0'), ('dense', '1', '1'), ('dense', '1', '3')],
                         ids=['dense_unpipelined', 'dense_pipelined', 'sparse_pipelined', 'dense_pipelined_geometry_trained', 'geometry_trained_refit_policy'])
@pytest.mark.parametrize('graph', ['0', '1'], ids=['eager', 'hip_graphs'])
def test_two_rank_training_follows_the_one_rank_run(graph, exchange, pipeline, geom, dev):
    """Four views dealt over two ranks (gloo for the collective, both on this GPU) train like four views on one rank: the chunked
    exchange sums what the ranks' backward passes scatter-added INTO its buckets, the fused Adam of every chunk sees the batch-mean
    gradient.  Per step the mean of the two ranks' losses is the one-rank loss (each rank's loss is the mean over its own views), and
    the trained textures and probe agree up to the order of the additions."""
    def run(world, rank, port):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), USE_GRAPH=graph,
                   EXCHANGE=exchange, PIPELINE=pipeline, GEOM=geom, HSA_ENABLE_IPC_MODE_LEGACY='0')
        return subprocess.Popen([sys.executable, '-c', _TRAIN_SNIPPET % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)

    def result(p):
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, out[-2000:] + err[-3000:]
        return json.loads([l for l in out.splitlines() if l.startswith('RESULT ')][-1][7:])
    one = result(run(1, 0, _free_port()))
    port = _free_port()
    procs = [run(2, r, port) for r in range(2)]
    two = sorted([result(p) for p in procs], key=lambda d: d['rank'])
    assert two[0]['resident'] and two[1]['resident'] and not one['resident']
    assert two[0]['graph'] == (graph == '1')
    for it in range(8):
        pair = 0.5 * (two[0]['losses'][it] + two[1]['losses'][it])
        assert abs(pair - one['losses'][it]) <= 2e-4 * abs(one['losses'][it]), (it, pair, one['losses'][it])
    assert one['losses'][-1] < one['losses'][0]
    if geom != '0':     # v_pos travels in the first chunk; the pipelined geometry stage (rebuild, vertex frames, G-buffer) runs from the reduced vertices
        assert one['n_params'] == 5 and ex_chunk_sizes(two[0]) [0] > 256 * 256 * 3 * 4
    for k in ('kd_sum', 'ks_sum', 'nrm_sum', 'light_sum') + (('vpos_sum',) if geom != '0' else ()):
        assert abs(two[0][k] - two[1][k]) <= 1e-9 * abs(two[0][k])             # the ranks hold the same parameters ...
        assert abs(two[0][k] - one[k]) <= 1e-4 * abs(one[k]), (k, two[0][k], one[k])      # ... and they are the one-rank parameters
    ex = two[0]['exchange']
    assert ex['mode'] == exchange and ex['chunk_modes'][0] == 'dense'          # [probe] first, dense; then the textures
    if exchange == 'sparse':
        assert 0 < ex['tiles_touched'] < 0.5 * ex['tiles_total'] and ex['bytes_sent'] < 0.6 * ex['bytes_dense']
        assert two[0]['exchange']['tiles_touched'] == two[1]['exchange']['tiles_touched']
    else:
        assert ex['bytes_sent'] == ex['bytes_dense']


def ex_chunk_sizes(res):
    return res['exchange']['chunk_bytes_dense']


_FORCED_SNIPPET = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
from nvdiffrecmc_amd.trainer import DirectLightingStep
res, n = 96, 4
out = {}
for tag, kw in (('plain', {}), ('forced', {'force_exchange': True, 'exchange_mode': os.environ.get('EXCHANGE', 'sparse'), 'union_views': [1, 2]})):
    st = DirectLightingStep('bob', res, n, view=[1], n_views=4, device='cuda:0', lr=0.03, tex_res=1024, pixel_index_offset=res * res,
                            use_graph=(os.environ.get('USE_GRAPH') == '1'), **kw)
    losses = [float(st.step(1).item()) for _ in range(7)]
    st.finish()
    torch.cuda.synchronize()
    out[tag] = {'losses': losses, 'sums': [float(p.detach().double().sum()) for p in st.params], 'graph': st._graphs is not None,
                'exchange': st._ex.report() if kw else None, 'split': bool(getattr(st, 'split_stage2', False))}
print('RESULT ' + json.dumps(out))
'''


@pytest.mark.parametrize('exchange', ['sparse', 'dense'])
@pytest.mark.parametrize('graph', ['0', '1'], ids=['eager', 'hip_graphs'])
def test_the_several_rank_schedule_with_one_rank_is_the_plain_iteration(graph, exchange, dev):
    """force_exchange: stage 2 | exchange of [probe] | its Adam | stage 1 of the NEXT iteration | tile-sparse texture chunk (flags, union list,
    gather, scatter) | texture Adam in front of the next lookup -- with one rank and no collective this must train exactly like the plain
    iteration: the same losses step by step and the same parameters up to the order of atomic additions (the pipelining moves launches, not arithmetic)."""
    env = dict(os.environ, USE_GRAPH=graph, EXCHANGE=exchange, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', _FORCED_SNIPPET % ROOT], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('RESULT ')][-1][7:])
    a, b = out['plain'], out['forced']
    assert a['graph'] == b['graph'] == (graph == '1')
    for x, y in zip(a['losses'], b['losses']):         # (up to the order of the lookup adjoint's atomic additions)
        assert abs(x - y) <= 2e-5 * abs(x), (a['losses'], b['losses'])
    for x, y in zip(a['sums'], b['sums']):
        assert abs(x - y) <= 1e-4 * abs(x), (a["sums"], b["sums"])      # (seven Adam steps at lr 0.03 amplify the last bits of a gradient)
    assert a['losses'][-1] < a['losses'][0]
    ex = b['exchange']
    assert ex['mode'] == exchange
    # with locked geometry the dense schedule replays the rebuild as a graph of its own on a side stream and cuts stage 2 in front of the traversal;
    # the tile-sparse exchange keeps the un-split graphs (trainer._capture)
    assert b['split'] == (graph == '1' and exchange == 'dense')
    if exchange == 'sparse':
        assert 0 < ex['tiles_touched'] < ex['tiles_total']          # the union of two views' tiles, not only this view's


_FROZEN_SNIPPET = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
from nvdiffrecmc_amd.trainer import DirectLightingStep
res, n = 96, 4
out = {}
for tag, kw in (('plain', {}), ('forced', {'force_exchange': True, 'exchange_mode': 'dense'})):
    st = DirectLightingStep('bob', res, n, view=[1], n_views=4, device='cuda:0', lr=0.03, tex_res=512, pixel_index_offset=res * res, **kw)
    st.step(1)
    st.set_lr_scale('kd', 0.0)                      # (completes the pending texture update first)
    kd0 = st.parameters()['kd'].detach().clone()
    ks0 = st.parameters()['ks'].detach().clone()
    peak = []
    for _ in range(6):
        st.step(1)
        st.finish()
        torch.cuda.synchronize()
        peak.append(float(st._tex_grad[0].abs().max()))          # the persistent scatter-add buffer (= the exchange bucket when forced) of kd
    frozen_same = bool(torch.equal(st.parameters()['kd'], kd0))
    ks_moved = float((st.parameters()['ks'] - ks0).abs().max())
    st.set_lr_scale('kd', 1.0)
    for _ in range(2):
        st.step(1)
    st.finish()
    torch.cuda.synchronize()
    kd1 = st.parameters()['kd']
    out[tag] = {'peak': peak, 'frozen_same': frozen_same, 'ks_moved': ks_moved, 'finite': bool(torch.isfinite(kd1).all()),
                'moved': float((kd1 - kd0).abs().max()), 'resident': bool(getattr(st, '_tex_grad_resident', False))}
print('RESULT ' + json.dumps(out))
'''


def test_a_frozen_texture_still_has_its_gradient_consumed(dev):
    """set_lr_scale(name, 0): no update, but the persistent scatter-add buffer of the texture lookup's adjoint -- the exchange bucket under
    the several-rank schedule, all-reduced in place every step -- must still be re-zeroed behind every iteration (it grew without
    bound and fed garbage into Adam when the tensor was unfrozen)."""
    r = subprocess.run([sys.executable, '-c', _FROZEN_SNIPPET % ROOT], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('RESULT ')][-1][7:])
    assert out['forced']['resident'] and not out['plain']['resident']
    for tag in ('plain', 'forced'):
        o = out[tag]
        assert o['peak'] == [0.0] * 6, (tag, o['peak'])          # consumed and cleared after every iteration
        assert o['frozen_same'] and o['ks_moved'] > 0.0           # kd stood still, the rest of the set trained
        assert o['finite'] and 0.0 < o['moved'] < 0.2             # unfrozen: two ordinary Adam steps at lr 0.03, not an explosion


@pytest.mark.gpu
def test_graph_capture_tolerates_event_queries_of_other_threads(dev):
    """The harness captures its HIP graphs with thread-local capture errors (trainer._capture_graph): the process group's watchdog thread polls its
    pending collectives with hipEventQuery, which the default ('global') mode turns into "operation not permitted when stream is capturing" whenever a
    capture is open -- what ended one in ~25 runs of the several-rank schedule in round 6.  Here another thread queries an event for the whole length
    of a capture made through the harness's helper: no error on either side, and the graph replays."""
    import threading
    import time
    from nvdiffrecmc_amd import trainer
    x = torch.zeros(1 << 18, device=dev)
    side, ev = torch.cuda.Stream(), torch.cuda.Event()
    with torch.cuda.stream(side):
        for _ in range(100):
            x.add_(1.0)
        ev.record()
    errors, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                ev.query()
            except RuntimeError as e:
                errors.append(str(e).split('\n')[0])
                return
            time.sleep(0.0005)

    y = torch.zeros(1024, device=dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    t = threading.Thread(target=poll)
    t.start()
    with trainer._capture_graph(g):
        for _ in range(20):
            y.add_(1.0)
        time.sleep(0.05)
    stop.set()
    t.join()
    assert not errors, errors
    g.replay()
    torch.cuda.synchronize()
    assert float(y[0]) == 20.0
