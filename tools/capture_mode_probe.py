"""HIP-graph capture and another thread's hipEventQuery: what the process group's watchdog thread does while the harness captures.
With capture_error_mode='global' (torch's default) the query is an error -- the failure seen once in ~25 runs of the several-rank
schedule (round 6) --, with 'thread_local' (what trainer._capture_graph uses) it is not.
    python tools/capture_mode_probe.py"""
import threading
import time
import torch

dev = torch.device('cuda:0')
x = torch.zeros(1 << 20, device=dev)
side = torch.cuda.Stream()
ev = torch.cuda.Event()
with torch.cuda.stream(side):
    for _ in range(200):
        x.add_(1.0)
    ev.record()


def poll(result, stop):
    while not stop.is_set():
        try:
            ev.query()
        except RuntimeError as e:            # noqa: PERF203
            result.append(str(e).split('\n')[0])
            return
        time.sleep(0.0005)


for mode in ('thread_local', 'global'):
    result, stop = [], threading.Event()
    t = threading.Thread(target=poll, args=(result, stop))
    y = torch.zeros(1024, device=dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    t.start()
    try:
        with torch.cuda.graph(g, capture_error_mode=mode):
            for _ in range(50):
                y.add_(1.0)
            time.sleep(0.05)                 # (the capture stays open while the other thread polls)
    except RuntimeError as e:
        result.append('capture: ' + str(e).split('\n')[0])
    stop.set()
    t.join()
    print('%-12s: %s' % (mode, 'event queries from another thread during the capture: no error' if not result else 'ERROR -- ' + result[0]))
    torch.cuda.synchronize()
