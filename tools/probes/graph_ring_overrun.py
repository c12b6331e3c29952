"""Does a host that runs many hipGraphLaunch calls of a LONG graph ahead of the device still get in-order execution?  (round 6)

Pure torch, nothing of this repository: a graph of ``--nodes`` dependent kernels (x = x * a + 1 on a buffer big enough that a launch takes
``--us`` microseconds) is replayed ``--replays`` times with NO host synchronisation, with at most ``--depth`` replays in flight (0 = unbounded),
and the result is compared with the closed form.  On ROCm 7.2 / gfx950 the TimeT step graph (~600 nodes, 7.4 ms) produced wrong results with an
unbounded host run-ahead and exact ones with depth 1 - 3 or with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; this probe asks whether the
runtime alone reproduces it.
"""
import argparse

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=600)
ap.add_argument("--replays", type=int, default=60)
ap.add_argument("--depth", type=int, default=0)
ap.add_argument("--mb", type=float, default=64.0, help="buffer size: sets the duration of a node")
a = ap.parse_args()

n = int(a.mb * 2 ** 20 / 8)
x = torch.zeros(n, dtype=torch.float64, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")


def body():
    for _ in range(a.nodes):
        x.add_(1.0)        # in place: every node depends on its predecessor
    cnt.add_(1)


body(); torch.cuda.synchronize(); x.zero_(); cnt.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
torch.cuda.synchronize()
evs = []
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for i in range(a.replays):
    if a.depth and len(evs) >= a.depth:
        evs.pop(0).synchronize()
    g.replay()
    x.mul_(1.0)            # an eager kernel between replays, like the optimizer
    if a.depth:
        e = torch.cuda.Event(); e.record(); evs.append(e)
t1.record()
torch.cuda.synchronize()
want = float(a.nodes * a.replays)
got_min, got_max, c = x.min().item(), x.max().item(), cnt.item()
print(f"nodes {a.nodes} replays {a.replays} depth {a.depth}: {t0.elapsed_time(t1) / a.replays:.3f} ms per replay; expected {want}, got [{got_min}, {got_max}], "
      f"replays counted {c} -> {'OK' if (got_min == want and got_max == want and c == a.replays) else 'WRONG'}")
