"""K renders captured into ONE hipGraph (torch.cuda.CUDAGraph over the library's launches on the capturing stream) against the same K
renders enqueued one by one: per-step time of the replay, and whether the capture works at all (the library must not synchronise, allocate
or query inside a warm render).  usage: graph_probe.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
cfg = config.cfg2(); S = 2880000
plan = api.Plan(cfg).upload()
xs = [torch.from_numpy(synth.gen(2 + k, 48000, S, 2)).to(dev) for k in range(13)]
F = plan.num_frames(S); rgba = torch.empty((F, 1024, 4), dtype=torch.uint8, device=dev)
turn = [0]
def step():
    turn[0] = (turn[0] + 1) % 13
    plan.render(xs[turn[0]], rgba=rgba)
for _ in range(3000): step()
torch.cuda.synchronize()
def plain():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e6
print("one by one: median %.2f us per step over 40 regions" % np.median([plain() for _ in range(40)]))
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        for _ in range(K): step()
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300]); sys.exit(0)
want = rgba.clone()
def replay():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e6
for _ in range(20): replay()
print("one graph : median %.2f us per step over 40 replays" % np.median([replay() for _ in range(40)]))
plan2 = api.Plan(cfg).upload(); ref = plan2.render(xs[turn[0]])
torch.cuda.synchronize()
print("image of the last captured step equals a plain render:", bool(torch.equal(rgba, ref)))
