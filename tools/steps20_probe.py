"""What the driver's 20-step timed region (0.7 ms) carries besides twenty renders: the two HIP event records bench.py used to put inside
it, and the wake-up of the final torch.cuda.synchronize().  Per variant: median us per step over 40 regions of 20 steps (spin-up first).
usage: steps20_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth

cfg = config.cfg2(); S = 2880000; dev = torch.device("cuda", 0)
plan = api.Plan(cfg).upload()
xs = [torch.from_numpy(synth.gen(2 + k, 48000, S, 2)).to(dev) for k in range(13)]
F = plan.num_frames(S); rgba = torch.empty((F, 1024, 4), dtype=torch.uint8, device=dev)
hip = C.CDLL("libamdhip64.so"); stream = torch.cuda.current_stream().cuda_stream
e0, e1 = C.c_void_p(), C.c_void_p(); hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
turn = [0]
def step():
    turn[0] = (turn[0] + 1) % 13
    plan.render(xs[turn[0]], rgba=rgba)
def region(events, spin):
    torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if events: hip.hipEventRecord(e0, C.c_void_p(stream))
    for _ in range(20): step()
    if events: hip.hipEventRecord(e1, C.c_void_p(stream))
    if spin:
        while hip.hipStreamQuery(C.c_void_p(stream)) != 0: pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e6
for _ in range(3000): step()
torch.cuda.synchronize()
for name, ev, sp in (("events inside, plain wait (bench.py so far)", True, False), ("no events", False, False), ("no events, stream polled before the wait", False, True), ("events inside, polled", True, True)):
    for _ in range(200): step()
    r = [region(ev, sp) for _ in range(40)]
    print(f"{name:48s}: median {np.median(r):6.2f}  min {np.min(r):6.2f}  max {np.max(r):6.2f} us per step")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): step()
torch.cuda.synchronize(); print(f"2000 steps: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per step")
