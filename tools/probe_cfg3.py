import sys, os, time, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from signalizer_amd import api, synth
L = api.lib()
sr, W, nch = 192000.0, 19200, 2
per_frame = 3200
width = 8 * W + 1
h = api.Scope(sample_rate=sr, window_size=float(W), num_channels=nch, trigger_mode=4, channel_mode=0, envelope_mode=2,
              interpolation=3, max_block=512, trigger_threshold=0.05, trigger_channel=1.0, envelope_window=0.3)
view = api.ScopeView(float(W), 0.0, 1.0, 1.0, width, 0)
x = synth.gen(31, int(sr), per_frame * 64, nch)
pinned = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory().numpy()
nv = L.sgz_scope_vertex_count(h.h, ctypes.byref(view))
outs = [(pinned((nv, 3), torch.float32), pinned((nv, 4), torch.uint8)) for _ in (0, 1)]
T = np.zeros(5); frame = 0
def step(acc):
    global frame
    a = (frame % 64) * per_frame
    t0 = time.perf_counter()
    for pos in range(a, a + per_frame, 512):
        while h.push(x[:, pos:min(pos + 512, a + per_frame)]) == api.SGZ_BUSY: pass
    t1 = time.perf_counter()
    h.flush(); t2 = time.perf_counter()
    h.peak_filter(1 / 60, 8); t3 = time.perf_counter()
    h.vertices(view, 0, 0, out=outs[0]); t4 = time.perf_counter()
    h.vertices(view, 1, 0, out=outs[1]); t5 = time.perf_counter()
    frame += 1
    if acc: T[:] += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
for _ in range(70): step(False)
for _ in range(100): step(True)
print("per frame (us): pushes %.0f  flush %.0f  peak_filter %.0f  vertices0 %.0f  vertices1 %.0f  total %.0f" % (*(T / 100 * 1e6), T.sum() / 100 * 1e6))
