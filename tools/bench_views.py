"""SURVEY 8(d) side numbers (not the headline metric): us per call and algorithmic GB/s of the Oscilloscope / Vectorscope
kernels on their BASELINE configs (cfg3, cfg4), and the PCIe-inclusive spectrogram render (host buffers in, RGBA8 out)."""
import sys, os, ctypes as C, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
L = api.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream

def timeit(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us

out = {}
# cfg3: oscilloscope, stereo 192 kHz, 100 ms window, 8 points per sample
W = 19200
ring = torch.from_numpy(synth.gen(3, 192000, W, 2)).to(dev)
v = api.ScopeView(window_size=float(W), left=0.0, right=1.0, rendering_scale=8.0, width=W)
npts = L.sgz_scope_num_points(C.byref(v))
verts = torch.zeros((2, npts, 2), dtype=torch.float32, device=dev)
us = timeit(lambda: api.check(L.sgz_scope_lanczos_device(C.byref(v), ring.data_ptr(), W, ring.stride(0), 2, verts.data_ptr(), stream)))
b = 2 * (W * 4 + npts * 8)
out["scope_lanczos_cfg3"] = {"us": us, "points_per_channel": int(npts), "algorithmic_bytes": b, "GBps": b / us * 1e-3}
# cfg4: vectorscope, 4 pairs 96 kHz, 9600 samples per frame
n = 9600
x = torch.from_numpy(synth.gen(4, 96000, n, 8)).to(dev)
pol = torch.zeros((4, n, 3), dtype=torch.float32, device=dev)
us = timeit(lambda: api.check(L.sgz_vector_polar_device(x.data_ptr(), x.stride(0), 4, n, 8, pol.data_ptr(), stream)))
b = 4 * (2 * n * 4 + n * 12)
out["vector_polar_cfg4"] = {"us": us, "algorithmic_bytes": b, "GBps": b / us * 1e-3}
# spectrogram from host buffers (sgz_spectrogram_render): h2d / kernels / d2h as the library times them
cfg = config.cfg2()
xh = synth.gen(config.CFG2_SEED, 48000, int(config.CFG2_SECONDS * 48000), 2)
best = None
for _ in range(5):
    rgba, _, t = api.render_spectrogram(cfg, xh)
    tot = t["h2d_ms"] + t["kernel_ms"] + t["d2h_ms"]
    if best is None or tot < best[0]:
        best = (tot, t)
t = best[1]
out["spectrogram_host_buffers_cfg2"] = {"h2d_ms": t["h2d_ms"], "kernel_ms": t["kernel_ms"], "d2h_ms": t["d2h_ms"],
                                        "frames_per_s_pcie_inclusive": t["frames"] / (best[0] * 1e-3)}
print(json.dumps(out, indent=1))
