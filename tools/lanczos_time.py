"""Sustained time of the Lanczos-10 stage call (scopeLanczosKernel: the same tap-weight code as the handle's scopeWaveLanczosKernel)
at the cfg3 view: 153 594 points x 2 channels, launches back to back on one stream between two events.
usage: python tools/lanczos_time.py [launches]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signalizer_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, width, scale = 19200.0, 19200, 8.0
ring = synth.gen(3, 192000, int(W), 2)
vg = api.ScopeView(window_size=W, left=0.0, right=1.0, rendering_scale=scale, width=width)
L = api.lib()
npts = L.sgz_scope_num_points(C.byref(vg))
d = torch.from_numpy(ring).cuda()
out = torch.zeros((2, npts, 2), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def go(k):
    for _ in range(k):
        api.check(L.sgz_scope_lanczos_device(C.byref(vg), d.data_ptr(), ring.shape[1], d.stride(0), 2, out.data_ptr(), s))
go(2000)
torch.cuda.synchronize()
best = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); go(n); b.record(); torch.cuda.synchronize()
    best.append(a.elapsed_time(b) / n * 1e3)
best.sort()
print("points %d x 2 channels: %.2f us per launch (median of 5 x %d), %.1f ns per 1000 point-channels" % (npts, best[2], n, best[2] * 1e3 / (2 * npts / 1000)))
