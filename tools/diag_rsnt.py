import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import pyoracle as po
from signalizer_amd import api, config as cf, synth
from test_gpu_resonator import _planes
d = cf.spectrum_config(algorithm=1, sample_rate=48000.0, window_size=32768, hop=171, axis_points=1073, channel_mode=0, window_type=1,
                       view_left=0.0435953682291898, view_right=0.9049718651982309, pole=(0.52, 0.91), free_q=1)
rng = np.random.default_rng(5)
F = 12
x = synth.gen(7, 48000, F * 171 + 30, 2)
p = po.params_from_dict(d)
plan = api.Plan(d).upload()
xs = torch.from_numpy(x).cuda()
got = plan.stage_mapped(xs).cpu().numpy()
r = po.resonator_spectrogram(p, x, want_mapped=True)
ref = _planes(r["mapped"], 0, 1073)
print("frame0 equal", np.array_equal(got[0], ref[0]))
co, g, w = po.resonator_map(p)
print("gain range", g.min(), g.max(), "r range", np.abs(co[1]).min(), np.abs(co[1]).max())
for f in range(F):
    e = np.abs(got[f, 0, 0] - ref[f, 0, 0]); i = int(np.argmax(e))
    print(f, "maxerr", e.max(), "at", i, "ref", ref[f, 0, 0, i], "max ref", ref[f, 0, 0].max(), "gain", g[i])
