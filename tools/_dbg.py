import sys, os, json
sys.path.insert(0, '.')
import numpy as np, torch
from signalizer_amd import api, config, synth
from oracle import pyoracle as po; po.build()
over = {"sample_rate": 48000.0, "window_size": 512, "hop": 512, "axis_points": 1024, "channel_mode": 4, "bin_interp": 1, "view_scaling": 1, "window_type": 2, "window_symmetry": 1, "num_pairs": 2, "window_alpha": 0.10920770711850136, "window_beta": 1.2878770615942225, "view_left": 0.1, "view_right": 0.6, "min_log_freq": 10.0, "low_db": -90.0, "high_db": 6.0, "slope_a": 0.0, "slope_b": 0.7, "pole": (0.0, 0.9)}
cfg = config.spectrum_config(**over)
W, hop, P = cfg["window_size"], cfg["hop"], cfg["axis_points"]
frames = 11
x = synth.gen(219, cfg["sample_rate"], W + (frames - 1) * hop + 100, 4)
r = po.spectrogram(po.params_from_dict(cfg), x, want_mapped=True, want_lines=True)
plan = api.Plan(cfg).upload()
m = r["mapped"][:, 0, :P]          # Phase: wsp floats (mag, cancel) interleaved as complex
ref_mag, ref_can = m.real.astype(np.float32), m.imag.astype(np.float32)
got = plan.stage_mapped(torch.from_numpy(x).cuda()).cpu().numpy()   # [F][C][2][P]
print("break", plan.break_pixel, "shape", got.shape)
dm = np.abs(got[:, 0, 0] - ref_mag) / np.abs(ref_mag).max()
dc = np.abs(got[:, 0, 1] - ref_can)
print("mag rel err max", dm.max(), "px", np.nonzero(dm[0] > 1e-5)[0][:10], "cancel abs err max", dc.max(), "px", np.nonzero(dc.max(axis=0) > 1e-3)[0][:20])
for px in np.nonzero(dc.max(axis=0) > 1e-3)[0][:5]:
    print("  px", px, "got", got[:, 0, 1, px], "ref", ref_can[:, px], "mag", got[:,0,0,px], ref_mag[:,px])
rgba = plan.render(torch.from_numpy(x).cuda()).cpu().numpy()
d = np.abs(rgba.astype(int) - r["rgba"].astype(int))
print("rgba max", d.max(), "bad (frame,px)", list(zip(*np.nonzero(d.max(axis=2) > 2)))[:20])
lines = torch.empty((rgba.shape[0], 2, 2, P, 2), dtype=torch.float32, device="cuda")
plan.render(torch.from_numpy(x).cuda(), lines=lines)
L = lines.cpu().numpy(); RL = r["lines"]
for (f, px) in list(zip(*np.nonzero(d.max(axis=2) > 2)))[:4]:
    print("  f", f, "px", px, "rgba", rgba[f, px], r["rgba"][f, px], "lines got", L[f, :, 0, px], "ref", RL[f, :, 0, px])
    print("     mapped got", got[f, :, :, px], "ref", r["mapped"][f, :, px])
