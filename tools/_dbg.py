import sys, os, json
sys.path.insert(0, '.')
import numpy as np, torch
from signalizer_amd import api, config, synth
from oracle import pyoracle as po; po.build()
cfgs = [
 {"sample_rate": 192000.0, "window_size": 1000, "hop": 1000, "axis_points": 300, "channel_mode": 0, "bin_interp": 2, "view_scaling": 1, "window_type": 6, "window_symmetry": 0, "num_pairs": 1, "window_alpha": 0.8157172852983607, "window_beta": 8.886762407695427, "view_left": 0.1, "view_right": 1.0, "min_log_freq": 10.0, "low_db": -90.0, "high_db": 6.0, "slope_a": 0.0, "slope_b": 1.0, "pole": (0.0, 0.999)},
 {"sample_rate": 96000.0, "window_size": 5000, "hop": 1500, "axis_points": 1024, "channel_mode": 1, "bin_interp": 2, "view_scaling": 1, "window_type": 6, "window_symmetry": 0, "num_pairs": 1, "window_alpha": 2.6551368416127543, "window_beta": 6.927766644014668, "view_left": 0.0, "view_right": 1.0, "min_log_freq": 5.0, "low_db": -60.0, "high_db": 6.0, "slope_a": 0.0, "slope_b": 0.7, "pole": (0.0, 0.999)},
 {"sample_rate": 48000.0, "window_size": 1024, "hop": 307, "axis_points": 1024, "channel_mode": 1, "bin_interp": 1, "view_scaling": 0, "window_type": 4, "window_symmetry": 1, "num_pairs": 1, "window_alpha": 0.11, "window_beta": 7.58, "view_left": 0.0, "view_right": 1.0, "min_log_freq": 20.0, "low_db": -120.0, "high_db": 0.0, "slope_a": 0.3, "slope_b": 1.0, "pole": (0.97, 0.9)},
]
for over in cfgs:
    cfg = config.spectrum_config(**over)
    W, hop, P = cfg["window_size"], cfg["hop"], cfg["axis_points"]
    frames = 3
    x = synth.gen(143, cfg["sample_rate"], W + (frames - 1) * hop, 2)
    r = po.spectrogram(po.params_from_dict(cfg), x, want_mapped=True)
    plan = api.Plan(cfg).upload()
    sides = plan.sides
    m = r["mapped"][:, :, :sides * P].reshape(frames, 1, sides, P)
    ref = np.sqrt((m.real.astype(np.float32) ** 2 + m.imag.astype(np.float32) ** 2).astype(np.float32))
    got = plan.stage_mapped(torch.from_numpy(x).cuda()).cpu().numpy()
    d = np.abs(got - ref) / np.abs(ref).max()
    bad = np.nonzero(d[0, 0, 0] > 1e-5)[0]
    print("N", plan.N, "path", plan.path, "break", plan.break_pixel, "rel err max", d.max(), "bad px:", bad[:12], len(bad), "imag!=0 px", np.nonzero(m[0,0,0].imag != 0)[0][:8])
    for b in bad[:4]: print("    px", b, got[0,0,0,b], ref[0,0,0,b], m[0,0,0,b])
