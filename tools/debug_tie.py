import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from signalizer_amd import api, config, synth
from oracle import pyoracle as po
import parity_chain as pc
cases = [
 (266395, 273, {"sample_rate": 192000.0, "window_size": 41576, "hop": 28126, "axis_points": 1276, "channel_mode": 4, "bin_interp": 1, "view_scaling": 0, "window_type": 5, "window_symmetry": 0, "num_pairs": 3, "window_alpha": 1.3655544878677526, "window_beta": 4.306534423327001, "view_left": 0.6286670910667574, "view_right": 1.0, "min_log_freq": 101.4876458107646, "low_db": -143.5860671359662, "high_db": -1.054302015974292, "clip_db": -384.0, "slope_a": 0.0, "slope_b": 1.0, "pole": [0.0, 0.9], "ratios": [0.2, 0.2, 0.2, 0.2, 0.2]}, (7,1,1229)),
]
for S, seed, over, (f, c, px) in cases:
    cfg = config.spectrum_config(**over)
    cfg["pole"] = tuple(cfg["pole"]); cfg["ratios"] = tuple(cfg["ratios"])
    x = synth.gen(seed, cfg["sample_rate"], S, 2 * cfg["num_pairs"])
    plan = api.Plan(cfg).upload()
    p = po.params_from_dict(cfg)
    got = plan.stage_mapped(torch.from_numpy(x).cuda()).cpu().numpy()
    r = po.spectrogram(p, x, want_mapped=True)
    P = plan.P
    print("break", plan.break_pixel, "N", plan.N, "got", got[f, c, :, px], "ref", r["mapped"][f, c, px])
    W, hop, N = p.window_size, cfg["hop"], plan.N
    raw, _, _ = po.frame_bins(p, x[2 * c, f * hop:f * hop + W], x[2 * c + 1, f * hop:f * hop + W])
    Z = raw[:N].astype(np.complex128)
    mf = plan.mapped_frequencies()
    nb = N // 2
    f2b = np.float32(nb / np.float32(p.sample_rate / 2))
    lo = int(np.float32(mf[px - 1]) * f2b); hi = int(np.float32(mf[px]) * f2b)
    print("bins", lo, hi, mf[px-1]*f2b, mf[px]*f2b)
    k = np.arange(max(1, lo - 2), hi + 3)
    Lk = (Z[k] + np.conj(Z[N - k])) * 0.5
    Rk = (Z[k] - np.conj(Z[N - k])) * (-0.5j)
    key = np.maximum(np.abs(Lk) ** 2, np.abs(Rk) ** 2)
    inv = plan.window_scale / (W * 0.5)
    val = (np.abs(Lk) + np.abs(Rk)) * inv
    canc = 1 - np.abs(Lk + Rk) / (np.abs(Lk) + np.abs(Rk))
    for i in range(len(k)):
        print("  k", k[i], "key/max", key[i] / key.max(), "val", val[i], "canc", canc[i])
    problems, stats = pc.check_render(po, plan, cfg, x, torch.device("cuda:0"))
    print("chain:", problems, stats)
