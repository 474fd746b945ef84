import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from signalizer_amd import api, config, synth
from oracle import pyoracle as po
import parity_chain as pc
cases = [
 (219134, 227, {"sample_rate": 48000.0, "window_size": 30951, "hop": 24199, "axis_points": 3501, "channel_mode": 4, "bin_interp": 1, "view_scaling": 0, "window_type": 3, "window_symmetry": 0, "num_pairs": 4, "window_alpha": 0.9984475263199435, "window_beta": 5.940136518994648, "view_left": 0.32076750693059997, "view_right": 0.7800275537619439, "min_log_freq": 21.948234491973718, "low_db": -26.720321558724876, "high_db": 5.678858040526887, "clip_db": -384.0, "slope_a": 0.0, "slope_b": 0.7, "pole": [0.9, 0.999], "ratios": [0.2, 0.2, 0.2, 0.2, 0.2]}, (5,1,1566)),
 (36078, 387, {"sample_rate": 44100.0, "window_size": 16384, "hop": 9088, "axis_points": 3221, "channel_mode": 4, "bin_interp": 1, "view_scaling": 1, "window_type": 0, "window_symmetry": 0, "num_pairs": 2, "window_alpha": 0.10917435600777625, "window_beta": 2.7452406144578054, "view_left": 0.24545190254984617, "view_right": 1.0, "min_log_freq": 192.91710683396533, "low_db": -128.73327234069512, "high_db": -3.8151905534295905, "clip_db": -384.0, "slope_a": 0.0, "slope_b": 0.7, "pole": [0.5, 0.99], "ratios": [0.2, 0.2, 0.2, 0.2, 0.2]}, (2,0,2261)),
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
