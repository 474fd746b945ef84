import sys, os, json
sys.path.insert(0, '.')
import numpy as np, torch
from signalizer_amd import api, config, synth
from oracle import pyoracle as po; po.build()
cases = [
 ({"sample_rate": 44100.0, "window_size": 4905, "hop": 2130, "axis_points": 1771, "channel_mode": 4, "bin_interp": 2, "view_scaling": 0, "window_type": 5, "window_symmetry": 1, "num_pairs": 1, "window_alpha": 2.5772116414134625, "window_beta": 5.865243326531589, "view_left": 0.17586159272877772, "view_right": 1.0, "min_log_freq": 7.257216811416272, "low_db": -126.73220617842847, "high_db": 5.571174747166207, "slope_a": 0.0, "slope_b": 0.7, "pole": (0.5, 0.9)}, 106, 6),
 ({"sample_rate": 96000.0, "window_size": 24315, "hop": 28043, "axis_points": 2400, "channel_mode": 4, "bin_interp": 1, "view_scaling": 0, "window_type": 6, "window_symmetry": 0, "num_pairs": 1, "window_alpha": 0.8509542949556593, "window_beta": 2.344978791337828, "view_left": 0.16793628641887173, "view_right": 0.5932429559424243, "min_log_freq": 131.06672841304444, "low_db": -101.66078169366199, "high_db": -6.233128467620861, "slope_a": 0.0, "slope_b": 0.7, "pole": (0.0, 0.9)}, 172, 11)]
for over, seed, frames in cases:
    cfg = config.spectrum_config(**over)
    W, hop, P = cfg["window_size"], cfg["hop"], cfg["axis_points"]
    x = synth.gen(seed, cfg["sample_rate"], 15885 if seed == 106 else W + (frames - 1) * hop + 7, 2)
    r = po.spectrogram(po.params_from_dict(cfg), x, want_mapped=True)
    plan = api.Plan(cfg).upload()
    m = r["mapped"][:, 0, :P]
    ref_mag, ref_can = m.real.astype(np.float32), m.imag.astype(np.float32)
    got = plan.stage_mapped(torch.from_numpy(x).cuda()).cpu().numpy()
    dm = np.abs(got[:, 0, 0] - ref_mag) / np.abs(ref_mag).max()
    dc = np.abs(got[:, 0, 1] - ref_can)
    print("N", plan.N, "break", plan.break_pixel, "P", P, "mag rel err max", dm.max(), "at px", np.unique(np.nonzero(dm > 1e-5)[1])[:10],
          "cancel abs err max", dc.max(), "at px", np.unique(np.nonzero(dc > 1e-3)[1])[:10])
    for px in np.unique(np.nonzero(dc > 1e-3)[1])[:3]:
        print("   px", px, "cancel got", got[:3, 0, 1, px], "ref", ref_can[:3, px], "mag got", got[:3, 0, 0, px], "ref", ref_mag[:3, px])
    if seed == 106:
        px, f = 95, 4
        print("   frame 4 px 95: mapped got", got[f, 0, :, px], "ref", ref_mag[f, px], ref_can[f, px])
        p = po.params_from_dict(cfg)
        raw, csf, csp = po.frame_bins(p, x[0, f * hop:f * hop + W], x[1, f * hop:f * hop + W])
        # which bins compete for this pixel?  print the top candidates of max(|L|^2,|R|^2) over the pixel's run
        Z = raw[:plan.N].astype(np.complex128)
        k = np.arange(1, plan.N // 2)
        Lk = (Z[k] + np.conj(Z[plan.N - k])) / 2; Rk = (Z[k] - np.conj(Z[plan.N - k])) / 2j
        mx = np.maximum(np.abs(Lk) ** 2, np.abs(Rk) ** 2)
        mf = plan.mapped_frequencies() if hasattr(plan, "mapped_frequencies") else None
        f2b = (plan.N / 2) / (cfg["sample_rate"] / 2)
        b0, b1, b2 = int(mf[px - 1] * f2b), int(mf[px] * f2b), int(mf[px + 1] * f2b)
        print("   bins", b0, b1, b2, "candidates:", [(int(kk), float(mx[kk - 1]), float(abs(Lk[kk-1]) + abs(Rk[kk-1]))) for kk in range(b0, b2 + 1)])
    rgba = plan.render(torch.from_numpy(x).cuda()).cpu().numpy()
    d = np.abs(rgba.astype(int) - r["rgba"].astype(int))
    print("   rgba max", d.max(), "bad (frame,px)", list(zip(*np.nonzero(d.max(axis=2) > 2)))[:10])
