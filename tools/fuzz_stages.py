"""Randomised BIT-EXACT sweeps of the stages whose bar is exactness:
  (i)  pixel mapping given identical csf magnitudes (Separate / MidSide: every csf entry is a magnitude) -- whichever map
       implementation the configuration selects: the fused kernel's balanced scan or its serial fallback, the per-side LDS map,
       the generic map kernel;
  (ii) peak decay + dB + colour given identical mapped magnitudes: line values within 2 ulp (one libm-vs-fp64 log ulp, one more
       rounding), RGBA8 within 1 LSB on <= 1e-4 of the bytes.
usage: fuzz_stages.py [count] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from signalizer_amd import api, config, synth
from oracle import pyoracle as po
from fuzzcfg import random_config

def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    po.build()
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(count):
        cfg = random_config(rng, wild=rng.random() < 0.5)
        cfg["channel_mode"] = int(rng.choice([config.CH_SEPARATE, config.CH_MIDSIDE]))
        cfg["num_pairs"] = 1
        W, P = cfg["window_size"], cfg["axis_points"]
        try:
            plan = api.Plan(cfg).upload()
        except api.SgzError:
            continue
        p = po.params_from_dict(cfg)
        frames = 2
        x = synth.gen(1700 + it, cfg["sample_rate"], W * frames, 2)
        csfs = np.zeros((frames, 1, plan.N + 1), np.float32)
        want = np.zeros((frames, 1, 2, P), np.float32)
        for f in range(frames):
            raw, csf, csp = po.frame_bins(p, x[0, f * W:(f + 1) * W], x[1, f * W:(f + 1) * W])
            csfs[f, 0] = csf.real
            v = csp.reshape(2, P)
            want[f, 0] = np.sqrt((v.real * v.real + v.imag * v.imag).astype(np.float32)).astype(np.float32)
        got = plan.stage_map_from_bins(torch.from_numpy(csfs).cuda()).cpu().numpy()
        ok1 = np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # (ii) decay + colour from random magnitudes
        C_ = int(rng.integers(1, 5))
        cfg2 = dict(cfg, num_pairs=C_, channel_mode=int(rng.choice([config.CH_SEPARATE, config.CH_MIDSIDE, config.CH_LEFT, config.CH_MERGE])),
                    ratios=tuple(float(v) for v in rng.uniform(0.05, 1.0, 5)))
        plan2 = api.Plan(cfg2).upload()
        F2 = int(rng.choice([1, 3, 8, 9, 37, 70, int(rng.integers(1, 200))]))
        if W <= 4096 and rng.random() < 0.25:
            F2 = int(rng.choice([513, 700, 1500]))            # > 64 time chunks: the two-kernel K_B path
            cfg2["hop"] = max(1, W // 8)
        S2 = W + (F2 - 1) * cfg2["hop"]
        x2 = synth.gen(1900 + it, cfg2["sample_rate"], S2, 2 * C_)
        if S2 > 3000 and rng.random() < 0.5:
            x2[:, S2 // 3:S2 // 2] = 0
        r = po.spectrogram(po.params_from_dict(cfg2), x2, want_lines=True, want_mapped=True)
        sides = plan2.sides
        m = r["mapped"][:, :, :sides * P].reshape(F2, C_, sides, P)
        mag = np.sqrt((m.real.astype(np.float32) ** 2 + m.imag.astype(np.float32) ** 2).astype(np.float32)).astype(np.float32)
        rgba, lines = plan2.stage_decay_colour(torch.from_numpy(mag).cuda(), want_lines=True)
        rgba, lines = rgba.cpu().numpy(), lines.cpu().numpy()
        ref = np.stack([r["lines"].real, r["lines"].imag], axis=-1).astype(np.float32)
        if sides == 1:
            lines, ref = lines[..., 0], ref[..., 0]
        both = np.isfinite(lines) & np.isfinite(ref)
        ulp = np.abs(lines.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))[both]
        mism = rgba != r["rgba"]
        ok2 = (ulp.size == 0 or ulp.max() <= 2) and mism.mean() <= 1e-4 and np.abs(rgba.astype(int) - r["rgba"].astype(int)).max() <= 1 and np.array_equal(np.isfinite(lines), np.isfinite(ref))
        print(it, "ok " if ok1 and ok2 else "BAD", "N", plan.N, "path", plan.path, "interp", cfg["bin_interp"], "view", cfg["view_scaling"], "P", P,
              "map", ok1, "| pairs", C_, "mode", cfg2["channel_mode"], "frames", F2, "decay", ok2, "ulp", int(ulp.max()) if ulp.size else 0,
              "mism", float(mism.mean()))
        bad += 0 if ok1 and ok2 else 1
    print("bad:", bad, "of", count)
    sys.exit(1 if bad else 0)

main()
