"""Randomised end-to-end parity sweep: GPU render (C ABI) against the oracle over random configurations -- window sizes on every
K_A path, channel modes, interpolation, view scaling and zoom, window functions, pixel counts, pairs, slope, dB range, poles.
usage: fuzz_parity.py [count] [seed]      (needs a GPU; the oracle is test infrastructure)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
from oracle import pyoracle as po

def random_config(rng):
    W = int(rng.choice([64, 100, 512, 1000, 1024, 2048, 3000, 4096, 5000, 8192, 16384, 20000, 32768, 40000, 65536]))
    if rng.random() < 0.15:
        W = int(rng.integers(33, 9000))
    hop = max(1, int(W * rng.choice([0.25, 0.5, 0.3, 1.0])))
    mode = int(rng.choice([config.CH_LEFT, config.CH_RIGHT, config.CH_MERGE, config.CH_SIDE, config.CH_PHASE, config.CH_SEPARATE,
                           config.CH_MIDSIDE, config.CH_COMPLEX]))
    left = float(rng.choice([0.0, 0.0, 0.1, 0.35]))
    right = float(rng.choice([1.0, 1.0, 0.9, 0.6]))
    cfg = config.spectrum_config(
        sample_rate=float(rng.choice([44100.0, 48000.0, 96000.0, 192000.0])), window_size=W, hop=hop,
        axis_points=int(rng.choice([16, 77, 256, 300, 1024, 1500])), channel_mode=mode,
        bin_interp=int(rng.integers(0, 3)), view_scaling=int(rng.integers(0, 2)),
        window_type=int(rng.integers(0, 8)), window_symmetry=int(rng.integers(0, 2)), window_alpha=float(rng.uniform(0, 3)),
        window_beta=float(rng.uniform(0.5, 9)), num_pairs=int(rng.choice([1, 1, 2, 3])), view_left=left, view_right=right,
        min_log_freq=float(rng.choice([10.0, 20.0, 5.0])), low_db=float(rng.choice([-120.0, -90.0, -60.0])),
        high_db=float(rng.choice([0.0, 6.0])), slope_a=float(rng.choice([0.0, 0.3])), slope_b=float(rng.choice([1.0, 0.7])),
        pole=(float(rng.choice([0.0, 0.5, 0.9, 0.97])), float(rng.choice([0.9, 0.99, 0.999]))))
    return cfg

def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    po.build()
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(count):
        cfg = random_config(rng)
        frames = int(rng.integers(1, 12))
        W, hop = cfg["window_size"], cfg["hop"]
        S = W + (frames - 1) * hop + int(rng.integers(0, hop))
        x = synth.gen(100 + it, cfg["sample_rate"], S, 2 * cfg["num_pairs"])
        try:
            plan = api.Plan(cfg)
        except api.SgzError as e:
            print(it, "rejected:", str(e)[:80]); continue
        plan.upload()
        r = po.spectrogram(po.params_from_dict(cfg), x)
        rgba = plan.render(torch.from_numpy(x).cuda()).cpu().numpy()
        d = np.abs(rgba.astype(int) - r["rgba"].astype(int))
        phase = cfg["channel_mode"] == config.CH_PHASE
        ok = rgba.shape == r["rgba"].shape and d.max() <= (2 if phase else 1) and (d > 0).mean() <= (2e-2 if phase else 5e-3)
        print(it, "ok " if ok else "BAD", "N", plan.N, "path", plan.path, "mode", cfg["channel_mode"], "interp", cfg["bin_interp"], "view",
              cfg["view_scaling"], "P", cfg["axis_points"], "pairs", cfg["num_pairs"], "frames", frames, "max", int(d.max()), "frac", float((d > 0).mean()))
        if not ok:
            bad += 1
            print("   ", json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items() if k not in ("colours",)}))
    print("bad:", bad, "of", count)
    sys.exit(1 if bad else 0)

main()
