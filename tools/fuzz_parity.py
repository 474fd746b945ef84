"""Randomised end-to-end parity sweep: GPU render (C ABI) against the oracle over random configurations -- window sizes on every
K_A path, channel modes, interpolation, view scaling and zoom, window functions, pixel counts, pairs, slope, dB range, poles.
usage: fuzz_parity.py [count] [seed] [wild] [mode=N]      (needs a GPU; the oracle is test infrastructure)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
from oracle import pyoracle as po
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from fuzzcfg import random_config
from parity_chain import check_render

def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    wild = len(sys.argv) > 3 and sys.argv[3] == "wild"
    force_mode = next((int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("mode=")), None)      # e.g. mode=4: Phase only
    po.build()
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(count):
        cfg = random_config(rng, wild)
        if force_mode is not None:
            cfg["channel_mode"] = force_mode
        frames = int(rng.integers(1, 12))
        W, hop = cfg["window_size"], cfg["hop"]
        S = W + (frames - 1) * hop + int(rng.integers(0, hop))
        x = synth.gen(100 + it, cfg["sample_rate"], S, 2 * cfg["num_pairs"])
        try:
            plan = api.Plan(cfg)
        except api.SgzError as e:
            print(it, "rejected:", str(e)[:80]); continue
        plan.upload()
        # the parity chain (tests/parity_chain.py): mapped pixels within the FFT tolerance (Phase near-tie flips verified against the
        # oracle's own bins), colour bytes and line values exact given the mapped pixels
        problems, stats = check_render(po, plan, cfg, x, torch.device("cuda:0"), want_lines=(it % 4 == 0))
        ok = not problems
        if ok and it % 8 == 0:
            rgba2, _, _ = api.render_spectrogram(cfg, x)              # the host-buffer entry point renders the same bytes
            # (it copies the channels into rows of a 64-sample multiple; the same layout here, so that both renders take the same
            # kernels -- before the real-input kernels took dword-aligned rows a plan fell back to the complex ones on odd strides)
            xt = torch.zeros((x.shape[0], (x.shape[1] + 63) // 64 * 64), dtype=torch.float32, device="cuda")
            xt[:, :x.shape[1]] = torch.from_numpy(x).cuda()
            ok = np.array_equal(rgba2, plan.render(xt[:, :x.shape[1]]).cpu().numpy())
            if not ok: problems = ["host-buffer entry point differs from the device render"]
        print(it, "ok " if ok else "BAD", "N", plan.N, "path", plan.path, "mode", cfg["channel_mode"], "interp", cfg["bin_interp"], "view",
              cfg["view_scaling"], "P", cfg["axis_points"], "pairs", cfg["num_pairs"], "frames", frames, stats)
        if not ok:
            bad += 1
            print("   ", "S", S, "synth seed", 100 + it, json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items() if k not in ("colours",)}))
            for pr in problems[:5]: print("   ", pr)
    print("bad:", bad, "of", count)
    sys.exit(1 if bad else 0)

main()
