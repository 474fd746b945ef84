"""Special input signals through the spectrum path against the oracle: silence, DC, impulses, full-scale square, denormal-level,
very large, and (reported only) NaN / Inf.   usage: fuzz_inputs.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth
from oracle import pyoracle as po

def signals(S, nch, rng):
    base = synth.gen(5, 48000, S, nch)
    out = {"silence": np.zeros((nch, S), np.float32), "dc": np.full((nch, S), 0.5, np.float32)}
    imp = np.zeros((nch, S), np.float32); imp[:, S // 3] = 1.0; imp[1, S // 2] = -1.0; out["impulses"] = imp
    out["square"] = np.sign(np.sin(np.arange(S) * 0.01))[None, :].repeat(nch, 0).astype(np.float32)
    out["denormal"] = (base * 1e-38).astype(np.float32)
    out["tiny"] = (base * 1e-20).astype(np.float32)
    out["huge"] = (base * 1e15).astype(np.float32)
    out["half_silent"] = base.copy(); out["half_silent"][:, S // 2:] = 0
    out["one_channel"] = base.copy(); out["one_channel"][1::2] = 0
    nan = base.copy(); nan[0, S // 2] = np.nan; out["nan (report only)"] = nan
    inf = base.copy(); inf[0, S // 2] = np.inf; out["inf (report only)"] = inf
    return out

def main():
    po.build()
    rng = np.random.default_rng(1)
    bad = 0
    for cfg in (config.spectrum_config(window_size=4096, hop=1024, axis_points=300),
                config.spectrum_config(window_size=8192, hop=2048, axis_points=300, channel_mode=config.CH_MERGE),
                config.spectrum_config(window_size=2048, hop=512, axis_points=300, channel_mode=config.CH_MIDSIDE, bin_interp=config.INTERP_LINEAR),
                config.spectrum_config(window_size=4096, hop=1024, axis_points=300, channel_mode=config.CH_PHASE),
                config.spectrum_config(window_size=32768, hop=8192, axis_points=300, view_scaling=config.VIEW_LINEAR)):
        W, hop = cfg["window_size"], cfg["hop"]
        S = W + 9 * hop
        plan = api.Plan(cfg).upload()
        for name, x in signals(S, 2, rng).items():
            ref = po.spectrogram(po.params_from_dict(cfg), x)["rgba"]
            got = plan.render(torch.from_numpy(x).cuda()).cpu().numpy()
            d = np.abs(got.astype(int) - ref.astype(int))
            phase = cfg["channel_mode"] == config.CH_PHASE
            ok = d.max() <= (2 if phase else 1) and (d > 0).mean() <= (2e-2 if phase else 5e-3)
            report = "report only" in name
            print("ok " if ok else ("DIFF" if report else "BAD"), "N", plan.N, "mode", cfg["channel_mode"], name, "max", int(d.max()), "frac", float((d > 0).mean()))
            bad += 0 if ok or report else 1
    print("bad:", bad)
    sys.exit(1 if bad else 0)

main()
