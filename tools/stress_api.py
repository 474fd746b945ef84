"""API usage stress: several plans alive at once (every K_A path), renders of varying length interleaved on a non-default stream,
stage hooks mixed with renders on the same plan, carried state -- every result must equal that of a fresh plan used once."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config, synth

def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    cfgs = [config.spectrum_config(window_size=4096, hop=1024, num_pairs=2),
            config.spectrum_config(window_size=8192, hop=2048, channel_mode=config.CH_MERGE),
            config.spectrum_config(window_size=32768, hop=8192),
            config.spectrum_config(window_size=65536, hop=16384, num_pairs=2, sample_rate=96000.0),
            config.spectrum_config(window_size=2048, hop=512, channel_mode=config.CH_PHASE),
            config.spectrum_config(window_size=16384, hop=4096, channel_mode=config.CH_COMPLEX),
            config.spectrum_config(window_size=3000, hop=750, axis_points=3000)]
    plans = [api.Plan(c).upload() for c in cfgs]
    side = torch.cuda.Stream()
    bad = 0
    for it in range(60):
        i = int(rng.integers(0, len(cfgs)))
        cfg, plan = cfgs[i], plans[i]
        W, hop = cfg["window_size"], cfg["hop"]
        frames = int(rng.choice([1, 2, 9, 40, 130, int(rng.integers(1, 300))]))
        if W >= 32768: frames = min(frames, 60)
        x = torch.from_numpy(synth.gen(2000 + it, cfg["sample_rate"], W + (frames - 1) * hop, 2 * cfg["num_pairs"])).cuda()
        what = int(rng.integers(0, 4))
        with torch.cuda.stream(side if it % 2 else torch.cuda.current_stream()):
            if what == 0:
                got = plan.render(x)
                torch.cuda.current_stream().synchronize()
                ref = api.Plan(cfg).upload().render(x)
            elif what == 1:
                got = plan.stage_mapped(x)
                torch.cuda.current_stream().synchronize()
                ref = api.Plan(cfg).upload().stage_mapped(x)
            elif what == 2:
                got = plan.stage_bins(x)
                torch.cuda.current_stream().synchronize()
                ref = api.Plan(cfg).upload().stage_bins(x)
            else:
                st1 = torch.rand((cfg["num_pairs"], 2, cfg["axis_points"], 2), device="cuda") * 1e-3
                st2 = st1.clone()
                got = plan.render(x, state=st1)
                torch.cuda.current_stream().synchronize()
                ref = api.Plan(cfg).upload().render(x, state=st2)
                if not torch.equal(st1, st2): bad += 1; print(it, "BAD state")
            torch.cuda.synchronize()
        ok = torch.equal(got, ref)
        print(it, "ok " if ok else "BAD", "N", plan.N, "frames", frames, "what", what)
        bad += 0 if ok else 1
    print("bad:", bad)
    sys.exit(1 if bad else 0)

main()
