"""Bit-identity under concurrency for every kernel family: whole renders (K_A + K_B) of three fixed buffers over four plans / streams,
against the quiet run, for a list of configurations that together reach every K_A / K_B form.
usage: [LOAD=n] overlap_stress_cfgs.py [rounds]        LOAD=n: n other PROCESSES render beside it as well (tools/gpu_load.py; the quiet
references of every case are taken before they start) -- the regime in which the RSNT carried-state race of round 6 showed"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, config as cf, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
gpu = torch.device("cuda", 0)
CASES = {
    "real N=32768 separate (bench)": dict(),
    "real N=16384 midside": dict(window_size=16384, hop=4096, channel_mode=cf.CH_MIDSIDE),
    "real N=65536 two pairs": dict(window_size=65536, hop=16384, num_pairs=2, sample_rate=96000.0),
    "real mono merge N=32768": dict(channel_mode=cf.CH_MERGE),
    "wide groups N=32768": dict(_wide=1),
    "whole-frame complex N=32768": dict(channel_mode=cf.CH_COMPLEX),
    "whole-frame N=4096 zero-padded": dict(window_size=3000, hop=750),
    "halves N=8192": dict(window_size=8192, hop=2048),
    "generic N=2048": dict(window_size=2048, hop=512),
    "phase N=4096": dict(window_size=4096, hop=1024, channel_mode=cf.CH_PHASE),
    "phase N=32768": dict(channel_mode=cf.CH_PHASE),
    "rsnt hop 1024 (matrix cores)": dict(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1024),
    "rsnt hop 1000 (vector form)": dict(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1000),
    "three pairs N=4096 (scan/emit K_B)": dict(window_size=4096, hop=1024, num_pairs=3),
    "fetch window blackman N=32768": dict(window_type=cf.WIN_BLACKMAN),
    "real N=65536 mono merge": dict(window_size=65536, hop=16384, channel_mode=cf.CH_MERGE, sample_rate=96000.0),
    "real N=65536 midside": dict(window_size=65536, hop=16384, channel_mode=cf.CH_MIDSIDE, sample_rate=96000.0),
    "real N=65536 fetched window (blackman)": dict(window_size=65536, hop=16384, window_type=cf.WIN_BLACKMAN, sample_rate=96000.0),
    "rsnt segmented chain (fuzz 1005 / 12)": dict(_fuzz=(1005, 12)),
    "rsnt matrix worst case (fuzz 2008 / 52)": dict(_fuzz=(2008, 52)),
}
load = int(os.environ.get("LOAD", "0"))
NS = int(os.environ.get("STREAMS", "4"))                    # plans / streams the launches rotate over (1: one launch at a time)
STAGE = os.environ.get("STAGE", "render")                  # STAGE=mapped: K_A alone (sgz_stage_mapped), compared as bit patterns
if os.environ.get("CASES"):                                # CASES=substring,substring: only those
    CASES = {k: v for k, v in CASES.items() if any(w in k for w in os.environ["CASES"].split(","))}
prepared = []
for name, over in CASES.items():
    over = dict(over)
    wide = over.pop("_wide", 0)
    fuzz = over.pop("_fuzz", None)
    if fuzz:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        import fuzzcfg
        cfg, _, x = fuzzcfg.rsnt_case(*fuzz)
        xs = [torch.from_numpy(x).to(gpu)] * 3
    else:
        cfg = cf.spectrum_config(**over)
        frames = 120 if cfg["window_size"] >= 16384 else 200
        S = cfg["window_size"] + cfg["hop"] * (frames - 1)
        xs = [torch.from_numpy(synth.gen(500 + k, int(cfg["sample_rate"]), S, 2 * cfg["num_pairs"])).to(gpu) for k in range(3)]
    def mk(cfg=cfg, wide=wide):
        p = api.Plan(cfg)
        if wide: p.set_option(api.OPT_WIDE_GROUPS, 1)
        return p.upload()
    ref = mk()
    want = [(ref.render(x) if STAGE == "render" else ref.stage_mapped(x).view(torch.int32)).clone() for x in xs]
    torch.cuda.synchronize()
    prepared.append((name, cfg, xs, mk, ref.path, want))
procs = []
if load:
    here = os.path.dirname(os.path.abspath(__file__))
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "gpu_load.py"), "1200", "rsnt" if k % 2 else "spectrum"], stdout=subprocess.PIPE, text=True) for k in range(load)]
    for p in procs:
        assert p.stdout.readline().strip() == "READY"
    print(f"{load} load processes render beside every case", flush=True)
total_bad = 0
try:
    for name, cfg, xs, mk, path, want in prepared:
        plans = [mk() for _ in range(NS)]
        streams = [torch.cuda.Stream(device=gpu) for _ in range(NS)]
        bad = 0
        n = rounds if cfg["algorithm"] == 0 else max(20, rounds // 4)
        for r in range(n):
            outs = []
            torch.cuda.synchronize()
            for k in range(9):
                if STAGE == "render":
                    outs.append(plans[k % NS].render(xs[k % 3], stream=streams[k % NS].cuda_stream))
                else:
                    with torch.cuda.stream(streams[k % NS]):
                        outs.append(plans[k % NS].stage_mapped(xs[k % 3]).view(torch.int32))
            torch.cuda.synchronize()
            for k in range(9):
                if not torch.equal(outs[k], want[k % 3]):
                    bad += 1
                    if STAGE != "render":
                        g, w = outs[k].view(torch.float32), want[k % 3].view(torch.float32)
                        idx = torch.nonzero(outs[k] != want[k % 3]).cpu().numpy()           # [n][frame, pair, side, pixel]
                        rel = ((g - w).abs() / w.abs().clamp_min(1e-30))[outs[k] != want[k % 3]].cpu().numpy()
                        units = sorted({(int(a), int(b), int(c)) for a, b, c, _ in idx})
                        print(f"    round {r} launch {k} (plan {k % NS}, input {k % 3}): {len(idx)} magnitudes differ in (frame, pair, side) {units[:6]}{' ...' if len(units) > 6 else ''}, "
                              f"pixels {idx[:, 3].min()}..{idx[:, 3].max()}, relative difference median {np.median(rel):.2e} max {rel.max():.2e}", flush=True)
                        continue
                    d = (outs[k].to(torch.int16) - want[k % 3].to(torch.int16)).abs().amax(dim=2).cpu().numpy()      # [frame][pixel]
                    fr = np.nonzero(d.max(axis=1))[0]
                    px = np.nonzero(d.max(axis=0))[0]
                    print(f"    round {r} render {k} (plan {k % NS}, input {k % 3}): frames {fr.min()}..{fr.max()} ({len(fr)}), pixels {px.min()}..{px.max()} ({len(px)}), "
                          f"largest byte difference {d.max()}; first frame's differing pixels {np.nonzero(d[fr.min()])[0][:12].tolist()}", flush=True)
        total_bad += bad
        print(f"{name:40s}: {bad} of {n * 9} renders differ (path {path})", flush=True)
    assert all(p.poll() is None for p in procs), "a load process ended early"
finally:
    for p in procs:
        p.kill()
        p.wait()
print("total differing:", total_bad)
