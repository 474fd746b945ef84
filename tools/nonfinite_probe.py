"""NaN / +-Inf / near-overflow samples through every entry point of the library: a plug-in host CAN hand such blocks over (a misbehaving
plug-in in front of the analyser), so the one thing that must never happen is a device fault -- an out-of-range table index made from a
NaN magnitude would take the whole process down.  Checked per case:
  * every call returns, the device is alive afterwards (a closing synchronize);
  * frames in front of the first contaminated frame are bit-identical to the render of the same audio with the bad samples zeroed
    (nothing non-finite leaks backwards or sideways);
  * a render of clean audio on the SAME plan / handle-less path afterwards is bit-identical to the clean reference (nothing sticks to
    the plan's scratch);
  * the real-time handles (spectrum stream, Oscilloscope, Vectorscope) take the blocks, answer every read call, and a NEW handle
    afterwards behaves like one on a fresh process (its outputs equal a handle that never saw the bad blocks).
What the contaminated frames themselves show is the reference's C semantics on NaN (comparisons are false, so NaN magnitudes never
enter the peak decay) as far as the oracle goes; they are reported, not asserted: tools/fuzz_inputs.py.
Runs in a process of its own under tests/test_gpu_nonfinite.py (a fault must fail one test, not end the suite).
usage: nonfinite_probe.py [quick]    prints one line per case and `problems: N` (quick: NaN and +Inf only -- the suite's form)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from signalizer_amd import api, config as cf, synth

gpu = torch.device("cuda", 0)
problems = 0
BAD = {"nan": np.nan, "+inf": np.inf, "-inf": -np.inf, "3e38": 3e38}
if "quick" in sys.argv[1:]:
    BAD = {"nan": np.nan, "+inf": np.inf}


def report(ok, what):
    global problems
    problems += 0 if ok else 1
    print(("ok   " if ok else "BAD  ") + what, flush=True)


def spectrum_cases():
    return {
        "real N=32768 (bench kernel)": cf.cfg2(),
        "real N=65536 walking": cf.spectrum_config(window_size=65536, hop=16384, sample_rate=96000.0),
        "real N=16384 mid/side": cf.spectrum_config(window_size=16384, hop=4096, channel_mode=cf.CH_MIDSIDE),
        "whole-frame complex N=4096": cf.spectrum_config(window_size=4096, hop=1024, channel_mode=cf.CH_COMPLEX),
        "halves N=8192 merge": cf.spectrum_config(window_size=8192, hop=2048, channel_mode=cf.CH_MERGE),
        "generic N=2048 linear interpolation": cf.spectrum_config(window_size=2048, hop=512, bin_interp=cf.INTERP_LINEAR),
        "phase N=4096": cf.spectrum_config(window_size=4096, hop=1024, channel_mode=cf.CH_PHASE),
        "three pairs N=4096": cf.spectrum_config(window_size=4096, hop=1024, num_pairs=3),
        "rsnt hop 1024 (matrix cores)": cf.spectrum_config(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1024),
        "rsnt hop 1000 (vector form)": cf.spectrum_config(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1000),
    }


def spectrum():
    for name, cfg in spectrum_cases().items():
        W, hop = cfg["window_size"], cfg["hop"]
        frames = 40
        S = W + hop * (frames - 1)
        nch = 2 * cfg["num_pairs"]
        clean = synth.gen(31, int(cfg["sample_rate"]), S, nch)
        plan = api.Plan(cfg).upload()
        ref_clean = plan.render(torch.from_numpy(clean).to(gpu)).clone()
        lines_plan = api.Plan(cfg).upload()
        for label, v in BAD.items():
            at = W + hop * 20 + 17                                   # first frame whose window holds it: frame 21 - W/hop ... -> frames < first are clean
            first = max(0, (at - W) // hop + 1)
            x = clean.copy()
            x[0, at] = v
            x[nch - 1, at + 3 * hop] = -v
            zeroed = x.copy()
            zeroed[~np.isfinite(zeroed)] = 0
            zeroed[np.abs(zeroed) > 1e30] = 0
            want = plan.render(torch.from_numpy(zeroed).to(gpu)).clone()
            got = plan.render(torch.from_numpy(x).to(gpu)).clone()
            torch.cuda.synchronize()
            # an RSNT bank has memory longer than one window: what is in front of the bad sample is still clean, nothing behind it is asserted
            ok = torch.equal(got[:first], want[:first])
            # with the line results and a carried state (the K_B forms a streaming caller runs)
            F = plan.num_frames(S)
            state = torch.zeros((cfg["num_pairs"], 2, cfg["axis_points"], 2), dtype=torch.float32, device=gpu)
            lines = torch.empty((F, cfg["num_pairs"], 2, cfg["axis_points"], 2), dtype=torch.float32, device=gpu)
            lines_plan.render(torch.from_numpy(x).to(gpu), lines=lines, state=state)
            torch.cuda.synchronize()
            again = plan.render(torch.from_numpy(clean).to(gpu))
            torch.cuda.synchronize()
            ok2 = torch.equal(again, ref_clean)
            report(ok and ok2, f"spectrum  {name:38s} {label:5s}: frames < {first} {'equal' if ok else 'DIFFER'}, clean render afterwards {'equal' if ok2 else 'DIFFERS'}")


def spectrum_stream():
    L = api.lib()
    for name, cfg in (("stream N=4096 separate", cf.spectrum_config(window_size=4096, hop=1024, axis_points=300)),
                      ("stream N=4096 phase", cf.spectrum_config(window_size=4096, hop=1024, axis_points=300, channel_mode=cf.CH_PHASE)),
                      ("stream rsnt", cf.spectrum_config(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1024, axis_points=300))):
        x = synth.gen(8, 48000, 24 * 512, 2)

        def run(sig):
            c = api.config_from_dict(cfg)
            h = C.c_void_p()
            api.check(L.sgz_spectrum_create(C.byref(c), C.byref(h)))
            cols = []
            buf = np.zeros((300, 4), np.uint8)
            ap = C.c_uint32(0)
            try:
                for b in range(sig.shape[1] // 512):
                    blk = np.ascontiguousarray(sig[:, b * 512:(b + 1) * 512])
                    ptrs = (C.c_void_p * 2)(blk[0].ctypes.data, blk[1].ctypes.data)
                    st = L.sgz_spectrum_push(h, ptrs, 2, 512)
                    assert st in (api.SGZ_OK, api.SGZ_BUSY), st
                L.sgz_spectrum_flush.argtypes = [C.c_void_p]
                api.check(L.sgz_spectrum_flush(h))
                import time
                t0 = time.time()
                while time.time() - t0 < 5.0 and len(cols) < sig.shape[1] // 1024:
                    st = L.sgz_spectrum_pop_column(h, buf.ctypes.data_as(C.c_void_p), C.byref(ap))
                    if st == api.SGZ_OK:
                        cols.append(buf.copy())
                    else:
                        time.sleep(0.001)
            finally:
                L.sgz_spectrum_destroy(h)
            return np.stack(cols) if cols else np.zeros((0, 300, 4), np.uint8)
        want = run(x)
        for label, v in BAD.items():
            y = x.copy()
            y[0, 7000] = v
            y[1, 9000] = v
            got = run(y)
            after = run(x)                                             # a new handle afterwards
            ok = len(got) == len(want) and np.array_equal(got[:2], want[:2]) and np.array_equal(after, want)
            report(ok, f"{name:48s} {label:5s}: {len(got)} columns, the first two equal, a new handle afterwards equals the first run")


def scope():
    SR = 192000.0
    base = dict(sample_rate=SR, window_size=19200.0, num_channels=2, trigger_mode=4, channel_mode=0, envelope_mode=1, interpolation=3,
                max_block=4096, trigger_threshold=0.05, trigger_channel=1.0, envelope_window=0.3)
    t = np.arange(40000) / SR
    x = np.stack([0.6 * np.sin(2 * np.pi * 441.7 * t), 0.5 * np.sin(2 * np.pi * 577.0 * t + 0.3)]).astype(np.float32)
    view = api.ScopeView(19200.0, 0.0, 1.0, 1.0, 1200, 0)

    def run(sig, **over):
        cfg = dict(base)
        cfg.update(over)
        dev = api.Scope(**cfg)
        try:
            for p in range(0, sig.shape[1], 512):
                while dev.push(sig[:, p:p + 512]) != api.SGZ_OK:
                    pass
            st = dev.state()
            g = dev.peak_filter(1.0 / 60.0)
            xyz, rgba = dev.vertices(view, 0, 0)
            ts = dev.analyse(0, 0)
            return st, g, xyz.copy(), rgba.copy()
        finally:
            dev.close()
    for over_name, over in (("zero-crossing, Lanczos, RMS envelope", dict()), ("spectral trigger", dict(trigger_mode=1)),
                            ("envelope trigger, linear", dict(trigger_mode=3, interpolation=1)), ("peak envelope", dict(envelope_mode=2))):
        try:
            want = run(x, **over)
        except api.SgzError as e:
            report(True, f"scope     {over_name:38s}: configuration refused ({e}) -- skipped")
            continue
        for label, v in BAD.items():
            y = x.copy()
            y[0, 20000] = v
            y[1, 20500:20600] = v
            run(y, **over)
            after = run(x, **over)
            ok = after[0] == want[0] and np.array_equal(after[2].view(np.uint32), want[2].view(np.uint32)) and np.array_equal(after[3], want[3])
            report(ok, f"scope     {over_name:38s} {label:5s}: every call returned, a new handle afterwards equals the first run")


def vector():
    x = synth.gen(4, 96000, 30000, 8)

    def run(sig, env_mode):
        dev = api.Vector(sample_rate=96000.0, num_channels=8, window_size=9600, envelope_mode=env_mode, lanes=8, fade_history=1,
                         max_block=4096, envelope_window=0.3, stereo_window=0.05)
        try:
            for p in range(0, sig.shape[1], 512):
                while dev.push(sig[:, p:p + 512]) != api.SGZ_OK:
                    pass
            f, gain = dev.filters()
            g = dev.peak_filter(1.0 / 60.0)
            xyz, rgb = dev.vertices_all()
            return [xyz.copy(), rgb.copy()]
        finally:
            dev.close()
    for env_mode in (0, 1, 2):
        want = run(x, env_mode)
        for label, v in BAD.items():
            y = x.copy()
            y[0, 15000] = v
            y[3, 15100:15200] = v
            run(y, env_mode)
            after = run(x, env_mode)
            ok = all(np.array_equal(a.view(np.uint32), w.view(np.uint32)) for a, w in zip(after, want))
            report(ok, f"vector    envelope mode {env_mode}                        {label:5s}: every call returned, a new handle afterwards equals the first run")


for part in (spectrum, spectrum_stream, scope, vector):
    part()
    torch.cuda.synchronize()
print("problems:", problems)
sys.exit(1 if problems else 0)
