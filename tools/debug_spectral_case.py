"""one case of tools/fuzz_spectral.py, frame by frame: the device's TriggerState next to the oracle's, and the oracle's median ring.
usage: debug_spectral_case.py <seed> <case>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from signalizer_amd import api
from oracle import pyoracle as po

seed0, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0 * 100000 + case)
sr = float(rng.choice([44100.0, 48000.0, 96000.0, 192000.0]))
window = float(np.round(rng.uniform(200, 30000), int(rng.integers(0, 2))))
f0 = float(np.exp(rng.uniform(np.log(30.0), np.log(5000.0))))
hyst = float(rng.choice([0.0, 0.1, 0.5, 0.9]))
thr = float(rng.choice([0.0, 0.02, 0.3]))
evaluator = int(rng.integers(0, 4))
em = {0: 0, 1: 0, 2: 1, 3: 2}[evaluator]
nh = int(rng.integers(1, 6))
amps = rng.uniform(0.05, 1.0, nh)
n = int(sr * 1.5)
t = np.arange(n) / sr
x = sum(a * np.sin(2 * np.pi * f0 * (k + 1) * t + rng.uniform(0, 6.28)) for k, a in enumerate(amps))
noise = float(rng.choice([0.0, 0.01, 0.2]))
xs = np.stack([x + noise * rng.standard_normal(n), 0.5 * x * (1 if rng.random() < 0.5 else -1) + noise * rng.standard_normal(n)]).astype(np.float32)
cfg = dict(sample_rate=sr, window_size=window, num_channels=2, trigger_mode=1, channel_mode=0, envelope_mode=0, interpolation=3, max_block=4096,
           trigger_threshold=thr, trigger_channel=1.0, envelope_window=0.3, trigger_hysteresis=hyst, trigger_phase_offset=float(rng.uniform(-180, 180)))
print(f"sr {sr} W {window} f0 {f0:.3f} amps {np.round(amps, 3)} hyst {hyst} thr {thr} ev {evaluator} noise {noise}")
dev = api.Scope(**cfg)
ref = po.ScopeStream(2, sr, window, 1, thr, 0, 1.0, 0, 0.3)
ts = po.SpectralState()
sz = max(int(0.5 + np.ceil(window)), 8192)
pos = 0
for frame in range(14):
    for _ in range(int(rng.integers(1, 6))):
        m = int(rng.integers(1, 4096))
        blk = xs[:, pos:pos + m]
        if blk.shape[1] == 0:
            break
        while dev.push(blk) == api.SGZ_BUSY:
            pass
        ref.audio(blk); pos += blk.shape[1]
    mem = [ref.logical(c, sz) for c in (0, 1)]
    a, b = (mem[1], mem[1]) if evaluator == 1 else (mem[0], mem[1]) if em else (mem[0], mem[0])
    # the oracle on the DEVICE's ring memory (front buffers + cursor), from the same filter state: separates the ring from the analysis
    tsd = po.SpectralState.from_buffer_copy(ts)
    fm = [dev.front(c) for c in (0, 1)]
    da, db = (fm[1][0], fm[1][0]) if evaluator == 1 else (fm[0][0], fm[1][0]) if em else (fm[0][0], fm[0][0])
    po.scope_analyse(tsd, da, db, em, fm[0][1], window, sr, thr, hyst, cfg["trigger_phase_offset"])
    po.scope_analyse(ts, a, b, em, 0, window, sr, thr, hyst, cfg["trigger_phase_offset"])
    got = dev.analyse(evaluator, 0)
    newest = (ts.median_pos - 1) % 8
    print(f"   frame winner: oracle/logical ({ts.median[newest].index}, {ts.median[newest].value:.6g})  oracle/device-ring ({tsd.median[newest].index}, {tsd.median[newest].value:.6g})"
          f"  device ring len {fm[0][0].size} cursor {fm[0][1]}; logical len {a.size}; rings equal: {np.array_equal(np.roll(da, -fm[0][1]), a) if da.size == a.size else 'size differs'}")
    ring = [(int(r.index), round(r.value, 3), round(r.offset, 6)) for r in ts.median]
    print(f"frame {frame} pos {pos} sz {sz}: dev rec ({got.record_index}, {got.record_value:.6g}, {got.record_offset:.6g}) f {got.fundamental:.6f} ring {got.ring_size}"
          f" | ora rec ({ts.record.index}, {ts.record.value:.6g}, {ts.record.offset:.6g}) f {ts.fundamental:.6f} medpos {ts.median_pos}")
    print("     oracle median ring:", ring)
    if hasattr(api.lib(), "sgz_scope_debug_median"):
        import ctypes as C
        buf = (C.c_double * 24)()
        api.lib().sgz_scope_debug_median(dev.h, buf)
        print("     device median ring:", [(int(buf[3 * i]), round(buf[3 * i + 1], 3), round(buf[3 * i + 2], 6)) for i in range(8)])
    sz = max(int(0.5 + ts.cycle_samples + np.ceil(window)), 8192)
dev.close()
