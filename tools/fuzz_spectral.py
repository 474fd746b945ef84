"""random sweep of sgz_scope_analyse (Spectral triggering) against the oracle: sample rates, windows, fundamentals, harmonic mixes, noise,
hysteresis, evaluators, block schedules.  usage: fuzz_spectral.py [cases] [seed]

A case that reports a different winning BIN is not automatically a disagreement: the candidate test compares omega(current) / omega(max)
with the nearest integer at a quarter-semitone bar (OscilloscopeDSP.inl:155-175), and while max is still the seeded bin 1 its omega is
1 + quadDelta(1) -- a ratio of differences of leakage-level bins 0, 1, 2 that can come out near -1 (omega 0.06).  The ratio is then
amplified 16 x 2767-fold and the two transforms' rounding (1e-9 of the offset) decides the test: seed 77 case 72 is such a frame
(tools/debug_spectral_case.py 77 72 prints both median rings: the device keeps (1, 409.6, -0.937457) where the oracle takes bin 2767; the
oracle run on the device's ring memory takes 2767 too).  Conditioning of the reference's own rule, 1 case in 300 of this sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from signalizer_amd import api
from oracle import pyoracle as po

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for case in range(cases):
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
    dev = api.Scope(**cfg)
    ref = po.ScopeStream(2, sr, window, 1, thr, 0, 1.0, 0, 0.3)
    ts = po.SpectralState()
    sz = max(int(0.5 + np.ceil(window)), 8192)
    pos = 0
    problems = []
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
        po.scope_analyse(ts, a, b, em, 0, window, sr, thr, hyst, cfg["trigger_phase_offset"])
        got = dev.analyse(evaluator, 0)
        if got.record_index != ts.record.index:
            problems.append(f"frame {frame}: bin {got.record_index} vs {ts.record.index} (values {got.record_value:.6g} / {ts.record.value:.6g})")
            break                                                   # the median filters have diverged: later frames mean nothing
        tau = 2 * np.pi
        dph = abs(((got.phase - ts.phase) + np.pi) % tau - np.pi)
        # the winning bin's fractional offset is a ratio of differences of neighbouring bins: in a noise-free harmonic signal the winner can
        # sit in a spectral null (leakage only, ~1e-7 of the peak), where the two transforms' rounding (1e-16 of the peak) moves the offset
        # by 1e-9 and the phase by 1e-6 -- conditioning, not a disagreement; the bars here are a decade above that
        if abs(got.fundamental - ts.fundamental) > 1e-6 * ts.fundamental or dph > 1e-4 or got.ring_size != max(int(0.5 + ts.cycle_samples + np.ceil(window)), 8192):
            problems.append(f"frame {frame}: fundamental {got.fundamental!r} vs {ts.fundamental!r}, phase {got.phase!r} vs {ts.phase!r}")
        sz = max(int(0.5 + ts.cycle_samples + np.ceil(window)), 8192)
    dev.close()
    if problems:
        bad += 1
        print(f"case {case}: sr {sr} W {window} f0 {f0:.2f} hyst {hyst} thr {thr} ev {evaluator} noise {noise}: " + "; ".join(problems[:3]))
print(f"{cases - bad} of {cases} cases clean")
