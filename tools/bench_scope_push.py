"""us per sgz_scope_push (512-sample stereo blocks, 192 kHz, 19 200-sample window) by trigger mode / colouring: GPU time per block is
the loop time once the staging ring is saturated (push returns SGZ_BUSY until a slot frees)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from signalizer_amd import api, synth
x = synth.gen(3, 192000, 512 * 64, 2)
BANDS = [(1.0, 0.25, 0.1), (0.2, 1.0, 0.3), (0.15, 0.35, 1.0)]
for name, kw in [("None", dict(trigger_mode=0)), ("ZeroCrossing", dict(trigger_mode=4)), ("EnvelopeHold", dict(trigger_mode=3)), ("Spectral", dict(trigger_mode=1)),
                 ("ZeroCrossing + colours", dict(trigger_mode=4, colour_by_frequency=1, frequency_colouring_blend=0.8, colour_smoothing_ms=4.0, band_colours=BANDS)),
                 ("ZeroCrossing + RMS", dict(trigger_mode=4, envelope_mode=1)), ("None + RMS", dict(trigger_mode=0, envelope_mode=1)),
                 ("None + RMS, Separate", dict(trigger_mode=0, envelope_mode=1, channel_mode=4)),
                 ("None + colours", dict(trigger_mode=0, colour_by_frequency=1, frequency_colouring_blend=0.8, colour_smoothing_ms=4.0, band_colours=BANDS)), ("ZeroCrossing, W=1000", dict(trigger_mode=4, window_size=1000.0))]:
    cfg = dict(sample_rate=192000.0, window_size=19200.0, num_channels=2, channel_mode=0, envelope_mode=0, interpolation=3, max_block=512,
               trigger_threshold=0.05, trigger_channel=1.0, envelope_window=0.3)
    cfg.update(kw)
    h = api.Scope(**cfg)
    def run(n):
        done = 0
        while done < n:
            b = (done % 64) * 512
            if h.push(x[:, b:b + 512]) == api.SGZ_OK:
                done += 1
        h.state()                      # waits for the stream
    run(200)
    t0 = time.perf_counter(); run(2000); dt = time.perf_counter() - t0
    print(f"{name:28s} {dt / 2000 * 1e6:8.1f} us per 512-sample block   swaps {h.state()['swaps']}")
    h.close()
