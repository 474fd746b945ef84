import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from signalizer_amd import api
sr = 48000.0
t = np.arange(512 * 64) / sr
for name, f, thr in [("1 kHz tone", 1000.0, 0.05), ("10 kHz tone", 10000.0, 0.05), ("20 kHz tone", 20000.0, 0.05)]:
    x = np.stack([0.5 * np.sin(2 * np.pi * f * t), 0.5 * np.sin(2 * np.pi * f * t + 0.3)]).astype(np.float32)
    h = api.Scope(sample_rate=sr, window_size=4800.0, num_channels=2, channel_mode=0, envelope_mode=0, interpolation=3, max_block=512,
                  trigger_threshold=thr, trigger_channel=1.0, envelope_window=0.3, trigger_mode=4)
    def run(n):
        done = 0
        while done < n:
            b = (done % 64) * 512
            if h.push(x[:, b:b + 512]) == api.SGZ_OK:
                done += 1
        h.state()
    run(100)
    s0 = h.state()['swaps']
    t0 = time.perf_counter(); run(1000); dt = time.perf_counter() - t0
    print(f"{name:14s} {dt / 1000 * 1e6:8.1f} us per 512-sample callback, {(h.state()['swaps'] - s0) / 1000:.1f} swaps per callback")
    h.close()
