"""Phase clocks of scopeIngestKernel (debug build, -DSGZ_DEBUG): wall-clock (100 MHz) stamps of one lane at the phase boundaries,
averaged over the blocks of a few cfg3 frames.  usage: SGZ_LIB=<debug build> ingest_clocks.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from signalizer_amd import api, synth
L = api.lib()
sr, W, nch = 192000.0, 19200, 2
extra = {}
if os.environ.get("SGZ_CLOCKS_COLOURS") == "1":                    # phase F as well
    extra = dict(colour_by_frequency=1, frequency_colouring_blend=0.8, colour_smoothing_ms=4.0, band_colours=[(1.0, 0.25, 0.1), (0.2, 1.0, 0.3), (0.15, 0.35, 1.0)])
if os.environ.get("SGZ_CLOCKS_MODE"):
    extra["trigger_mode"] = int(os.environ["SGZ_CLOCKS_MODE"])
cfgd = dict(sample_rate=sr, window_size=float(W), num_channels=nch, trigger_mode=4, channel_mode=0, envelope_mode=2,
            interpolation=3, max_block=512, trigger_threshold=0.05, trigger_channel=1.0, envelope_window=0.3)
cfgd.update(extra)
h = api.Scope(**cfgd)
x = synth.gen(31, int(sr), 3200 * 16, nch)
names = ["state in LDS -> A done (zero crossings)", "B done (automaton)", "(colours)", "swap list read", "swaps copied", "back ring", "envelope", "state written back"]
acc = np.zeros(7); cnt = 0
out = (C.c_ulonglong * 8)()
for pos in range(0, x.shape[1], 512):
    while h.push(x[:, pos:pos + 512]) == api.SGZ_BUSY: pass
    api.check(L.sgz_debug_ingest_clocks(out))
    t = np.array(list(out), dtype=np.int64)
    if pos >= 512 * 20:
        acc += np.diff(t) * 0.01; cnt += 1
for n, v in zip(names[0:], acc / cnt): print(f"{n:45s} {v:6.2f} us")
print(f"{'total':45s} {acc.sum() / cnt:6.2f} us over {cnt} blocks")
