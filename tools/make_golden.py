"""Generate tests/golden/*.npz from the CPU oracle (oracle/libsgz_oracle.so) in THIS container.

The reference ships no tests or golden vectors and cannot be built here, so these
fixtures pin the oracle itself (against silent regressions) and give the GPU tests a data-only target
that travels to the GPU box.  Exception: hsb_table.npz and colour_tables.npz come from the reference's own
juce::Colour::withRotatedHue, compiled from /root/reference (oracle/ref_juce_colour.cpp).  Inputs are re-synthesised from seeds (signalizer_amd.synth), only
parameters and expected outputs are stored.  Run:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from signalizer_amd import config, synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def main():
    po.build()
    # cfg1: one 4096-pt Hann frame (BASELINE configs[0])
    cfg = config.cfg1()
    x = synth.gen(1, 48000, 4096, 2)
    r = po.spectrogram(po.params_from_dict(cfg), x, want_lines=True, want_mapped=True)
    np.savez_compressed(os.path.join(OUT, "cfg1_frame.npz"), seed=1, rgba=r["rgba"],
                        lines=r["lines"], mapped=r["mapped"])
    # cfg2: first 8 frames of the 32768-pt job
    cfg = config.cfg2()
    S = 32768 + 7 * 8192
    x = synth.gen(2, 48000, S, 2)
    r = po.spectrogram(po.params_from_dict(cfg), x, want_lines=True)
    np.savez_compressed(os.path.join(OUT, "cfg2_8frames.npz"), seed=2, nsamples=S, rgba=r["rgba"],
                        lines_main=r["lines"][:, 0, 0, :])
    # Phase mode (TransformDSP.inl:643-853): 6 frames of 4096 points, Lanczos bins, both graphs' (magnitude, phase) lines
    cfgp = config.spectrum_config(window_size=4096, hop=1024, channel_mode=config.CH_PHASE, axis_points=256)
    S = 4096 + 5 * 1024
    r = po.spectrogram(po.params_from_dict(cfgp), synth.gen(15, 48000, S, 2), want_lines=True, want_mapped=True)
    np.savez_compressed(os.path.join(OUT, "phase_6frames.npz"), seed=15, nsamples=S, rgba=r["rgba"], lines=r["lines"],
                        mapped=r["mapped"][:, :, :256])
    # csf entries left complex (complex_dc.hpp): Left and Complex mode at the reference's default size, 48 kHz, a view from 10 Hz --
    # the first pixels' Lanczos windows wrap below bin 0 onto csf[N-1..] (mono) / include the complex csf[0] (Complex)
    for name, mode in (("left", config.CH_LEFT), ("complex", config.CH_COMPLEX)):
        cfgm = config.spectrum_config(window_size=4096, hop=1024, channel_mode=mode, axis_points=256)
        S = 4096 + 3 * 1024
        r = po.spectrogram(po.params_from_dict(cfgm), synth.gen(16, 48000, S, 2) + np.float32(0.2), want_mapped=True)
        np.savez_compressed(os.path.join(OUT, "mono_complex_entries_%s.npz" % name), seed=16, nsamples=S, mode=mode, rgba=r["rgba"],
                            mapped=r["mapped"][:, :, :256])
    # colour map cases (KA8): intensities around every branch, two-pair additive blend
    cfgc = config.spectrum_config(num_pairs=2, axis_points=16, ratios=(0.1, 0.3, 0.2, 0.25, 0.15))
    p = po.params_from_dict(cfgc)
    I = np.array([-1e-6, 0.0, 1e-7, 0.1, 0.0999, 0.25, 0.4, 0.5, 0.6, 0.85, 0.998, 0.999, 0.9991, 1.0, 1.7, -384.0], np.float32)
    frames = np.zeros((2, 16), np.complex64)
    frames[0].real = I
    frames[1].real = I[::-1]
    np.savez_compressed(os.path.join(OUT, "colormap_cases.npz"), intensity=I, rgba=po.blend_column(p, frames),
                        ratios=po.colour_ratios(cfgc["ratios"]), table0=po.colour_table(p, 0), table1=po.colour_table(p, 1))
    # HSB round trip (KA9): 4096 random colours x rotation amounts
    rng = np.random.default_rng(9)
    cols = rng.integers(0, 256, (4096, 3)).astype(np.uint8)
    cols[:8] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128], [0, 128, 255], [255, 64, 0]]
    amt = rng.choice(np.array([0.0, 0.25, 0.5, 1.0 / 3.0, 1.0 / 32.0, 0.75], np.float32), 4096).astype(np.float32)
    # ... from the REFERENCE ITSELF: juce::Colour::withRotatedHue compiled from /root/reference (oracle/ref_juce_colour.cpp, oracle/pyref.py).
    # The one fixture family in this directory that does not come from the oracle: it pins the oracle's and the product's HSB round trip.
    from oracle import pyref
    if not pyref.available():
        raise SystemExit("tools/make_golden.py needs /root/reference for the colour fixtures (oracle/_ref)")
    outc = np.stack([pyref.rotate_hue(cols[i], float(amt[i])) for i in range(4096)])
    np.savez_compressed(os.path.join(OUT, "hsb_table.npz"), rgb=cols, amount=amt, rotated=outc, source="juce::Colour::withRotatedHue, compiled reference")
    # generateSpectrogramColourRotation(p) for 1 .. 16 pairs through ColourRotation::operator[]'s float index / size
    # (TransformConstant.h:55-65, CommonSignalizer.h:931-937, Spectrum.cpp:398-402), the default colours and a random set
    sets = {"default": np.array(config.spectrum_config()["colours"], np.uint8), "random": rng.integers(0, 256, (6, 3)).astype(np.uint8)}
    tables = {}
    for name, base in sets.items():
        t = np.zeros((16, 16, 6, 3), np.uint8)                                   # [pairs - 1][rotation][stop][rgb]
        for pairs in range(1, 17):
            for rot in range(pairs):
                t[pairs - 1, rot] = pyref.spectrogram_colours(base, pairs, rot)
        tables["base_" + name] = base
        tables["table_" + name] = t
    np.savez_compressed(os.path.join(OUT, "colour_tables.npz"), **tables)
    # zero-crossing triggers (KA11)
    sig = synth.gen(3, 192000, 48000, 2)
    st = po.ZeroCrossingState(state=0.0, threshold=0.05, steady_clock=0, cross_origin=0, count=0, armed=0)
    trig = po.zero_crossing(st, 0, sig[0])
    np.savez_compressed(os.path.join(OUT, "trigger_cases.npz"), seed=3, threshold=0.05, triggers=trig)
    # Lanczos scope points (KA10) - subsampled
    v = po.ScopeView(window_size=19200.0, left=0.0, right=1.0, rendering_scale=8.0, width=19200)
    xx, yy = po.scope_lanczos(v, synth.gen(3, 192000, 19200, 2)[0])
    np.savez_compressed(os.path.join(OUT, "lanczos_cases.npz"), seed=3, npoints=xx.size, idx=np.arange(0, xx.size, 97),
                        x=xx[::97], y=yy[::97])
    # polar (KA12)
    L = np.array([1, 0, 1, 1, 0, 0.5, -0.25, 1e-8], np.float32)
    R = np.array([0, 1, 1, -1, 0, 0.25, 0.5, -1e-8], np.float32)
    np.savez_compressed(os.path.join(OUT, "polar_cases.npz"), L=L, R=R, xyz=po.vector_polar(L, R))
    print("golden fixtures written to", OUT, {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})


if __name__ == "__main__":
    main()
