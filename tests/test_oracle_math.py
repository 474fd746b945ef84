"""CPU tests that pin the oracle against independent mathematics and closed-form known answers
 -- the reference ships no vectors, so this is what "pinned" can mean here."""
import ctypes as C

import numpy as np
import pytest

from signalizer_amd import config


def test_fft_vs_numpy_fp64(oracle):
    po = oracle
    rng = np.random.default_rng(0)
    for N in (32, 64, 1024, 4096, 32768, 65536):
        x = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
        ref = np.fft.fft(x.astype(np.complex128))
        assert np.abs(po.fft32(x) - ref).max() <= 4e-7 * np.abs(ref).max()
        assert np.abs(po.fft64(x) - ref).max() <= 1e-14 * np.abs(ref).max()


def test_simd_baseline_transform_vs_numpy_and_the_restated_one(oracle):
    """oracle/fft_simd.c (bench.py's cpu_baseline.simd_value only -- timing, not parity) computes the same forward, unnormalised,
    natural-order transform as the restated radix-2, and a whole render on it stays within a byte of the port's."""
    po = oracle
    rng = np.random.default_rng(1)
    for N in (2, 4, 8, 32, 64, 1024, 4096, 32768, 65536):
        x = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
        ref = np.fft.fft(x.astype(np.complex128))
        buf = x.copy()
        assert po.lib().sgzo_fft_forward_simd(buf.ctypes.data_as(C.c_void_p), C.c_uint32(N)) == 0
        assert np.abs(buf - ref).max() <= 4e-7 * np.abs(ref).max()
        assert np.abs(buf - po.fft32(x)).max() <= 6e-7 * np.abs(ref).max()
    assert po.lib().sgzo_fft_forward_simd(x.ctypes.data_as(C.c_void_p), C.c_uint32(48)) == -1       # not a power of two: refused
    cfg = config.spectrum_config(window_size=4096, hop=1024)
    p = po.params_from_dict(cfg)
    from signalizer_amd import synth
    xs = synth.gen(7, 48000, 4096 + 1024 * 11, 2)
    want = po.spectrogram_range(p, xs, 0, 12)
    got = np.zeros_like(want)
    ptrs = (C.c_void_p * 2)(*[xs[c].ctypes.data for c in range(2)])
    fn = po.lib().sgzo_spectrogram_range_simd
    fn.restype = C.c_long
    assert fn(C.byref(p), ptrs, C.c_size_t(xs.shape[1]), C.c_long(0), C.c_long(12), got.ctypes.data_as(C.c_void_p)) == 12
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3


def test_windows_vs_scipy(oracle):
    import scipy.signal.windows as w
    po = oracle
    W = 1000
    cases = [(po.WIN_HANN, w.hann), (po.WIN_HAMMING, w.hamming), (po.WIN_BLACKMAN, w.blackman),
             (po.WIN_BLACKMAN_NUTTALL, w.nuttall),   # scipy's `nuttall` is the 4-term Blackman-Nuttall
              (po.WIN_BLACKMAN_HARRIS, w.blackmanharris), (po.WIN_TRIANGULAR, None)]
    for typ, fn in cases:
        for sym in (po.WIN_SYMMETRIC, po.WIN_PERIODIC):
            got, scale = po.window(typ, sym, W)
            if fn is not None:
                ref = fn(W, sym=(sym == po.WIN_SYMMETRIC))
                assert np.abs(got - ref).max() < 2e-7, (typ, sym)
            assert abs(scale - W / got.astype(np.float64).sum()) < 1e-12
    got, _ = po.window(po.WIN_KAISER, po.WIN_SYMMETRIC, W, beta=8.0)
    assert np.abs(got - w.kaiser(W, 8.0)).max() < 2e-7
    got, _ = po.window(po.WIN_RECT, po.WIN_PERIODIC, W)
    assert (got == 1).all()


@pytest.mark.parametrize("win", [config.WIN_RECT, config.WIN_HANN, config.WIN_BLACKMAN_HARRIS])
def test_KA1_KA4_unit_sine_reads_one(oracle, win):
    """a unit sine on an exact bin reads 1.0 (0 dBFS) after invSize, for any window (definition of the scale)"""
    po = oracle
    N, k0 = 4096, 100
    sig = np.sin(2 * np.pi * k0 * np.arange(N) / N).astype(np.float32)
    cfg = config.spectrum_config(window_size=N, hop=N, window_type=win, channel_mode=config.CH_LEFT,
                                 bin_interp=config.INTERP_NONE, view_scaling=config.VIEW_LINEAR, axis_points=N // 2 + 1)
    raw, csf, csp = po.frame_bins(po.params_from_dict(cfg), sig, sig)
    assert abs(csp[k0].real - 1.0) < 2e-6 and int(np.argmax(csp[:N // 2 + 1].real)) == k0


def test_KA2_dc_is_halved(oracle):
    po = oracle
    N = 1024
    cfg = config.spectrum_config(window_size=N, hop=N, window_type=config.WIN_RECT, channel_mode=config.CH_LEFT,
                                 bin_interp=config.INTERP_NONE, view_scaling=config.VIEW_LINEAR, axis_points=N // 2 + 1)
    one = np.ones(N, np.float32)
    raw, csf, csp = po.frame_bins(po.params_from_dict(cfg), one, one)
    # |X[0]| = N, halved (TransformDSP.inl:553), times invSize = 1/(N/2)  => exactly 1.0
    assert csp[0].real == 1.0


def test_KA3_two_for_one_equals_two_mono_runs(oracle):
    po = oracle
    N = 4096
    n = np.arange(N)
    L = np.sin(2 * np.pi * 300 * n / N).astype(np.float32)
    R = (0.5 * np.sin(2 * np.pi * 1000 * n / N + 0.3)).astype(np.float32)
    base = dict(window_size=N, hop=N, bin_interp=config.INTERP_NONE, view_scaling=config.VIEW_LINEAR, axis_points=N // 2 + 1)
    _, csfS, _ = po.frame_bins(po.params_from_dict(config.spectrum_config(channel_mode=config.CH_SEPARATE, **base)), L, R)
    _, csfL, _ = po.frame_bins(po.params_from_dict(config.spectrum_config(channel_mode=config.CH_LEFT, **base)), L, R)
    _, csfR, _ = po.frame_bins(po.params_from_dict(config.spectrum_config(channel_mode=config.CH_RIGHT, **base)), L, R)
    k = np.arange(1, N // 2 - 1)                          # excludes DC, N/2 and the quirk bin N/2-1 (Q3)
    tol = 1e-6 * max(np.abs(csfL.real).max(), np.abs(csfR.real).max())
    assert np.abs(csfS.real[k] - csfL.real[k]).max() < tol
    assert np.abs(csfS.real[N - k] - csfR.real[k]).max() < tol
    assert int(np.argmax(csfS.real[1:N // 2])) + 1 == 300 and int(np.argmax(csfS.real[N // 2 + 1:N])) + N // 2 + 1 == N - 1000
    # Q3: bin N/2-1 of the first channel is halved
    assert abs(csfS.real[N // 2 - 1] - 0.5 * csfL.real[N // 2 - 1]) <= tol


def test_KA5_log_mapping_endpoints(oracle):
    po = oracle
    p = po.params_from_dict(config.cfg2())
    mf = po.remap_frequencies(p)
    assert mf[0] == np.float32(10.0) and mf[-1] == np.float32(24000.0) and (np.diff(mf) > 0).all()


def test_KA6_KA7_decay_and_db_map(oracle):
    po = oracle
    cfg = config.spectrum_config(axis_points=4, low_db=-120.0, high_db=0.0, pole=(0.5, 0.25))
    p = po.params_from_dict(cfg)
    states = np.zeros((2, 4), np.complex64)
    csp = np.zeros(8, np.complex64)
    csp[:4] = [1e-6, 1.0, 0.0, 0.5]                       # lowFrac, hiFrac, silence, -6 dB
    r = po.filters(p, csp, states)
    assert abs(r[0, 0].real - 0.0) < 1e-6 and abs(r[0, 1].real - 1.0) < 1e-6
    assert r[0, 2].real == np.float32(-384.0)             # clipDB sentinel
    assert abs(r[0, 3].real - (1 + 20 * np.log10(0.5) / 120)) < 1e-6
    # impulse then silence: state_t = mag * pole^t exactly
    zero = np.zeros(8, np.complex64)
    for t in range(1, 6):
        po.filters(p, zero, states)
        assert states[0, 1].real == np.float32(0.5) ** t and states[1, 1].real == np.float32(0.25) ** t


def test_KA8_colour_map_hand_computed(oracle):
    po = oracle
    cfg = config.spectrum_config(axis_points=6, colours=[(10, 20, 30), (0, 0, 64), (0, 128, 255), (0, 255, 128), (255, 255, 0), (255, 64, 0)])
    p = po.params_from_dict(cfg)
    ratios = po.colour_ratios(cfg["ratios"])
    assert ratios[0] == 0 and abs(ratios[1:].sum() - 1) < 1e-6 and ratios[1:].sum() < 1.0
    frames = np.zeros((1, 6), np.complex64)
    frames[0].real = [-0.5, 0.0, 0.1, 0.2, 0.999, 2.0]
    rgba = po.blend_column(p, frames)
    assert (rgba[:, 3] == 255).all()
    assert tuple(rgba[0, :3]) == (0, 0, 0)                              # I < 0: skipped, buffer stays 0
    assert tuple(rgba[1, :3]) == (10, 20, 30)                           # I = 0: first stop = background
    # I = 0.1 : halfway between background and (0,0,64) (hue-rotated by 0 for pair 0)
    a, b = np.float32([10, 20, 30]) / np.float32(255), np.float32([0, 0, 64]) / np.float32(255)
    mix = np.float32(0.1) / ratios[1]
    want = (a * (np.float32(1) - mix) + b * mix) * np.float32(255)
    assert tuple(rgba[2, :3]) == tuple(want.astype(np.uint8))
    assert tuple(rgba[4, :3]) == (255, 64, 0) and tuple(rgba[5, :3]) == (255, 64, 0)


def test_KA9_hsb_rotation_properties(oracle):
    """juce HSB round trip: not the identity for all RGB (rounding +1e-5, roundToInt), greys are fixed points,
    rotation by 1/3 maps pure red to pure green; brightness (max channel) is preserved within 1."""
    import colorsys
    po = oracle
    assert tuple(po.rotate_hue((255, 0, 0), 1.0 / 3.0)) == (0, 255, 0)
    assert tuple(po.rotate_hue((0, 255, 0), 1.0 / 3.0)) == (0, 0, 255)
    for g in (0, 1, 77, 255):
        assert tuple(po.rotate_hue((g, g, g), 0.37)) == (g, g, g)
    rng = np.random.default_rng(1)
    worst = 0
    for _ in range(2000):
        rgb = tuple(int(v) for v in rng.integers(0, 256, 3))
        amt = float(rng.random())
        got = po.rotate_hue(rgb, amt)
        h, s, v = colorsys.rgb_to_hsv(*(c / 255 for c in rgb))
        ref = np.array(colorsys.hsv_to_rgb((h + amt) % 1.0, s, v)) * 255
        worst = max(worst, np.abs(got - ref).max())
        assert abs(int(got.max()) - max(rgb)) <= 1
    assert worst <= 2.0          # integer rounding at three places


def test_KA9_hsb_rotation_equals_the_compiled_reference(oracle):
    """The one link of the chain that is pinned to the reference itself (container-only): juce::Colour::withRotatedHue compiled from
    the sources under /root/reference (oracle/ref_juce_colour.cpp -> oracle/_ref/, oracle/pyref.py) against the oracle's restatement
    (primitives.c) and the product's host function (plan.cpp rotateHueRgb8) over 200 000 random colours and rotation amounts, plus
    ColourRotation::operator[]'s own amounts index / size for 1 .. 16 pairs.  Skipped where the reference is not present (the GPU
    box): there tests/test_golden.py holds both to the committed vectors this library generated."""
    from oracle import pyref
    if not pyref.available():
        pytest.skip("the reference's sources are not in this environment (oracle/_ref is container-only)")
    from signalizer_amd import api
    po = oracle
    L, O, A = pyref.lib(), po.lib(), api.lib()
    import ctypes as C
    rng = np.random.default_rng(2026)
    n = 200000
    cols = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    amt = rng.random(n).astype(np.float32)
    amt[: 16 * 17 // 2] = [np.float32(i) / np.float32(p) for p in range(1, 17) for i in range(p)]        # index / size as ColourRotation computes it
    ref, got_o, got_p = np.zeros((n, 3), np.uint8), np.zeros((n, 3), np.uint8), np.zeros((n, 3), np.uint8)
    vp = C.c_void_p
    for i in range(n):
        a = C.c_float(float(amt[i]))
        L.sgzref_rotate_hue_rgb8(vp(cols[i].ctypes.data), a, vp(ref[i].ctypes.data))
        O.sgzo_rotate_hue_rgb8(vp(cols[i].ctypes.data), a, vp(got_o[i].ctypes.data))
        A.sgz_rotate_hue_rgb8(vp(cols[i].ctypes.data), a, vp(got_p[i].ctypes.data))
    assert np.array_equal(got_o, ref), int((got_o != ref).any(axis=1).sum())
    assert np.array_equal(got_p, ref), int((got_p != ref).any(axis=1).sum())


def test_KA10_lanczos_kernel(oracle):
    po = oracle
    L = po.lib()
    assert L.sgzo_lanczos_kernel(0.0, 10) == 1.0
    for k in range(1, 10):
        assert abs(L.sgzo_lanczos_kernel(float(k), 10)) < 1e-15
    assert L.sgzo_lanczos_kernel(10.0, 10) == 0.0 and L.sgzo_lanczos_kernel(-10.5, 10) == 0.0
    v = np.sin(0.05 * np.arange(21)).astype(np.float32)
    for i in (3, 10, 17):
        assert abs(L.sgzo_lanczos_filter_f64(v.ctypes.data_as(C.c_void_p), 21, float(i), 10) - v[i]) < 1e-15
    mid = L.sgzo_lanczos_filter_f64(v.ctypes.data_as(C.c_void_p), 21, 10.5, 10)
    assert abs(mid - np.sin(0.05 * 10.5)) < 2e-3


def test_KA11_zero_crossing(oracle):
    po = oracle
    x = np.array([-1, -0.5, 0.2, 0.6, 0.1, -0.3, 0.01, 0.02, 0.9, -1, 1], np.float32)
    st = po.ZeroCrossingState(state=0.0, threshold=0.5, steady_clock=100, cross_origin=0, count=0, armed=0)
    # arms at i=2 (fires at i=3 -> 102), re-arms at i=6 (fires at i=8 -> 106), arms+fires at i=10 (110)
    assert list(po.zero_crossing(st, 0, x)) == [102, 106, 110]
    st = po.ZeroCrossingState(state=0.0, threshold=5.0, steady_clock=0, cross_origin=0, count=0, armed=0)
    assert po.zero_crossing(st, 0, x).size == 0 and st.armed == 1 and st.cross_origin == 10
    # exact zeros neither arm nor count as negative
    st = po.ZeroCrossingState(state=0.0, threshold=0.0, steady_clock=0, cross_origin=0, count=0, armed=0)
    assert po.zero_crossing(st, 0, np.array([0, 1, 0, 1, -1, 0, 1], np.float32)).size == 0


def test_KA12_polar_closed_form(oracle):
    po = oracle
    L = np.array([1, 0, 1, 1, 0], np.float32)
    R = np.array([0, 1, 1, -1, 0], np.float32)
    xyz = po.vector_polar(L, R)
    s = np.sqrt(0.5)
    # (1,0): Y=-s, X=s -> atan(-1) = -pi/4 ; (0,1): Y=-s, X=-s -> atan(1) = pi/4
    assert np.allclose(xyz[0, :2], [-s, s], atol=1e-6) and np.allclose(xyz[1, :2], [s, s], atol=1e-6)
    # (1,1): Y=-2s, X=0 -> angle 0 (atan(-0.0) = -0) -> (0, 1) ; (1,-1): Y=0, X=2s -> atan(+inf)=pi/2 -> (1, 0)
    assert np.allclose(xyz[2, :2], [0, 1], atol=1e-6) and np.allclose(xyz[3, :2], [1, 0], atol=1e-6)
    assert np.allclose(xyz[4, :2], [0, 0])


def test_KA13_one_pole_envelope(oracle):
    po = oracle
    n = 4096
    x = np.full(n, 0.5, np.float32)
    f = po.VectorFilters()
    a = float(np.float32(np.exp(-1.0 / 200.0)))
    po.vector_audio_processing(f, x, x, a, a)
    want = 0.25 * (1 - np.float64(np.float32(a)) ** n)           # y_n = x (1 - a^n), x = 0.25 = 0.5^2
    assert abs(f.env[0] - want) < 1e-5 and f.env[0] == f.env[1]
    assert abs(f.balance[0][0] - want) < 1e-5


def test_spectrogram_frame_count_and_edge_cases(oracle):
    po = oracle
    L = po.lib()
    assert L.sgzo_num_frames(2880000, 32768, 8192) == 348           # BASELINE cfg2
    assert L.sgzo_num_frames(5760000, 65536, 16384) == 348          # BASELINE cfg5
    assert L.sgzo_num_frames(100, 4096, 1024) == 0 and L.sgzo_num_frames(4096, 4096, 1024) == 1
    assert L.sgzo_transform_size(1) == 32 and L.sgzo_transform_size(33) == 64 and L.sgzo_transform_size(3000) == 4096
    # silence renders the background colour, not garbage
    cfg = config.spectrum_config(window_size=64, hop=16, axis_points=8, colours=[(3, 2, 1)] + config.DEFAULT_COLOURS[1:])
    r = po.spectrogram(po.params_from_dict(cfg), np.zeros((2, 128), np.float32))
    assert r["frames"] == 5 and (r["rgba"][..., :3] == 0).all()      # I = clipDB < 0 -> pixel skipped -> 0


@pytest.mark.parametrize("interp", [config.INTERP_NONE, config.INTERP_LINEAR, config.INTERP_LANCZOS])
def test_KA14_phase_mode_cancellation(oracle, interp):
    """Phase mode (TransformDSP.inl:643-853): wsp[2x+1] = 1 - |L+R| / (|L|+|R|) per pixel.  Identical channels cancel
    nothing (0), opposite channels cancel completely (1), and the magnitude plane is |L| + |R| either way."""
    po = oracle
    cfg = config.spectrum_config(window_size=1024, hop=1024, channel_mode=config.CH_PHASE, bin_interp=interp, axis_points=200)
    p = po.params_from_dict(cfg)
    rng = np.random.default_rng(3)
    L = rng.uniform(-1, 1, 1024).astype(np.float32)
    _, _, same = po.frame_bins(p, L, L)
    _, _, opp = po.frame_bins(p, L, -L)
    P = 200
    ms, cs = same[:P].real, same[:P].imag
    mo, co = opp[:P].real, opp[:P].imag
    # (interpolating at pos and at fl(N - pos) does not sample exactly mirrored points: ~1e-4 of residue in the filtered modes)
    tol = 2e-6 if interp == config.INTERP_NONE else 1e-3
    assert np.abs(cs).max() < tol and np.abs(co - 1).max() < tol
    # (the few pixels whose taps straddle the lazily normalised / still complex boundary differ, Q6)
    close = np.abs(ms - mo) <= 1e-5 * np.abs(ms) + 1e-7
    assert close.mean() >= 0.9 and np.allclose(ms, mo, rtol=0.2) and ms.max() > 0
    # the filters: magnitude halves (mag *= 0.5), the smoothed phase starts from 0 towards cancellation * mag (Q7: * mag again
    # for the second graph)
    st = np.zeros((2, P), np.complex64)
    res = po.filters(p, opp, st)
    mag = (mo * np.float32(0.5)).astype(np.float32)
    assert np.array_equal(st[0].real, mag) and np.array_equal(st[1].real, mag)
    pf = np.float32(np.float32(cfg["pole"][0]) ** np.float32(0.3))
    ph0 = (co * mag).astype(np.float32)
    want0 = (ph0 + pf * (np.float32(0) - ph0)).astype(np.float32)
    assert np.allclose(st[0].imag, want0, rtol=1e-6, atol=1e-12)
    assert res.shape == (2, P)


@pytest.mark.parametrize("mode", [config.CH_LEFT, config.CH_MERGE, config.CH_SIDE, config.CH_COMPLEX])
def test_KA15_entries_left_complex(oracle, mode):
    """The csf entries mapToLinearSpace leaves complex (mono modes: csf[N/2..N-1], TransformDSP.inl:553-560; Complex: csf[0],
    :993) enter the Lanczos sums as complex numbers: an independent numpy restatement (fp64 FFT, complex arithmetic) of the
    interpolated pixels of a view that starts at 0 Hz, where the filter window wraps below bin 0."""
    po = oracle
    W = N = 1024
    P = 120
    cfg = config.spectrum_config(sample_rate=48000.0, window_size=W, hop=W, channel_mode=mode, bin_interp=config.INTERP_LANCZOS,
                                 view_scaling=config.VIEW_LINEAR, axis_points=P, view_left=0.0, view_right=0.02,
                                 window_type=config.WIN_HANN)
    p = po.params_from_dict(cfg)
    rng = np.random.default_rng(15)
    L = rng.uniform(-1, 1, W).astype(np.float32) + 0.3
    R = rng.uniform(-1, 1, W).astype(np.float32) - 0.2
    _, _, csp = po.frame_bins(p, L, R)
    win, scale = po.window(cfg["window_type"], cfg["window_symmetry"], W)
    w = win[:W].astype(np.float64)
    l, r = L.astype(np.float64), R.astype(np.float64)
    z = {config.CH_LEFT: l * w, config.CH_MERGE: (l + r) * w * 0.5, config.CH_SIDE: (l - r) * w * 0.5,
         config.CH_COMPLEX: (l + 1j * r) * w}[mode]
    Z = np.fft.fft(z)
    csf = np.zeros(N + 1, np.complex128)
    if mode == config.CH_COMPLEX:
        csf[:N] = np.abs(Z)
        csf[0] = 0.5 * Z[0]
    else:
        csf[:N] = Z
        csf[0] *= 0.5
        csf[N // 2] *= 0.5
        csf[:N // 2] = np.abs(csf[:N // 2])
    mf = po.remap_frequencies(p).astype(np.float64)
    f2b = (N / 2) / (48000.0 / 2)
    inv = scale / (W * 0.5)
    a = 5
    checked = 0
    for x in range(20):                                       # well inside the interpolated region of this view
        pos = mf[x] * f2b
        fl = int(np.floor(pos))
        acc = 0j
        for i in range(fl - a + 1, fl + a + 1):
            d = pos - i
            wt = 1.0 if d == 0 else (0.0 if abs(d) >= a else a * np.sin(np.pi * d) * np.sin(np.pi * d / a) / (np.pi * d) ** 2)
            acc += csf[i % (N + 1)] * wt
        want = inv * abs(acc)
        got = abs(complex(csp[x]))
        assert abs(got - want) <= 2e-5 * max(want, 1e-3), (x, got, want)
        checked += 1 if abs(acc.imag) > 1e-6 * abs(acc) else 0
    assert checked >= 3                                      # the complex entries really took part
