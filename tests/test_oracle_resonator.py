"""RSNT (the resonator algorithm): the oracle's restatement pinned against independent mathematics, and the library's host tables
against the oracle.  cpl's CComplexResonator is absent (parity unpinned): these are known answers of the published mathematics
(complex one-pole resonator = exponentially windowed DFT bin; cosine-sum windows as frequency-domain kernels)."""
import numpy as np
import pytest

from signalizer_amd import api, config as cf


def _cfg(**over):
    d = dict(algorithm=cf.ALGO_RSNT, window_size=4096, hop=1024, axis_points=256, window_type=cf.WIN_RECT)
    d.update(over)
    return cf.spectrum_config(**d)


def test_window_terms_sum_to_one_and_alternate(oracle):
    # a cosine-sum window peaks at w[N/2] = sum a_m = 1; the frequency-domain kernel's weights are a_0, -a_1/2, +a_2/2, ...
    for wt, K in ((cf.WIN_RECT, 1), (cf.WIN_HANN, 2), (cf.WIN_HAMMING, 2), (cf.WIN_BLACKMAN, 3), (cf.WIN_NUTTALL, 4),
                  (cf.WIN_BLACKMAN_HARRIS, 4), (cf.WIN_FLATTOP, 5), (cf.WIN_KAISER, 1)):
        a = np.zeros(5)
        assert oracle.lib().sgzo_window_cosine_terms(wt, a.ctypes.data) == K
        assert abs(a.sum() - 1.0) < 2e-6 and np.all(a[:K] > 0) and np.all(a[K:] == 0)
        p = oracle.params_from_dict(_cfg(window_type=wt))
        _, _, w = oracle.resonator_map(p)
        assert len(w) == 2 * K - 1 and np.allclose(w, w[::-1])
        assert w[K - 1] == np.float32(a[0])
        for m in range(1, K):
            assert w[K - 1 + m] == np.float32((-0.5 if m & 1 else 0.5) * a[m])


def test_bandwidth_follows_the_axis_and_the_window_bound(oracle):
    p = oracle.params_from_dict(_cfg())
    mf = oracle.remap_frequencies(p).astype(np.float64)
    coeff, gain, _ = oracle.resonator_map(p)
    r = np.abs(coeff[0].astype(np.complex128))
    length = 48000.0 / np.abs(np.diff(mf))
    length = np.append(length, length[-1])                       # the last filter reuses the spacing before it
    want = np.exp(-np.pi / np.clip(length, 2.0, 4096.0))         # Q not free: never longer than the window
    assert np.allclose(r, want, rtol=2e-7)
    assert np.allclose(gain, 1.0 - want, rtol=1e-4, atol=1e-9)
    assert np.allclose(np.angle(coeff[0].astype(np.complex128)), 2 * np.pi * mf / 48000.0, atol=3e-7)
    pf = oracle.params_from_dict(_cfg(free_q=1))
    rf = np.abs(oracle.resonator_map(pf)[0][0].astype(np.complex128))
    assert np.all(rf >= r - 1e-7) and rf[0] > r[0]                # free Q: the low filters get their full (longer) windows


def test_state_is_the_exponentially_windowed_dft(oracle):
    rng = np.random.default_rng(5)
    p = oracle.params_from_dict(_cfg(axis_points=64, hop=500))
    coeff, gain, w = oracle.resonator_map(p)
    x = rng.standard_normal((2, 1500)).astype(np.float32)
    got = oracle.resonator_spectrogram(p, x, want_mapped=True)["mapped"]          # [3][1][2P]
    n = np.arange(1500)
    for f in range(3):
        end = 500 * (f + 1)
        for ch in range(2):
            # s = sum_k c^(end-1-k) x[k], times gain (rectangular: one vector)
            c = coeff[0].astype(np.complex128)[:, None]
            s = (c ** (end - 1 - n[:end])[None, :] * x[ch, :end][None, :]).sum(axis=1) * gain
            ref = got[f, 0, ch * 64:(ch + 1) * 64].astype(np.complex128)
            assert np.max(np.abs(ref - s)) <= 2e-5 * np.max(np.abs(s))


def test_full_scale_sine_reads_one_half_unwindowed_and_hann_is_its_three_tap_kernel(oracle):
    p = oracle.params_from_dict(_cfg())
    mf = oracle.remap_frequencies(p)
    i = 150
    t = np.arange(48000) / 48000.0
    x = np.sin(2 * np.pi * float(mf[i]) * t).astype(np.float32)
    m = oracle.resonator_spectrogram(p, np.stack([x, 0.25 * x]), want_mapped=True)["mapped"][-1, 0]
    assert abs(abs(m[i]) - 0.5) < 0.01 and abs(abs(m[256 + i]) - 0.125) < 0.0025      # Separate: right channel at [P, 2P)
    # Hann: a0 s0 - a1/2 (s-1 + s+1) on resonators detuned by the bandwidth, evaluated from independently computed states
    ph = oracle.params_from_dict(_cfg(window_type=cf.WIN_HANN))
    coeff, gain, w = oracle.resonator_map(ph)
    mh = oracle.resonator_spectrogram(ph, np.stack([x, x]), want_mapped=True)["mapped"][-1, 0]
    n = np.arange(x.size // 1024 * 1024)
    c = coeff[:, i].astype(np.complex128)
    s = np.array([(cv ** (n.size - 1 - n) * x[:n.size]).sum() for cv in c])
    want = (w.astype(np.float64) * s).sum() * gain[i]
    assert abs(mh[i] - want) <= 3e-5 * abs(want)


def test_dispatch_and_phase_post_processing(oracle):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 2048)).astype(np.float32)
    base = _cfg(axis_points=32, hop=512)

    def run(mode, data):
        return oracle.resonator_spectrogram(oracle.params_from_dict(dict(base, channel_mode=mode)), data, want_mapped=True)["mapped"][:, 0]
    sep = run(cf.CH_SEPARATE, x)
    # TransformDSP.inl:1250-1293: Left / Right pick a channel, Mid = L + R (not halved), Side = L - R, MidSide = (L - R, L + R)
    assert np.array_equal(run(cf.CH_LEFT, x)[:, :32], sep[:, :32]) and np.array_equal(run(cf.CH_RIGHT, x)[:, :32], sep[:, 32:])
    mid = run(cf.CH_SEPARATE, np.stack([x[0] + x[1], x[0] - x[1]]))
    assert np.array_equal(run(cf.CH_MERGE, x)[:, :32], mid[:, :32]) and np.array_equal(run(cf.CH_SIDE, x)[:, :32], mid[:, 32:])
    ms = run(cf.CH_MIDSIDE, x)
    assert np.array_equal(ms[:, :32], mid[:, 32:]) and np.array_equal(ms[:, 32:], mid[:, :32])
    # Phase (:1111-1127): (|L| + |R|, 1 - |L + R| / (|L| + |R|)): identical channels cancel nothing, opposite ones everything
    same = run(cf.CH_PHASE, np.stack([x[0], x[0]]))
    opp = run(cf.CH_PHASE, np.stack([x[0], -x[0]]))
    assert np.allclose(same[:, :32].imag, 0.0, atol=2e-6) and np.allclose(opp[:, :32].imag, 1.0, atol=1e-6)
    assert np.allclose(same[:, :32].real, 2 * np.abs(sep[:, :32]), rtol=1e-6)


@pytest.mark.parametrize("over", [dict(), dict(window_type=cf.WIN_HANN, free_q=1), dict(window_type=cf.WIN_FLATTOP, view_scaling=cf.VIEW_LINEAR),
                                  dict(window_type=cf.WIN_BLACKMAN_HARRIS, axis_points=1024, window_size=32768, hop=8192, sample_rate=96000.0)])
def test_library_tables_equal_the_oracle(oracle, over):
    d = _cfg(**over)
    plan = api.Plan(d)
    coeff, gain, w = plan.resonator()
    oc, og, ow = oracle.resonator_map(oracle.params_from_dict(d))
    assert np.array_equal(coeff.view(np.float32), oc.view(np.float32)) and np.array_equal(gain, og) and np.array_equal(w, ow)
    assert plan.num_frames(10 * d["hop"] + 5) == 10 and plan.num_frames(d["hop"] - 1) == 0


def test_fp64_walk_helper_against_lfilter_and_the_oracle(oracle):
    """tests/rsnt_truth.py (the "truth" the GPU tests hold the device to at the 4-unit bar): its block-sum evaluation equals a plain
    complex128 recurrence (scipy.signal.lfilter), and the oracle's sequential fp32 evaluation of a recorded worst case
    (profiles/r05e/fuzz_campaign_2.txt: seed 2021, case 23) stays within 8 units of it (recorded: 1.09 of the 4-unit bar)."""
    import fuzzcfg
    import rsnt_truth
    from scipy.signal import lfilter
    po = oracle
    d, F, x = fuzzcfg.rsnt_case(2021, 23)
    p = po.params_from_dict(d)
    pair, sig, hop, P = 1, 0, d["hop"], d["axis_points"]
    truth, scale = rsnt_truth.frame_magnitudes(po, p, x, pair, sig, F)
    coeff, gain, weights = po.resonator_map(p)
    xin = rsnt_truth.dispatch_signal(po, d["channel_mode"], x[2 * pair], x[2 * pair + 1], sig)[:F * hop].astype(np.complex128)
    for i in (0, 11, P // 2, P - 1):
        acc = np.zeros(F, np.complex128)
        for v in range(coeff.shape[0]):
            acc += float(weights[v]) * lfilter([1.0], [1.0, -complex(coeff[v, i])], xin)[hop - 1::hop][:F]
        assert np.allclose(np.abs(acc) * float(gain[i]), truth[:, i], rtol=1e-7, atol=1e-9 * float(scale[:, i].max()))
    r = po.resonator_spectrogram(p, x, want_mapped=True, want_scale=True)
    assert np.allclose(r["scale"][:, pair, sig], scale, rtol=1e-4, atol=1e-12)                   # the same bar either way
    mag = lambda z: np.sqrt(z.real * z.real + z.imag * z.imag)
    ref = mag(r["mapped"][:, pair, :P])
    state_tol = 8.0 * 2.0 ** -24 * np.sqrt(1.0 / np.maximum(gain.astype(np.float64), 1e-12))
    for f in range(F):
        bar = 2e-5 * max(float(truth[f].max()), 1e-30) + state_tol * scale[f]
        assert float(np.max(np.abs(ref[f] - truth[f]) / bar)) <= 1.0
