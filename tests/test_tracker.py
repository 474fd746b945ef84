"""Frequency tracker (SpectrumRendering.cpp:379-469): known answers for the oracle's restatement, and the HIP kernel against it."""
import ctypes as C

import numpy as np
import pytest

from signalizer_amd import config, synth


def _tone(freq, sr, n, amp=0.5):
    t = np.arange(n) / sr
    return np.stack([amp * np.sin(2 * np.pi * freq * t), 0.25 * np.sin(2 * np.pi * 3.3 * freq * t)]).astype(np.float32)


def _mouse_fraction_of(freq, cfg):
    # log view [0, 1]: f = minFreq (top / minFreq)^fraction
    top = cfg["sample_rate"] / 2
    return float(np.log(freq / cfg["min_log_freq"]) / np.log(top / cfg["min_log_freq"]))


@pytest.mark.parametrize("freq", [440.0, 997.3, 5000.5])
def test_oracle_tracker_finds_the_tone(oracle, freq):
    """KA: a Hann-windowed sine between two bins: the parabolic fit in the dB domain lands within a few hundredths of a bin of the
    true frequency, and the peak level within 0.3 dB of the tone's 20 log10(amp) (the parabola's known bias on a Hann main lobe)"""
    po = oracle
    cfg = config.spectrum_config(window_size=8192, hop=2048)
    p = po.params_from_dict(cfg)
    x = _tone(freq, 48000, 8192)
    raw, csf, csp = po.frame_bins(p, x[0], x[1])
    _, scale = po.window(p.window_type, p.window_symmetry, p.window_size)
    r = po.track_peak(p, csf, scale, _mouse_fraction_of(freq * 1.01, cfg))
    assert abs(r["peak_frequency"] - freq) < 0.05 * 48000 / 8192, r
    assert abs(r["peak_dbs"] - 20 * np.log10(0.5)) < 0.3, r
    assert r["peak_offset"] == round(freq * 8192 / 48000)


@pytest.mark.gpu
@pytest.mark.parametrize("W,mode", [(4096, config.CH_SEPARATE), (32768, config.CH_SEPARATE), (65536, config.CH_SEPARATE), (3000, config.CH_LEFT),
                                    (8192, config.CH_MIDSIDE)])
def test_gpu_tracker_against_the_oracle(gpu, oracle, W, mode):
    """peak bin (integer work): exact; the fit (log10f of libm vs ocml): 1e-4 dB / 1e-5 of a bin"""
    import torch
    from signalizer_amd import api
    po = oracle
    cfg = config.spectrum_config(window_size=W, hop=max(2, W // 4 // 2 * 2), channel_mode=mode, sample_rate=96000.0 if W == 65536 else 48000.0)
    p = po.params_from_dict(cfg)
    sr = int(cfg["sample_rate"])
    x = synth.gen(31, sr, W, 2)
    plan = api.Plan(cfg).upload()
    bins = plan.stage_bins(torch.from_numpy(x).to(gpu))                       # [1][1][N+1]
    raw, csf, csp = po.frame_bins(p, x[0], x[1])
    _, scale = po.window(p.window_type, p.window_symmetry, p.window_size)
    L = api.lib()
    for mf in (0.0, 0.07, 0.33, 0.5, 0.61, 0.8, 0.97, 1.0):
        want = po.track_peak(p, csf, scale, mf)
        got = api.Peak()
        api.check(L.sgz_stage_track_peak(plan.h, bins.data_ptr(), mf, C.byref(got), torch.cuda.current_stream().cuda_stream))
        g = got.asdict()
        # the two FFTs differ by rounding: a range whose top two bins tie to 1e-7 could pick either; not in this signal
        assert g["peak_offset"] == want["peak_offset"], (mf, g, want)
        for k in ("alpha", "beta", "gamma", "peak_dbs"):
            assert abs(g[k] - want[k]) <= 2e-3, (mf, k, g[k], want[k])             # 20 log10 of magnitudes that agree to 4e-6 of the maximum
        assert abs(g["peak_fraction"] - want["peak_fraction"]) <= 1e-3 * 2 / plan.N + 1e-9 or abs(g["phi"] - want["phi"]) <= 1e-3


@pytest.mark.gpu
def test_gpu_tracker_on_the_realtime_handle(gpu, oracle):
    """sgz_spectrum_track_peak: the newest window of the device ring, transformed on demand"""
    from signalizer_amd import api
    cfg = config.spectrum_config(window_size=8192, hop=2048, axis_points=512)
    c = api.config_from_dict(cfg)
    h = C.c_void_p()
    api.check(api.lib().sgz_spectrum_create(C.byref(c), C.byref(h)))
    try:
        x = _tone(1234.5, 48000, 8192 * 3)
        ptrs = (C.c_void_p * 2)(x[0].ctypes.data, x[1].ctypes.data)
        api.check(api.lib().sgz_spectrum_push(h, ptrs, 2, x.shape[1]))
        got = api.Peak()
        api.check(api.lib().sgz_spectrum_track_peak(h, 0, _mouse_fraction_of(1250.0, cfg), C.byref(got)))
        assert abs(got.peak_frequency - 1234.5) < 0.05 * 48000 / 8192
        assert abs(got.peak_dbs - 20 * np.log10(0.5)) < 0.3
    finally:
        api.lib().sgz_spectrum_destroy(h)


@pytest.mark.parametrize("over", [dict(), dict(bin_interp=config.INTERP_LINEAR), dict(channel_mode=config.CH_COMPLEX, bin_interp=config.INTERP_NONE),
                                  dict(algorithm=1, window_size=4096, hop=1024), dict(axis_points=10), dict(axis_points=33, view_scaling=config.VIEW_LINEAR),
                                  dict(axis_points=3000, window_size=32768, hop=8192)])
def test_line_results_tracker_equals_the_oracle(oracle, over):
    """The tracker's OTHER branch (Complex mode, RSNT, LineMain / LineSecond: SpectrumRendering.cpp:300-377) picks the peak from the
    displayed line.  It is host arithmetic on host-resident results in the reference and here (sgz_track_peak_lines): the library's
    function against the oracle's line-by-line restatement, every output bit for bit -- random lines, plateaus (max_element takes the
    FIRST largest), monotone ramps (the walks along a rising edge at either boundary of the +-3 % range) and the ends of the axis."""
    from signalizer_amd import api
    po = oracle
    cfg = config.spectrum_config(**{**dict(window_size=8192, hop=2048), **over})
    p = po.params_from_dict(cfg)
    plan = api.Plan(cfg)
    P = plan.P
    rng = np.random.default_rng(5)
    lines = []
    for kind in range(6):
        r = np.zeros((P, 2), np.float32)
        if kind == 0: r[:, 0] = rng.random(P)
        elif kind == 1: r[:, 0] = np.round(rng.random(P) * 4) / 4                      # plateaus: ties
        elif kind == 2: r[:, 0] = np.linspace(0, 1, P)                                  # rising to the right: the upward walk to the end
        elif kind == 3: r[:, 0] = np.linspace(1, 0, P)                                  # rising to the left: the downward walk (stops at 1, not 0)
        elif kind == 4: r[:, 0] = np.exp(-0.5 * ((np.arange(P) - 0.37 * P) / (0.02 * P + 1)) ** 2)
        else: r[:, 0] = 0.25
        r[:, 1] = rng.random(P)                                                         # (the right magnitudes are never looked at)
        lines.append(r)
    n = 0
    for r in lines:
        for mf in (-0.3, 0.0, 0.01, 0.029, 0.03, 0.2, 0.37, 0.5, 0.77, 0.969, 0.97, 0.99, 1.0, 1.4):
            want = po.track_peak_lines(p, r, plan.N, mf)
            got = plan.track_peak_lines(r, mf)
            for k, v in want.items():
                assert np.float64(got[k]).view(np.uint64) == np.float64(v).view(np.uint64), (over, mf, k, got, want)
            n += 1
    assert n == 84


@pytest.mark.gpu
@pytest.mark.parametrize("over,graph", [(dict(channel_mode=config.CH_COMPLEX), 0), (dict(algorithm=1), 0), (dict(), 1), (dict(channel_mode=config.CH_MIDSIDE), 1)])
def test_handle_tracks_the_peak_of_its_line_results(gpu, oracle, over, graph):
    """sgz_spectrum_track_peak_lines = sgz_track_peak_lines on what sgz_spectrum_line_results returns at that moment: Complex mode, RSNT
    and the LineSecond graph -- the cases the raw-bin tracker does not serve (SpectrumRendering.cpp:301)"""
    from signalizer_amd import api
    po = oracle
    cfg = config.spectrum_config(**{**dict(window_size=4096, hop=1024, axis_points=512), **over})
    p = po.params_from_dict(cfg)
    x = _tone(2500.0, 48000, 4096 * 6)
    L = api.lib()
    c = api.config_from_dict(cfg)
    h = C.c_void_p()
    api.check(L.sgz_spectrum_create(C.byref(c), C.byref(h)))
    try:
        for o in range(0, x.shape[1], 512):
            blk = np.ascontiguousarray(x[:, o:o + 512])
            api.check(L.sgz_spectrum_push(h, (C.c_void_p * 2)(blk[0].ctypes.data, blk[1].ctypes.data), 2, 512))
        L.sgz_spectrum_flush.argtypes = [C.c_void_p]
        api.check(L.sgz_spectrum_flush(h))
        import torch
        torch.cuda.synchronize()
        res = np.zeros((512, 2), np.float32)
        for _ in range(50):                                              # (the newest results reach the host behind the producer's stream)
            api.check(L.sgz_spectrum_line_results(h, 0, graph, res.ctypes.data_as(C.c_void_p)))
            if np.abs(res[:, 0]).max() > 0:
                break
        assert np.abs(res[:, 0]).max() > 0
        N = api.Plan(cfg).N
        for mf in (0.1, 0.45, 0.6, 0.9):
            got = api.LinePeak()
            api.check(L.sgz_spectrum_track_peak_lines(h, 0, graph, mf, C.byref(got)))
            want = po.track_peak_lines(p, res, N, mf)
            for k, v in want.items():
                assert np.float64(getattr(got, k)).view(np.uint64) == np.float64(v).view(np.uint64), (mf, k, got.asdict(), want)
    finally:
        L.sgz_spectrum_destroy(h)
