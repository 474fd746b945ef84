"""SpectrumChannels::Phase (TransformDSP.inl:643-853, :1393-1432) on the GPU against the oracle: complex bins within the fp32
FFT tolerance, the pixel mapping (magnitude + cancellation) bit for bit on identical bins, the filters / colour on identical
mapped values, and the whole path end to end."""
import numpy as np
import pytest

from signalizer_amd import api, config, synth

pytestmark = pytest.mark.gpu
BIN_TOL = 4e-6


def _cuda(x, gpu):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to(gpu)


def _split(raw):
    """separateTransformsIPL + the DC / Nyquist fix-ups in fp32, exactly as oracle/primitives.c + spectrum.c (:646-652)"""
    N = raw.size - 1
    csf = raw.astype(np.complex64).copy()
    k = np.arange(1, N // 2)
    a, b = raw[k], raw[N - k]
    half = np.float32(0.5)
    x1 = ((a.real + b.real) * half) + 1j * ((a.imag - b.imag) * half)
    x2 = ((a.imag + b.imag) * half) + 1j * ((b.real - a.real) * half)
    csf[k] = x1.astype(np.complex64)
    csf[N - k] = x2.astype(np.complex64)
    csf[N] = np.complex64(raw[0].imag * half)
    csf[0] = np.complex64(raw[0].real * half)
    csf[N // 2] = np.complex64(complex(raw[N // 2].real * half, raw[N // 2].imag * half))
    v = csf[N // 2 - 1]
    csf[N // 2 - 1] = np.complex64(complex(v.real * half, v.imag * half))
    return csf


def _cfg(interp, view=config.VIEW_LOG, W=4096, P=500, **kw):
    return config.spectrum_config(window_size=W, hop=W // 4, channel_mode=config.CH_PHASE, bin_interp=interp,
                                  view_scaling=view, axis_points=P, **kw)


@pytest.mark.parametrize("W", [4096, 1024, 3000])
def test_phase_complex_bins(gpu, oracle, W):
    po = oracle
    cfg = _cfg(config.INTERP_LANCZOS, W=W)
    p = po.params_from_dict(cfg)
    hop = cfg["hop"]
    x = synth.gen(12, 48000, W + 2 * hop, 2)
    plan = api.Plan(cfg).upload()
    bins = plan.stage_bins(_cuda(x, gpu)).cpu().numpy()           # [F][1][N+1][2]
    for f in range(3):
        raw, _, _ = po.frame_bins(p, x[0, f * hop:f * hop + W], x[1, f * hop:f * hop + W])
        ref = _split(raw)
        got = bins[f, 0, :, 0] + 1j * bins[f, 0, :, 1]
        assert np.abs(got - ref).max() <= BIN_TOL * np.abs(ref).max()


@pytest.mark.parametrize("interp", [config.INTERP_NONE, config.INTERP_LINEAR, config.INTERP_LANCZOS])
@pytest.mark.parametrize("view,P", [(config.VIEW_LOG, 500), (config.VIEW_LINEAR, 700), (config.VIEW_LOG, 4000)])
def test_phase_mapping_bit_exact_given_bins(gpu, oracle, interp, view, P):
    """magnitude and cancellation of every pixel from identical complex bins (P = 4000: a view that never leaves the
    interpolation branch)"""
    import torch
    po = oracle
    cfg = _cfg(interp, view, P=P)
    p = po.params_from_dict(cfg)
    x = synth.gen(6, 48000, 4096 * 2, 2)
    plan = api.Plan(cfg).upload()
    frames = 2
    csfs = np.zeros((frames, 1, plan.N + 1, 2), np.float32)
    want = np.zeros((frames, 1, 2, plan.P), np.float32)
    for f in range(frames):
        raw, _, csp = po.frame_bins(p, x[0, f * 4096:(f + 1) * 4096], x[1, f * 4096:(f + 1) * 4096])
        c = _split(raw)
        csfs[f, 0, :, 0], csfs[f, 0, :, 1] = c.real, c.imag
        want[f, 0, 0], want[f, 0, 1] = csp[:plan.P].real, csp[:plan.P].imag
    got = plan.stage_map_from_bins(torch.from_numpy(csfs).to(gpu)).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (np.abs(got - want).max(), np.argwhere(got != want)[:5])


def test_phase_filters_and_colour_given_mapped(gpu, oracle):
    import torch
    po = oracle
    cfg = _cfg(config.INTERP_LANCZOS, P=333, num_pairs=2, pole=(0.9, 0.99))
    p = po.params_from_dict(cfg)
    frames = 29
    S = 4096 + (frames - 1) * 1024
    x = synth.gen(10, 48000, S, 4)
    x[:, 12000:20000] = 0
    r = po.spectrogram(p, x, want_lines=True, want_mapped=True)
    P, C = 333, 2
    m = r["mapped"].reshape(frames, C, 2, P)[:, :, 0, :]              # csp[0..P) = (mag, cancellation) pairs
    mapped = np.stack([m.real, m.imag], axis=2).astype(np.float32)   # planes: magnitude, cancellation
    plan = api.Plan(cfg).upload()
    state = torch.zeros((C, 2, P, 2), dtype=torch.float32, device=gpu)
    rgba, lines = plan.stage_decay_colour(torch.from_numpy(mapped).to(gpu), want_lines=True, state=state)
    lines = lines.cpu().numpy()
    ref = np.stack([r["lines"].real, r["lines"].imag], axis=-1).astype(np.float32)
    # given identical mapped pixels the filters, the dB map (glibc's logf on the device) and the colours are bit-identical
    assert np.array_equal(lines.view(np.uint32), ref.view(np.uint32)), np.abs(lines - ref).max()
    assert np.array_equal(rgba.cpu().numpy(), r["rgba"])


@pytest.mark.parametrize("interp", [config.INTERP_NONE, config.INTERP_LINEAR, config.INTERP_LANCZOS])
def test_phase_end_to_end(gpu, oracle, interp):
    po = oracle
    cfg = _cfg(interp, P=400)
    frames = 9
    x = synth.gen(14, 48000, 4096 + (frames - 1) * 1024, 2)
    # the parity chain (tests/parity_chain.py): mapped magnitudes / cancellations within the FFT tolerance of the oracle's -- an
    # arg-max pixel that differs more must be a verified near-tie in the oracle's own bins -- and the colours exact given them
    from parity_chain import check_render
    problems, stats = check_render(po, api.Plan(cfg).upload(), cfg, x, gpu, want_lines=True)
    assert not problems, (problems[:5], stats)


@pytest.mark.parametrize("frames", [1, 9, 70])
def test_phase_colour_only_render(gpu, oracle, frames):
    """the image alone (no line results, no state): the main graph's magnitude state is a plain peak decay of plane 0 x 0.5, so the
    request runs the one-launch chunked K_B instead of the sequential walk the phase smoother needs -- same bytes as that walk and as the
    oracle given the mapped planes"""
    import torch
    po = oracle
    cfg = _cfg(config.INTERP_LANCZOS, P=400)
    x = synth.gen(15, 48000, 4096 + (frames - 1) * 1024, 2)
    plan = api.Plan(cfg).upload()
    xg = torch.from_numpy(x).to(gpu)
    fused = plan.render(xg).cpu().numpy()
    lines = torch.empty((fused.shape[0], plan.C, 2, plan.P, 2), dtype=torch.float32, device=gpu)
    walked = plan.render(xg, lines=lines).cpu().numpy()               # (lines requested: the sequential kernels)
    assert np.array_equal(fused, walked)
    from parity_chain import check_render
    problems, stats = check_render(po, plan, cfg, x, gpu, want_lines=False)
    assert not problems, (problems[:5], stats)


def test_phase_carry_fold_covers_the_image_only(gpu):
    """the magnitude half of the Phase state is a peak decay and folds exactly (the image is coloured from it alone); the cancellation
    smoother is a linear recurrence without an exact fold: line results / end state from a folded carry are refused"""
    import torch
    plan = api.Plan(_cfg(config.INTERP_LINEAR, P=64)).upload()
    aggs = torch.zeros((2, 1, 2, 64, 2), dtype=torch.float32, device=gpu)
    carry = torch.zeros((1, 2, 64, 2), dtype=torch.float32, device=gpu)
    plan.fold_carry(aggs, [4, 4], 1, carry)
    mapped = torch.rand((4, 1, 2, 64), dtype=torch.float32, device=gpu)
    end = torch.zeros((1, 2, 64, 2), dtype=torch.float32, device=gpu)
    lines = torch.zeros((4, 1, 2, 64, 2), dtype=torch.float32, device=gpu)
    s = torch.cuda.current_stream().cuda_stream
    api.check(api.lib().sgz_stage_decay_scan(plan.h, mapped.data_ptr(), 4, end.data_ptr(), s))
    st = api.lib().sgz_stage_decay_emit(plan.h, mapped.data_ptr(), 4, carry.data_ptr(), None, lines.data_ptr(), None, s)
    assert st == api.SGZ_EUNSUPPORTED
