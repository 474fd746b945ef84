"""GPU parity tests of the Spectrum path: HIP kernels (through the C ABI) vs the CPU oracle.

Bars (north star): bit-exact for the integer colour map (and every stage whose inputs are identical),
a stated fp32 tolerance for spectral magnitudes.
  * bins (window x FFT x split x |.|): |gpu - oracle| <= 4e-6 * max|X| per bin   (different but correct
    fp32 butterfly orders; the oracle itself is 2e-7*max away from numpy fp64)
  * pixel mapping given identical bins: bit-exact
  * decay + dB + colour given identical mapped magnitudes: bit-exact, lines and RGBA8 (std::log(float) is glibc's logf
    algorithm on the device, checked against libm over every positive float)
  * end to end: RGBA8 channel values differ by at most 1 LSB on at most 0.5 % of the bytes
"""
import numpy as np
import pytest

from signalizer_amd import api, config, synth

pytestmark = pytest.mark.gpu

BIN_TOL = 4e-6


def _chain(po, plan, cfg, x, gpu, want_lines=False):
    """the end-to-end bar of every named configuration: the parity chain (tests/parity_chain.py) -- mapped pixels within the FFT's
    tolerance of the oracle's, colours (and lines) byte for byte given the HIP path's own pixels -- not a fraction of differing bytes"""
    from parity_chain import check_render
    problems, stats = check_render(po, plan, cfg, x, gpu, want_lines=want_lines)
    assert not problems, problems
    return stats


def _planar_cuda(x, gpu):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to(gpu)


def _oracle_bins(po, p, x, frames, hop, W):
    """oracle csf[0..N] after split/abs for each frame (pair 0.. )"""
    C = p.num_pairs
    N = po.lib().sgzo_transform_size(W)
    out = np.zeros((frames, C, N + 1), np.float32)
    raws = []
    for f in range(frames):
        for c in range(C):
            L = x[2 * c, f * hop:f * hop + W]
            R = x[2 * c + 1, f * hop:f * hop + W]
            raw, csf, csp = po.frame_bins(p, L, R)
            out[f, c] = csf.real
            if p.channel_mode == config.CH_COMPLEX:
                out[f, c, 0] = np.abs(csf[0])      # Complex keeps csf[0] = 0.5 Z[0] complex (TransformDSP.inl:993); the hook reports |.|
    return out


@pytest.mark.parametrize("cfgname", ["cfg1", "cfg2small", "midside", "left", "w3000", "n32", "n1024", "n8192", "n65536", "w100",
                                     "n65536pad", "n8192sep", "n65536midside", "n8192pad_left", "n65536complex"])
def test_bins_tolerance(gpu, oracle, cfgname):
    po = oracle
    cfg = {
        "cfg1": config.cfg1(),
        "cfg2small": config.cfg2(),
        "midside": config.spectrum_config(channel_mode=config.CH_MIDSIDE, window_type=config.WIN_BLACKMAN_HARRIS),
        "left": config.spectrum_config(channel_mode=config.CH_LEFT, window_size=4096, hop=1024),
        "w3000": config.spectrum_config(window_size=3000, hop=750, window_type=config.WIN_KAISER, window_beta=8.0,
                                        window_symmetry=config.WIN_SYMMETRIC),
        # generic path (spectrum_generic.hip): every other power-of-two transform size
        "n32": config.spectrum_config(window_size=20, hop=7, axis_points=16),
        "n1024": config.spectrum_config(window_size=1024, hop=256, channel_mode=config.CH_MIDSIDE),
        "n8192": config.spectrum_config(window_size=8192, hop=2048, channel_mode=config.CH_MERGE),
        "n65536": config.cfg5(pairs=1),
        "w100": config.spectrum_config(window_size=100, hop=50, window_type=config.WIN_TRIANGULAR, axis_points=33),
        # N = 2 R^3 (two half-frame workgroups, stft_body.hpp HALF): zero-padded windows and every channel-mix family
        "n65536pad": config.spectrum_config(window_size=40000, hop=9000, sample_rate=96000.0),
        "n8192sep": config.spectrum_config(window_size=8192, hop=2048),
        "n65536midside": config.spectrum_config(window_size=65536, hop=16384, channel_mode=config.CH_MIDSIDE,
                                                window_type=config.WIN_BLACKMAN_HARRIS),
        "n8192pad_left": config.spectrum_config(window_size=5000, hop=1000, channel_mode=config.CH_LEFT),
        "n65536complex": config.spectrum_config(window_size=65536, hop=16384, channel_mode=config.CH_COMPLEX),
    }[cfgname]
    W, hop = cfg["window_size"], cfg["hop"]
    frames = 3
    S = W + (frames - 1) * hop
    x = synth.gen(11, cfg["sample_rate"], S, 2)
    p = po.params_from_dict(cfg)
    plan = api.Plan(cfg).upload()
    bins = plan.stage_bins(_planar_cuda(x, gpu)).cpu().numpy()
    ref = _oracle_bins(po, p, x, frames, hop, W)
    assert bins.shape == ref.shape
    if plan.sides == 1:
        # mono modes: the oracle keeps bins >= N/2 as raw complex (TransformDSP.inl:557-560); compare the |.| region
        nb = plan.N // 2
        bins, ref = bins[..., :nb], ref[..., :nb]
    err = np.abs(bins - ref).max()
    scale = np.abs(ref).max()
    assert err <= BIN_TOL * scale, (err, scale, err / scale)


@pytest.mark.parametrize("interp", [config.INTERP_NONE, config.INTERP_LINEAR, config.INTERP_LANCZOS])
@pytest.mark.parametrize("view", [config.VIEW_LOG, config.VIEW_LINEAR])
def test_mapping_bit_exact_given_bins(gpu, oracle, interp, view):
    """mapToLinearSpace (TransformDSP.inl:871-985) on identical csf magnitudes must match bit for bit."""
    import torch
    po = oracle
    cfg = config.spectrum_config(window_size=4096, hop=4096, bin_interp=interp, view_scaling=view, axis_points=700)
    p = po.params_from_dict(cfg)
    x = synth.gen(5, 48000, 4096 * 2, 2)
    plan = api.Plan(cfg).upload()
    frames = 2
    csfs = np.zeros((frames, 1, plan.N + 1), np.float32)
    want = np.zeros((frames, 1, 2, plan.P), np.float32)
    for f in range(frames):
        raw, csf, csp = po.frame_bins(p, x[0, f * 4096:(f + 1) * 4096], x[1, f * 4096:(f + 1) * 4096])
        csfs[f, 0] = csf.real
        v = csp.reshape(2, plan.P)
        want[f, 0] = np.sqrt((v.real * v.real + v.imag * v.imag).astype(np.float32)).astype(np.float32)
    got = plan.stage_map_from_bins(torch.from_numpy(csfs).to(gpu)).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()


def test_decay_colour_bit_exact_given_mapped(gpu, oracle):
    """mapAndTransformDFTFilters + blendAndDispatchSpectrums on identical mapped magnitudes."""
    import torch
    po = oracle
    cfg = config.spectrum_config(window_size=4096, hop=1024, num_pairs=3, axis_points=333,
                                 ratios=(0.1, 0.3, 0.2, 0.25, 0.15), low_db=-100.0, high_db=6.0)
    p = po.params_from_dict(cfg)
    frames = 37                                    # > 4 time chunks: exercises the exact carry fix-up
    S = 4096 + (frames - 1) * 1024
    x = synth.gen(9, 48000, S, 6)
    x[:, 20000:30000] = 0                          # silence: pure decay + clip path
    r = po.spectrogram(p, x, want_lines=True, want_mapped=True)
    P, C = 333, 3
    mapped = r["mapped"].reshape(frames, C, 2, P)
    mag = np.sqrt((mapped.real ** 2 + mapped.imag ** 2).astype(np.float32)).astype(np.float32)
    plan = api.Plan(cfg).upload()
    rgba, lines = plan.stage_decay_colour(torch.from_numpy(mag).to(gpu), want_lines=True)
    rgba = rgba.cpu().numpy()
    lines = lines.cpu().numpy()                    # [F][C][G][P][2]
    ref_lines = r["lines"]                         # [F][C][G][P] complex (left, right)
    ref = np.stack([ref_lines.real, ref_lines.imag], axis=-1).astype(np.float32)
    # std::log(float) is glibc's logf algorithm on the device (decay_body.hpp glibcLogf): line values and colours are bit-identical
    assert np.array_equal(lines.view(np.uint32), ref.view(np.uint32)), np.abs(lines - ref).max()
    assert np.array_equal(rgba, r["rgba"])


@pytest.mark.parametrize("pairs,frames,P", [(1, 1, 1024), (1, 8, 1024), (1, 9, 333), (1, 65, 1024), (1, 348, 1024), (1, 512, 40), (1, 513, 1024),
                                              (2, 17, 1024), (9, 70, 333), (32, 24, 96)])
def test_decay_colour_every_kernel(gpu, oracle, pairs, frames, P):
    """K_B's launch forms on identical magnitudes, bit for bit against the oracle: the fused colour kernel and the fused kernel with
    lines + state (one pair, <= 64 chunks: XCD-ordered pixel groups, emissions pipelined beside the fold -- 1 frame, whole and partial
    chunks, the 64-chunk limit), the scan / emit kernels beyond it, and the several-pairs emit kernel (partial passes of 8 pairs)."""
    import torch
    po = oracle
    cfg = config.spectrum_config(window_size=512, hop=128, num_pairs=pairs, axis_points=P, low_db=-90.0, high_db=3.0)
    p = po.params_from_dict(cfg)
    S = 512 + (frames - 1) * 128
    x = synth.gen(31 + pairs, 48000, S, 2 * pairs)
    x[:, S // 3: S // 2] = 0                        # silence: pure decay + clip path
    r = po.spectrogram(p, x, want_lines=True, want_mapped=True)
    mapped = r["mapped"].reshape(frames, pairs, 2, P)
    mag = torch.from_numpy(np.sqrt((mapped.real ** 2 + mapped.imag ** 2).astype(np.float32)).astype(np.float32)).to(gpu)
    plan = api.Plan(cfg).upload()
    ref = np.stack([r["lines"].real, r["lines"].imag], axis=-1).astype(np.float32)
    img, _ = plan.stage_decay_colour(mag)                                         # image only
    assert np.array_equal(img.cpu().numpy(), r["rgba"])
    for px in (8, 16):                                                            # the fused colour kernel with 8 / 16 pixels per workgroup (sgz_render_queue's lanes)
        wide = api.Plan(cfg)
        wide.set_option(api.OPT_FUSED_COLOUR, px)
        img_w, _ = wide.upload().stage_decay_colour(mag)
        assert np.array_equal(img_w.cpu().numpy(), r["rgba"]), px
    state = torch.zeros((pairs, 2, P, 2), dtype=torch.float32, device=gpu)
    img2, lines = plan.stage_decay_colour(mag, want_lines=True, state=state)      # with lines and state
    assert np.array_equal(img2.cpu().numpy(), r["rgba"])
    lines = lines.cpu().numpy()
    assert np.array_equal(lines.view(np.uint32), ref.view(np.uint32)), np.abs(lines - ref).max()
    # the end state continues the recurrence: two calls with the state carried == the one call (which equals the oracle)
    if frames >= 2:
        h = frames // 2
        st = torch.zeros((pairs, 2, P, 2), dtype=torch.float32, device=gpu)
        a_img, a_lines = plan.stage_decay_colour(mag[:h].contiguous(), want_lines=True, state=st)
        b_img, b_lines = plan.stage_decay_colour(mag[h:].contiguous(), want_lines=True, state=st)
        assert np.array_equal(torch.cat([a_img, b_img]).cpu().numpy(), r["rgba"])
        assert np.array_equal(torch.cat([a_lines, b_lines]).cpu().numpy().view(np.uint32), ref.view(np.uint32))
        assert torch.equal(st, state)


def test_logf_equals_libm_over_every_positive_float(gpu, oracle):
    """dB map's std::log(float): the device port of glibc's logf against libm's logf (what the oracle -- and the reference on
    Linux -- call), over all 2^31 - 2^23 - 1 positive finite floats, bit for bit."""
    import torch
    po = oracle
    step = 1 << 26
    y = torch.empty(step, dtype=torch.float32, device=gpu)
    for lo in range(0, 0x7f800000, step):
        hi = min(lo + step, 0x7f800000)
        bits = torch.arange(max(lo, 1), hi, dtype=torch.int32, device=gpu)
        x = bits.view(torch.float32)
        api.check(api.lib().sgz_stage_logf(x.data_ptr(), y.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream))
        got = y[:x.numel()].cpu().numpy()
        want = po.logf(x.cpu().numpy())
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), lo
    # +inf and the first normal / largest subnormal neighbourhood were inside the sweep; inf itself:
    xi = torch.tensor([float("inf")], dtype=torch.float32, device=gpu)
    api.check(api.lib().sgz_stage_logf(xi.data_ptr(), y.data_ptr(), 1, torch.cuda.current_stream().cuda_stream))
    assert np.isposinf(y[:1].cpu().numpy()[0])


def test_finish_pixel_equals_sqrt_of_square_over_every_float(gpu):
    """K_A's last step per pixel, sqrt(x * x + 0) (TransformDSP.inl:1331 with im == 0), is evaluated as |x| where the square is a normal
    float: bit-identical to the correctly rounded root of the rounded square for EVERY float (both signs, denormals, inf; NaN stays NaN)."""
    import torch
    step = 1 << 26
    y = torch.empty(step, dtype=torch.float32, device=gpu)
    for lo in range(0, 1 << 32, step):
        bits = torch.arange(lo, lo + step, dtype=torch.int64, device=gpu).to(torch.int32)      # wraps into the negative patterns
        x = bits.view(torch.float32)
        api.check(api.lib().sgz_stage_finish_pixel(x.data_ptr(), y.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream))
        got = y.cpu().numpy()
        xc = x.cpu().numpy()
        with np.errstate(over="ignore", invalid="ignore", under="ignore"):
            want = np.sqrt((xc * xc + np.float32(0)).astype(np.float32))
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(got), nan), lo
        assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32)), lo


def test_end_to_end_cfg1(gpu, oracle):
    po = oracle
    cfg = config.cfg1()
    x = synth.gen(1, 48000, 4096, 2)
    r = po.spectrogram(po.params_from_dict(cfg), x)
    plan = api.Plan(cfg).upload()
    rgba = plan.render(_planar_cuda(x, gpu)).cpu().numpy()
    _chain(po, plan, cfg, x, gpu)
    diff = np.abs(rgba.astype(int) - r["rgba"].astype(int))
    assert diff.max() <= 1, diff.max()                                    # (raw bytes against the oracle's own render: a sanity line, the bar is the chain)


def test_end_to_end_cfg2_8frames_and_host_wrapper(gpu, oracle):
    po = oracle
    cfg = config.cfg2()
    S = 32768 + 7 * 8192
    x = synth.gen(2, 48000, S, 2)
    r = po.spectrogram(po.params_from_dict(cfg), x, want_lines=True)
    rgba, lines, timing = api.render_spectrogram(cfg, x, want_lines=True)   # host-buffer C entry point
    assert timing["frames"] == 8
    diff = np.abs(rgba.astype(int) - r["rgba"].astype(int))
    assert diff.max() <= 1, diff.max()                                    # (sanity line; the bar is the chain below)
    ref = np.stack([r["lines"].real, r["lines"].imag], axis=-1)
    ok = ref > -100                                                       # not the clip sentinel
    assert np.abs(lines - ref)[ok].max() <= 2e-4                          # normalised dB units (1.0 = 120 dB)
    # the plan-keeping form: same bytes as the device render, call after call, on buffers of changing length and odd row lengths
    plan = api.Plan(cfg).upload()
    _chain(po, plan, cfg, x, gpu, want_lines=True)
    for n in (S, S - 8192 - 3, S):
        xs = np.ascontiguousarray(x[:, :n])
        got, glines, t = api.render_spectrogram_host(plan, xs, want_lines=True)
        want = plan.render(_planar_cuda(xs, gpu)).cpu().numpy()
        assert t["frames"] == want.shape[0] and np.array_equal(got, want)
        assert np.isfinite(glines).all()
    assert np.array_equal(got, rgba)


def test_multi_pair_blend(gpu, oracle):
    po = oracle
    cfg = config.spectrum_config(window_size=4096, hop=2048, num_pairs=4, axis_points=257)
    x = synth.gen(21, 48000, 4096 + 5 * 2048, 8)
    r = po.spectrogram(po.params_from_dict(cfg), x)
    plan = api.Plan(cfg).upload()
    rgba = plan.render(_planar_cuda(x, gpu)).cpu().numpy()
    _chain(po, plan, cfg, x, gpu)
    diff = np.abs(rgba.astype(int) - r["rgba"].astype(int))
    assert diff.max() <= 1, diff.max()                                    # (sanity line)


def test_state_carry_across_calls(gpu, oracle):
    """rendering [0,T) in one call == rendering two halves with the decay state carried (exactly)."""
    import torch
    cfg = config.spectrum_config(window_size=4096, hop=1024)
    plan = api.Plan(cfg).upload()
    frames = 24
    S = 4096 + (frames - 1) * 1024
    x = _planar_cuda(synth.gen(3, 48000, S, 2), gpu)
    full = plan.render(x).cpu().numpy()
    state = torch.zeros((1, 2, plan.P, 2), dtype=torch.float32, device=gpu)
    h = 11
    a = plan.render(x[:, :4096 + (h - 1) * 1024].contiguous(), state=state).cpu().numpy()
    b = plan.render(x[:, h * 1024:].contiguous(), state=state).cpu().numpy()
    assert np.array_equal(np.concatenate([a, b]), full)


@pytest.mark.parametrize("pairs,frames", [(1, 61), (3, 37), (2, 5)])
def test_decay_state_only_pass(gpu, pairs, frames):
    """A state-only K_B pass (no colour, no lines: what every rank runs for the multi-GPU carry exchange) leaves exactly
    the end state of the full pass, from a non-zero carry-in state."""
    import torch
    cfg = config.spectrum_config(window_size=4096, hop=1024, num_pairs=pairs, axis_points=333)
    plan = api.Plan(cfg).upload()
    S = 4096 + (frames - 1) * 1024
    x = _planar_cuda(synth.gen(5, 48000, S, 2 * pairs), gpu)
    mapped = plan.stage_mapped(x)
    state = torch.full((pairs, 2, plan.P, 2), 0.01, dtype=torch.float32, device=gpu)
    plan.stage_decay_colour(mapped, state=state, want_lines=True)
    state2 = torch.full((pairs, 2, plan.P, 2), 0.01, dtype=torch.float32, device=gpu)
    plan.stage_decay_colour(mapped, state=state2, want_rgba=False)
    assert np.array_equal(state2.cpu().numpy(), state.cpu().numpy())


def test_full_size_cfg2_properties(gpu):
    """BASELINE cfg2 at full size (348 frames): size-independent properties.
    (i) shift: frames [k, k+m) of the full render == render of the shifted buffer when decay is off;
    (ii) linearity in dB: scaling the input by 0.5 moves every unclipped line value by 20log10(0.5)/120."""
    import torch
    cfg = config.cfg2()
    cfg["pole"] = (0.0, 0.0)
    S = int(config.CFG2_SECONDS * 48000)
    x = _planar_cuda(synth.gen(config.CFG2_SEED, 48000, S, 2), gpu)
    plan = api.Plan(cfg).upload()
    F = plan.num_frames(S)
    assert F == 348
    lines = torch.empty((F, 1, 2, plan.P, 2), dtype=torch.float32, device=gpu)
    full = plan.render(x, lines=lines).cpu().numpy()
    k, m = 100, 16
    sub = plan.render(x[:, k * 8192:k * 8192 + 32768 + (m - 1) * 8192].contiguous()).cpu().numpy()
    assert np.array_equal(sub, full[k:k + m])
    lines2 = torch.empty_like(lines)
    plan.render((x * 0.5).contiguous(), lines=lines2)
    a, b = lines.cpu().numpy(), lines2.cpu().numpy()
    ok = (a > -1) & (b > -1)
    assert np.abs((a - b)[ok] - 20 * np.log10(2.0) / 120.0).max() < 1e-4
    assert full[..., 3].min() == 255


@pytest.mark.parametrize("W,hop,pairs,P", [(65536, 16384, 2, 1024), (1024, 256, 3, 200), (2048, 512, 1, 128), (600, 100, 1, 77)])
def test_end_to_end_generic_sizes(gpu, oracle, W, hop, pairs, P):
    """window sizes outside the fused kernel (incl. BASELINE cfg5's N = 65536) through the generic path"""
    po = oracle
    cfg = config.spectrum_config(sample_rate=96000.0, window_size=W, hop=hop, num_pairs=pairs, axis_points=P)
    frames = 5
    x = synth.gen(31, 96000, W + (frames - 1) * hop, 2 * pairs)
    r = po.spectrogram(po.params_from_dict(cfg), x, want_lines=True)
    plan = api.Plan(cfg).upload()
    rgba = plan.render(_planar_cuda(x, gpu)).cpu().numpy()
    diff = np.abs(rgba.astype(int) - r["rgba"].astype(int))
    assert rgba.shape == r["rgba"].shape
    _chain(po, plan, cfg, x, gpu)
    assert diff.max() <= 1, diff.max()                                    # (sanity line)


@pytest.mark.parametrize("W,pairs,frames,P,mode", [(8192, 3, 1500, 300, config.CH_SEPARATE), (65536, 2, 6, 4000, config.CH_SEPARATE),
                                                  (8192, 2, 9, 256, config.CH_COMPLEX), (7000, 1, 7, 512, config.CH_MIDSIDE)])
def test_end_to_end_halves(gpu, oracle, W, pairs, frames, P, mode):
    """N = 2 R^3 (two half-frame workgroups + map kernel): several slabs of tasks walked pair-major, a view too tall for the
    LDS-staged map (falls back to the generic map kernel), Complex (whole-spectrum view), zero-padded window"""
    po = oracle
    hop = W // 4
    cfg = config.spectrum_config(sample_rate=96000.0, window_size=W, hop=hop, num_pairs=pairs, axis_points=P, channel_mode=mode)
    x = synth.gen(37, 96000, W + (frames - 1) * hop, 2 * pairs)
    r = po.spectrogram(po.params_from_dict(cfg), x)
    plan = api.Plan(cfg).upload()
    rgba = plan.render(_planar_cuda(x, gpu)).cpu().numpy()
    diff = np.abs(rgba.astype(int) - r["rgba"].astype(int))
    assert rgba.shape == r["rgba"].shape
    _chain(po, plan, cfg, x, gpu)
    assert diff.max() <= 1, diff.max()                                    # (sanity line)


@pytest.mark.parametrize("W", [4096, 2048, 8192, 32768])
@pytest.mark.parametrize("interp,view", [(config.INTERP_LANCZOS, config.VIEW_LOG), (config.INTERP_LINEAR, config.VIEW_LINEAR),
                                         (config.INTERP_NONE, config.VIEW_LOG)])
def test_complex_mode_dc_bin(gpu, oracle, W, interp, view):
    """SpectrumChannels::Complex keeps csf[0] complex (TransformDSP.inl:993): the pixels whose filter window reaches bin 0 are
    complex sums.  Fused (4096, 32768), halves (8192) and generic (2048) paths against the oracle's mapped values."""
    po = oracle
    P, frames = 300, 3
    cfg = config.spectrum_config(sample_rate=96000.0, window_size=W, hop=W // 4, axis_points=P, channel_mode=config.CH_COMPLEX,
                                 bin_interp=interp, view_scaling=view)
    x = synth.gen(41, 96000, W + (frames - 1) * (W // 4), 2)
    x += 0.25                                            # a DC offset: makes csf[0] large
    x[1] -= 0.6
    r = po.spectrogram(po.params_from_dict(cfg), x, want_mapped=True)
    m = r["mapped"][:, :, :P]
    ref = np.sqrt((m.real.astype(np.float32) ** 2 + m.imag.astype(np.float32) ** 2).astype(np.float32))
    plan = api.Plan(cfg).upload()
    got = plan.stage_mapped(_planar_cuda(x, gpu)).cpu().numpy()[:, :, 0, :]
    if interp != config.INTERP_NONE:
        assert (np.abs(m.imag) > 0).any()                # the case under test exists in this view
    err = np.abs(got - ref).max()
    assert err <= BIN_TOL * np.abs(ref).max(), (err, np.abs(ref).max())


def test_cfg5_defining_shape_against_the_oracle(gpu, oracle):
    """BASELINE configs[4] in its defining form on one GPU: 64 channels = 32 pairs, 96 kHz, N = W = 65536, hop 16384, and the 7.5 s
    one rank of eight owns (40 frames => 1280 transforms through the halves path, all 32 pairs blended into each column),
    against the oracle by the parity chain: mapped pixels within the FFT tolerance, colour bytes exact given them."""
    from parity_chain import check_render
    cfg = config.cfg5()
    S = int(7.5 * 96000)
    x = synth.gen(5, 96000, S, 64)
    plan = api.Plan(cfg).upload()
    assert plan.N == 65536 and plan.C == 32 and (plan.path & 7) == 2 | 4 and plan.num_frames(S) == 40
    problems, stats = check_render(oracle, plan, cfg, x, gpu, want_lines=True)
    assert not problems, (problems[:5], stats)
    assert stats["max_byte_diff"] <= 1 and stats["frac"] <= 5e-3, stats        # raw bytes against the oracle's own render


def test_full_size_cfg5_properties(gpu):
    """BASELINE cfg5 sizes on one GPU (N = 65536, 8 of the 32 pairs, 10 s => 55 frames per pair, several slabs of the halves
    path): (i) shift: frames [k, k+m) of the full render == render of the shifted buffer when decay is off; (ii) a pair rendered
    alone equals its lines inside the multi-pair render (the blend is the only cross-pair step); (iii) dB linearity."""
    import torch
    cfg = config.cfg5(pairs=8)
    cfg["pole"] = (0.0, 0.0)
    S = 10 * 96000
    x = _planar_cuda(synth.gen(5, 96000, S, 16), gpu)
    plan = api.Plan(cfg).upload()
    assert (plan.path & 7) == 2 | 4
    F = plan.num_frames(S)
    assert F == 55
    lines = torch.empty((F, 8, 2, plan.P, 2), dtype=torch.float32, device=gpu)
    full = plan.render(x, lines=lines).cpu().numpy()
    k, m = 20, 7
    sub = plan.render(x[:, k * 16384:k * 16384 + 65536 + (m - 1) * 16384].contiguous()).cpu().numpy()
    assert np.array_equal(sub, full[k:k + m])
    one = dict(cfg, num_pairs=1)
    plan1 = api.Plan(one).upload()
    lines1 = torch.empty((F, 1, 2, plan.P, 2), dtype=torch.float32, device=gpu)
    plan1.render(x[6:8].contiguous(), lines=lines1)
    assert np.array_equal(lines1.cpu().numpy()[:, 0], lines.cpu().numpy()[:, 3])
    lines2 = torch.empty_like(lines)
    plan.render((x * 0.5).contiguous(), lines=lines2)
    a, b = lines.cpu().numpy(), lines2.cpu().numpy()
    ok = (a > -1) & (b > -1)
    assert np.abs((a - b)[ok] - 20 * np.log10(2.0) / 120.0).max() < 1e-4


@pytest.mark.parametrize("W,P", [(32768, 3000), (4096, 2500)])
def test_tall_view_serial_map_fallback(gpu, oracle, W, P):
    """views whose arg-max pieces do not fit in LDS beside the csf array: the fused kernel's serial per-pixel scan"""
    po = oracle
    cfg = config.spectrum_config(window_size=W, hop=W // 4, axis_points=P)
    frames = 3
    x = synth.gen(43, 48000, W + (frames - 1) * (W // 4), 2)
    r = po.spectrogram(po.params_from_dict(cfg), x, want_mapped=True)
    m = r["mapped"].reshape(frames, 1, 2, P)
    ref = np.sqrt((m.real.astype(np.float32) ** 2 + m.imag.astype(np.float32) ** 2).astype(np.float32))
    got = api.Plan(cfg).upload().stage_mapped(_planar_cuda(x, gpu)).cpu().numpy()
    assert np.abs(got - ref).max() <= BIN_TOL * np.abs(ref).max()


def test_degenerate_inputs(gpu):
    """a buffer shorter than one window renders zero frames (the reference skips the frame, TransformDSP.inl:45-46); exactly
    one window renders one; a ragged tail is ignored"""
    import torch
    for W in (4096, 8192, 2048):
        plan = api.Plan(config.spectrum_config(window_size=W, hop=W // 4)).upload()
        x = torch.zeros((2, W - 1), dtype=torch.float32, device=gpu)
        assert plan.num_frames(W - 1) == 0 and plan.render(x).shape[0] == 0
        y = torch.from_numpy(synth.gen(3, 48000, W + W // 4 + 17, 2)).to(gpu)
        full = plan.render(y).cpu().numpy()
        assert full.shape[0] == 2
        assert np.array_equal(plan.render(y[:, :W].contiguous()).cpu().numpy(), full[:1])


def test_unsupported_and_errors(gpu):
    with pytest.raises(api.SgzError):
        api.Plan(config.spectrum_config(axis_points=1))


@pytest.mark.parametrize("N,sr,pairs,frames,over", [
    (32768, 48000.0, 1, 5, {}), (32768, 48000.0, 3, 4, {}), (65536, 96000.0, 2, 3, {}), (16384, 24000.0, 1, 9, {}), (16384, 24000.0, 2, 6, {}),
    # tap windows that reach over bin 0 into the other channel's entries (settled by the later workgroup): the reference's default view
    # at N = 16384 / 48 kHz, linear views from 0 Hz, linear interpolation
    (16384, 48000.0, 1, 9, {}), (16384, 44100.0, 2, 5, {}), (32768, 48000.0, 2, 4, dict(view_scaling=0)),
    (32768, 48000.0, 1, 4, dict(view_scaling=0, bin_interp=1)), (65536, 96000.0, 1, 3, dict(view_scaling=0)),
    # MidSide: the same two workgroups on (l + r) / 2 and (l - r) / 2
    (32768, 48000.0, 2, 5, dict(channel_mode=config.CH_MIDSIDE)), (65536, 96000.0, 1, 3, dict(channel_mode=config.CH_MIDSIDE)),
    (16384, 48000.0, 1, 7, dict(channel_mode=config.CH_MIDSIDE)),
    # the 1024-thread form of the N = 32768 kernel (spectrum_real16.hip; plan option SGZ_OPT_WIDE_GROUPS)
    (32768, 48000.0, 1, 5, dict(wide=1)), (32768, 48000.0, 3, 4, dict(wide=1)), (32768, 48000.0, 2, 4, dict(view_scaling=0, wide=1)),
    (32768, 48000.0, 2, 5, dict(channel_mode=config.CH_MIDSIDE, wide=1)), (32768, 44100.0, 1, 4, dict(window_type=config.WIN_BLACKMAN_HARRIS, wide=1))])
def test_channel_split_kernel_against_the_oracle(gpu, oracle, monkeypatch, N, sr, pairs, frames, over):
    """spectrum_real.hip (one workgroup per (frame, pair, channel), real-input FFT; the default at N = 16384 and 65536, forced here at
    N = 32768 too) through the parity chain, and bin for bin against the whole-frame kernels: same csf within the FFT tolerance -- including
    csf[0], csf[N], csf[N/2 - 1] (quirk Q3) and csf[N/2], the one entry that needs both channels and is settled by whichever
    workgroup finishes second -- and identical pixels given identical bins."""
    from parity_chain import check_render
    over = dict(over)
    wide = over.pop("wide", 0)
    cfg = config.spectrum_config(sample_rate=sr, window_size=N, hop=N // 4, num_pairs=pairs, **over)
    S = N + (frames - 1) * (N // 4)
    x = synth.gen(23, int(sr), S, 2 * pairs)
    split = api.Plan(cfg).set_option(api.OPT_WIDE_GROUPS, wide).upload()
    whole = api.Plan(cfg).set_option(api.OPT_CHANNEL_SPLIT, 0).upload()
    assert split.path & 8 and not whole.path & 8
    xg = _planar_cuda(x, gpu)
    a, b = split.stage_bins(xg).cpu().numpy(), whole.stage_bins(xg).cpu().numpy()
    assert np.abs(a - b).max() <= BIN_TOL * np.abs(b).max()
    for k in (0, N, N // 2, N // 2 - 1):
        assert np.abs(a[..., k] - b[..., k]).max() <= BIN_TOL * np.abs(b).max(), k
    problems, stats = check_render(oracle, split, cfg, x, gpu, want_lines=True)
    assert not problems, (problems[:5], stats)
    # digital silence on both channels of a pair: every arg-max run takes the literal-scan fallback (all squares zero), nothing but zeros
    # may come out, and the other pairs are untouched
    x[0] = 0; x[1] = 0
    m = split.stage_mapped(_planar_cuda(x, gpu)).cpu().numpy()
    assert not m[:, 0].any() and np.isfinite(m).all()
    if pairs > 1:
        problems, stats = check_render(oracle, split, cfg, x, gpu)
        assert not problems, (problems[:5], stats)


@pytest.mark.parametrize("N,sr,over", [
    (16384, 48000.0, {}), (32768, 48000.0, {}), (65536, 96000.0, {}),
    (32768, 48000.0, dict(channel_mode=config.CH_MIDSIDE, bin_interp=config.INTERP_LINEAR)),
    # a linear view from 0 Hz: the first pixels' tap windows reach over bin 0 into the other channel's bins (realLateKernel's low pixels)
    (32768, 48000.0, dict(view_scaling=config.VIEW_LINEAR, view_left=0.0, view_right=0.02, axis_points=777)),
    (16384, 48000.0, dict(bin_interp=config.INTERP_NONE, axis_points=2500)),       # more pixels than two per thread: the map's further rounds
    (65536, 96000.0, dict(axis_points=300, min_log_freq=40.0)),                    # long runs: many chunks per pixel
    # the 1024-thread form's map (two lanes per row of 32 magnitudes: spectrum_real16.hip ChunkMap16)
    (32768, 48000.0, dict(wide=1)), (32768, 48000.0, dict(channel_mode=config.CH_MIDSIDE, bin_interp=config.INTERP_LINEAR, wide=1)),
    (32768, 48000.0, dict(view_scaling=config.VIEW_LINEAR, view_left=0.0, view_right=0.02, axis_points=777, wide=1)),
    (32768, 48000.0, dict(bin_interp=config.INTERP_NONE, axis_points=2500, wide=1)), (32768, 48000.0, dict(axis_points=300, min_log_freq=40.0, wide=1))])
def test_channel_split_mapping_bit_exact_given_bins(gpu, oracle, monkeypatch, N, sr, over):
    """Chain link 2 on the kernels the bench runs: sgz_stage_map_from_bins on a channel-split plan feeds the oracle's csf magnitudes to
    realMapFromBinsKernel -- the chunk-scan map, the late-pixel bookkeeping and realLateKernel are the very functions stftRealKernel
    runs behind its transform (spectrum_real.hip realMapSettle) -- and every pixel, incl. the top pixels that csf[N/2] can win and
    the pixels whose taps reach over bin 0, must equal the oracle's mapToLinearSpace (TransformDSP.inl:871-985) bit for bit."""
    import torch
    po = oracle
    over = dict(over)
    wide = over.pop("wide", 0)
    cfg = config.spectrum_config(sample_rate=sr, window_size=N, hop=N // 4, **over)
    p = po.params_from_dict(cfg)
    frames = 3
    x = synth.gen(23, int(sr), N + (frames - 1) * (N // 4), 2)
    x[0, 1::2] -= 0.4                                   # energy at Nyquist in the left channel: csf[N/2] wins the top pixels
    x[0, 0::2] += 0.4
    plan = api.Plan(cfg).set_option(api.OPT_WIDE_GROUPS, wide).upload()
    assert plan.path & 8
    csfs = np.zeros((frames, 1, plan.N + 1), np.float32)
    want = np.zeros((frames, 1, 2, plan.P), np.float32)
    for f in range(frames):
        o = f * (N // 4)
        raw, csf, csp = po.frame_bins(p, x[0, o:o + N], x[1, o:o + N])
        csfs[f, 0] = csf.real
        v = csp.reshape(2, plan.P)
        want[f, 0] = np.sqrt((v.real * v.real + v.imag * v.imag).astype(np.float32)).astype(np.float32)
    got = plan.stage_map_from_bins(torch.from_numpy(csfs).to(gpu)).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (int((got != want).sum()), np.nonzero(got != want)[-1][:8])
    # and the transform kernel itself, through the chain (its own bins within the FFT's tolerance, colours given its pixels byte for byte)
    from parity_chain import check_render
    problems, _ = check_render(po, plan, cfg, x, gpu)
    assert not problems, problems


@pytest.mark.parametrize("N,sr,mode,pairs,over", [
    (32768, 48000.0, config.CH_LEFT, 1, {}), (32768, 48000.0, config.CH_RIGHT, 2, {}), (32768, 48000.0, config.CH_MERGE, 1, {}),
    (32768, 48000.0, config.CH_SIDE, 3, {}), (65536, 96000.0, config.CH_MERGE, 2, {}), (65536, 96000.0, config.CH_LEFT, 1, {}),
    (16384, 24000.0, config.CH_SIDE, 1, {}), (16384, 24000.0, config.CH_RIGHT, 2, {}),
    (32768, 48000.0, config.CH_MERGE, 1, dict(bin_interp=1)), (32768, 44100.0, config.CH_LEFT, 1, dict(window_type=2)),
    # tap windows that wrap below bin 0 (csf[N - j] = conj X[j] stays complex in the mono modes, csf[N] = 0) or reach csf[N/2 ..]:
    # redone from the kernel's complex entries (complex_dc.hpp)
    (16384, 48000.0, config.CH_SIDE, 1, {}), (16384, 48000.0, config.CH_MERGE, 2, dict(bin_interp=1)),
    (32768, 48000.0, config.CH_LEFT, 1, dict(view_scaling=0, view_left=0.0, view_right=1.0)),
    (32768, 48000.0, config.CH_MERGE, 1, dict(view_scaling=0, view_left=0.0, view_right=0.01, bin_interp=1)),
    (65536, 96000.0, config.CH_RIGHT, 1, dict(min_log_freq=2.0)), (16384, 48000.0, config.CH_LEFT, 1, dict(view_scaling=0, view_left=0.9, view_right=1.0))])
def test_mono_modes_on_the_real_input_kernel(gpu, oracle, monkeypatch, N, sr, mode, pairs, over):
    """Left / Right / Merge / Side transform one real signal per frame: spectrum_real.hip's MONO form (one workgroup per (frame, pair)) against
    the oracle through the parity chain, and bin for bin (csf[0 .. N/2], incl. the halved csf[0] and the signed csf[N/2]) against the
    complex whole-frame / halves / generic kernels"""
    from parity_chain import check_render
    cfg = config.spectrum_config(sample_rate=sr, window_size=N, hop=N // 4, num_pairs=pairs, channel_mode=mode, **over)
    frames = 6
    x = synth.gen(29, int(sr), N + (frames - 1) * (N // 4), 2 * pairs)
    real = api.Plan(cfg).upload()
    other = api.Plan(cfg).set_option(api.OPT_CHANNEL_SPLIT, 0).upload()
    assert real.path & 8 and not other.path & 8
    xg = _planar_cuda(x, gpu)
    a, b = real.stage_bins(xg).cpu().numpy()[..., :N // 2 + 1], other.stage_bins(xg).cpu().numpy()[..., :N // 2 + 1]
    # the complex kernels leave csf[N/2] complex; as a real number it is +- its magnitude
    assert np.abs(a[..., :N // 2] - b[..., :N // 2]).max() <= BIN_TOL * np.abs(b).max()
    assert np.abs(np.abs(a[..., N // 2]) - np.abs(b[..., N // 2])).max() <= BIN_TOL * np.abs(b).max()
    problems, stats = check_render(oracle, real, cfg, x, gpu, want_lines=True)
    assert not problems, (problems[:5], stats)
    m1, m2 = real.stage_mapped(xg).cpu().numpy(), other.stage_mapped(xg).cpu().numpy()
    assert np.abs(m1 - m2).max() <= 4e-6 * np.abs(b).max() * real.window_scale / (N * 0.5) * 4


@pytest.mark.parametrize("N,sr,mode", [(32768, 48000.0, config.CH_SEPARATE), (16384, 48000.0, config.CH_MERGE), (65536, 96000.0, config.CH_MIDSIDE)])
def test_result_does_not_depend_on_the_row_layout(gpu, N, sr, mode):
    """rows at their natural 4-byte alignment (odd row stride, base shifted by one sample) run the same real-input kernels as aligned rows:
    identical bits, whatever buffer the caller hands over"""
    import torch
    cfg = config.spectrum_config(sample_rate=sr, window_size=N, hop=N // 4, channel_mode=mode)
    plan = api.Plan(cfg).upload()
    assert plan.path & 8
    S = N + 5 * (N // 4)
    x = torch.from_numpy(synth.gen(41, int(sr), S, 2)).to(gpu)
    even = torch.zeros((2, S + 2), dtype=torch.float32, device=gpu)
    odd = torch.zeros((2, S + 3), dtype=torch.float32, device=gpu)
    even[:, :S] = x
    odd[:, 1:S + 1] = x                                                # odd stride AND a base one sample past an aligned address
    a = plan.stage_mapped(even[:, :S]).cpu().numpy()
    b = plan.stage_mapped(odd[:, 1:S + 1]).cpu().numpy()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(plan.render(even[:, :S]).cpu().numpy(), plan.render(odd[:, 1:S + 1]).cpu().numpy())


@pytest.mark.parametrize("depth", [1, 2, 3, 5])
def test_render_queue_equals_one_render_after_the_other(gpu, depth):
    """sgz_render_queue (round 6): independent buffers submitted round-robin over `depth` lanes -- every image byte for byte the one
    sgz_spectrogram_render_device gives for that buffer alone (decay states from zero), whatever overlaps on the device; tickets,
    wait, join, the input-ready dependency on the caller's stream, and the refusals."""
    import ctypes as C
    import torch
    cfg = config.spectrum_config(window_size=4096, hop=1024, axis_points=300)
    S = 4096 + 1024 * 40
    n = 11
    plan = api.Plan(cfg).upload()
    F = plan.num_frames(S)
    xs = [torch.from_numpy(synth.gen(100 + k, 48000, S, 2)).to(gpu) for k in range(n)]
    want = [plan.render(x).cpu().numpy() for x in xs]
    q = api.RenderQueue(cfg, depth)
    outs = [torch.zeros((F, 300, 4), dtype=torch.uint8, device=gpu) for _ in range(n)]
    torch.cuda.synchronize()                     # the zero fills run on torch's stream: a lane must not write an image before its fill has (sgz.h: after_stream)
    tickets = [q.submit(xs[k], outs[k]) for k in range(n)]
    assert tickets == list(range(1, n + 1))
    q.wait(tickets[0])
    assert np.array_equal(outs[0].cpu().numpy(), want[0])                         # the first one is done; the others may still run
    q.wait()
    for k in range(n):
        assert np.array_equal(outs[k].cpu().numpy(), want[k]), k
    # the samples are produced on the caller's stream right before the submit: the lane must wait for them
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fresh = torch.zeros_like(xs[0])
        big = torch.randn((4096, 4096), device=gpu)
        for _ in range(8):
            big = big @ big * 1e-3                                                 # keep the stream busy in front of the copy
        fresh.copy_(xs[3])
        out = torch.zeros((F, 300, 4), dtype=torch.uint8, device=gpu)
        t = q.submit(fresh, out, after_stream=side.cuda_stream)
        q.join(side.cuda_stream)                                                  # ... and the caller's stream can wait for the image without the host
        got = out.clone()
    side.synchronize()
    assert np.array_equal(got.cpu().numpy(), want[3]) and t == n + 1
    # refusals: an unknown ticket, a buffer shorter than one window (nothing enqueued, no ticket), options once work is in
    assert api.lib().sgz_render_queue_wait(q.h, C.c_uint64(t + 1)) == api.SGZ_EINVAL
    short = torch.zeros((2, 100), dtype=torch.float32, device=gpu)
    tk = C.c_uint64(0)
    assert api.lib().sgz_render_queue_submit(q.h, C.c_void_p(short.data_ptr()), short.stride(0), 100, C.c_void_p(out.data_ptr()), None, C.byref(tk)) == api.SGZ_SKIPPED_FRAME
    assert tk.value == 0
    assert api.lib().sgz_render_queue_set_option(q.h, api.OPT_FUSED_COLOUR, 4) == api.SGZ_EINVAL
    q.close()
    h = C.c_void_p()
    c = api.config_from_dict(cfg)
    assert api.lib().sgz_render_queue_create(C.byref(c), 0, C.byref(h)) == api.SGZ_EINVAL
    assert api.lib().sgz_render_queue_create(C.byref(c), 17, C.byref(h)) == api.SGZ_EINVAL


def test_render_queue_at_the_bench_size(gpu):
    """the queue at BASELINE configs[1] (348 frames, N = 32768: the channel-split K_A + the 16-pixel fused K_B), 9 buffers over 3 lanes"""
    import torch
    cfg = config.cfg2()
    S = 32768 + 8192 * 347
    plan = api.Plan(cfg).upload()
    q = api.RenderQueue(cfg, 3)
    xs = [torch.from_numpy(synth.gen(200 + k, 48000, S, 2)).to(gpu) for k in range(3)]
    want = [plan.render(x).cpu().numpy() for x in xs]
    outs = [torch.zeros((348, 1024, 4), dtype=torch.uint8, device=gpu) for _ in range(9)]
    torch.cuda.synchronize()
    for k in range(9):
        q.submit(xs[k % 3], outs[k])
    q.wait()
    for k in range(9):
        assert np.array_equal(outs[k].cpu().numpy(), want[k % 3]), k
    q.close()


@pytest.mark.parametrize("over", [dict(num_pairs=3, channel_mode=config.CH_MIDSIDE), dict(channel_mode=config.CH_PHASE), dict(channel_mode=config.CH_LEFT, window_size=3000),
                                  dict(algorithm=config.ALGO_RSNT, hop=1024, window_type=config.WIN_HANN), dict(window_size=32768, hop=8192, num_pairs=2)])
def test_render_queue_on_other_plans(gpu, over):
    """the queue is plan-agnostic: several pairs (the scan / emit K_B), Phase, a mono mode with a zero-padded window, the RSNT algorithm
    (every buffer from rest: the resonators' state lives in the lane's plan and is reset per render), N = 32768 with two pairs -- each
    image equals the single render's"""
    import torch
    cfg = config.spectrum_config(**{**dict(window_size=4096, hop=1024, axis_points=257), **over})
    S = cfg["window_size"] + cfg["hop"] * 30
    plan = api.Plan(cfg).upload()
    F = plan.num_frames(S)
    xs = [torch.from_numpy(synth.gen(300 + k, 48000, S, 2 * cfg["num_pairs"])).to(gpu) for k in range(4)]
    want = [plan.render(x).cpu().numpy() for x in xs]
    q = api.RenderQueue(cfg, 3)
    outs = [torch.zeros((F, 257, 4), dtype=torch.uint8, device=gpu) for _ in range(8)]
    torch.cuda.synchronize()
    for k in range(8):
        q.submit(xs[k % 4], outs[k])
    q.wait()
    for k in range(8):
        assert np.array_equal(outs[k].cpu().numpy(), want[k % 4]), k
    q.close()


@pytest.mark.parametrize("size", ["cfg2", "cfg5-like"])
def test_launches_in_flight_on_several_streams_are_bit_identical(gpu, size):
    """Round 6's find.  The barrier behind exchange 1's second store round had no s_waitcnt in front of it (the stores are inline
    assembly: invisible to the compiler's wait counts) -- a wave could pass it with its ds_write_b128 in flight and a reader of another wave
    be served first.  One launch at a time that practically never happened (no fuzz campaign ever saw it); with launches of several
    streams sharing the CUs one workgroup in ~1 000 launches transformed a stale value: a faint broadband error in one frame.  Here:
    K_A (sgz_stage_mapped) of three fixed buffers over four plans / streams, thousands of launches, every output bit for bit the quiet
    run's (fft_common.hpp ldsBarrier; tools/ka_overlap_stress.py is the same loop with diagnostics)."""
    import torch
    if size == "cfg2":
        cfg, S, rounds = config.cfg2(), 32768 + 8192 * 347, 500
    else:                                                             # N = 65536 (the walking kernel, ldsWrite64 stores), two pairs
        cfg, S, rounds = config.cfg5(2), 65536 + 16384 * 99, 150
    xs = [torch.from_numpy(synth.gen(400 + k, int(cfg["sample_rate"]), S, 2 * cfg["num_pairs"])).to(gpu) for k in range(3)]
    ref = api.Plan(cfg).upload()
    want = [ref.stage_mapped(x).clone() for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=gpu) for _ in range(4)]
    plans = [api.Plan(cfg).upload() for _ in range(4)]
    bad = 0
    for r in range(rounds):
        outs = []
        torch.cuda.synchronize()
        for k in range(9):
            with torch.cuda.stream(streams[k % 4]):
                outs.append(plans[k % 4].stage_mapped(xs[k % 3]))
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(outs[k].view(torch.int32), want[k % 3].view(torch.int32)) else 1 for k in range(9))
    assert bad == 0, f"{bad} of {rounds * 9} launches differ from the quiet run"


_OVERLAP_CASES = {
    "real-16384-midside": dict(window_size=16384, hop=4096, channel_mode=config.CH_MIDSIDE),
    "real-mono-merge": dict(channel_mode=config.CH_MERGE),
    "wide-groups": dict(_wide=1),
    "whole-frame-complex": dict(channel_mode=config.CH_COMPLEX),
    "whole-frame-4096-padded": dict(window_size=3000, hop=750),
    "halves-8192": dict(window_size=8192, hop=2048),
    "generic-2048": dict(window_size=2048, hop=512),
    "phase-32768": dict(channel_mode=config.CH_PHASE),
    "rsnt-matrix": dict(algorithm=config.ALGO_RSNT, window_size=4096, hop=1024),
    "three-pairs": dict(window_size=4096, hop=1024, num_pairs=3),
    "fetched-window": dict(window_type=config.WIN_BLACKMAN),
}


@pytest.mark.parametrize("case", sorted(_OVERLAP_CASES))
def test_every_kernel_family_is_bit_identical_with_launches_in_flight(gpu, case):
    """the concurrency axis for the other K_A / K_B forms (tools/overlap_stress_cfgs.py is the long version: 15 configurations x 1 800
    renders, 0 differing): whole renders of three fixed buffers over four plans / streams against the quiet run"""
    import torch
    over = dict(_OVERLAP_CASES[case])
    wide = over.pop("_wide", 0)
    cfg = config.spectrum_config(**over)
    frames = 100 if cfg["window_size"] >= 16384 else 160
    S = cfg["window_size"] + cfg["hop"] * (frames - 1)
    xs = [torch.from_numpy(synth.gen(500 + k, int(cfg["sample_rate"]), S, 2 * cfg["num_pairs"])).to(gpu) for k in range(3)]

    def make():
        p = api.Plan(cfg)
        if wide:
            p.set_option(api.OPT_WIDE_GROUPS, 1)
        return p.upload()

    ref = make()
    want = [ref.render(x).clone() for x in xs]
    torch.cuda.synchronize()
    plans = [make() for _ in range(4)]
    streams = [torch.cuda.Stream(device=gpu) for _ in range(4)]
    bad = 0
    for r in range(25 if cfg["algorithm"] else 80):
        torch.cuda.synchronize()
        outs = [plans[k % 4].render(xs[k % 3], stream=streams[k % 4].cuda_stream) for k in range(9)]
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(outs[k], want[k % 3]) else 1 for k in range(9))
    assert bad == 0, bad
