"""The end-to-end parity argument as a chain (tests/test_gpu_fuzz.py, tools/fuzz_parity.py).

The HIP path's FFT is a different -- equally valid -- fp32 butterfly order than the oracle's (and than pffft's, which
nobody can reproduce here), so raw RGBA8 bytes of a whole render can differ where a 1e-7 relative difference of a
magnitude crosses a uint8 truncation.  Instead of a blanket "x % of bytes may differ" the sweep proves three links:

  1. mapped pixels:  |K_A(x) - oracle_map(x)| <= MAP_TOL * (the frame's largest bin)  per pixel  (the FFT's rounding, nothing else);
     exceptions must be *explained*: Phase mode's arg-max takes the bin with the largest max(|L|^2, |R|^2) but shows
     |L| + |R| of it, so two candidate bins whose keys tie to ~1e-7 can be picked differently by two fp32 FFTs -- for each
     such pixel the candidate the HIP path picked is looked up in the ORACLE's own bins and must tie with the oracle's
     winner to <= TIE_REL;
  2. mapping given bins: bit-exact (tests/test_gpu_spectrum.py::test_mapping_bit_exact_given_bins, test_gpu_phase.py);
  3. colour given mapped pixels: render(x) == oracle_decay_colour(K_A(x)) byte for byte -- checked here on every case.
"""
import numpy as np

MAP_TOL = 4e-6          # same bar as the bins (tests/test_gpu_spectrum.py BIN_TOL)
# K_A evaluates a periodic Hann / Hamming window inside the kernel as p0 - p1 cos(phase) (plan option SGZ_OPT_FETCH_WINDOW = 0): every
# coefficient carries an ABSOLUTE rounding of about one ulp of the window's peak, where the oracle's table (fp64, rounded once)
# carries a relative one.  On a bin that is sum_n dw_n x_n e^{..}: a random walk of ~ 2^-25 |x_n|, i.e. <= WIN_ABS * invSize * ||x||_2
# at six sigma (per coefficient: the angle addition's two products and the subtraction, ~ 2.5e-8 rms).  Invisible next to MAP_TOL * (largest bin) for any frame whose energy is spread over the window (8.8e-10 against
# 1e-6 at cfg2); it is the larger term only where a frame's whole energy sits under the window's feet (the first frames of a stream
# that starts from silence: tests/test_gpu_stream_modes.py renders one with 480 of 32768 samples non-zero).
WIN_ABS = 2e-7
TIE_REL = 1e-5          # relative key distance that counts as an FFT-rounding tie (or the bins' own bar, see _phase_tie_ok)
CH_PHASE = 4


def _ref_planes(po, p, r_mapped, sides, P):
    """oracle csp [F][C][2P] complex -> the planes K_A writes, [F][C][sides][P] float32"""
    F, Cn = r_mapped.shape[:2]
    out = np.zeros((F, Cn, sides, P), np.float32)
    if p.channel_mode == CH_PHASE:
        out[:, :, 0] = r_mapped[:, :, :P].real
        out[:, :, 1] = r_mapped[:, :, :P].imag
    else:
        for s in range(sides):
            v = r_mapped[:, :, s * P:(s + 1) * P]
            out[:, :, s] = np.sqrt((v.real.astype(np.float32) ** 2 + v.imag.astype(np.float32) ** 2).astype(np.float32))
    return out


def _phase_tie_ok(po, p, plan, x, hop, f, pair, px, got, ref, tol, tolc):
    """Phase arg-max pixel px shows (|L| + |R|, cancellation) of the bin with the largest max(|L|^2, |R|^2) of its run.  Is the HIP
    path's (magnitude, cancellation) `got` that of a bin g, and the oracle's `ref` that of a bin o != g, whose keys tie to
    <= TIE_REL in the ORACLE's own bins (fp64 from its raw transform)?  Candidates: the run's bins and a margin of two either side."""
    W = p.window_size
    N = plan.N
    raw, _, _ = po.frame_bins(p, x[2 * pair, f * hop:f * hop + W], x[2 * pair + 1, f * hop:f * hop + W])
    Z = raw[:N].astype(np.complex128)
    mf = plan.mapped_frequencies()
    num_bins = N // 2
    f2b = np.float32(num_bins / np.float32(p.sample_rate / 2))
    lo = max(1, int(np.float32(mf[max(px - 1, 0)]) * f2b) - 2)
    hi = min(num_bins - 2, int(np.float32(mf[px]) * f2b) + 2)
    k = np.arange(lo, hi + 1)
    Lk = (Z[k] + np.conj(Z[N - k])) * 0.5
    Rk = (Z[k] - np.conj(Z[N - k])) * (-0.5j)
    key = np.maximum(np.abs(Lk) ** 2, np.abs(Rk) ** 2)
    mid = np.abs(Lk) + np.abs(Rk)
    val = mid * (plan.window_scale / (W * 0.5))
    canc = 1.0 - np.abs(Lk + Rk) / np.maximum(mid, 1e-300)
    is_g = (np.abs(val - got[0]) <= tol) & (np.abs(canc - got[1]) <= tolc)
    is_o = (np.abs(val - ref[0]) <= tol) & (np.abs(canc - ref[1]) <= tolc)
    # a tie: the keys differ by no more than TIE_REL, or by no more than link 1's own bar allows -- `tol` on a mapped value is
    # tol / scale on a raw |L| or |R|, i.e. 2 sqrt(key) tol / scale on a key (low bins of a loud frame: the bar is absolute, the
    # keys are small; seed 600 of the wild sweep met two keys 1.6e-5 apart at 0.2 % of the frame's maximum)
    tol_raw = tol / (plan.window_scale / (W * 0.5))
    for g in np.nonzero(is_g)[0]:
        for o in np.nonzero(is_o)[0]:
            top = max(key[g], key[o])
            if g != o and abs(key[g] - key[o]) <= max(TIE_REL * top, 2.0 * np.sqrt(top) * tol_raw):
                return True
    return False


def check_render(po, plan, cfg, x, gpu, want_lines=False, win_abs=WIN_ABS):
    """Returns (problems: list[str], stats: dict).  x: host float32 [2C][S]."""
    import torch
    p = po.params_from_dict(cfg)
    xg = torch.from_numpy(x).to(gpu)
    P, sides = plan.P, plan.sides
    got_mapped = plan.stage_mapped(xg).cpu().numpy()
    lines_t = torch.empty((got_mapped.shape[0], plan.C, 2, P, 2), dtype=torch.float32, device=gpu) if want_lines else None
    got_rgba = plan.render(xg, lines=lines_t).cpu().numpy()
    r = po.spectrogram(p, x, want_mapped=True)
    problems = []
    if got_rgba.shape != r["rgba"].shape:
        return ["shape %s vs %s" % (got_rgba.shape, r["rgba"].shape)], {}
    # link 3: colour (and lines) given the HIP path's own mapped pixels -- byte for byte
    chain_rgba, chain_lines = po.decay_colour(p, got_mapped, want_lines=want_lines)
    if not np.array_equal(got_rgba, chain_rgba):
        d = np.abs(got_rgba.astype(int) - chain_rgba.astype(int))
        problems.append("colour given mapped: %d bytes differ (max %d)" % (int((d > 0).sum()), int(d.max())))
    if want_lines:
        gl = lines_t.cpu().numpy()                                    # [F][C][G][P][2]
        rl = np.stack([chain_lines.real, chain_lines.imag], axis=-1).astype(np.float32)
        # (sqrt(m*m) == m needs m*m normal: the oracle re-derives the magnitude from (re, 0))
        loud = np.moveaxis(got_mapped, 2, 3)[:, :, None] > 1e-18 if p.channel_mode != CH_PHASE else np.ones_like(gl, bool)
        same = (gl.view(np.uint32) == rl.view(np.uint32)) | ~np.broadcast_to(loud, gl.shape)
        if p.channel_mode != CH_PHASE and sides == 1:
            same[..., 1] = gl[..., 1] == 0                           # results[i].phase = 0
        if not same.all():
            problems.append("lines given mapped: %d values differ" % int((~same).sum()))
    # link 1: mapped pixels against the oracle's, within the FFT tolerance; Phase near-ties explained one by one
    ref = _ref_planes(po, p, r["mapped"], sides, P)
    phase = p.channel_mode == CH_PHASE
    # the FFT's rounding error scales with the frame's largest bin, wherever it lies -- not with the largest pixel of a zoomed view:
    # scale[f][c] = invSize * max_k |Z[k]| of the windowed frame (numpy fp64)
    W, hop, N = p.window_size, cfg["hop"], plan.N
    win = plan.window()[:W].astype(np.float64)
    inv_size = plan.window_scale / (W * 0.5)
    scale = np.zeros((ref.shape[0], ref.shape[1], 1))
    for f in range(ref.shape[0]):
        for c in range(ref.shape[1]):
            z = (x[2 * c, f * hop:f * hop + W].astype(np.float64) + 1j * x[2 * c + 1, f * hop:f * hop + W])
            scale[f, c, 0] = inv_size * np.abs(np.fft.fft(z * win, N)).max() + (win_abs / MAP_TOL) * inv_size * np.sqrt((np.abs(z) ** 2).sum())
    scale = np.maximum(scale, 1e-30)
    ties = 0
    finite = np.isfinite(ref) & np.isfinite(got_mapped)
    if phase:
        # magnitude |L| + |R| within the FFT tolerance; cancellation = 1 - |L + R| / (|L| + |R|) within the FFT's rounding relative to
        # the pixel's own magnitude.  A pixel outside either bar must be an arg-max pixel whose two candidates tie in the oracle's bins.
        tolc = np.minimum(1.0, 8 * MAP_TOL * scale / np.maximum(ref[:, :, 0], 1e-30))
        dm = np.abs(got_mapped[:, :, 0] - ref[:, :, 0])
        dc = np.abs(got_mapped[:, :, 1] - ref[:, :, 1])
        bad = ((dm > MAP_TOL * scale) | (dc > tolc)) & finite[:, :, 0] & finite[:, :, 1]
        for f, c, px in zip(*np.nonzero(bad)):
            if px >= plan.break_pixel and _phase_tie_ok(po, p, plan, x, cfg["hop"], int(f), int(c), int(px), got_mapped[f, c, :, px],
                                                         ref[f, c, :, px], float(MAP_TOL * scale[f, c, 0]), float(tolc[f, c, px])):
                ties += 1
            else:
                problems.append("phase frame %d pair %d pixel %d: (mag, canc) %s vs %s (frame max %g)" %
                                (f, c, px, got_mapped[f, c, :, px], ref[f, c, :, px], scale[f, c, 0]))
    else:
        dm = np.abs(got_mapped - ref)
        bad = (dm > MAP_TOL * scale[:, :, None]) & finite
        if bad.any():
            f, c, s, px = [int(v[0]) for v in np.nonzero(bad)]
            problems.append("mapped: %d pixels off, first frame %d pair %d side %d pixel %d: %g vs %g (max %g)" %
                            (int(bad.sum()), f, c, s, px, got_mapped[f, c, s, px], ref[f, c, s, px], scale[f, c, 0]))
    d = np.abs(got_rgba.astype(int) - r["rgba"].astype(int))
    return problems, {"ties": ties, "bytes_differing": int((d > 0).sum()), "max_byte_diff": int(d.max()) if d.size else 0,
                      "frac": float((d > 0).mean()) if d.size else 0.0}
