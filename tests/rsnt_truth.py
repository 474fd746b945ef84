"""An fp64 walk of an RSNT render's resonators -- the "truth" both fp32 evaluations (the oracle's sequential recurrence, the device's
chained block sums) are approximations of.  Test infrastructure (numpy only).

The plan's fp32 poles, gains and window weights are taken as exact; the signal is the one the reference's dispatch mixes in fp32
(resonatingDispatch, Source/Spectrum/TransformDSP.inl:1213-1295: sgzo_resonator_dispatch restates it).  Every resonator is
s[n] = c s[n - 1] + x[n] from rest; the frame value at the end of hop f is |sum_v w_v s_v[(f + 1) hop - 1]| * gain
(mapToLinearSpace's RSNT branch :1103-1133 / getWholeWindowedState).  Evaluated in complex128 as block sums against the pole's powers,
chained with c^hop -- exact to ~1e-13 relative, eight orders below the fp32 evaluations' differences."""
import ctypes as C

import numpy as np


def dispatch_signal(po, mode: int, L: np.ndarray, R: np.ndarray, sig: int) -> np.ndarray:
    L, R = np.ascontiguousarray(L, np.float32), np.ascontiguousarray(R, np.float32)
    w0, w1 = np.zeros_like(L), np.zeros_like(L)
    nsig = po.lib().sgzo_resonator_dispatch(C.c_uint32(mode), L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), C.c_size_t(L.size),
                                           w0.ctypes.data_as(C.c_void_p), w1.ctypes.data_as(C.c_void_p))
    assert sig < nsig, (sig, nsig)
    return (w0 if sig == 0 else w1).astype(np.float64)


def frame_magnitudes(po, p, x: np.ndarray, pair: int, sig: int, frames: int, chunk: int = 512):
    """([frames][P] float64: the windowed magnitudes of (pair, signal) after every hop,
        [frames][P] float64: gain * sum_v |w_v| |s_v|, the size of the terms the window sums -- the error bars' scale)"""
    coeff, gain, weights = po.resonator_map(p)
    coeff = np.asarray(coeff).astype(np.complex128)                      # [V][P], the fp32 values widened (exact)
    V, P = coeff.shape
    hop = int(p.hop)
    xin = dispatch_signal(po, int(p.channel_mode), x[2 * pair], x[2 * pair + 1], sig)[:frames * hop]
    X = xin.reshape(frames, hop).T[::-1].copy()                             # X[k, f] = x[f hop + hop - 1 - k]: multiplies c^k
    out = np.zeros((frames, P))
    scale = np.zeros((frames, P))
    w = np.asarray(weights, np.float64)
    g = np.asarray(gain, np.float64)
    for a in range(0, P, chunk):
        c = coeff[:, a:a + chunk].reshape(-1)                               # resonators of this chunk, all vectors
        # c^k: cumulative products in blocks of 64 (exp(k log c) would lose ~k eps)
        pw = np.empty((c.size, hop), np.complex128)
        step = 64
        base = np.cumprod(np.concatenate([np.ones((c.size, 1), np.complex128), np.repeat(c[:, None], step - 1, axis=1)], axis=1), axis=1)   # c^0..c^63
        cstep = base[:, -1] * c                                             # c^64
        lead = np.ones(c.size, np.complex128)
        for b in range(0, hop, step):
            m = min(step, hop - b)
            pw[:, b:b + m] = lead[:, None] * base[:, :m]
            lead = lead * cstep
        local = pw @ X                                                      # [res][frames]: each frame from rest
        chop = pw[:, hop - 1] * c                                           # c^hop
        s = np.zeros(c.size, np.complex128)
        acc = np.zeros((frames, c.size), np.complex128)
        for f in range(frames):
            s = chop * s + local[:, f]
            acc[f] = s
        acc = acc.reshape(frames, V, -1)
        out[:, a:a + chunk] = np.abs(np.einsum("v,fvp->fp", w, acc)) * g[a:a + chunk]
        scale[:, a:a + chunk] = np.einsum("v,fvp->fp", np.abs(w), np.abs(acc)) * g[a:a + chunk]
    return out, scale
