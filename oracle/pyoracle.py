"""ctypes binding of the CPU oracle (oracle/libsgz_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package `signalizer_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsgz_oracle.so")

NUM_SPEC_COLOURS = 5
NUM_GRAPHS = 2

CH_LEFT, CH_RIGHT, CH_MERGE, CH_SIDE, CH_PHASE, CH_SEPARATE, CH_MIDSIDE, CH_COMPLEX = range(8)
INTERP_NONE, INTERP_LINEAR, INTERP_LANCZOS = range(3)
VIEW_LINEAR, VIEW_LOG = range(2)
(WIN_RECT, WIN_HANN, WIN_HAMMING, WIN_FLATTOP, WIN_BLACKMAN, WIN_EXACT_BLACKMAN, WIN_NUTTALL,
 WIN_BLACKMAN_NUTTALL, WIN_BLACKMAN_HARRIS, WIN_TRIANGULAR, WIN_WELCH, WIN_GAUSSIAN, WIN_KAISER) = range(13)
WIN_SYMMETRIC, WIN_PERIODIC = range(2)


class SpectrumParams(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_float),
        ("window_size", C.c_uint32),
        ("hop", C.c_uint32),
        ("axis_points", C.c_uint32),
        ("channel_mode", C.c_uint32),
        ("bin_interp", C.c_uint32),
        ("view_scaling", C.c_uint32),
        ("window_type", C.c_uint32),
        ("window_symmetry", C.c_uint32),
        ("num_pairs", C.c_uint32),
        ("window_alpha", C.c_double),
        ("window_beta", C.c_double),
        ("view_left", C.c_double),
        ("view_right", C.c_double),
        ("min_log_freq", C.c_double),
        ("low_db", C.c_double),
        ("high_db", C.c_double),
        ("clip_db", C.c_double),
        ("slope_a", C.c_double),
        ("slope_b", C.c_double),
        ("pole", C.c_float * NUM_GRAPHS),
        ("colours", (C.c_uint8 * 3) * (NUM_SPEC_COLOURS + 1)),
        ("_pad", C.c_uint8 * 2),
        ("ratios", C.c_double * NUM_SPEC_COLOURS),
        ("algorithm", C.c_uint32),
        ("free_q", C.c_uint32),
    ]


class ZeroCrossingState(C.Structure):
    _fields_ = [("state", C.c_double), ("threshold", C.c_double), ("steady_clock", C.c_uint64),
                ("cross_origin", C.c_uint64), ("count", C.c_uint64), ("armed", C.c_int)]


class ScopeView(C.Structure):
    _fields_ = [("window_size", C.c_double), ("left", C.c_double), ("right", C.c_double),
                ("rendering_scale", C.c_double), ("width", C.c_uint32), ("_pad", C.c_uint32)]


class VectorFilters(C.Structure):
    _fields_ = [("env", C.c_float * 2), ("balance", (C.c_float * 2) * 2), ("phase", C.c_float * 2)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile). Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("primitives.c", "spectrum.c", "scope_vector.c", "scope_stream.c", "scope_spectral.c", "resonator.c", "spectrum_stream.c", "fft_simd.c", "sgz_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs if os.path.exists(s))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libsgz_oracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        vp = C.c_void_p
        L.sgzo_window.restype = C.c_double
        L.sgzo_window.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_uint32, vp]
        L.sgzo_transform_size.restype = C.c_uint32
        L.sgzo_transform_size.argtypes = [C.c_uint32]
        L.sgzo_fft_forward.argtypes = [vp, C.c_uint32]
        L.sgzo_fft_forward_f64.argtypes = [vp, C.c_uint32]
        L.sgzo_separate_transforms_ipl.argtypes = [vp, C.c_uint32]
        L.sgzo_lanczos_kernel.restype = C.c_double
        L.sgzo_lanczos_kernel.argtypes = [C.c_double, C.c_int]
        L.sgzo_lanczos_filter_f64.restype = C.c_double
        L.sgzo_lanczos_filter_f64.argtypes = [vp, C.c_size_t, C.c_double, C.c_int]
        L.sgzo_remap_frequencies.argtypes = [C.POINTER(SpectrumParams), vp]
        L.sgzo_slope_map.argtypes = [C.POINTER(SpectrumParams), vp, vp]
        L.sgzo_colour_ratios.argtypes = [vp, vp]
        L.sgzo_rotate_hue_rgb8.argtypes = [vp, C.c_float, vp]
        L.sgzo_colour_table.argtypes = [C.POINTER(SpectrumParams), C.c_uint32, vp]
        L.sgzo_prepare_transform.argtypes = [C.c_uint32, vp, vp, vp, C.c_uint32, C.c_uint32, vp]
        L.sgzo_map_to_linear_space.restype = C.c_int
        L.sgzo_map_to_linear_space.argtypes = [C.POINTER(SpectrumParams), vp, C.c_double, vp, C.c_uint32, vp]
        L.sgzo_map_and_transform_filters.argtypes = [C.POINTER(SpectrumParams), vp, vp, vp, vp]
        L.sgzo_blend_column.argtypes = [C.POINTER(SpectrumParams), vp, vp, C.c_uint32, vp]
        L.sgzo_spectrogram.restype = C.c_long
        L.sgzo_spectrogram.argtypes = [C.POINTER(SpectrumParams), vp, C.c_size_t, vp, vp, vp]
        L.sgzo_spectrogram_range.restype = C.c_long
        L.sgzo_spectrogram_range.argtypes = [C.POINTER(SpectrumParams), vp, C.c_size_t, C.c_long, C.c_long, vp]
        L.sgzo_window_cosine_terms.restype = C.c_int
        L.sgzo_window_cosine_terms.argtypes = [C.c_uint32, vp]
        L.sgzo_resonator_map.argtypes = [C.POINTER(SpectrumParams), vp, vp, vp, vp, C.POINTER(C.c_int)]
        L.sgzo_resonate_real.argtypes = [vp, C.c_uint32, C.c_int, vp, vp, C.c_int, C.c_size_t]
        L.sgzo_resonator_dispatch.restype = C.c_int
        L.sgzo_resonator_dispatch.argtypes = [C.c_uint32, vp, vp, C.c_size_t, vp, vp]
        L.sgzo_resonator_windowed_state.argtypes = [C.POINTER(SpectrumParams), vp, vp, vp, C.c_int, C.c_int, vp]
        L.sgzo_resonator_num_frames.restype = C.c_long
        L.sgzo_resonator_num_frames.argtypes = [C.c_size_t, C.c_uint32]
        L.sgzo_resonator_spectrogram_scaled.restype = C.c_long
        L.sgzo_resonator_spectrogram_scaled.argtypes = [C.POINTER(SpectrumParams), vp, C.c_size_t, vp, vp, vp, vp]
        L.sgzo_resonator_spectrogram.restype = C.c_long
        L.sgzo_resonator_spectrogram.argtypes = [C.POINTER(SpectrumParams), vp, C.c_size_t, vp, vp, vp]
        L.sgzo_track_peak.argtypes = [C.POINTER(SpectrumParams), vp, C.c_uint32, vp, C.c_double, C.c_double, vp]
        L.sgzo_track_peak_lines.argtypes = [C.POINTER(SpectrumParams), vp, vp, vp, C.c_uint32, C.c_double, vp]
        L.sgzo_decay_colour.restype = C.c_long
        L.sgzo_decay_colour.argtypes = [C.POINTER(SpectrumParams), vp, C.c_long, vp, vp]
        L.sgzo_logf_array.argtypes = [vp, vp, C.c_size_t]
        L.sgzo_num_frames.restype = C.c_long
        L.sgzo_num_frames.argtypes = [C.c_size_t, C.c_uint32, C.c_uint32]
        L.sgzo_zero_crossing_process.restype = C.c_size_t
        L.sgzo_zero_crossing_process.argtypes = [C.POINTER(ZeroCrossingState), C.c_uint32, vp, vp, C.c_size_t, vp, C.c_size_t]
        L.sgzo_scope_lanczos.restype = C.c_size_t
        L.sgzo_scope_lanczos.argtypes = [C.POINTER(ScopeView), vp, C.c_size_t, vp, vp, C.c_size_t]
        L.sgzo_scope_num_points.restype = C.c_size_t
        L.sgzo_scope_num_points.argtypes = [C.POINTER(ScopeView)]
        L.sgzo_peak_filter.restype = C.c_double
        L.sgzo_peak_filter.argtypes = [vp, C.c_uint32, C.c_size_t, C.c_uint32, C.c_double, vp]
        L.sgzo_vector_polar.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
        L.sgzo_vector_audio_processing.argtypes = [C.POINTER(VectorFilters), vp, vp, C.c_size_t, C.c_uint32,
                                                   C.c_float, C.c_float, C.c_float, C.c_int, vp]
        L.sgzo_vector_polar_view.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_uint32, C.c_int, vp, vp, vp]
        L.sgzo_scope_wave_plot.restype = C.c_size_t
        L.sgzo_scope_wave_plot.argtypes = [C.POINTER(ScopeView), C.c_int, C.c_int, vp, vp, C.c_int, C.c_size_t, C.c_size_t, vp, C.c_size_t]
        L.sgzo_scope_stream_create.restype = vp
        L.sgzo_scope_stream_create.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_int, C.c_double, C.c_uint32, C.c_double,
                                               C.c_uint32, C.c_double]
        L.sgzo_scope_stream_destroy.argtypes = [vp]
        L.sgzo_scope_stream_audio.argtypes = [vp, vp, C.c_size_t]
        L.sgzo_scope_stream_size.restype = C.c_size_t
        L.sgzo_scope_stream_size.argtypes = [vp]
        L.sgzo_scope_stream_front.restype = C.c_size_t
        L.sgzo_scope_stream_front.argtypes = [vp, C.c_uint32, vp]
        L.sgzo_scope_stream_envelope_gain.restype = C.c_double
        L.sgzo_scope_stream_envelope_gain.argtypes = [vp]
        L.sgzo_scope_stream_envelopes.argtypes = [vp, vp]
        L.sgzo_scope_stream_state.argtypes = [vp, vp]
        L.sgzo_scope_stream_peak_filter.restype = C.c_double
        L.sgzo_scope_stream_peak_filter.argtypes = [vp, C.c_uint32, C.c_double]
        L.sgzo_scope_stream_enable_colours.restype = None
        L.sgzo_scope_stream_enable_colours.argtypes = [vp, vp, C.c_double, C.c_double, vp]
        L.sgzo_scope_stream_front_colours.restype = C.c_size_t
        L.sgzo_scope_stream_front_colours.argtypes = [vp, C.c_uint32, C.c_int, vp]
        L.sgzo_scope_wave_plot_ex2.restype = C.c_size_t
        L.sgzo_scope_wave_plot_ex2.argtypes = [C.POINTER(ScopeView), C.c_int, C.c_int, vp, vp, C.c_int, C.c_size_t, C.c_size_t, C.c_double,
                                               C.c_double, C.c_int64, C.c_uint32, vp, vp, vp, C.c_size_t]
        L.sgzo_scope_stream_set_hysteresis.argtypes = [vp, C.c_double]
        L.sgzo_scope_wave_plot_ex.restype = C.c_size_t
        L.sgzo_scope_wave_plot_ex.argtypes = [C.POINTER(ScopeView), C.c_int, C.c_int, vp, vp, C.c_int, C.c_size_t, C.c_size_t, C.c_double,
                                              C.c_double, vp, vp, vp, C.c_size_t]
        L.sgzo_nth_element_by_index.restype = None
        L.sgzo_nth_element_by_index.argtypes = [vp, C.c_int, C.c_int]
        L.sgzo_scope_fundamental.restype = None
        L.sgzo_scope_fundamental.argtypes = [C.POINTER(SpectralState), vp, vp, C.c_int, C.c_size_t, C.c_size_t, C.c_double, C.c_double,
                                             C.c_double, C.c_double]
        L.sgzo_scope_fundamental_custom.restype = None
        L.sgzo_scope_fundamental_custom.argtypes = [C.POINTER(SpectralState), C.c_double, C.c_double]
        L.sgzo_scope_trigger_offset.restype = None
        L.sgzo_scope_trigger_offset.argtypes = [C.POINTER(SpectralState), vp, vp, C.c_int, C.c_size_t, C.c_size_t, C.c_double, C.c_double,
                                                C.c_double]
        L.sgzo_lr_design.restype = None
        L.sgzo_lr_design.argtypes = [C.c_double, C.c_double, C.c_double, vp]
        L.sgzo_lr_process.restype = None
        L.sgzo_lr_process.argtypes = [vp, vp, C.c_float, vp]
        L.sgzo_colour_smooth_pole.restype = C.c_float
        L.sgzo_colour_smooth_pole.argtypes = [C.c_double, C.c_double]
        L.sgzo_colour_accumulate.restype = None
        L.sgzo_colour_accumulate.argtypes = [vp, vp, vp, C.c_float, vp]
        L.sgzo_stream_create.restype = vp
        L.sgzo_stream_create.argtypes = [C.POINTER(SpectrumParams), C.c_uint32, C.c_size_t]
        L.sgzo_stream_destroy.restype = None
        L.sgzo_stream_destroy.argtypes = [vp]
        L.sgzo_stream_audio.restype = C.c_long
        L.sgzo_stream_audio.argtypes = [vp, vp, C.c_size_t, vp, vp, vp, C.c_size_t]
        L.sgzo_stream_render_lines.restype = C.c_int
        L.sgzo_stream_render_lines.argtypes = [vp, vp, vp]
        L.sgzo_stream_filters_given.restype = None
        L.sgzo_stream_filters_given.argtypes = [vp, vp, vp, vp]
        L.sgzo_stream_history.restype = None
        L.sgzo_stream_history.argtypes = [vp, C.c_uint32, C.c_size_t, vp]
        L.sgzo_stream_counter.restype = C.c_size_t
        L.sgzo_stream_counter.argtypes = [vp]
        _lib = L
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


_PARAM_FIELDS = {f[0] for f in SpectrumParams._fields_}


def params_from_dict(d: dict) -> SpectrumParams:
    """Fill a SpectrumParams from the plain dict produced by signalizer_amd.config.spectrum_config()."""
    p = SpectrumParams()
    for k, v in d.items():
        if k == "pole":
            for i in range(NUM_GRAPHS):
                p.pole[i] = v[i]
        elif k == "colours":
            for i in range(NUM_SPEC_COLOURS + 1):
                for c in range(3):
                    p.colours[i][c] = int(v[i][c])
        elif k == "ratios":
            for i in range(NUM_SPEC_COLOURS):
                p.ratios[i] = float(v[i])
        elif k in _PARAM_FIELDS:                      # (keys of the handle's configuration the chain does not read, e.g. display_mode, are not params)
            setattr(p, k, v)
    return p


# ----------------------------------------------------------------------------- convenience wrappers
def window(type_: int, symmetry: int, W: int, alpha: float = 0.0, beta: float = 0.0):
    out = np.zeros(W, np.float32)
    scale = lib().sgzo_window(type_, symmetry, alpha, beta, W, _ptr(out))
    return out, scale


def fft32(x: np.ndarray) -> np.ndarray:
    buf = np.ascontiguousarray(x, dtype=np.complex64).copy()
    lib().sgzo_fft_forward(_ptr(buf), buf.size)
    return buf


def fft64(x: np.ndarray) -> np.ndarray:
    buf = np.ascontiguousarray(x, dtype=np.complex128).copy()
    lib().sgzo_fft_forward_f64(_ptr(buf), buf.size)
    return buf


def remap_frequencies(p: SpectrumParams) -> np.ndarray:
    out = np.zeros(p.axis_points, np.float32)
    lib().sgzo_remap_frequencies(C.byref(p), _ptr(out))
    return out


def slope_map(p: SpectrumParams, mapped: np.ndarray) -> np.ndarray:
    out = np.zeros(p.axis_points, np.float32)
    lib().sgzo_slope_map(C.byref(p), _ptr(mapped), _ptr(out))
    return out


def colour_ratios(ratios) -> np.ndarray:
    r = np.asarray(ratios, np.float64)
    out = np.zeros(NUM_SPEC_COLOURS + 1, np.float32)
    lib().sgzo_colour_ratios(_ptr(r), _ptr(out))
    return out


def colour_table(p: SpectrumParams, pair: int) -> np.ndarray:
    out = np.zeros((NUM_SPEC_COLOURS + 1, 3), np.float32)
    lib().sgzo_colour_table(C.byref(p), pair, _ptr(out))
    return out


def rotate_hue(rgb, amount: float) -> np.ndarray:
    a = np.asarray(rgb, np.uint8)
    out = np.zeros(3, np.uint8)
    lib().sgzo_rotate_hue_rgb8(_ptr(a), C.c_float(amount), _ptr(out))
    return out


def frame_bins(p: SpectrumParams, L: np.ndarray, R: np.ndarray):
    """window x frame -> FFT -> (split, |.|): returns (csf_after_map [N+1] complex64, csp [2P] complex64)."""
    W = p.window_size
    N = lib().sgzo_transform_size(W)
    win = np.zeros(N, np.float32)
    scale = lib().sgzo_window(p.window_type, p.window_symmetry, p.window_alpha, p.window_beta, W, _ptr(win))
    csf = np.zeros(N + 1, np.complex64)
    Lc = np.ascontiguousarray(L[:W], np.float32)
    Rc = np.ascontiguousarray(R[:W], np.float32)
    lib().sgzo_prepare_transform(p.channel_mode, _ptr(Lc), _ptr(Rc), _ptr(win), W, N, _ptr(csf))
    lib().sgzo_fft_forward(_ptr(csf), N)
    raw = csf.copy()
    mapped = remap_frequencies(p)
    csp = np.zeros(2 * p.axis_points, np.complex64)
    lib().sgzo_map_to_linear_space(C.byref(p), _ptr(mapped), scale, _ptr(csf), N, _ptr(csp))
    return raw, csf, csp


def map_to_linear_space(p: SpectrumParams, csf_raw: np.ndarray, scale: float):
    N = csf_raw.size - 1
    csf = np.ascontiguousarray(csf_raw, np.complex64).copy()
    mapped = remap_frequencies(p)
    csp = np.zeros(2 * p.axis_points, np.complex64)
    rc = lib().sgzo_map_to_linear_space(C.byref(p), _ptr(mapped), scale, _ptr(csf), N, _ptr(csp))
    return rc, csf, csp


def filters(p: SpectrumParams, csp: np.ndarray, states: np.ndarray):
    """states: [graphs][P] complex64 in/out. returns results [graphs][P] complex64."""
    mapped = remap_frequencies(p)
    slope = slope_map(p, mapped)
    results = np.zeros((NUM_GRAPHS, p.axis_points), np.complex64)
    cspc = np.ascontiguousarray(csp, np.complex64)
    lib().sgzo_map_and_transform_filters(C.byref(p), _ptr(slope), _ptr(cspc), _ptr(states), _ptr(results))
    return results


def blend_column(p: SpectrumParams, frames: np.ndarray) -> np.ndarray:
    """frames: [pairs][P] complex64 (magnitude in .real). returns RGBA8 [P][4]."""
    fr = np.ascontiguousarray(frames, np.complex64)
    ratios = colour_ratios([p.ratios[i] for i in range(NUM_SPEC_COLOURS)])
    out = np.zeros((p.axis_points, 4), np.uint8)
    lib().sgzo_blend_column(C.byref(p), _ptr(ratios), _ptr(fr), fr.shape[0], _ptr(out))
    return out


def spectrogram(p: SpectrumParams, planar: np.ndarray, want_lines: bool = False, want_mapped: bool = False):
    """planar: [2*pairs][S] float32. returns dict(rgba [F][P][4], lines [F][C][G][P] c64, mapped [F][C][2P] c64)."""
    planar = np.ascontiguousarray(planar, np.float32)
    nch, S = planar.shape
    assert nch == 2 * p.num_pairs
    F = lib().sgzo_num_frames(S, p.window_size, p.hop)
    P = p.axis_points
    rgba = np.zeros((F, P, 4), np.uint8)
    lines = np.zeros((F, p.num_pairs, NUM_GRAPHS, P), np.complex64) if want_lines else None
    mapped = np.zeros((F, p.num_pairs, 2 * P), np.complex64) if want_mapped else None
    ptrs = (C.c_void_p * nch)(*[planar[c].ctypes.data for c in range(nch)])
    n = lib().sgzo_spectrogram(C.byref(p), ptrs, S, _ptr(rgba),
                               _ptr(lines) if want_lines else None, _ptr(mapped) if want_mapped else None)
    assert n == F, (n, F)
    return {"rgba": rgba, "lines": lines, "mapped": mapped, "frames": F}


RES_MAX_TERMS = 5


def resonator_map(p: SpectrumParams):
    """mapSystemHz restated: (coeff [V][P] complex64, gain [P], weights [V]) for the plan's mapped frequencies."""
    P = p.axis_points
    mf = remap_frequencies(p)
    coeff = np.zeros((2 * RES_MAX_TERMS - 1, P), np.complex64)
    gain = np.zeros(P, np.float32)
    weights = np.zeros(2 * RES_MAX_TERMS - 1, np.float32)
    V = C.c_int(0)
    lib().sgzo_resonator_map(C.byref(p), _ptr(mf), _ptr(coeff), _ptr(gain), _ptr(weights), C.byref(V))
    return coeff[:V.value].copy(), gain, weights[:V.value].copy()


def resonator_spectrogram(p: SpectrumParams, planar: np.ndarray, want_lines: bool = False, want_mapped: bool = False, want_scale: bool = False):
    """RSNT offline job: one frame per hop samples, resonators from rest.  Same result dict as spectrogram(); want_scale adds
    "scale" [F][C][2][P]: gain * sum_v |w_v| |s_v| per signal -- the size of the terms the window kernel sums (the tests' error bar)"""
    planar = np.ascontiguousarray(planar, np.float32)
    nch, S = planar.shape
    assert nch == 2 * p.num_pairs
    F = lib().sgzo_resonator_num_frames(S, p.hop)
    P = p.axis_points
    rgba = np.zeros((F, P, 4), np.uint8)
    lines = np.zeros((F, p.num_pairs, NUM_GRAPHS, P), np.complex64) if want_lines else None
    mapped = np.zeros((F, p.num_pairs, 2 * P), np.complex64) if want_mapped else None
    ptrs = (C.c_void_p * nch)(*[planar[c].ctypes.data for c in range(nch)])
    scale = np.zeros((F, p.num_pairs, 2, P), np.float32) if want_scale else None
    n = lib().sgzo_resonator_spectrogram_scaled(C.byref(p), ptrs, S, _ptr(rgba), _ptr(lines) if want_lines else None,
                                                _ptr(mapped) if want_mapped else None, _ptr(scale) if want_scale else None)
    assert n == F, (n, F)
    return {"rgba": rgba, "lines": lines, "mapped": mapped, "frames": F, "scale": scale}


def decay_colour(p: SpectrumParams, mapped: np.ndarray, want_lines: bool = False):
    """mapAndTransformDFTFilters + blendAndDispatchSpectrums on GIVEN mapped pixels (states start from zero).
    mapped: [F][C][sides][P] float32 as the HIP path's K_A writes them (two-plane modes: left / right magnitudes, Phase: magnitude /
    cancellation; one-plane modes: the magnitude).  Returns (rgba [F][P][4], lines [F][C][G][P] complex64 or None)."""
    m = np.ascontiguousarray(mapped, np.float32)
    F, Cn, sides, P = m.shape
    assert Cn == p.num_pairs and P == p.axis_points
    csp = np.zeros((F, Cn, 2 * P), np.complex64)
    if p.channel_mode == CH_PHASE:
        csp[:, :, :P] = m[:, :, 0] + 1j * m[:, :, 1]            # wsp[2x] = magnitude, wsp[2x+1] = cancellation
    else:
        for s_ in range(sides):
            csp[:, :, s_ * P:(s_ + 1) * P] = m[:, :, s_]
    rgba = np.zeros((F, P, 4), np.uint8)
    lines = np.zeros((F, Cn, NUM_GRAPHS, P), np.complex64) if want_lines else None
    n = lib().sgzo_decay_colour(C.byref(p), _ptr(csp), F, _ptr(rgba), _ptr(lines) if want_lines else None)
    assert n == F
    return rgba, lines


def logf(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().sgzo_logf_array(_ptr(x), _ptr(y), x.size)
    return y


def spectrogram_range(p: SpectrumParams, planar: np.ndarray, f0: int, f1: int) -> np.ndarray:
    planar = np.ascontiguousarray(planar, np.float32)
    nch, S = planar.shape
    rgba = np.zeros((f1 - f0, p.axis_points, 4), np.uint8)
    ptrs = (C.c_void_p * nch)(*[planar[c].ctypes.data for c in range(nch)])
    lib().sgzo_spectrogram_range(C.byref(p), ptrs, S, f0, f1, _ptr(rgba))
    return rgba


def zero_crossing(st: ZeroCrossingState, mode: int, a: np.ndarray, b: np.ndarray | None = None, max_out: int = 1 << 20):
    a = np.ascontiguousarray(a, np.float32)
    b = a if b is None else np.ascontiguousarray(b, np.float32)
    out = np.zeros(max_out, np.uint64)
    n = lib().sgzo_zero_crossing_process(C.byref(st), mode, _ptr(a), _ptr(b), a.size, _ptr(out), max_out)
    return out[:min(n, max_out)].copy()


def scope_lanczos(view: ScopeView, ring: np.ndarray):
    ring = np.ascontiguousarray(ring, np.float32)
    n = lib().sgzo_scope_num_points(C.byref(view))
    x = np.zeros(n, np.float32)
    y = np.zeros(n, np.float32)
    m = lib().sgzo_scope_lanczos(C.byref(view), _ptr(ring), ring.size, _ptr(x), _ptr(y), n)
    assert m == n
    return x, y


def peak_filter(channels: np.ndarray, coeff_pow: float, env: np.ndarray, lanes: int = 8) -> float:
    ch = np.ascontiguousarray(channels, np.float32)
    ptrs = (C.c_void_p * ch.shape[0])(*[ch[c].ctypes.data for c in range(ch.shape[0])])
    return lib().sgzo_peak_filter(ptrs, ch.shape[0], ch.shape[1], lanes, coeff_pow, _ptr(env))


def vector_polar(L: np.ndarray, R: np.ndarray) -> np.ndarray:
    L = np.ascontiguousarray(L, np.float32)
    R = np.ascontiguousarray(R, np.float32)
    out = np.zeros((L.size, 3), np.float32)
    lib().sgzo_vector_polar(_ptr(L), _ptr(R), L.size, 0, _ptr(out))
    return out


def vector_audio_processing(f: VectorFilters, L, R, envelope_coeff, stereo_coeff, second_speed=0.25,
                            env_mode=1, lanes=8):
    L = np.ascontiguousarray(L, np.float32)
    R = np.ascontiguousarray(R, np.float32)
    gain = C.c_float(float("nan"))
    lib().sgzo_vector_audio_processing(C.byref(f), _ptr(L), _ptr(R), L.size, lanes, envelope_coeff,
                                       stereo_coeff, second_speed, env_mode, C.byref(gain))
    return gain.value


TRIG_NONE, TRIG_SPECTRAL, TRIG_ZERO_CROSSING = 0, 1, 4
ENV_NONE, ENV_RMS, ENV_PEAK_DECAY = 0, 1, 2
OSC_LEFT, OSC_RIGHT, OSC_MID, OSC_SIDE, OSC_SEPARATE, OSC_MIDSIDE = range(6)


class ScopeStream:
    """Oscilloscope::StreamState for TriggeringMode None / ZeroCrossing (oracle/scope_stream.c)."""

    def __init__(self, channels, sample_rate, window_size, trigger_mode=TRIG_ZERO_CROSSING, threshold=0.0, osc_mode=OSC_LEFT,
                 trigger_channel=1.0, env_mode=ENV_NONE, envelope_window=0.3):
        self.channels = channels
        self.h = lib().sgzo_scope_stream_create(channels, sample_rate, window_size, trigger_mode, threshold, osc_mode,
                                                trigger_channel, env_mode, envelope_window)

    def __del__(self):
        try:
            lib().sgzo_scope_stream_destroy(self.h)
        except Exception:
            pass

    def set_hysteresis(self, hysteresis: float):
        lib().sgzo_scope_stream_set_hysteresis(self.h, hysteresis)

    def audio(self, block: np.ndarray):
        b = np.ascontiguousarray(block, np.float32)
        assert b.shape[0] == self.channels
        ptrs = (C.c_void_p * self.channels)(*[b[c].ctypes.data for c in range(self.channels)])
        lib().sgzo_scope_stream_audio(self.h, ptrs, b.shape[1])

    @property
    def size(self) -> int:
        return lib().sgzo_scope_stream_size(self.h)

    def front(self, c: int):
        """(raw ring memory [size], cursor)"""
        out = np.zeros(self.size, np.float32)
        cur = lib().sgzo_scope_stream_front(self.h, c, _ptr(out))
        return out, int(cur)

    def front_in_time_order(self, c: int) -> np.ndarray:
        m, cur = self.front(c)
        return np.concatenate([m[cur:], m[:cur]])

    @property
    def envelope_gain(self) -> float:
        return lib().sgzo_scope_stream_envelope_gain(self.h)

    def envelopes(self) -> np.ndarray:
        out = np.zeros(self.channels, np.float32)
        lib().sgzo_scope_stream_envelopes(self.h, _ptr(out))
        return out

    def state(self) -> dict:
        out = np.zeros(8, np.uint64)
        lib().sgzo_scope_stream_state(self.h, _ptr(out))
        keys = ("frontOrigin", "bufferedSamples", "oldPeak", "currentPeak", "steadyClock", "peaks", "isWorkingOnPeak", "swaps")
        return {k: int(v) for k, v in zip(keys, out)}

    def peak_filter(self, lanes: int, coeff: float) -> float:
        return lib().sgzo_scope_stream_peak_filter(self.h, lanes, coeff)

    def enable_colours(self, band_colours, blend: float, smoothing_ms: float, keys):
        """band_colours [3][3] float r,g,b of low / mid / high; keys [channels][4] RGBA8 (defaultKey per channel)"""
        bc = np.ascontiguousarray(band_colours, np.float32).reshape(3, 3)
        k = np.ascontiguousarray(keys, np.uint8).reshape(self.channels, 4)
        lib().sgzo_scope_stream_enable_colours(self.h, _ptr(bc), float(blend), float(smoothing_ms), _ptr(k))

    def front_colours(self, c: int, aux: bool = False):
        """(raw colour ring memory [size] as RGBA8 words, cursor)"""
        out = np.zeros(self.size, np.uint32)
        cur = lib().sgzo_scope_stream_front_colours(self.h, c, int(aux), _ptr(out))
        return out, int(cur)

    def logical(self, c: int, size: int, colours=None) -> np.ndarray:
        """the reference's Spectral-mode ring of the moment: the newest `size` samples, oldest first (a ring whose cursor is 0).
        colours: None -> audio; False / True -> colourData / auxColourData"""
        m, cur = self.front(c) if colours is None else self.front_colours(c, colours)
        t = np.concatenate([m[cur:], m[:cur]])
        return np.ascontiguousarray(t[len(t) - size:])


def scope_wave_plot(view: ScopeView, trigger_mode: int, interpolation: int, mem_a: np.ndarray, mem_b: np.ndarray, eval_mode: int,
                    cursor: int, max_points: int = 1 << 22) -> np.ndarray:
    """drawWavePlot for one evaluator -> vertices [n][3]"""
    a = np.ascontiguousarray(mem_a, np.float32)
    b = np.ascontiguousarray(mem_b, np.float32)
    n = lib().sgzo_scope_wave_plot(C.byref(view), trigger_mode, interpolation, _ptr(a), _ptr(b), eval_mode, a.size, cursor, None, 0)
    out = np.zeros((n, 3), np.float32)
    m = lib().sgzo_scope_wave_plot(C.byref(view), trigger_mode, interpolation, _ptr(a), _ptr(b), eval_mode, a.size, cursor, _ptr(out), n)
    assert m == n
    return out


def scope_wave_plot_ex(view: ScopeView, trigger_mode: int, interpolation: int, mem_a, mem_b, eval_mode: int, cursor: int,
                       cycle_samples: float = 0.0, sample_offset: float = 0.0, colour_mem=None):
    """drawWavePlot incl. Spectral triggering and per-vertex colours -> (vertices [n][3], rgba [n][4] or None)"""
    a = np.ascontiguousarray(mem_a, np.float32)
    b = np.ascontiguousarray(mem_b, np.float32)
    cm = None if colour_mem is None else np.ascontiguousarray(colour_mem, np.uint32)
    args = (C.byref(view), trigger_mode, interpolation, _ptr(a), _ptr(b), eval_mode, a.size, cursor, cycle_samples, sample_offset,
            None if cm is None else _ptr(cm))
    n = lib().sgzo_scope_wave_plot_ex(*args, None, None, 0)
    out = np.zeros((n, 3), np.float32)
    rgba = None if cm is None else np.zeros(n, np.uint32)
    m = lib().sgzo_scope_wave_plot_ex(*args, _ptr(out), None if rgba is None else _ptr(rgba), n)
    assert m == n
    return out, (None if rgba is None else rgba.view(np.uint8).reshape(n, 4))


def scope_wave_plot_ex2(view: ScopeView, trigger_mode: int, interpolation: int, mem_a, mem_b, eval_mode: int, cursor: int,
                        cycle_samples: float = 0.0, sample_offset: float = 0.0, transport_position: int = 0, key: int = 0xFFFFFFFF,
                        colour_mem=None):
    """drawWavePlot with every trigger mode (Window: transport_position) and interpolation (None / Rectangular / Linear / Lanczos)
    -> (vertices [n][3], rgba [n][4])"""
    a = np.ascontiguousarray(mem_a, np.float32)
    b = np.ascontiguousarray(mem_b, np.float32)
    cm = None if colour_mem is None else np.ascontiguousarray(colour_mem, np.uint32)
    args = (C.byref(view), trigger_mode, interpolation, _ptr(a), _ptr(b), eval_mode, a.size, cursor, cycle_samples, sample_offset,
            int(transport_position), int(key), None if cm is None else _ptr(cm))
    n = lib().sgzo_scope_wave_plot_ex2(*args, None, None, 0)
    out = np.zeros((n, 3), np.float32)
    rgba = np.zeros(n, np.uint32)
    m = lib().sgzo_scope_wave_plot_ex2(*args, _ptr(out), _ptr(rgba), n)
    assert m == n
    return out, rgba.view(np.uint8).reshape(n, 4)


class BinRecord(C.Structure):
    _fields_ = [("index", C.c_uint64), ("value", C.c_double), ("offset", C.c_double)]


class SpectralState(C.Structure):
    """triggerState + medianTriggerFilter / medianPos (Oscilloscope.h:176-196, :324-325)"""
    _fields_ = [("median", BinRecord * 8), ("median_pos", C.c_uint64), ("record", BinRecord), ("fundamental", C.c_double),
                ("cycle_samples", C.c_double), ("sample_offset", C.c_double), ("phase", C.c_double)]


def scope_analyse(ts: SpectralState, mem_a, mem_b, eval_mode: int, cursor: int, window_size: float, sample_rate: float,
                  threshold: float, hysteresis: float, phase_offset_degrees: float, custom_frequency: float = 0.0):
    """calculateFundamentalPeriod + calculateTriggeringOffset (Spectral) on a ring; updates ts.  custom_frequency > 0: state.customTrigger"""
    a = np.ascontiguousarray(mem_a, np.float32)
    b = np.ascontiguousarray(mem_b, np.float32)
    if custom_frequency > 0:
        lib().sgzo_scope_fundamental_custom(C.byref(ts), custom_frequency, sample_rate)
    else:
        lib().sgzo_scope_fundamental(C.byref(ts), _ptr(a), _ptr(b), eval_mode, a.size, cursor, window_size, sample_rate, threshold, hysteresis)
    lib().sgzo_scope_trigger_offset(C.byref(ts), _ptr(a), _ptr(b), eval_mode, a.size, cursor, window_size, sample_rate, phase_offset_degrees)
    return ts


def nth_element_by_index(records: np.ndarray, nth: int) -> np.ndarray:
    """records: structured [(index u8, value f8, offset f8)]; returns the array as std::nth_element leaves it"""
    r = np.ascontiguousarray(records).copy()
    lib().sgzo_nth_element_by_index(_ptr(r), r.shape[0], nth)
    return r


def lr_bands(x: np.ndarray, sample_rate: float, low=300.0, high=3000.0) -> np.ndarray:
    """the 3-band Linkwitz-Riley split of a signal -> [n][3]"""
    co = np.zeros(20, np.float32)
    st = np.zeros(16, np.float32)
    lib().sgzo_lr_design(low, high, sample_rate, _ptr(co))
    out = np.zeros((x.size, 3), np.float32)
    for i, v in enumerate(np.asarray(x, np.float32)):
        lib().sgzo_lr_process(_ptr(st), _ptr(co), float(v), _ptr(out[i]))
    return out


def vector_polar_view(mem_l: np.ndarray, mem_r: np.ndarray, cursor: int, lanes: int = 8, fade_history: bool = False,
                      colour=(1.0, 1.0, 1.0)):
    """drawPolarPlot over the two sections of a history ring -> (xyz [size][3], rgb [size][3])"""
    L = np.ascontiguousarray(mem_l, np.float32)
    R = np.ascontiguousarray(mem_r, np.float32)
    col = np.asarray(colour, np.float32)
    xyz = np.zeros((L.size, 3), np.float32)
    rgb = np.zeros((L.size, 3), np.float32)
    lib().sgzo_vector_polar_view(_ptr(L), _ptr(R), L.size, cursor, lanes, int(fade_history), _ptr(col), _ptr(xyz), _ptr(rgb))
    return xyz, rgb


def track_peak(p: SpectrumParams, source: np.ndarray, scale: float, mouse_fraction: float) -> dict:
    """raw-FFT frequency tracker on csf (complex64 [N+1], as mapToLinearSpace left it)"""
    src = np.ascontiguousarray(source, np.complex64)
    mapped = remap_frequencies(p)
    out = np.zeros(8, np.float64)
    lib().sgzo_track_peak(C.byref(p), _ptr(src), src.size - 1, _ptr(mapped), scale, mouse_fraction, _ptr(out))
    keys = ("peak_offset", "peak_fraction", "peak_frequency", "peak_dbs", "alpha", "beta", "gamma", "phi")
    return dict(zip(keys, out.tolist()))


def track_peak_lines(p: SpectrumParams, results: np.ndarray, transform_size: int, mouse_fraction: float) -> dict:
    """the tracker's line-results branch (SpectrumRendering.cpp:300-377) on lineGraphs[g].results: float2 [P] (left, right)"""
    r = np.ascontiguousarray(results, np.float32)
    mapped = remap_frequencies(p)
    slope = slope_map(p, mapped)
    out = np.zeros(6, np.float64)
    lib().sgzo_track_peak_lines(C.byref(p), _ptr(r), _ptr(mapped), _ptr(slope), int(transform_size), mouse_fraction, _ptr(out))
    keys = ("peak_offset", "peak_frequency", "peak_deviance", "peak_fraction_y", "peak_dbs", "peak_slope")
    return dict(zip(keys, out.tolist()))


# ----------------------------------------------------------------------------- the Spectrum view as a stream (spectrum_stream.c)
DISPLAY_LINE_GRAPH, DISPLAY_COLOUR_SPECTRUM = 0, 1


class SpectrumStream:
    """The reference's two threads on one pair list: audio(block) = onStreamAudio (audioEntryPoint with quirks Q1 / Q2, frames blended
    into columns), render_lines() = vectorGLRendering's LineGraph case (whole-ring transform, filters advanced once per call)."""

    def __init__(self, p: SpectrumParams, display_mode: int = DISPLAY_COLOUR_SPECTRUM, history: int = 0):
        self.p = p
        self.h = lib().sgzo_stream_create(C.byref(p), display_mode, history)
        assert self.h, "sgzo_stream_create failed"

    def __del__(self):
        if getattr(self, "h", None):
            lib().sgzo_stream_destroy(self.h)
            self.h = None

    def audio(self, block: np.ndarray, max_frames: int | None = None):
        """block [2*pairs][n] float32 -> dict(frames, rgba [frames][P][4], lines [C][G][P] c64, mapped [frames][C][2P] c64)"""
        block = np.ascontiguousarray(block, np.float32)
        nch, n = block.shape
        assert nch == 2 * self.p.num_pairs
        P, Cn = self.p.axis_points, self.p.num_pairs
        if max_frames is None:
            max_frames = n // max(1, self.p.hop) + 2
        rgba = np.zeros((max_frames, P, 4), np.uint8)
        lines = np.zeros((Cn, NUM_GRAPHS, P), np.complex64)
        mapped = np.zeros((max_frames, Cn, 2 * P), np.complex64)
        ptrs = (C.c_void_p * nch)(*[block[c].ctypes.data for c in range(nch)])
        F = lib().sgzo_stream_audio(self.h, ptrs, n, _ptr(rgba), _ptr(lines), _ptr(mapped), max_frames)
        F = min(int(F), max_frames)
        return dict(frames=F, rgba=rgba[:F], lines=lines, mapped=mapped[:F])

    def render_lines(self):
        P, Cn = self.p.axis_points, self.p.num_pairs
        results = np.zeros((Cn, NUM_GRAPHS, P), np.complex64)
        mapped = np.zeros((Cn, 2 * P), np.complex64)
        ok = lib().sgzo_stream_render_lines(self.h, _ptr(results), _ptr(mapped))
        return dict(ok=bool(ok), results=results, mapped=mapped)

    def filters_given(self, csp_all: np.ndarray, want_rgba: bool = False):
        P, Cn = self.p.axis_points, self.p.num_pairs
        csp = np.ascontiguousarray(csp_all, np.complex64).reshape(Cn, 2 * P)
        results = np.zeros((Cn, NUM_GRAPHS, P), np.complex64)
        rgba = np.zeros((P, 4), np.uint8) if want_rgba else None
        lib().sgzo_stream_filters_given(self.h, _ptr(csp), _ptr(results), _ptr(rgba) if want_rgba else None)
        return results, rgba

    def history(self, channel: int, count: int) -> np.ndarray:
        out = np.zeros(count, np.float32)
        lib().sgzo_stream_history(self.h, channel, count, _ptr(out))
        return out

    def counter(self) -> int:
        return int(lib().sgzo_stream_counter(self.h))
