"""Host-side binding of libsgz.so (include/sgz.h) for Python callers, tests and bench.py.

A thin ctypes layer over the C ABI: `Plan` (the TransformConstant mirror and the batch / stage entry points),
`render_spectrogram` (host buffers), and the argtypes of the real-time handles (sgz_spectrum_* / sgz_scope_* / sgz_vector_*:
onStreamAudio in, columns / vertices out), which the tests drive directly.
PyTorch is used only as the device allocator / stream provider.  No CPU fallback: if libsgz.so is
missing or no GPU is visible, compute calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

NUM_SPEC_COLOURS = 5
NUM_GRAPHS = 2

SGZ_OK, SGZ_EMPTY, SGZ_SKIPPED_FRAME, SGZ_BUSY = 0, 1, 2, 3
SGZ_EINVAL, SGZ_EHIP, SGZ_ENOMEM, SGZ_EUNSUPPORTED = -1, -2, -3, -4


class SgzError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"sgz status {status}: {msg}")
        self.status = status


class SpectrumConfig(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_float), ("window_size", C.c_uint32), ("hop", C.c_uint32), ("axis_points", C.c_uint32),
        ("channel_mode", C.c_uint32), ("bin_interp", C.c_uint32), ("view_scaling", C.c_uint32),
        ("window_type", C.c_uint32), ("window_symmetry", C.c_uint32), ("num_pairs", C.c_uint32),
        ("window_alpha", C.c_double), ("window_beta", C.c_double), ("view_left", C.c_double),
        ("view_right", C.c_double), ("min_log_freq", C.c_double), ("low_db", C.c_double), ("high_db", C.c_double),
        ("clip_db", C.c_double), ("slope_a", C.c_double), ("slope_b", C.c_double),
        ("pole", C.c_float * NUM_GRAPHS), ("colours", (C.c_uint8 * 3) * (NUM_SPEC_COLOURS + 1)),
        ("_pad", C.c_uint8 * 2), ("ratios", C.c_double * NUM_SPEC_COLOURS),
        ("algorithm", C.c_uint32), ("free_q", C.c_uint32), ("display_mode", C.c_uint32), ("_reserved", C.c_uint32),
    ]


class Peak(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("peak_offset", "peak_fraction", "peak_frequency", "peak_dbs", "alpha", "beta", "gamma", "phi")]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class LinePeak(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("peak_offset", "peak_frequency", "peak_deviance", "peak_fraction_y", "peak_dbs", "peak_slope")]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Timing(C.Structure):
    _fields_ = [("h2d_ms", C.c_double), ("kernel_ms", C.c_double), ("d2h_ms", C.c_double), ("frames", C.c_uint64)]


class ScopeView(C.Structure):
    _fields_ = [("window_size", C.c_double), ("left", C.c_double), ("right", C.c_double),
                ("rendering_scale", C.c_double), ("width", C.c_uint32), ("_pad", C.c_uint32)]


class ZeroCrossingState(C.Structure):
    _fields_ = [("state", C.c_double), ("threshold", C.c_double), ("steady_clock", C.c_uint64),
                ("cross_origin", C.c_uint64), ("count", C.c_uint64), ("armed", C.c_int32), ("_pad", C.c_int32)]


class ScopeConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_double), ("window_size", C.c_double), ("num_channels", C.c_uint32), ("trigger_mode", C.c_uint32),
                ("channel_mode", C.c_uint32), ("envelope_mode", C.c_uint32), ("interpolation", C.c_uint32), ("max_block", C.c_uint32),
                ("trigger_threshold", C.c_double), ("trigger_channel", C.c_double), ("envelope_window", C.c_double),
                ("colours", (C.c_uint8 * 4) * 64),
                ("trigger_hysteresis", C.c_double), ("trigger_phase_offset", C.c_double), ("colour_by_frequency", C.c_uint32),
                ("frequency_colouring_blend", C.c_float), ("colour_smoothing_ms", C.c_double), ("band_colours", (C.c_float * 3) * 3),
                ("custom_trigger", C.c_uint32), ("custom_trigger_frequency", C.c_double)]


class TriggerState(C.Structure):
    _fields_ = [("record_index", C.c_uint64), ("record_value", C.c_double), ("record_offset", C.c_double), ("fundamental", C.c_double),
                ("cycle_samples", C.c_double), ("sample_offset", C.c_double), ("phase", C.c_double), ("ring_size", C.c_uint64)]


class VectorFilters(C.Structure):
    _fields_ = [("env", C.c_float * 2), ("balance", (C.c_float * 2) * 2), ("phase", C.c_float * 2)]


class VectorConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_double), ("num_channels", C.c_uint32), ("window_size", C.c_uint32), ("envelope_mode", C.c_uint32),
                ("lanes", C.c_uint32), ("fade_history", C.c_uint32), ("max_block", C.c_uint32), ("envelope_window", C.c_double),
                ("stereo_window", C.c_double), ("colours", (C.c_float * 3) * 32)]


def config_from_dict(d: dict) -> SpectrumConfig:
    c = SpectrumConfig()
    for k, v in d.items():
        if k == "pole":
            for i in range(NUM_GRAPHS):
                c.pole[i] = v[i]
        elif k == "colours":
            for i in range(NUM_SPEC_COLOURS + 1):
                for j in range(3):
                    c.colours[i][j] = int(v[i][j])
        elif k == "ratios":
            for i in range(NUM_SPEC_COLOURS):
                c.ratios[i] = float(v[i])
        else:
            setattr(c, k, v)
    return c


_lib = None
LIB_PATH = _build.LIB

# every symbol include/sgz.h declares (tests check that the library exports all of them)
EXPORTS = [
    "sgz_last_error", "sgz_abi_version", "sgz_device_count", "sgz_set_device",
    "sgz_plan_create", "sgz_plan_destroy", "sgz_plan_upload", "sgz_plan_transform_size",
    "sgz_plan_window_scale", "sgz_plan_break_pixel", "sgz_plan_path", "sgz_plan_dc_pixels", "sgz_plan_get_window",
    "sgz_plan_get_mapped_frequencies", "sgz_plan_get_slope_map", "sgz_plan_get_colour_ratios",
    "sgz_plan_get_colour_table", "sgz_rotate_hue_rgb8", "sgz_num_frames", "sgz_plan_num_frames", "sgz_plan_get_resonator", "sgz_plan_reset_resonator",
    "sgz_render_queue_create", "sgz_render_queue_destroy", "sgz_render_queue_submit", "sgz_render_queue_wait", "sgz_render_queue_join", "sgz_render_queue_set_option", "sgz_render_queue_distinct_lanes",
    "sgz_spectrogram_render_device", "sgz_spectrogram_render", "sgz_stage_bins", "sgz_stage_mapped", "sgz_stage_mapped_dominant", "sgz_plan_set_option",
    "sgz_stage_map_from_bins", "sgz_stage_track_peak", "sgz_spectrum_track_peak", "sgz_track_peak_lines", "sgz_spectrum_track_peak_lines", "sgz_stage_decay_colour", "sgz_stage_decay_scan", "sgz_stage_decay_emit", "sgz_stage_logf", "sgz_stage_finish_pixel", "sgz_decay_fold_carry", "sgz_comm_unique_id", "sgz_comm_create", "sgz_comm_destroy", "sgz_shard_layout", "sgz_spectrogram_render_sharded_on",
    "sgz_spectrogram_render_sharded", "sgz_peer_group_create", "sgz_peer_group_destroy", "sgz_peer_transport", "sgz_peer_transport_release",
    "sgz_spectrum_create", "sgz_spectrum_destroy", "sgz_spectrum_configure", "sgz_spectrum_push",
    "sgz_spectrum_pop_column", "sgz_spectrum_line_results", "sgz_spectrum_clear_state", "sgz_spectrum_set_mix",
    "sgz_spectrogram_render_host", "sgz_spectrum_stats", "sgz_spectrum_history", "sgz_spectrum_bind_image", "sgz_spectrum_create_image", "sgz_spectrum_bind_gl_buffer",
    "sgz_spectrum_flush_columns", "sgz_spectrum_render_lines", "sgz_spectrum_set_option",
    "sgz_scope_create", "sgz_scope_destroy", "sgz_scope_configure", "sgz_scope_set_option", "sgz_vector_set_option", "sgz_scope_stream", "sgz_vector_stream", "sgz_scope_push", "sgz_scope_peak_filter", "sgz_scope_gains",
    "sgz_scope_vertex_count", "sgz_scope_vertices", "sgz_scope_vertices_all", "sgz_scope_front", "sgz_scope_debug_state", "sgz_scope_analyse",
    "sgz_scope_front_colours", "sgz_scope_vertices_device", "sgz_vector_vertices_device", "sgz_export_alloc", "sgz_export_free",
    "sgz_vector_create", "sgz_vector_destroy", "sgz_vector_configure", "sgz_vector_push", "sgz_vector_peak_filter",
    "sgz_vector_filters_get", "sgz_vector_vertices", "sgz_vector_vertices_all", "sgz_spectrum_backlog", "sgz_spectrum_stream", "sgz_spectrum_flush", "sgz_scope_flush", "sgz_scope_set_transport", "sgz_vector_flush", "sgz_vector_history",
    "sgz_scope_num_points", "sgz_scope_lanczos_device", "sgz_scope_zero_crossing_device",
    "sgz_peak_filter_device", "sgz_vector_polar_device", "sgz_vector_audio_processing_device",
]


def lib() -> C.CDLL:
    """Load libsgz.so (building it with hipcc if the in-tree .so is missing or stale)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SGZ_LIB", LIB_PATH)          # (SGZ_LIB: another build of the library, for A/B timing on one box: tools/ab.sh)
    if not os.path.exists(path):
        _build.build()
    # PyTorch-ROCm bundles its own HIP/HSA runtime: if torch is going to share this process it must be
    # loaded first so that libsgz.so binds to the same runtime instance (two runtimes cannot both own the GPU).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(path)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    L.sgz_last_error.restype = C.c_char_p
    L.sgz_plan_create.argtypes = [C.POINTER(SpectrumConfig), C.POINTER(vp)]
    L.sgz_plan_destroy.argtypes = [vp]
    L.sgz_plan_destroy.restype = None
    L.sgz_plan_upload.argtypes = [vp]
    L.sgz_render_queue_create.argtypes = [C.POINTER(SpectrumConfig), u32, C.POINTER(vp)]
    L.sgz_render_queue_destroy.argtypes = [vp]
    L.sgz_render_queue_destroy.restype = None
    L.sgz_render_queue_submit.argtypes = [vp, vp, sz, sz, vp, vp, C.POINTER(C.c_uint64)]
    L.sgz_render_queue_wait.argtypes = [vp, C.c_uint64]
    L.sgz_render_queue_join.argtypes = [vp, vp]
    L.sgz_render_queue_set_option.argtypes = [vp, u32, u32]
    L.sgz_render_queue_distinct_lanes.argtypes = [vp]
    L.sgz_render_queue_distinct_lanes.restype = u32
    L.sgz_plan_transform_size.argtypes = [vp]
    L.sgz_plan_transform_size.restype = u32
    L.sgz_plan_window_scale.argtypes = [vp]
    L.sgz_plan_window_scale.restype = C.c_double
    L.sgz_plan_break_pixel.argtypes = [vp]
    L.sgz_plan_break_pixel.restype = u32
    L.sgz_plan_path.argtypes = [vp]
    L.sgz_plan_path.restype = u32
    L.sgz_plan_dc_pixels.argtypes = [vp, vp, u32]
    L.sgz_plan_dc_pixels.restype = u32
    for f in ("sgz_plan_get_window", "sgz_plan_get_mapped_frequencies", "sgz_plan_get_slope_map",
              "sgz_plan_get_colour_ratios"):
        getattr(L, f).argtypes = [vp, vp]
    L.sgz_plan_get_colour_table.argtypes = [vp, u32, vp]
    L.sgz_rotate_hue_rgb8.argtypes = [vp, C.c_float, vp]
    L.sgz_rotate_hue_rgb8.restype = None
    L.sgz_num_frames.argtypes = [sz, u32, u32]
    L.sgz_num_frames.restype = C.c_long
    L.sgz_plan_num_frames.argtypes = [vp, sz]
    L.sgz_plan_num_frames.restype = C.c_uint64
    L.sgz_plan_get_resonator.argtypes = [vp, vp, vp, vp, vp]
    L.sgz_plan_reset_resonator.argtypes = [vp, vp]
    L.sgz_scope_set_transport.argtypes = [vp, C.c_int64]
    L.sgz_spectrogram_render_device.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp]
    L.sgz_spectrogram_render.argtypes = [C.POINTER(SpectrumConfig), vp, u32, sz, vp, vp, C.POINTER(Timing)]
    L.sgz_spectrogram_render_host.argtypes = [vp, vp, u32, sz, vp, vp, C.POINTER(Timing)]
    L.sgz_stage_bins.argtypes = [vp, vp, sz, sz, vp, vp]
    L.sgz_stage_mapped.argtypes = [vp, vp, sz, sz, vp, vp]
    L.sgz_stage_mapped_dominant.argtypes = [vp, vp, sz, sz, vp, vp]
    L.sgz_stage_map_from_bins.argtypes = [vp, vp, sz, vp, vp]
    L.sgz_stage_decay_colour.argtypes = [vp, vp, sz, vp, vp, vp, vp]
    L.sgz_decay_fold_carry.argtypes = [vp, vp, vp, u32, u32, vp, vp]
    L.sgz_stage_logf.argtypes = [vp, vp, sz, vp]
    L.sgz_plan_set_option.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.sgz_stage_finish_pixel.argtypes = [vp, vp, sz, vp]
    L.sgz_stage_track_peak.argtypes = [vp, vp, C.c_double, C.POINTER(Peak), vp]
    L.sgz_spectrum_track_peak.argtypes = [vp, u32, C.c_double, C.POINTER(Peak)]
    L.sgz_track_peak_lines.argtypes = [vp, vp, C.c_double, C.POINTER(LinePeak)]
    L.sgz_spectrum_track_peak_lines.argtypes = [vp, u32, u32, C.c_double, C.POINTER(LinePeak)]
    L.sgz_comm_unique_id.argtypes = [vp]
    L.sgz_comm_create.argtypes = [vp, u32, u32, C.POINTER(vp)]
    L.sgz_comm_destroy.argtypes = [vp]
    L.sgz_comm_destroy.restype = None
    L.sgz_shard_layout.argtypes = [vp, u32, u32, sz] + [C.POINTER(C.c_uint64)] * 4
    L.sgz_spectrogram_render_sharded.argtypes = [vp, vp, u32, u32, vp, sz, sz, vp, C.POINTER(C.c_uint64), vp]
    L.sgz_stage_decay_scan.argtypes = [vp, vp, sz, vp, vp]
    L.sgz_stage_decay_emit.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp]
    L.sgz_spectrum_create.argtypes = [C.POINTER(SpectrumConfig), C.POINTER(vp)]
    L.sgz_spectrum_destroy.argtypes = [vp]
    L.sgz_spectrum_destroy.restype = None
    L.sgz_spectrum_configure.argtypes = [vp, C.POINTER(SpectrumConfig)]
    L.sgz_spectrum_push.argtypes = [vp, vp, u32, u32]
    L.sgz_spectrum_pop_column.argtypes = [vp, vp, C.POINTER(u32)]
    L.sgz_spectrum_line_results.argtypes = [vp, u32, u32, vp]
    L.sgz_spectrum_render_lines.argtypes = [vp, vp, vp]
    L.sgz_spectrum_set_option.argtypes = [vp, u32, C.c_uint64]
    L.sgz_spectrum_clear_state.argtypes = [vp]
    L.sgz_spectrum_set_mix.argtypes = [vp, u32, vp]
    L.sgz_spectrum_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sgz_spectrum_history.argtypes = [vp, u32, vp]
    L.sgz_spectrum_bind_image.argtypes = [vp, vp, u32, sz]
    L.sgz_spectrum_create_image.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_int)]
    L.sgz_spectrum_bind_gl_buffer.argtypes = [vp, C.c_uint, u32, sz]
    L.sgz_spectrum_flush_columns.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.sgz_scope_create.argtypes = [C.POINTER(ScopeConfig), C.POINTER(vp)]
    L.sgz_scope_set_option.argtypes = [vp, u32, C.c_uint64]
    L.sgz_scope_stream.argtypes = [vp]
    L.sgz_scope_stream.restype = vp
    L.sgz_vector_stream.argtypes = [vp]
    L.sgz_vector_stream.restype = vp
    L.sgz_vector_set_option.argtypes = [vp, u32, C.c_uint64]
    L.sgz_scope_destroy.argtypes = [vp]
    L.sgz_scope_destroy.restype = None
    L.sgz_scope_configure.argtypes = [vp, C.POINTER(ScopeConfig)]
    L.sgz_scope_push.argtypes = [vp, vp, u32, u32]
    L.sgz_scope_peak_filter.argtypes = [vp, C.c_double, u32, C.POINTER(C.c_double)]
    L.sgz_scope_gains.argtypes = [vp, C.POINTER(C.c_double), vp]
    L.sgz_scope_vertex_count.argtypes = [vp, C.POINTER(ScopeView)]
    L.sgz_scope_vertex_count.restype = sz
    L.sgz_scope_vertices.argtypes = [vp, C.POINTER(ScopeView), u32, u32, vp, vp, C.POINTER(u32)]
    L.sgz_scope_front.argtypes = [vp, u32, vp, C.POINTER(u32), C.POINTER(u32)]
    L.sgz_scope_debug_state.argtypes = [vp, vp]
    L.sgz_scope_analyse.argtypes = [vp, u32, u32, C.POINTER(TriggerState)]
    L.sgz_scope_front_colours.argtypes = [vp, u32, u32, vp]
    L.sgz_scope_vertices_device.argtypes = [vp, C.POINTER(ScopeView), u32, u32, vp, vp, C.POINTER(u32)]
    L.sgz_scope_vertices_all.argtypes = [vp, C.POINTER(ScopeView), u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(vp), C.POINTER(vp), C.POINTER(u32)]
    L.sgz_vector_vertices_device.argtypes = [vp, u32, vp, vp, C.POINTER(u32)]
    L.sgz_export_alloc.argtypes = [sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_int)]
    L.sgz_export_free.argtypes = [vp]
    L.sgz_export_free.restype = None
    L.sgz_vector_create.argtypes = [C.POINTER(VectorConfig), C.POINTER(vp)]
    L.sgz_vector_destroy.argtypes = [vp]
    L.sgz_vector_destroy.restype = None
    L.sgz_vector_configure.argtypes = [vp, C.POINTER(VectorConfig)]
    L.sgz_vector_push.argtypes = [vp, vp, u32, u32]
    L.sgz_vector_peak_filter.argtypes = [vp, C.c_double, C.POINTER(C.c_double)]
    L.sgz_vector_filters_get.argtypes = [vp, C.POINTER(VectorFilters), C.POINTER(C.c_double)]
    L.sgz_vector_vertices.argtypes = [vp, u32, vp, vp, C.POINTER(u32)]
    L.sgz_vector_vertices_all.argtypes = [vp, vp, vp, C.POINTER(u32)]
    L.sgz_vector_history.argtypes = [vp, u32, vp, C.POINTER(u32), C.POINTER(u32)]
    L.sgz_scope_num_points.argtypes = [C.POINTER(ScopeView)]
    L.sgz_scope_num_points.restype = sz
    L.sgz_scope_lanczos_device.argtypes = [C.POINTER(ScopeView), vp, sz, sz, u32, vp, vp]
    L.sgz_scope_zero_crossing_device.argtypes = [C.POINTER(ZeroCrossingState), u32, vp, vp, sz, vp, sz,
                                                 C.POINTER(sz), vp]
    L.sgz_peak_filter_device.argtypes = [vp, sz, u32, sz, u32, C.c_double, vp, C.POINTER(C.c_double), vp]
    L.sgz_vector_polar_device.argtypes = [vp, sz, u32, sz, u32, vp, vp]
    L.sgz_vector_audio_processing_device.argtypes = [C.POINTER(VectorFilters), vp, vp, sz, u32, C.c_float,
                                                     C.c_float, C.c_float, C.c_int, C.POINTER(C.c_float), vp]
    _lib = L
    return L


def check(status: int) -> int:
    if status < 0:
        raise SgzError(status, (lib().sgz_last_error() or b"").decode(errors="replace"))
    return status


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


RT_OPT_STRICT_REFERENCE_QUIRKS, RT_OPT_AUDIO_HISTORY, RT_OPT_DEFER_SUBMIT, RT_OPT_PARK_PUSHES = 1, 2, 3, 4
OPT_CHANNEL_SPLIT, OPT_FUSED_COLOUR, OPT_FETCH_WINDOW, OPT_MATRIX_RESONATOR, OPT_RESONATOR_SLAB, OPT_WIDE_GROUPS = 1, 2, 3, 4, 5, 6
OPT_RESONATOR_SHARD_BOUND = 7
OPT_PIPELINED = 8


class Plan:
    """The Spectrum constant block (TransformConstant mirror). Host tables need no GPU."""

    def __init__(self, cfg: dict | SpectrumConfig):
        self.cfg = cfg if isinstance(cfg, SpectrumConfig) else config_from_dict(cfg)
        h = C.c_void_p()
        check(lib().sgz_plan_create(C.byref(self.cfg), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().sgz_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self):
        check(lib().sgz_plan_upload(self.h))
        return self

    def set_option(self, option: int, value: int):
        """sgz_plan_set_option: OPT_CHANNEL_SPLIT / OPT_FUSED_COLOUR / OPT_FETCH_WINDOW"""
        check(lib().sgz_plan_set_option(self.h, option, value))
        return self

    @property
    def N(self) -> int:
        return lib().sgz_plan_transform_size(self.h)

    @property
    def P(self) -> int:
        return self.cfg.axis_points

    @property
    def C(self) -> int:
        return self.cfg.num_pairs

    @property
    def sides(self) -> int:
        return 2 if self.cfg.channel_mode in (4, 5, 6) else 1      # Phase (4): magnitude and cancellation planes

    @property
    def window_scale(self) -> float:
        return lib().sgz_plan_window_scale(self.h)

    @property
    def break_pixel(self) -> int:
        return lib().sgz_plan_break_pixel(self.h)

    @property
    def path(self) -> int:
        """SGZ_PATH_*: 0 generic, 1 fused, 2 halves (+4: per-side LDS map usable)"""
        return lib().sgz_plan_path(self.h)

    def dc_pixels(self) -> np.ndarray:
        n = lib().sgz_plan_dc_pixels(self.h, None, 0)
        out = np.zeros(max(n, 1), np.uint32)
        lib().sgz_plan_dc_pixels(self.h, _np_ptr(out), n)
        return out[:n]

    def window(self) -> np.ndarray:
        out = np.zeros(self.N, np.float32)
        check(lib().sgz_plan_get_window(self.h, _np_ptr(out)))
        return out

    def mapped_frequencies(self) -> np.ndarray:
        out = np.zeros(self.P, np.float32)
        check(lib().sgz_plan_get_mapped_frequencies(self.h, _np_ptr(out)))
        return out

    def slope_map(self) -> np.ndarray:
        out = np.zeros(self.P, np.float32)
        check(lib().sgz_plan_get_slope_map(self.h, _np_ptr(out)))
        return out

    def colour_ratios(self) -> np.ndarray:
        out = np.zeros(NUM_SPEC_COLOURS + 1, np.float32)
        check(lib().sgz_plan_get_colour_ratios(self.h, _np_ptr(out)))
        return out

    def track_peak_lines(self, results: np.ndarray, mouse_fraction: float) -> dict:
        """sgz_track_peak_lines: the tracker's line-results branch on host results float2 [P] (no upload needed: host arithmetic)"""
        r = np.ascontiguousarray(results, np.float32)
        assert r.size == 2 * self.P
        out = LinePeak()
        check(lib().sgz_track_peak_lines(self.h, _np_ptr(r), float(mouse_fraction), C.byref(out)))
        return out.asdict()

    def colour_table(self, pair: int) -> np.ndarray:
        out = np.zeros((NUM_SPEC_COLOURS + 1, 3), np.float32)
        check(lib().sgz_plan_get_colour_table(self.h, pair, _np_ptr(out)))
        return out

    def num_frames(self, nsamples: int) -> int:
        return int(lib().sgz_plan_num_frames(self.h, nsamples))

    def resonator(self):
        """RSNT plans: (coeff [V][P] complex64, gain [P], weights [V]) of the resonator bank."""
        V = C.c_uint32(0)
        check(lib().sgz_plan_get_resonator(self.h, C.byref(V), None, None, None))
        coeff = np.zeros((V.value, self.P), np.complex64)
        gain = np.zeros(self.P, np.float32)
        weights = np.zeros(V.value, np.float32)
        check(lib().sgz_plan_get_resonator(self.h, None, _np_ptr(coeff), _np_ptr(gain), _np_ptr(weights)))
        return coeff, gain, weights

    def reset_resonator(self, stream=None):
        import torch
        check(lib().sgz_plan_reset_resonator(self.h, stream if stream is not None else torch.cuda.current_stream().cuda_stream))

    # ---- device entry points (torch tensors on the GPU) -------------------------------------------
    def render(self, planar, rgba=None, lines=None, state=None, stream=None):
        """planar: torch.float32 [2*C, S] (cuda, contiguous rows). Returns rgba uint8 [F, P, 4]."""
        import torch
        assert planar.is_cuda and planar.dtype == torch.float32 and planar.stride(1) == 1
        S = planar.shape[1]
        F = self.num_frames(S)
        if rgba is None:
            rgba = torch.empty((F, self.P, 4), dtype=torch.uint8, device=planar.device)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(lib().sgz_spectrogram_render_device(
            self.h, planar.data_ptr(), planar.stride(0), S, rgba.data_ptr(),
            lines.data_ptr() if lines is not None else None,
            state.data_ptr() if state is not None else None, s))
        return rgba

    def stage_bins(self, planar):
        import torch
        S = planar.shape[1]
        F = self.num_frames(S)
        # Phase mode keeps the bins complex: (re, im) pairs instead of magnitudes
        shape = (F, self.C, self.N + 1, 2) if self.cfg.channel_mode == 4 else (F, self.C, self.N + 1)
        out = torch.empty(shape, dtype=torch.float32, device=planar.device)
        check(lib().sgz_stage_bins(self.h, planar.data_ptr(), planar.stride(0), S, out.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream))
        return out

    def stage_mapped(self, planar):
        import torch
        S = planar.shape[1]
        F = self.num_frames(S)
        out = torch.empty((F, self.C, self.sides, self.P), dtype=torch.float32, device=planar.device)
        check(lib().sgz_stage_mapped(self.h, planar.data_ptr(), planar.stride(0), S, out.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream))
        return out

    def stage_map_from_bins(self, bins):
        import torch
        F = bins.shape[0]
        out = torch.empty((F, self.C, self.sides, self.P), dtype=torch.float32, device=bins.device)
        check(lib().sgz_stage_map_from_bins(self.h, bins.data_ptr(), F, out.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream))
        return out

    def stage_decay_colour(self, mapped, want_lines=False, state=None, want_rgba=True):
        import torch
        F = mapped.shape[0]
        rgba = torch.empty((F, self.P, 4), dtype=torch.uint8, device=mapped.device) if want_rgba else None
        lines = torch.empty((F, self.C, NUM_GRAPHS, self.P, 2), dtype=torch.float32, device=mapped.device) if want_lines else None
        check(lib().sgz_stage_decay_colour(self.h, mapped.data_ptr(), F, rgba.data_ptr() if want_rgba else None,
                                           lines.data_ptr() if want_lines else None,
                                           state.data_ptr() if state is not None else None,
                                           torch.cuda.current_stream().cuda_stream))
        return rgba, lines


    def fold_carry(self, aggs, frames_per_rank, rank: int, carry):
        """aggs: cuda float32 [world, C, G, P, 2]; carry: cuda float32 [C, G, P, 2] (out)."""
        import torch
        world = aggs.shape[0]
        fr = (C.c_int64 * world)(*[int(f) for f in frames_per_rank])
        check(lib().sgz_decay_fold_carry(self.h, aggs.data_ptr(), fr, world, rank, carry.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream))
        return carry


class RenderQueue:
    """sgz_render_queue: `depth` lanes of (plan, stream); renders of independent device buffers submitted round-robin (sgz.h)."""

    def __init__(self, cfg: dict, depth: int = 3):
        self.cfg = config_from_dict(cfg)
        self.h = C.c_void_p()
        check(lib().sgz_render_queue_create(C.byref(self.cfg), depth, C.byref(self.h)))
        self.depth = depth
        self.distinct_lanes = int(lib().sgz_render_queue_distinct_lanes(self.h))

    def submit(self, planar, rgba, after_stream=None) -> int:
        """planar: torch.float32 [2*C, S] (cuda), rgba: torch.uint8 [F, P, 4] (cuda); returns the ticket"""
        t = C.c_uint64(0)
        check(lib().sgz_render_queue_submit(self.h, C.c_void_p(planar.data_ptr()), C.c_size_t(planar.stride(0)), C.c_size_t(planar.shape[1]),
                                            C.c_void_p(rgba.data_ptr()), C.c_void_p(after_stream) if after_stream else None, C.byref(t)))
        return int(t.value)

    def wait(self, ticket: int = 0):
        check(lib().sgz_render_queue_wait(self.h, C.c_uint64(ticket)))

    def join(self, stream: int):
        check(lib().sgz_render_queue_join(self.h, C.c_void_p(stream)))

    def set_option(self, option: int, value: int):
        check(lib().sgz_render_queue_set_option(self.h, option, value))
        return self

    def close(self):
        if self.h:
            lib().sgz_render_queue_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:                                           # noqa: BLE001
            pass


def render_spectrogram(cfg: dict, planar: np.ndarray, want_lines: bool = False):
    """Host-buffer batch render (sgz_spectrogram_render). planar: float32 [2*C, S]."""
    c = config_from_dict(cfg)
    planar = np.ascontiguousarray(planar, np.float32)
    nch, S = planar.shape
    F = S // c.hop if c.algorithm == 1 else lib().sgz_num_frames(S, c.window_size, c.hop)
    rgba = np.zeros((F, c.axis_points, 4), np.uint8)
    lines = np.zeros((F, c.num_pairs, NUM_GRAPHS, c.axis_points, 2), np.float32) if want_lines else None
    ptrs = (C.c_void_p * nch)(*[planar[i].ctypes.data for i in range(nch)])
    t = Timing()
    check(lib().sgz_spectrogram_render(C.byref(c), ptrs, nch, S, _np_ptr(rgba),
                                       _np_ptr(lines) if want_lines else None, C.byref(t)))
    return rgba, lines, {"h2d_ms": t.h2d_ms, "kernel_ms": t.kernel_ms, "d2h_ms": t.d2h_ms, "frames": t.frames}


def render_spectrogram_host(plan, planar: np.ndarray, want_lines: bool = False):
    """Host-buffer render on a plan the caller keeps (sgz_spectrogram_render_host): no table rebuild, no allocation after the first call."""
    planar = np.ascontiguousarray(planar, np.float32)
    nch, S = planar.shape
    F = plan.num_frames(S)
    rgba = np.zeros((F, plan.P, 4), np.uint8)
    lines = np.zeros((F, plan.cfg.num_pairs, NUM_GRAPHS, plan.P, 2), np.float32) if want_lines else None
    ptrs = (C.c_void_p * nch)(*[planar[i].ctypes.data for i in range(nch)])
    t = Timing()
    check(lib().sgz_spectrogram_render_host(plan.h, ptrs, nch, S, _np_ptr(rgba), _np_ptr(lines) if want_lines else None, C.byref(t)))
    return rgba, lines, {"h2d_ms": t.h2d_ms, "kernel_ms": t.kernel_ms, "d2h_ms": t.d2h_ms, "frames": t.frames}


def rotate_hue(rgb, amount: float) -> np.ndarray:
    a = np.asarray(rgb, np.uint8)
    out = np.zeros(3, np.uint8)
    lib().sgz_rotate_hue_rgb8(_np_ptr(a), C.c_float(amount), _np_ptr(out))
    return out


class Scope:
    """sgz_scope_* handle: the Oscilloscope's audio-thread state machine in HBM + drawWavePlot vertices."""

    def __init__(self, **kw):
        self.cfg = ScopeConfig()
        colours = kw.pop("colours", None)
        bands = kw.pop("band_colours", None)
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        for c in range(64):
            col = colours[c] if colours is not None and c < len(colours) else (255, 255, 255, 255)
            for j in range(4):
                self.cfg.colours[c][j] = int(col[j])
        if bands is not None:
            for i in range(3):
                for j in range(3):
                    self.cfg.band_colours[i][j] = float(bands[i][j])
        self.h = C.c_void_p()
        check(lib().sgz_scope_create(C.byref(self.cfg), C.byref(self.h)))

    def set_option(self, option: int, value: int):
        """sgz_scope_set_option (RT_OPT_DEFER_SUBMIT)"""
        check(lib().sgz_scope_set_option(self.h, option, value))
        return self

    def close(self):
        if getattr(self, "h", None):
            lib().sgz_scope_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def configure(self, **kw):
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        check(lib().sgz_scope_configure(self.h, C.byref(self.cfg)))

    def push(self, block: np.ndarray) -> int:
        b = np.ascontiguousarray(block, np.float32)
        ptrs = (C.c_void_p * b.shape[0])(*[b[c].ctypes.data for c in range(b.shape[0])])
        return check(lib().sgz_scope_push(self.h, ptrs, b.shape[0], b.shape[1]))

    def flush(self):
        """blocks that waited for a staging slot are enqueued now (sgz_scope_flush): readers of results call it first"""
        check(lib().sgz_scope_flush(self.h))

    def set_transport(self, position_in_samples: int):
        """cs.transportPosition (TriggeringMode::Window)"""
        check(lib().sgz_scope_set_transport(self.h, C.c_int64(int(position_in_samples))))

    def front(self, channel: int):
        self.flush()
        size, cur = C.c_uint32(0), C.c_uint32(0)
        check(lib().sgz_scope_front(self.h, channel, None, C.byref(size), C.byref(cur)))
        out = np.zeros(size.value, np.float32)
        check(lib().sgz_scope_front(self.h, channel, _np_ptr(out), C.byref(size), C.byref(cur)))
        return out, int(cur.value)

    def front_colours(self, channel: int, aux: bool = False) -> np.ndarray:
        self.flush()
        """colour ring memory beside front(channel): uint32 RGBA8 words [size]"""
        size = C.c_uint32(0)
        check(lib().sgz_scope_front(self.h, channel, None, C.byref(size), None))
        out = np.zeros(size.value, np.uint32)
        check(lib().sgz_scope_front_colours(self.h, channel, int(aux), _np_ptr(out)))
        return out

    def analyse(self, evaluator: int = 0, channel: int = 0) -> TriggerState:
        self.flush()
        ts = TriggerState()
        check(lib().sgz_scope_analyse(self.h, evaluator, channel, C.byref(ts)))
        return ts

    def state(self) -> dict:
        self.flush()
        out = np.zeros(8, np.uint64)
        check(lib().sgz_scope_debug_state(self.h, _np_ptr(out)))
        keys = ("frontOrigin", "bufferedSamples", "oldPeak", "currentPeak", "steadyClock", "peaks", "isWorkingOnPeak", "swaps")
        return {k: int(v) for k, v in zip(keys, out)}

    def gains(self):
        self.flush()
        g = C.c_double(0)
        env = np.zeros(self.cfg.num_channels, np.float32)
        check(lib().sgz_scope_gains(self.h, C.byref(g), _np_ptr(env)))
        return g.value, env

    def peak_filter(self, delta_time: float, lanes: int = 8) -> float:
        self.flush()
        g = C.c_double(0)
        check(lib().sgz_scope_peak_filter(self.h, delta_time, lanes, C.byref(g)))
        return g.value

    def vertices(self, view: ScopeView, evaluator: int, channel: int = 0, want_colours: bool = True, out=None):
        """out: (xyz float32 [>= n][3], rgba uint8 [>= n][4] or None) buffers the caller keeps (pinned ones make the read-back a DMA);
        default: fresh arrays"""
        self.flush()
        n = lib().sgz_scope_vertex_count(self.h, C.byref(view))
        if out is not None:
            xyz, rgba = out
            assert xyz.shape[0] >= n and (rgba is None or rgba.shape[0] >= n)
            want_colours = rgba is not None
        else:
            xyz = np.empty((n, 3), np.float32)
            rgba = np.empty((n, 4), np.uint8) if want_colours else None
        cnt = C.c_uint32(n)
        check(lib().sgz_scope_vertices(self.h, C.byref(view), evaluator, channel, _np_ptr(xyz),
                                       _np_ptr(rgba) if want_colours else None, C.byref(cnt)))
        return xyz[:cnt.value], (rgba[:cnt.value] if want_colours else None)


    def vertices_all(self, view: ScopeView, evaluators, channels, out):
        """sgz_scope_vertices_all: out = [(xyz, rgba or None), ...] one per item, buffers the caller keeps; one wait for all of them"""
        self.flush()
        k = len(evaluators)
        n = lib().sgz_scope_vertex_count(self.h, C.byref(view))
        ev = (C.c_uint32 * k)(*evaluators); ch = (C.c_uint32 * k)(*channels)
        xs = (C.c_void_p * k)(*[o[0].ctypes.data for o in out])
        want = all(o[1] is not None for o in out)
        cs = (C.c_void_p * k)(*[o[1].ctypes.data if o[1] is not None else None for o in out])
        cnt = (C.c_uint32 * k)(*[o[0].shape[0] for o in out])
        check(lib().sgz_scope_vertices_all(self.h, C.byref(view), k, ev, ch, xs, cs if want else None, cnt))
        return [(o[0][:cnt[i]], o[1][:cnt[i]] if o[1] is not None else None) for i, o in enumerate(out)]


class Vector:
    """sgz_vector_* handle: history ring + audio-thread filters + polar vertices in HBM."""

    def __init__(self, **kw):
        self.cfg = VectorConfig()
        colours = kw.pop("colours", None)
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        for p in range(32):
            col = colours[p] if colours is not None and p < len(colours) else (1.0, 1.0, 1.0)
            for j in range(3):
                self.cfg.colours[p][j] = float(col[j])
        self.h = C.c_void_p()
        check(lib().sgz_vector_create(C.byref(self.cfg), C.byref(self.h)))

    def set_option(self, option: int, value: int):
        """sgz_vector_set_option (RT_OPT_DEFER_SUBMIT)"""
        check(lib().sgz_vector_set_option(self.h, option, value))
        return self

    def close(self):
        if getattr(self, "h", None):
            lib().sgz_vector_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, block: np.ndarray) -> int:
        b = np.ascontiguousarray(block, np.float32)
        ptrs = (C.c_void_p * b.shape[0])(*[b[c].ctypes.data for c in range(b.shape[0])])
        return check(lib().sgz_vector_push(self.h, ptrs, b.shape[0], b.shape[1]))

    def flush(self):
        """blocks that waited for a staging slot are enqueued now (sgz_vector_flush): readers of results call it first"""
        check(lib().sgz_vector_flush(self.h))

    def history(self, channel: int):
        self.flush()
        size, cur = C.c_uint32(0), C.c_uint32(0)
        out = np.zeros(self.cfg.window_size, np.float32)
        check(lib().sgz_vector_history(self.h, channel, _np_ptr(out), C.byref(size), C.byref(cur)))
        return out, int(cur.value)

    def filters(self):
        self.flush()
        f, g = VectorFilters(), C.c_double(0)
        check(lib().sgz_vector_filters_get(self.h, C.byref(f), C.byref(g)))
        return f, g.value

    def peak_filter(self, delta_time: float) -> float:
        self.flush()
        g = C.c_double(0)
        check(lib().sgz_vector_peak_filter(self.h, delta_time, C.byref(g)))
        return g.value

    def vertices_all(self, want_colours: bool = True, out=None):
        """out: (xyz float32 [pairs][n][3], rgb float32 [pairs][n][3] or None) buffers the caller keeps; default: fresh arrays"""
        self.flush()
        n, pairs = self.cfg.window_size, self.cfg.num_channels // 2
        if out is not None:
            xyz, rgb = out
            assert xyz.shape == (pairs, n, 3) and (rgb is None or rgb.shape == (pairs, n, 3))
            want_colours = rgb is not None
        else:
            xyz = np.empty((pairs, n, 3), np.float32)
            rgb = np.empty((pairs, n, 3), np.float32) if want_colours else None
        cnt = C.c_uint32(n)
        check(lib().sgz_vector_vertices_all(self.h, _np_ptr(xyz), _np_ptr(rgb) if want_colours else None, C.byref(cnt)))
        return xyz, rgb

    def vertices(self, pair: int = 0, want_colours: bool = True):
        self.flush()
        n = self.cfg.window_size
        xyz = np.zeros((n, 3), np.float32)
        rgb = np.zeros((n, 3), np.float32) if want_colours else None
        cnt = C.c_uint32(n)
        check(lib().sgz_vector_vertices(self.h, pair, _np_ptr(xyz), _np_ptr(rgb) if want_colours else None, C.byref(cnt)))
        return xyz, rgb
