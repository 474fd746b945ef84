"""Host-side configuration of the Spectrum path: a plain-dict mirror of Signalizer's
TransformConstant inputs (Source/Spectrum/TransformConstant.h:190-239, Spectrum.cpp:351-406),
and the BASELINE.json workloads."""
from __future__ import annotations

# SpectrumChannels (Source/Common/CommonSignalizer.h:495-539)
CH_LEFT, CH_RIGHT, CH_MERGE, CH_SIDE, CH_PHASE, CH_SEPARATE, CH_MIDSIDE, CH_COMPLEX = range(8)
INTERP_NONE, INTERP_LINEAR, INTERP_LANCZOS = range(3)
VIEW_LINEAR, VIEW_LOG = range(2)
(WIN_RECT, WIN_HANN, WIN_HAMMING, WIN_FLATTOP, WIN_BLACKMAN, WIN_EXACT_BLACKMAN, WIN_NUTTALL,
 WIN_BLACKMAN_NUTTALL, WIN_BLACKMAN_HARRIS, WIN_TRIANGULAR, WIN_WELCH, WIN_GAUSSIAN, WIN_KAISER) = range(13)
WIN_SYMMETRIC, WIN_PERIODIC = range(2)
ALGO_FFT, ALGO_RSNT = range(2)   # SpectrumContent::TransformAlgorithm (Source/Spectrum/SpectrumParameters.h:66-69)
DISPLAY_LINE_GRAPH, DISPLAY_COLOUR_SPECTRUM = range(2)   # SpectrumContent::DisplayMode (SpectrumParameters.h:60-64); real-time handle only

DEFAULT_COLOURS = [(0, 0, 0), (0, 0, 64), (0, 128, 255), (0, 255, 128), (255, 255, 0), (255, 64, 0)]


def spectrum_config(**over) -> dict:
    """Common spectrum parameters of SURVEY.md 8(d): Hann (periodic), Separate, Lanczos, log view [0,1],
    minFreq 10 Hz, P=1024, -120..0 dB, slope a=0 b=1, poles 0.9/0.99, 5-colour gradient on black."""
    cfg = dict(
        sample_rate=48000.0, window_size=32768, hop=8192, axis_points=1024,
        channel_mode=CH_SEPARATE, bin_interp=INTERP_LANCZOS, view_scaling=VIEW_LOG,
        window_type=WIN_HANN, window_symmetry=WIN_PERIODIC, num_pairs=1,
        window_alpha=0.0, window_beta=0.0, view_left=0.0, view_right=1.0, min_log_freq=10.0,
        low_db=-120.0, high_db=0.0, clip_db=-384.0, slope_a=0.0, slope_b=1.0,
        pole=(0.9, 0.99), colours=DEFAULT_COLOURS, ratios=(0.2, 0.2, 0.2, 0.2, 0.2),
        algorithm=ALGO_FFT, free_q=0, display_mode=DISPLAY_COLOUR_SPECTRUM,
    )
    cfg.update(over)
    return cfg


# BASELINE.json configs (index = position in BASELINE.json `configs`)
def cfg1() -> dict:   # stereo 48 kHz, single 4096-pt Hann frame
    return spectrum_config(window_size=4096, hop=4096)


def cfg2() -> dict:   # stereo 48 kHz, N=W=32768, hop 8192, 60 s => 348 frames
    return spectrum_config()


def cfg5(pairs: int = 32) -> dict:   # 64 ch 96 kHz, N=W=65536, hop 16384
    return spectrum_config(sample_rate=96000.0, window_size=65536, hop=16384, num_pairs=pairs)


CFG2_SECONDS = 60.0
CFG2_SEED = 2
