"""Random spectrum configurations for the parity sweeps (tools/fuzz_parity.py, tests/test_gpu_fuzz.py)."""
import numpy as np

from signalizer_amd import config


def random_config(rng, wild=False):
    W = int(rng.choice([64, 100, 512, 1000, 1024, 2048, 3000, 4096, 5000, 8192, 16384, 20000, 32768, 40000, 65536]))
    if rng.random() < 0.15:
        W = int(rng.integers(33, 9000))
    hop = max(1, int(W * rng.choice([0.25, 0.5, 0.3, 1.0])))
    mode = int(rng.choice([config.CH_LEFT, config.CH_RIGHT, config.CH_MERGE, config.CH_SIDE, config.CH_PHASE, config.CH_SEPARATE,
                           config.CH_MIDSIDE, config.CH_COMPLEX]))
    left = float(rng.choice([0.0, 0.0, 0.1, 0.35]))
    right = float(rng.choice([1.0, 1.0, 0.9, 0.6]))
    cfg = config.spectrum_config(
        sample_rate=float(rng.choice([44100.0, 48000.0, 96000.0, 192000.0])), window_size=W, hop=hop,
        axis_points=int(rng.choice([16, 77, 256, 300, 1024, 1500])), channel_mode=mode,
        bin_interp=int(rng.integers(0, 3)), view_scaling=int(rng.integers(0, 2)),
        window_type=int(rng.integers(0, 8)), window_symmetry=int(rng.integers(0, 2)), window_alpha=float(rng.uniform(0, 3)),
        window_beta=float(rng.uniform(0.5, 9)), num_pairs=int(rng.choice([1, 1, 2, 3])), view_left=left, view_right=right,
        min_log_freq=float(rng.choice([10.0, 20.0, 5.0])), low_db=float(rng.choice([-120.0, -90.0, -60.0])),
        high_db=float(rng.choice([0.0, 6.0])), slope_a=float(rng.choice([0.0, 0.3])), slope_b=float(rng.choice([1.0, 0.7])),
        pole=(float(rng.choice([0.0, 0.5, 0.9, 0.97])), float(rng.choice([0.9, 0.99, 0.999]))))
    if wild:       # arbitrary window lengths, zooms and heights
        W = int(rng.integers(1, 70000)) if rng.random() < 0.5 else W
        left = float(rng.uniform(0.0, 0.8))
        right = float(min(1.0, left + rng.uniform(0.02, 1.0)))
        cfg.update(window_size=W, hop=max(1, int(W * rng.uniform(0.05, 1.2))), view_left=left, view_right=right,
                   axis_points=int(rng.integers(2, 4000)), min_log_freq=float(rng.uniform(1.0, 200.0)),
                   num_pairs=int(rng.integers(1, 6)), low_db=float(rng.uniform(-150, -20)), high_db=float(rng.uniform(-10, 12)))
    return cfg



def rsnt_case(seed: int, index: int):
    """case `index` of tools/fuzz_rsnt.py's stream for `seed` (the same draws in the same order: the campaign records name cases this
    way): (config dict, frames, planar input)."""
    import numpy as np

    from signalizer_amd import config as cf, synth
    rng = np.random.default_rng(seed)
    for _ in range(index + 1):
        mode = int(rng.integers(0, 8))
        d = cf.spectrum_config(algorithm=cf.ALGO_RSNT, channel_mode=mode, window_type=int(rng.integers(0, 13)),
                               window_size=int(rng.choice([512, 4096, 32768])), hop=int(rng.choice([int(rng.integers(40, 3000)), 1024, 2048, 3072])),
                               axis_points=int(rng.integers(2, 1500)), num_pairs=int(rng.integers(1, 4)), free_q=int(rng.integers(0, 2)),
                               view_scaling=int(rng.integers(0, 2)), sample_rate=float(rng.choice([44100.0, 48000.0, 96000.0])),
                               view_left=float(rng.uniform(0, 0.3)), view_right=float(rng.uniform(0.5, 1.0)),
                               pole=(float(rng.uniform(0.5, 0.999)), float(rng.uniform(0.5, 0.999))))
        F = int(rng.integers(1, 20))
        x = synth.gen(int(rng.integers(1, 1000)), int(d["sample_rate"]), F * d["hop"] + int(rng.integers(0, d["hop"])), 2 * d["num_pairs"])
    return d, F, x
