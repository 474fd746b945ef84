"""frames/s of the whole spectrogram path for other configurations (not the headline metric): cfg1-like N = 4096,
cfg5 N = 65536 (two half-frame workgroups + map kernel), N = 8192, N = 16384 (generic HBM-resident FFT passes), Phase mode."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signalizer_amd import api, config, synth
only = sys.argv[1] if len(sys.argv) > 1 else ""       # substring filter on the configuration name
def run(name, cfg, seconds, sr):
    if only not in name: return
    nch = 2 * cfg["num_pairs"]
    x = torch.from_numpy(synth.gen(7, sr, int(seconds * sr), nch)).cuda()
    plan = api.Plan(cfg).upload()
    F = plan.num_frames(x.shape[1])
    rgba = torch.empty((F, plan.P, 4), dtype=torch.uint8, device="cuda")
    for _ in range(3): plan.render(x, rgba=rgba)
    torch.cuda.synchronize(); t0 = time.perf_counter(); it = 0
    while time.perf_counter() - t0 < 0.5:
        plan.render(x, rgba=rgba); it += 1
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    print(json.dumps({"config": name, "frames": F, "pairs": cfg["num_pairs"], "N": plan.N, "us_per_render": dt * 1e6,
                      "transforms_per_s": F * cfg["num_pairs"] / dt}))
run("N=4096 stereo (cfg1 sizes), 60 s", config.spectrum_config(window_size=4096, hop=1024), 60, 48000)
run("N=32768 stereo cfg2", config.cfg2(), 60, 48000)
run("N=32768 stereo, Merge mode (one real signal per frame)", config.spectrum_config(window_size=32768, hop=8192, channel_mode=config.CH_MERGE), 60, 48000)
run("N=32768 stereo, Left mode", config.spectrum_config(window_size=32768, hop=8192, channel_mode=config.CH_LEFT), 60, 48000)
run("N=65536, 32 pairs 96 kHz, Merge mode", dict(config.cfg5(pairs=32), channel_mode=config.CH_MERGE), 20, 96000)
run("N=16384 stereo, Side mode, 60 s", config.spectrum_config(window_size=16384, hop=4096, channel_mode=config.CH_SIDE), 60, 48000)
run("N=32768 stereo, Phase mode", config.spectrum_config(window_size=32768, hop=8192, channel_mode=config.CH_PHASE), 60, 48000)
run("N=65536, 4 pairs 96 kHz (cfg5 sizes), 10 s", config.cfg5(pairs=4), 10, 96000)
run("N=65536, 32 pairs 96 kHz (cfg5, one GPU's 20 s chunk)", config.cfg5(pairs=32), 20, 96000)
run("N=8192 stereo, 60 s", config.spectrum_config(window_size=8192, hop=2048), 60, 48000)
run("N=16384 stereo, 60 s", config.spectrum_config(window_size=16384, hop=4096), 60, 48000)
run("N=2048 stereo (generic passes), 60 s", config.spectrum_config(window_size=2048, hop=512), 60, 48000)
run("N=1024 stereo (generic passes), 60 s", config.spectrum_config(window_size=1024, hop=256), 60, 48000)
