"""Handles are created and destroyed for as long as a host runs (every reconfiguration of a view makes new ones): device and host memory
come back."""
import ctypes as C
import gc

import numpy as np
import pytest

from signalizer_amd import api, config, synth

pytestmark = pytest.mark.gpu


def _free_device_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _rss():
    import psutil
    return psutil.Process().memory_info().rss


def _cycle_plan(x):
    cfg = config.spectrum_config(window_size=4096, hop=1024, num_pairs=2)
    p = api.Plan(cfg).upload()
    p.render(x)
    p.close()
    r = api.Plan(config.spectrum_config(algorithm=config.ALGO_RSNT, window_size=4096, hop=1024)).upload()
    r.render(x[:2])
    r.close()


def _cycle_queue(x):
    import torch
    q = api.RenderQueue(config.spectrum_config(window_size=4096, hop=1024), depth=3)
    outs = [torch.empty(((x.shape[1] - 4096) // 1024 + 1, 1024, 4), dtype=torch.uint8, device=x.device) for _ in range(3)]
    torch.cuda.synchronize()                                    # (the queue's lanes are not torch's stream)
    for o in outs:
        q.submit(x[:2], o)
    q.wait()
    q.close()


def _cycle_spectrum_stream(x):
    L = api.lib()
    c = api.config_from_dict(config.spectrum_config(window_size=4096, hop=1024, axis_points=300))
    h = C.c_void_p()
    api.check(L.sgz_spectrum_create(C.byref(c), C.byref(h)))
    blk = np.ascontiguousarray(x[:2, :2048].cpu().numpy())
    ptrs = (C.c_void_p * 2)(blk[0].ctypes.data, blk[1].ctypes.data)
    assert L.sgz_spectrum_push(h, ptrs, 2, 2048) in (api.SGZ_OK, api.SGZ_BUSY)
    L.sgz_spectrum_destroy(h)


def _cycle_scope(x):
    dev = api.Scope(sample_rate=192000.0, window_size=19200.0, num_channels=2, trigger_mode=4, channel_mode=0, envelope_mode=1, interpolation=3,
                    max_block=4096, trigger_threshold=0.05, trigger_channel=1.0, envelope_window=0.3)
    blk = x[:2, :3000].cpu().numpy()
    while dev.push(blk) != api.SGZ_OK:
        pass
    dev.vertices(api.ScopeView(19200.0, 0.0, 1.0, 1.0, 800, 0), 0, 0)
    dev.close()


def _cycle_vector(x):
    dev = api.Vector(sample_rate=96000.0, num_channels=4, window_size=9600, envelope_mode=1, lanes=8, fade_history=1, max_block=4096,
                     envelope_window=0.3, stereo_window=0.05)
    blk = x[:4, :3000].cpu().numpy()
    while dev.push(blk) != api.SGZ_OK:
        pass
    dev.vertices_all()
    dev.close()


@pytest.mark.parametrize("kind", ["plan", "queue", "spectrum_stream", "scope", "vector"])
def test_create_use_destroy_gives_the_memory_back(gpu, kind):
    """60 create -> use -> destroy cycles per handle type after 5 to settle the allocators: the device's free memory and the process's
    resident set end where they were (64 MiB / 96 MiB of slack: the runtime's own pools move by a few MiB; one leaked plan is ~50 MiB)"""
    import torch
    x = torch.from_numpy(synth.gen(3, 48000, 4096 + 1024 * 63, 4)).to(gpu)
    cycle = {"plan": _cycle_plan, "queue": _cycle_queue, "spectrum_stream": _cycle_spectrum_stream, "scope": _cycle_scope, "vector": _cycle_vector}[kind]
    for _ in range(5):
        cycle(x)
    gc.collect()
    free0, rss0 = _free_device_bytes(), _rss()
    for _ in range(60):
        cycle(x)
    gc.collect()
    free1, rss1 = _free_device_bytes(), _rss()
    import os
    if "PYTEST_XDIST_WORKER" not in os.environ:                   # (the figure is the DEVICE's: under pytest -n the other workers' allocations move it)
        assert free0 - free1 < 64 << 20, f"device memory: {(free0 - free1) / 2**20:.1f} MiB fewer free after 60 cycles"
    assert rss1 - rss0 < 96 << 20, f"host memory: resident set grew by {(rss1 - rss0) / 2**20:.1f} MiB over 60 cycles"


def test_no_result_depends_on_what_fresh_memory_held(gpu):
    """tools/poison_probe_all.py in a process of its own: the driver's free memory and every output buffer are filled with 0x00 in one
    run and 0xFF (NaNs) in the other before plans / handles are created and used -- 60 random spectrum configurations (image, line
    results, carried state, mapped magnitudes), the spectrum stream, the Oscilloscope, the Vectorscope: every output of the two runs is
    bit-identical (no read of uninitialised scratch, no output byte left unwritten).  200 configurations: profiles/r06d/poison_probe_all.txt"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "poison_probe_all.py"), "60", "5"], capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0 and "dependences on uninitialised memory: 0" in r.stdout, tail
    assert "vectorscope: 3 handles run twice: 0 depend on it" in r.stdout, tail


def test_no_result_depends_on_what_the_lds_or_the_registers_held(gpu):
    """tools/lds_poison_probe.py: neither the LDS nor the vector registers are cleared between workgroups -- a new workgroup inherits what the
    last one (of any kernel, of any process) left.  Every CU's LDS and every SIMD's register file (arch + acc) are filled with zeros, then
    with NaN / Inf / largest-float patterns (the tool checks that fresh workgroups really find the pattern: 1.000 of the words), and 18
    configurations over every K_A / K_B / RSNT form are rendered behind each fill: mapped magnitudes, image, line results and state are
    bit-identical throughout.  (Written while looking for the cause of rare wrong frames beside other processes' work: it excluded this
    class; the cause was the platform's, NOTES.md round 6.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "lds_poison_probe.py"), "2"], capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-12:])
    assert r.returncode == 0 and "dependences on the earlier contents of the LDS or the registers: 0" in r.stdout, tail
    assert r.stdout.count("in 1.000 of their LDS words, waves in 1.000 of the registers") == 2, tail
