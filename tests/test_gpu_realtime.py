"""Real-time per-block path (sgz_spectrum_push / pop_column) vs the offline render and the oracle."""
import ctypes as C
import time

import numpy as np
import pytest

from signalizer_amd import api, config, synth

pytestmark = pytest.mark.gpu


def _pop_all(h, P, want, timeout=10.0):
    cols = []
    t0 = time.time()
    buf = np.zeros((P, 4), np.uint8)
    ap = C.c_uint32(0)
    while len(cols) < want and time.time() - t0 < timeout:
        st = api.lib().sgz_spectrum_pop_column(h, buf.ctypes.data_as(C.c_void_p), C.byref(ap))
        if st == api.SGZ_OK:
            assert ap.value == P
            cols.append(buf.copy())
        else:
            assert st == api.SGZ_EMPTY
            time.sleep(0.001)
    return cols


@pytest.mark.parametrize("block,mode,W", [(256, config.CH_SEPARATE, 4096), (1024, config.CH_SEPARATE, 4096), (480, config.CH_SEPARATE, 4096),
                                          (512, config.CH_PHASE, 4096), (512, config.CH_MIDSIDE, 2048)])
def test_push_pop_matches_offline(gpu, oracle, block, mode, W):
    """history starts as W samples of silence; a column fires every `hop` samples.  The stream of columns must equal
    the offline render of [W zeros ++ audio] (and therefore the oracle, within the end-to-end tolerance)."""
    po = oracle
    cfg = config.spectrum_config(window_size=W, hop=1024, axis_points=300, channel_mode=mode)
    hop, P = 1024, 300
    nblocks = (9 * hop) // block
    S = nblocks * block
    x = synth.gen(8, 48000, S, 2)
    c = api.config_from_dict(cfg)
    h = C.c_void_p()
    api.check(api.lib().sgz_spectrum_create(C.byref(c), C.byref(h)))
    cols = []
    try:
        for b in range(nblocks):
            blk = np.ascontiguousarray(x[:, b * block:(b + 1) * block])
            ptrs = (C.c_void_p * 2)(blk[0].ctypes.data, blk[1].ctypes.data)
            api.check(api.lib().sgz_spectrum_push(h, ptrs, 2, block))
            cols += _pop_all(h, P, 1, timeout=0.0)
        frames = S // hop
        cols += _pop_all(h, P, frames - len(cols))
        assert len(cols) == frames
        padded = np.concatenate([np.zeros((2, W), np.float32), x], axis=1)[:, hop:]
        got = np.stack(cols)
        # the per-block path runs the batch path's kernels frame by frame with the decay state carried: the columns are the batch
        # render's bytes exactly; the batch render of the same audio is held against the oracle by the parity chain
        import torch
        from parity_chain import check_render
        plan = api.Plan(cfg).upload()
        batch = plan.render(torch.from_numpy(np.ascontiguousarray(padded)).to(gpu)).cpu().numpy()[:frames]
        assert np.array_equal(got, batch), int((got != batch).sum())
        problems, stats = check_render(po, plan, cfg, np.ascontiguousarray(padded), gpu)
        assert not problems, (problems[:5], stats)
        # line results of the last frame
        line = np.zeros((P, 2), np.float32)
        api.check(api.lib().sgz_spectrum_line_results(h, 0, 0, line.ctypes.data_as(C.c_void_p)))
        assert np.isfinite(line).all()
        # wrong channel count is rejected like the reference's assertion (SpectrumDSP.cpp:65)
        assert api.lib().sgz_spectrum_push(h, ptrs, 3, block) == api.SGZ_EINVAL
    finally:
        api.lib().sgz_spectrum_destroy(h)


def test_queue_depth_drops_like_frame_queue(gpu):
    """frameQueue(10): if the consumer never pops, at most 10 columns are retained (SpectrumDSP.cpp:47,:185-186)."""
    cfg = config.spectrum_config(window_size=4096, hop=256, axis_points=64)
    c = api.config_from_dict(cfg)
    h = C.c_void_p()
    api.check(api.lib().sgz_spectrum_create(C.byref(c), C.byref(h)))
    try:
        x = synth.gen(1, 48000, 256 * 25, 2)
        ptrs = (C.c_void_p * 2)(x[0].ctypes.data, x[1].ctypes.data)
        api.check(api.lib().sgz_spectrum_push(h, ptrs, 2, x.shape[1]))
        cols = _pop_all(h, 64, 25, timeout=1.0)
        assert len(cols) == 10
    finally:
        api.lib().sgz_spectrum_destroy(h)
