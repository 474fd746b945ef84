"""The real-time Spectrum handle in the reference's two block-driven forms:

  * DisplayMode::LineGraph -- sgz_spectrum_render_lines = Spectrum::vectorGLRendering's per-video-frame transform of the current ring
    (SpectrumRendering.cpp:617-635), pushes of uneven blocks in between;
  * SGZ_RT_OPT_STRICT_REFERENCE_QUIRKS -- audioEntryPoint as written (TransformDSP.inl:1165-1211): quirks Q1 / Q2.

How they are tied to the oracle.  tests/stream_windows.py says WHICH W samples each frame sees; tests/test_oracle_stream.py proves (CPU)
that oracle/spectrum_stream.c -- the line-by-line restatement -- produces exactly the ideal-framing oracle render of those frames laid
end to end at hop == W.  Here the handle's output must equal, byte for byte, the library's own offline render of the same laid-out
frames (one K_A per frame either way), and that offline render is held against the oracle by the parity chain (tests/parity_chain.py:
bins within the FFT tolerance, mapping exact given bins, dB / colour / line results exact given the mapped pixels).  RSNT has no
transform in between: its line results are compared with the oracle stream directly."""
import ctypes as C
import time

import numpy as np
import pytest

from signalizer_amd import api, config, synth
from stream_windows import cut, newest_windows, strict_frames

pytestmark = pytest.mark.gpu

OPT_STRICT, OPT_HISTORY = 1, 2


def _create(cfg):
    c = api.config_from_dict(cfg)
    h = C.c_void_p()
    api.check(api.lib().sgz_spectrum_create(C.byref(c), C.byref(h)))
    return h


def _push(h, blk, flush=True):
    """push; then (the tests are single-threaded and look at the result of THIS block) make sure the block is not left waiting in the
    host FIFO behind a GPU that is still busy with the earlier ones -- push itself never waits (rt_common.hpp Backlog)"""
    ptrs = (C.c_void_p * blk.shape[0])(*[blk[c].ctypes.data for c in range(blk.shape[0])])
    api.check(api.lib().sgz_spectrum_push(h, ptrs, blk.shape[0], blk.shape[1]))
    if flush:
        api.check(api.lib().sgz_spectrum_flush(h))


def _pop(h, P, want, timeout=10.0):
    cols, col = [], np.zeros((P, 4), np.uint8)
    n = C.c_uint32(0)
    t0 = time.time()
    while len(cols) < want and time.time() - t0 < timeout:
        st = api.lib().sgz_spectrum_pop_column(h, col.ctypes.data_as(C.c_void_p), C.byref(n))
        if st == api.SGZ_OK:
            cols.append(col.copy())
        else:
            assert st == api.SGZ_EMPTY
            time.sleep(0.0005)
    return cols


def _offline(cfg, frames, gpu, want_lines=False):
    """the library's offline render of the frames laid end to end at hop == W: (rgba [F][P][4], lines [F][C][G][P][2] | None, plan, x, cfg')"""
    import torch
    c2 = dict(cfg, hop=cfg["window_size"])
    x = np.ascontiguousarray(np.concatenate(frames, axis=1))
    plan = api.Plan(c2).upload()
    F = len(frames)
    lines = torch.empty((F, plan.C, 2, plan.P, 2), dtype=torch.float32, device=gpu) if want_lines else None
    rgba = plan.render(torch.from_numpy(x).to(gpu), lines=lines).cpu().numpy()
    return rgba, (lines.cpu().numpy() if want_lines else None), plan, x, c2


@pytest.mark.parametrize("hop", [200, 333])
@pytest.mark.parametrize("block", [64, 512, 2048])
def test_strict_quirks_q1(gpu, oracle, hop, block):
    """a 512-sample block at hop 200 produces the reference's frames -- including the same frame twice (Q1)"""
    from parity_chain import check_render
    cfg = config.spectrum_config(window_size=2048, hop=hop, axis_points=256)
    W, P = 2048, 256
    x = synth.gen(41, 48000, 9000, 2)
    blocks = cut(x, [block])
    frames, per_block = strict_frames(blocks, W, hop)
    want, _, plan, xcat, c2 = _offline(cfg, frames, gpu)
    h = _create(cfg)
    try:
        api.check(api.lib().sgz_spectrum_set_option(h, OPT_STRICT, 1))
        got, expect = [], []
        at = 0
        for blk, n in zip(blocks, per_block):
            _push(h, blk)
            kept = min(n, 10)                                  # frameQueue(10): a callback that makes 11 frames loses the last (SpectrumDSP.cpp:185-186)
            cols = _pop(h, P, kept)
            assert len(cols) == kept
            got += cols
            expect += list(want[at:at + kept])
            at += n
        assert at == len(frames) > 0
        assert np.array_equal(np.stack(got), np.stack(expect))
        if max(per_block) >= 3:
            k = next(i for i, n in enumerate(per_block) if n >= 3)
            a = sum(per_block[:k])
            assert np.array_equal(frames[a + 1], frames[a + 2])           # the quirk itself was exercised
        # ... and the offline render of those frames against the oracle, through the chain
        problems, stats = check_render(oracle, plan, c2, xcat, gpu)
        assert not problems, (problems[:4], stats)
        # tripwire on raw bytes against the oracle STREAM (the restated audioEntryPoint itself)
        st = oracle.SpectrumStream(oracle.params_from_dict(cfg))
        ref = np.concatenate([st.audio(b)["rgba"] for b in blocks])
        d = np.abs(want.astype(int) - ref.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 2e-2, (int(d.max()), float((d > 0).mean()))
    finally:
        api.lib().sgz_spectrum_destroy(h)


@pytest.mark.parametrize("extra", [96, 700])
def test_strict_quirks_q2(gpu, oracle, extra):
    """audio history longer than the window: the frame is `extra` samples short (TransformDSP.inl:245-257)"""
    cfg = config.spectrum_config(window_size=2048, hop=200, axis_points=200, channel_mode=config.CH_MIDSIDE)
    W, P = 2048, 200
    x = synth.gen(42, 48000, 7000, 2)
    blocks = cut(x, [512, 64, 300])
    frames, per_block = strict_frames(blocks, W, 200, history=W + extra)
    assert any(not f[:, W - extra:].any() for f in frames)
    want, _, _, _, _ = _offline(cfg, frames, gpu)
    h = _create(cfg)
    try:
        api.check(api.lib().sgz_spectrum_set_option(h, OPT_STRICT, 1))
        assert api.lib().sgz_spectrum_set_option(h, OPT_HISTORY, W - 1) == api.SGZ_EINVAL
        api.check(api.lib().sgz_spectrum_set_option(h, OPT_HISTORY, W + extra))
        got = []
        for blk, n in zip(blocks, per_block):
            _push(h, blk)
            got += _pop(h, P, n)
        assert np.array_equal(np.stack(got), want)
    finally:
        api.lib().sgz_spectrum_destroy(h)


def test_default_framing_is_unchanged_by_the_option_when_blocks_divide_the_hop(gpu):
    cfg = config.spectrum_config(window_size=2048, hop=512, axis_points=128)
    x = synth.gen(43, 48000, 8192, 2)
    out = []
    for strict in (0, 1):
        h = _create(cfg)
        try:
            api.check(api.lib().sgz_spectrum_set_option(h, OPT_STRICT, strict))
            cols, pushed = [], 0
            for blk in cut(x, [256]):
                _push(h, blk)
                pushed += blk.shape[1]
                cols += _pop(h, 128, pushed // 512 - len(cols))       # (keeps the 10-deep frame queue from overflowing)
            out.append(np.stack(cols))
        finally:
            api.lib().sgz_spectrum_destroy(h)
    assert out[0].shape[0] == 16 and np.array_equal(out[0], out[1])


@pytest.mark.parametrize("mode,W,pairs", [(config.CH_SEPARATE, 4096, 2), (config.CH_MERGE, 2048, 1), (config.CH_PHASE, 4096, 1),
                                            (config.CH_SEPARATE, 32768, 1)])
def test_line_graph_render(gpu, oracle, mode, W, pairs):
    """pushes of uneven blocks interleaved with render calls; every call = the newest window of every pair, filters advanced once"""
    from parity_chain import check_render
    P = 300
    cfg = config.spectrum_config(window_size=W, hop=1024, axis_points=P, channel_mode=mode, num_pairs=pairs,
                                 display_mode=config.DISPLAY_LINE_GRAPH)
    x = synth.gen(44, 48000, 3 * W + 5000, 2 * pairs)
    blocks = cut(x, [480, 37, 4096, 1000, 20000])
    render_after = [0, 0, 2, 3, 5, 5, 5, len(blocks) - 1]
    wins = newest_windows(blocks, W, render_after)
    _, want, plan, xcat, c2 = _offline(cfg, wins, gpu, want_lines=True)
    problems, stats = check_render(oracle, plan, c2, xcat, gpu, want_lines=True)      # the offline render of those windows against the oracle
    assert not problems, (problems[:4], stats)
    if W == 32768:
        # the first window holds 480 samples under the Hann window's last 1.5 %: the case parity_chain.WIN_ABS exists for.  With the
        # window FETCHED from the plan's table instead of evaluated in the kernel the plain bar holds -- the term is the window's, not the FFT's
        tabled = api.Plan(c2)
        tabled.set_option(api.OPT_FETCH_WINDOW, 1)
        problems, stats = check_render(oracle, tabled.upload(), c2, xcat, gpu, win_abs=0.0)
        assert not problems, (problems[:4], stats)
    h = _create(cfg)
    L = api.lib()
    try:
        out = np.zeros((pairs, 2, P, 2), np.float32)
        got = []
        col = np.zeros((P, 4), np.uint8)
        for k, blk in enumerate(blocks):
            _push(h, blk)
            for _ in [q for q in render_after if q == k]:
                api.check(L.sgz_spectrum_render_lines(h, None, out.ctypes.data_as(C.c_void_p)))
                got.append(out.copy())
                one = np.zeros((P, 2), np.float32)
                api.check(L.sgz_spectrum_line_results(h, pairs - 1, 1, one.ctypes.data_as(C.c_void_p)))
                assert np.array_equal(one, out[pairs - 1, 1])                     # lineGraphs[k].getResults of the last render
        assert L.sgz_spectrum_pop_column(h, col.ctypes.data_as(C.c_void_p), None) == api.SGZ_EINVAL   # no frames on the audio thread (:1167): a LINE_GRAPH handle has no columns and says so
        got = np.stack(got)                                                       # [calls][C][G][P][2]
        assert got.shape == want.shape
        bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        assert not len(bad), (len(bad), sorted(set(bad[:, 0])), sorted(set(bad[:, 3]))[:4], sorted(set(bad[:, 3]))[-4:])
    finally:
        L.sgz_spectrum_destroy(h)


def test_line_graph_poles_per_call(gpu):
    """the decay of a video frame comes with the call (openGLDeltaTime, Spectrum.cpp:388-394): poles given per call == poles configured"""
    P, W = 128, 2048
    x = synth.gen(45, 48000, 9000, 2)
    outs = []
    for configured, per_call in (((0.5, 0.97), None), ((0.9, 0.99), (0.5, 0.97))):
        cfg = config.spectrum_config(window_size=W, hop=256, axis_points=P, pole=configured, display_mode=config.DISPLAY_LINE_GRAPH)
        h = _create(cfg)
        try:
            res, out = [], np.zeros((1, 2, P, 2), np.float32)
            poles = (C.c_float * 2)(*per_call) if per_call else None
            for blk in cut(x, [700]):
                _push(h, blk)
                api.check(api.lib().sgz_spectrum_render_lines(h, poles, out.ctypes.data_as(C.c_void_p)))
                res.append(out.copy())
            outs.append(np.stack(res))
        finally:
            api.lib().sgz_spectrum_destroy(h)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    assert outs[0][-1, 0, 0].max() > outs[0][-1, 0, 0].min()


def test_render_lines_refused_in_colour_mode(gpu):
    h = _create(config.spectrum_config(window_size=2048, hop=512, axis_points=64))
    try:
        out = np.zeros((1, 2, 64, 2), np.float32)
        assert api.lib().sgz_spectrum_render_lines(h, None, out.ctypes.data_as(C.c_void_p)) == api.SGZ_EINVAL
    finally:
        api.lib().sgz_spectrum_destroy(h)


@pytest.mark.parametrize("mode,win", [(config.CH_SEPARATE, config.WIN_HANN), (config.CH_MERGE, config.WIN_RECT), (config.CH_PHASE, config.WIN_HANN)])
def test_line_graph_rsnt(gpu, oracle, mode, win):
    """RSNT in the line-graph mode: the audio thread resonates whole blocks (TransformDSP.inl:1206-1209), the render thread windows the
    state as it is (:1103-1133) and advances the filters.  A block advance is the reference's recurrence sample by sample: bit-exact."""
    po = oracle
    P = 200
    cfg = config.spectrum_config(window_size=1024, hop=256, axis_points=P, channel_mode=mode, window_type=win, algorithm=config.ALGO_RSNT,
                                 display_mode=config.DISPLAY_LINE_GRAPH)
    x = synth.gen(46, 48000, 6000, 2)
    blocks = cut(x, [480, 37, 512, 1000])
    st = po.SpectrumStream(po.params_from_dict(cfg), po.DISPLAY_LINE_GRAPH)
    h = _create(cfg)
    try:
        out = np.zeros((1, 2, P, 2), np.float32)
        for k, blk in enumerate(blocks):
            _push(h, blk)
            assert st.audio(blk)["frames"] == 0
            if k % 2 == 1:
                api.check(api.lib().sgz_spectrum_render_lines(h, None, out.ctypes.data_as(C.c_void_p)))
                ref = st.render_lines()["results"]                                 # [C][G][P] complex = (left | magnitude, right | phase)
                want = np.stack([ref.real, ref.imag], axis=-1).astype(np.float32)
                assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (k, float(np.abs(out - want).max()))
    finally:
        api.lib().sgz_spectrum_destroy(h)


def test_colour_mode_line_results_are_the_newest_frames(gpu):
    """sgz_spectrum_line_results reads a host copy the producer's stream fills (no wait on that stream): once the stream has drained it is
    the last frame's lineGraphs[k].results, bit for bit the batch render's"""
    import torch
    W, hop, P = 4096, 1024, 160
    cfg = config.spectrum_config(window_size=W, hop=hop, axis_points=P)
    x = synth.gen(47, 48000, 10 * hop, 2)
    h = _create(cfg)
    L = api.lib()
    L.sgz_spectrum_stream.restype = C.c_void_p
    L.sgz_spectrum_stream.argtypes = [C.c_void_p]
    try:
        one = np.ones((P, 2), np.float32)
        api.check(L.sgz_spectrum_line_results(h, 0, 0, one.ctypes.data_as(C.c_void_p)))
        assert not one.any()                                                      # no frame yet: the graphs start zeroed
        for blk in cut(x, [480]):
            _push(h, blk)
            _pop(h, P, 1, timeout=0.0)
            api.check(L.sgz_spectrum_line_results(h, 0, 1, one.ctypes.data_as(C.c_void_p)))     # never blocks, always whole frames
            assert np.isfinite(one).all()
        _pop(h, P, 10, timeout=1.0)
        torch.cuda.synchronize()
        frames = x.shape[1] // hop
        padded = np.concatenate([np.zeros((2, W), np.float32), x], axis=1)[:, hop:]
        plan = api.Plan(cfg).upload()
        lines = torch.empty((frames, 1, 2, P, 2), dtype=torch.float32, device=gpu)
        plan.render(torch.from_numpy(np.ascontiguousarray(padded)).to(gpu), lines=lines)
        want = lines.cpu().numpy()[frames - 1, 0]
        for g in range(2):
            api.check(L.sgz_spectrum_line_results(h, 0, g, one.ctypes.data_as(C.c_void_p)))
            assert np.array_equal(one.view(np.uint32), want[g].view(np.uint32))
    finally:
        L.sgz_spectrum_destroy(h)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_block_structures_through_both_modes(gpu, seed):
    """seeded sweep: random window / hop / block-size mixes / audio histories / channel modes through the strict framing AND the line
    graph -- the handle's bytes against the offline render of the frames tests/stream_windows.py says each path sees (that render is
    held to the oracle by the named tests above and by tests/test_gpu_fuzz.py)"""
    rng = np.random.default_rng(1000 + seed)
    L = api.lib()
    for case in range(6):
        W = int(rng.choice([1024, 2048, 4096]))
        hop = int(rng.integers(50, W))
        P = int(rng.integers(40, 300))
        mode = int(rng.choice([config.CH_LEFT, config.CH_MERGE, config.CH_SEPARATE, config.CH_MIDSIDE, config.CH_COMPLEX]))
        pairs = int(rng.integers(1, 3))
        sizes = [int(v) for v in rng.integers(1, 3000, size=int(rng.integers(1, 5)))]
        x = synth.gen(int(rng.integers(1, 1 << 20)), 48000, int(rng.integers(3 * W, 6 * W)), 2 * pairs)
        blocks = cut(x, sizes)
        extra = int(rng.choice([0, 0, 1, 37, W // 2]))
        # --- strict framing
        cfg = config.spectrum_config(window_size=W, hop=hop, axis_points=P, channel_mode=mode, num_pairs=pairs)
        frames, per_block = strict_frames(blocks, W, hop, history=W + extra)
        if frames:
            want, _, _, _, _ = _offline(cfg, frames, gpu)
            h = _create(cfg)
            try:
                api.check(L.sgz_spectrum_set_option(h, OPT_STRICT, 1))
                api.check(L.sgz_spectrum_set_option(h, OPT_HISTORY, W + extra))
                got, expect, at = [], [], 0
                for blk, n in zip(blocks, per_block):
                    _push(h, blk)
                    kept = min(n, 10)
                    cols = _pop(h, P, kept)
                    assert len(cols) == kept, (seed, case)
                    got += cols
                    expect += list(want[at:at + kept])
                    at += n
                if got:
                    assert np.array_equal(np.stack(got), np.stack(expect)), (seed, case, W, hop, sizes, extra, mode)
            finally:
                L.sgz_spectrum_destroy(h)
        # --- line graph
        cfg = dict(cfg, display_mode=config.DISPLAY_LINE_GRAPH)
        render_after = sorted(int(v) for v in rng.integers(0, len(blocks), size=5))
        wins = newest_windows(blocks, W, render_after)
        _, want, _, _, _ = _offline(cfg, wins, gpu, want_lines=True)
        h = _create(cfg)
        try:
            out, got = np.zeros((pairs, 2, P, 2), np.float32), []
            for k, blk in enumerate(blocks):
                _push(h, blk)
                for _ in [q for q in render_after if q == k]:
                    api.check(L.sgz_spectrum_render_lines(h, None, out.ctypes.data_as(C.c_void_p)))
                    got.append(out.copy())
            got = np.stack(got)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, case, W, sizes, mode)
        finally:
            L.sgz_spectrum_destroy(h)


def test_strict_quirks_with_callbacks_longer_than_a_staged_piece(gpu):
    """a 40 000-sample callback is staged in three pieces and makes 57 frames, every one of them from the history before the callback and
    the first <= hop samples of the block (all after the first piece): the frame queue keeps ten of them, the states advance over all"""
    W, hop, P = 4096, 700, 120
    cfg = config.spectrum_config(window_size=W, hop=hop, axis_points=P)
    x = synth.gen(48, 48000, 93000, 2)
    blocks = cut(x, [40000, 20000, 33000])
    frames, per_block = strict_frames(blocks, W, hop)
    want, _, _, _, _ = _offline(cfg, frames, gpu)
    h = _create(cfg)
    try:
        api.check(api.lib().sgz_spectrum_set_option(h, OPT_STRICT, 1))
        at = 0
        for blk, n in zip(blocks, per_block):
            assert n > 10
            _push(h, blk)
            cols = _pop(h, P, 10)
            assert len(cols) == 10 and np.array_equal(np.stack(cols), want[at:at + 10]), at
            at += n
        dropped, refused = C.c_uint64(0), C.c_uint64(0)
        api.check(api.lib().sgz_spectrum_stats(h, C.byref(dropped), C.byref(refused)))
        assert dropped.value == sum(per_block) - 30 and refused.value == 0
    finally:
        api.lib().sgz_spectrum_destroy(h)


def test_render_thread_and_audio_thread_run_concurrently(gpu):
    """the render thread's calls (render_lines, track_peak, line_results, backlog / stats polling) take no lock against push: a producer
    thread pushes 3000 blocks flat out while the consumer renders as fast as it can.  No call may fail or refuse, and once the producer
    has stopped the next render is exactly the offline result for the ring's final window (nothing was torn on the way)."""
    import threading
    W, P = 4096, 200
    cfg = config.spectrum_config(window_size=W, hop=1024, axis_points=P, display_mode=config.DISPLAY_LINE_GRAPH, pole=(0.0, 0.0))
    x = synth.gen(49, 48000, 3000 * 160, 2)
    blocks = cut(x, [160])
    h = _create(cfg)
    L = api.lib()
    L.sgz_spectrum_backlog.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    errors, renders = [], [0]
    stop = threading.Event()

    def producer():
        try:
            for blk in blocks:
                ptrs = (C.c_void_p * 2)(blk[0].ctypes.data, blk[1].ctypes.data)
                st = L.sgz_spectrum_push(h, ptrs, 2, blk.shape[1])
                if st != api.SGZ_OK:
                    errors.append(("push", st))
                    return
        finally:
            stop.set()

    def consumer():
        out = np.zeros((1, 2, P, 2), np.float32)
        pk = api.Peak()
        d, w = C.c_uint64(), C.c_uint32()
        while not stop.is_set():
            for st in (L.sgz_spectrum_render_lines(h, None, out.ctypes.data_as(C.c_void_p)), L.sgz_spectrum_track_peak(h, 0, 0.5, C.byref(pk)),
                       L.sgz_spectrum_backlog(h, C.byref(d), C.byref(w))):
                if st != api.SGZ_OK:
                    errors.append(("consumer", st, L.sgz_last_error()))
                    return
            renders[0] += 1

    tp, tc = threading.Thread(target=producer), threading.Thread(target=consumer)
    tc.start(); tp.start(); tp.join(timeout=120); tc.join(timeout=120)
    try:
        assert not errors and not tp.is_alive() and not tc.is_alive(), errors[:3]
        assert renders[0] > 10
        api.check(L.sgz_spectrum_flush(h))
        out = np.zeros((1, 2, P, 2), np.float32)
        api.check(L.sgz_spectrum_render_lines(h, None, out.ctypes.data_as(C.c_void_p)))
        # poles 0: the filters hold no memory, so the result is the newest window's alone
        _, want, _, _, _ = _offline(cfg, [np.ascontiguousarray(x[:, -W:])], gpu, want_lines=True)
        assert np.array_equal(out.view(np.uint32), want[0].view(np.uint32))
    finally:
        L.sgz_spectrum_destroy(h)


def test_zero_initialised_config_is_a_line_graph_handle(gpu):
    """A C host that zero-initialises sgz_spectrum_config and fills in the sizes gets display_mode 0 = SGZ_DISPLAY_LINE_GRAPH (the reference's
    enum order and its default): pushes produce no columns, and sgz_spectrum_pop_column / sgz_spectrum_flush_columns say SGZ_EINVAL instead
    of staying empty for ever; sgz_spectrum_render_lines works."""
    L = api.lib()
    c = api.config_from_dict(config.spectrum_config(window_size=1024, hop=256, axis_points=64))
    c.display_mode = 0
    h = C.c_void_p()
    api.check(L.sgz_spectrum_create(C.byref(c), C.byref(h)))
    try:
        x = synth.gen(3, 48000, 4096, 2)
        api.check(L.sgz_spectrum_push(h, (C.c_void_p * 2)(x[0].ctypes.data, x[1].ctypes.data), 2, 4096))
        col = np.zeros((64, 4), np.uint8)
        assert L.sgz_spectrum_pop_column(h, col.ctypes.data_as(C.c_void_p), None) == api.SGZ_EINVAL
        assert L.sgz_spectrum_flush_columns(h, None, None) == api.SGZ_EINVAL
        out = np.zeros((1, 2, 64, 2), np.float32)
        api.check(L.sgz_spectrum_render_lines(h, None, out.ctypes.data_as(C.c_void_p)))
        assert np.isfinite(out).all() and out[0, 0, :, 0].max() > 0
    finally:
        L.sgz_spectrum_destroy(h)
