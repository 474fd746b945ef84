"""The Vectorscope real-time handle (sgz_vector_*, csrc/vector_stream.hip) against the oracle: history ring, the audio thread's
one-pole filters (Processor::audioProcessing), runPeakFilter, and drawPolarPlot's vertex / colour stream over the two sections of
the ring for every channel pair.

Bars: ring contents, cursors, the envelope / balance recurrences (plain fp32 multiply-adds) and the fade ramp: bit-exact.  The
phase recurrence and the polar coordinates go through atan / sincos (libm in the oracle, ocml on the device; cpl's SIMD
polynomials in the reference): <= 1e-5 / <= 2e-6."""
import numpy as np
import pytest

from signalizer_amd import api, synth

pytestmark = pytest.mark.gpu

SR = 96000.0


def _push(dev, blk):
    while True:
        st = dev.push(blk)
        if st == api.SGZ_OK:
            return
        assert st == api.SGZ_BUSY


class RefVector:
    """the oracle side: a CLIFOStream-like ring per channel + sgzo_vector_audio_processing + sgzo_peak_filter"""

    def __init__(self, po, channels, size, env_mode, envelope_window, stereo_window, lanes):
        self.po, self.size, self.cursor, self.lanes, self.env_mode = po, size, 0, lanes, env_mode
        self.mem = np.zeros((channels, size), np.float32)
        self.f = po.VectorFilters()
        self.gain = 1.0
        self.ec = float(np.float32(np.exp(-1.0 / (envelope_window * SR))))
        self.sc = float(np.float32(np.exp(-1.0 / (stereo_window * SR))))

    def audio(self, blk):
        n = blk.shape[1]
        idx = (self.cursor + np.arange(n)) % self.size
        self.mem[:, idx] = blk                      # later samples overwrite earlier ones where a long block laps the ring
        self.cursor = int((self.cursor + n) % self.size)
        g = self.po.vector_audio_processing(self.f, blk[0], blk[1], self.ec, self.sc, 0.25, 1 if self.env_mode == 1 else 0, self.lanes)
        if self.env_mode == 1 and np.isfinite(g):
            self.gain = g


@pytest.mark.parametrize("channels,size,env_mode,fade", [(2, 9600, 1, 0), (8, 9600, 1, 1), (2, 1000, 0, 1), (4, 777, 2, 0), (2, 13, 1, 1)])
def test_vector_stream_against_the_oracle(gpu, oracle, channels, size, env_mode, fade):
    po = oracle
    colours = [(1.0, 0.5, 0.25), (0.2, 0.9, 0.4), (0.3, 0.3, 1.0), (0.9, 0.9, 0.1)]
    dev = api.Vector(sample_rate=SR, num_channels=channels, window_size=size, envelope_mode=env_mode, lanes=8, fade_history=fade,
                     max_block=4096, envelope_window=0.3, stereo_window=0.05, colours=colours)
    ref = RefVector(po, channels, size, env_mode, 0.3, 0.05, 8)
    x = synth.gen(4, SR, 60000, channels)
    x[:, 5000:5200] = 0                                      # both channels silent: the angle is defined as 0 / pi/4 there
    rng = np.random.default_rng(9)
    pos = 0
    while pos < x.shape[1]:
        n = int(rng.integers(1, 3000))
        blk = x[:, pos:pos + n]
        _push(dev, blk); ref.audio(blk)
        pos += blk.shape[1]
    for c in range(channels):
        mem, cur = dev.history(c)
        assert cur == ref.cursor
        assert np.array_equal(mem.view(np.uint32), ref.mem[c].view(np.uint32))
    f, gain = dev.filters()
    if env_mode == 1:
        assert np.array_equal(np.array(f.env[:], np.float32).view(np.uint32), np.array(ref.f.env[:], np.float32).view(np.uint32))
        # the oracle hands the gain back as a float (the old stage entry point's type); the handle keeps Processor::envelopeGain's double
        assert np.float32(gain) == np.float32(ref.gain)
    gb = np.array([list(r) for r in f.balance], np.float32)
    rb = np.array([list(r) for r in ref.f.balance], np.float32)
    assert np.array_equal(gb.view(np.uint32), rb.view(np.uint32))
    assert np.abs(np.array(f.phase[:]) - np.array(ref.f.phase[:])).max() <= 1e-5
    # polar vertices of every pair
    for pair in range(channels // 2):
        xyz, rgb = dev.vertices(pair)
        want, wrgb = po.vector_polar_view(ref.mem[2 * pair], ref.mem[2 * pair + 1], ref.cursor, 8, bool(fade), colours[pair])
        assert np.abs(xyz[:, :2] - want[:, :2]).max() <= 2e-6
        assert np.array_equal(xyz[:, 2].view(np.uint32), want[:, 2].view(np.uint32))          # the fade ramp: bit-exact
        assert np.array_equal(rgb.view(np.uint32), wrgb.view(np.uint32))
    # runPeakFilter (PeakDecay): memory-order peak of channels 0 / 1, SIMD tail dropped
    if env_mode == 2:
        env = np.zeros(2, np.float64)
        for frame in range(4):
            dt = 1 / 60
            coeff = float(np.power(np.float64(np.float32(np.exp(-1.0 / (0.3 * SR)))), size * dt))
            want_gain = po.peak_filter(ref.mem[:2], coeff, env, 8)
            got_gain = dev.peak_filter(dt)
            assert got_gain == want_gain
            more = synth.gen(40 + frame, SR, 500, channels) * np.float32(0.3)
            _push(dev, more); ref.audio(more)


@pytest.mark.parametrize("channels,size,env_mode,max_block", [(2, 9600, 1, 700), (8, 9600, 1, 300), (4, 777, 0, 64), (2, 13, 1, 9), (6, 2048, 1, 1500)])
def test_batched_launches_are_the_callback_walk(gpu, oracle, channels, size, env_mode, max_block):
    """vectorIngestKernel takes every callback that waited in ONE launch, keeping the callbacks' boundaries (Vectorscope.cpp:268-377 runs
    once per callback: the envelope gain and the balance / phase filters are read at callback ends).  SGZ_RT_OPT_DEFER_SUBMIT makes EVERY
    launch a multi-callback one -- blocks wait for a full batch or a reader -- and the ring, the cursor, the envelopes, the balance and the
    phase filters must still be the oracle's callback-by-callback walk (ring, envelopes and balance bit for bit)."""
    po = oracle
    dev = api.Vector(sample_rate=SR, num_channels=channels, window_size=size, envelope_mode=env_mode, lanes=8, fade_history=1,
                     max_block=4096, envelope_window=0.3, stereo_window=0.05).set_option(api.RT_OPT_DEFER_SUBMIT, 1)
    ref = RefVector(po, channels, size, env_mode, 0.3, 0.05, 8)
    x = synth.gen(7, SR, 40000, channels)
    x[:, 3000:3100] = 0
    rng = np.random.default_rng(31)
    pos, pushes, reads = 0, 0, 0
    while pos < x.shape[1]:
        n = int(rng.integers(1, max_block))
        blk = x[:, pos:pos + n]
        _push(dev, blk); ref.audio(blk)
        pos += blk.shape[1]
        pushes += 1
        if rng.random() < 0.04:                                       # a reader in mid-stream: flush on read, then the comparison
            f, gain = dev.filters()
            reads += 1
            gb = np.array([list(r) for r in f.balance], np.float32)
            rb = np.array([list(r) for r in ref.f.balance], np.float32)
            assert np.array_equal(gb.view(np.uint32), rb.view(np.uint32)), pushes
            if env_mode == 1:
                assert np.float32(gain) == np.float32(ref.gain), pushes
    assert pushes > 3 * max(reads, 1)                                 # several callbacks per launch on average
    for c in range(channels):
        mem, cur = dev.history(c)
        assert cur == ref.cursor
        assert np.array_equal(mem.view(np.uint32), ref.mem[c].view(np.uint32)), c
    f, gain = dev.filters()
    if env_mode == 1:
        assert np.array_equal(np.array(f.env[:], np.float32).view(np.uint32), np.array(ref.f.env[:], np.float32).view(np.uint32))
        assert np.float32(gain) == np.float32(ref.gain)
    gb = np.array([list(r) for r in f.balance], np.float32)
    rb = np.array([list(r) for r in ref.f.balance], np.float32)
    assert np.array_equal(gb.view(np.uint32), rb.view(np.uint32))
    assert np.abs(np.array(f.phase[:]) - np.array(ref.f.phase[:])).max() <= 1e-5


def test_parked_blocks_reach_the_gpu_with_the_next_read(gpu, oracle):
    """A push that finds the render thread submitting (or no staging slot free) parks its block in the handle's host FIFO.  Flush on read
    covers that FIFO: when the transport stops -- no further push -- the newest audio still shows in the next read.  SGZ_RT_OPT_PARK_PUSHES
    sends EVERY block that way; ring, cursor, envelopes and balance must be the oracle's walk over all of them after one read, without a
    flush and without another push, and again after more blocks with readers in between."""
    po = oracle
    channels, size = 4, 3000
    dev = api.Vector(sample_rate=SR, num_channels=channels, window_size=size, envelope_mode=1, lanes=8, fade_history=1,
                     max_block=1024, envelope_window=0.3, stereo_window=0.05).set_option(api.RT_OPT_PARK_PUSHES, 1)
    ref = RefVector(po, channels, size, 1, 0.3, 0.05, 8)
    x = synth.gen(19, SR, 30000, channels)
    rng = np.random.default_rng(5)
    pos = 0
    for rnd in range(6):
        for _ in range(int(rng.integers(1, 24))):                        # more blocks than one staging batch holds, sometimes
            n = int(rng.integers(1, 1024))
            blk = x[:, pos:pos + n]
            if blk.shape[1] == 0: break
            _push(dev, blk); ref.audio(blk)
            pos += blk.shape[1]
        f, gain = dev.filters()                                         # the read: nothing else hands the parked blocks on
        gb = np.array([list(r) for r in f.balance], np.float32)
        rb = np.array([list(r) for r in ref.f.balance], np.float32)
        assert np.array_equal(gb.view(np.uint32), rb.view(np.uint32)), rnd
        assert np.float32(gain) == np.float32(ref.gain), rnd
        for c in range(channels):
            mem, cur = dev.history(c)
            assert cur == ref.cursor
            assert np.array_equal(mem.view(np.uint32), ref.mem[c].view(np.uint32)), (rnd, c)


def test_cfg4_shape(gpu, oracle):
    """BASELINE configs[3]: 8 channels 96 kHz, 100 ms window = 9600 samples per pair, blocks of 480"""
    po = oracle
    dev = api.Vector(sample_rate=SR, num_channels=8, window_size=9600, envelope_mode=1, lanes=8, fade_history=1, max_block=512,
                     envelope_window=0.3, stereo_window=0.05)
    ref = RefVector(po, 8, 9600, 1, 0.3, 0.05, 8)
    x = synth.gen(4, SR, 48000, 8)
    for pos in range(0, x.shape[1], 480):
        _push(dev, x[:, pos:pos + 480]); ref.audio(x[:, pos:pos + 480])
    for pair in range(4):
        xyz, rgb = dev.vertices(pair)
        want, wrgb = po.vector_polar_view(ref.mem[2 * pair], ref.mem[2 * pair + 1], ref.cursor, 8, True)
        assert np.abs(xyz[:, :2] - want[:, :2]).max() <= 2e-6 and np.array_equal(xyz[:, 2], want[:, 2]) and np.array_equal(rgb, wrgb)


def test_vector_push_rejects_bad_input(gpu):
    with pytest.raises(api.SgzError):
        api.Vector(sample_rate=SR, num_channels=3, window_size=100, envelope_mode=0, lanes=8, fade_history=0, max_block=0,
                   envelope_window=0.3, stereo_window=0.05)
    dev = api.Vector(sample_rate=SR, num_channels=2, window_size=100, envelope_mode=0, lanes=8, fade_history=0, max_block=64,
                     envelope_window=0.3, stereo_window=0.05)
    with pytest.raises(api.SgzError):
        dev.push(np.zeros((2, 65), np.float32))
    with pytest.raises(api.SgzError):
        dev.push(np.zeros((4, 8), np.float32))


def test_vertices_all_equals_per_pair_calls(gpu):
    """sgz_vector_vertices_all (every pair's stream, one wait) writes the same floats as one sgz_vector_vertices call per pair"""
    from signalizer_amd import api
    sr, W, nch = 96000.0, 2400, 6
    h = api.Vector(sample_rate=sr, num_channels=nch, window_size=W, envelope_mode=2, lanes=8, fade_history=1, max_block=512,
                   envelope_window=0.3, stereo_window=0.1, colours=[(1.0, 0.5, 0.25), (0.2, 0.9, 0.4), (0.3, 0.3, 1.0)])
    x = synth.gen(77, int(sr), 512 * 9 + 100, nch)
    for pos in range(0, x.shape[1], 512):
        while h.push(x[:, pos:pos + 512]) == api.SGZ_BUSY:
            pass
    h.peak_filter(1 / 60)
    all_xyz, all_rgb = h.vertices_all()
    for p in range(nch // 2):
        xyz, rgb = h.vertices(p)
        assert np.array_equal(all_xyz[p].view(np.uint32), xyz.view(np.uint32)), p
        assert np.array_equal(all_rgb[p].view(np.uint32), rgb.view(np.uint32)), p
    # pinned caller buffers are written by the DMA engine directly (rt_common.hpp readBack): the same floats as through the bounce buffer
    import torch
    pin = lambda: torch.empty((nch // 2, W, 3), dtype=torch.float32).pin_memory().numpy()
    out = (pin(), pin())
    out[0][:] = np.nan; out[1][:] = np.nan
    got_xyz, got_rgb = h.vertices_all(out=out)
    assert got_xyz is out[0] and np.array_equal(got_xyz.view(np.uint32), all_xyz.view(np.uint32))
    assert np.array_equal(got_rgb.view(np.uint32), all_rgb.view(np.uint32))
    h.close()


@pytest.mark.parametrize("size,lanes", [(9600, 8), (4096, 8), (4096, 4), (1000, 16), (777, 8), (65536, 8), (12345, 2), (9600, 32), (3001, 64),
                                        (100000, 8), (33, 8), (17, 16)])
def test_fade_ramp_is_the_reference_sum(gpu, oracle, size, lanes):
    """the fade ramp (z of every vertex) at many (size, cursor, lanes): the handle evaluates it from the arithmetic progressions of the
    fp32 running sum (fade_chain.hpp; lanes > 16 walk the sum as the reference does) -- bit for bit the oracle's SIMD-lane replay,
    at cursors that put the section boundary everywhere (sizes that are powers of two make every step exact, the others cross binades
    with ties on the way)"""
    po = oracle
    dev = api.Vector(sample_rate=SR, num_channels=2, window_size=size, envelope_mode=0, lanes=lanes, fade_history=1, max_block=4096,
                     envelope_window=0.3, stereo_window=0.05, colours=[(1.0, 0.5, 0.25)])
    mem = np.zeros((2, size), np.float32)
    cursor = 0
    rng = np.random.default_rng(size + lanes)
    x = synth.gen(6, SR, 4096, 2)
    for step in range(14):
        n = [1, lanes - 1, lanes, lanes + 1, 37, 511, 1000, 2047, 4096, 3, 64, 999, 4095, 2][step]
        blk = np.ascontiguousarray(x[:, :n])
        _push(dev, blk)
        idx = (cursor + np.arange(n)) % size
        mem[:, idx] = blk
        cursor = int((cursor + n) % size)
        xyz, rgb = dev.vertices(0)
        want, wrgb = po.vector_polar_view(mem[0], mem[1], cursor, lanes, True, (1.0, 0.5, 0.25))
        assert np.array_equal(xyz[:, 2].view(np.uint32), want[:, 2].view(np.uint32)), (step, cursor, int((xyz[:, 2] != want[:, 2]).sum()))
        assert np.array_equal(rgb.view(np.uint32), wrgb.view(np.uint32))
    dev.close()


def test_audio_and_render_threads_run_concurrently(gpu, oracle):
    """a producer thread pushes 1500 callbacks flat out while the render thread draws (peak filter, every pair's vertices): no call
    fails and no callback is lost or reordered -- history rings, balance and phase filters end where the oracle's walk ends"""
    import threading
    import torch
    po = oracle
    channels, size = 4, 4800
    dev = api.Vector(sample_rate=SR, num_channels=channels, window_size=size, envelope_mode=2, lanes=8, fade_history=1, max_block=512,
                     envelope_window=0.3, stereo_window=0.05, colours=[(1.0, 0.5, 0.25), (0.2, 0.9, 0.4)])
    ref = RefVector(po, channels, size, 2, 0.3, 0.05, 8)
    x = synth.gen(12, SR, 1500 * 200, channels)
    errors, frames = [], [0]
    done = threading.Event()

    def producer():
        try:
            for pos in range(0, x.shape[1], 200):
                blk = np.ascontiguousarray(x[:, pos:pos + 200])
                while True:
                    st = dev.push(blk)
                    if st == api.SGZ_OK:
                        break
                    if st != api.SGZ_BUSY:
                        errors.append(("push", st)); return
        finally:
            done.set()

    def render():
        outs = (torch.zeros((channels // 2, size, 3), dtype=torch.float32).pin_memory().numpy(),
                torch.zeros((channels // 2, size, 3), dtype=torch.float32).pin_memory().numpy())
        try:
            while not done.is_set() or frames[0] < 8:            # (at least eight frames however fast the producer is: the count is not a speed test)
                dev.peak_filter(1 / 60)
                dev.vertices_all(out=outs)
                frames[0] += 1
        except Exception as e:                                     # noqa: BLE001
            errors.append(("render", repr(e)))

    tp, tr = threading.Thread(target=producer), threading.Thread(target=render)
    tr.start(); tp.start(); tp.join(timeout=180); tr.join(timeout=180)
    assert not errors and not tp.is_alive() and not tr.is_alive(), errors[:3]
    assert frames[0] >= 8
    for pos in range(0, x.shape[1], 200):
        ref.audio(np.ascontiguousarray(x[:, pos:pos + 200]))
    for c in range(channels):
        mem, cur = dev.history(c)
        assert cur == ref.cursor and np.array_equal(mem.view(np.uint32), ref.mem[c].view(np.uint32))
    f, _ = dev.filters()
    gb = np.array([list(r) for r in f.balance], np.float32)
    rb = np.array([list(r) for r in ref.f.balance], np.float32)
    assert np.array_equal(gb.view(np.uint32), rb.view(np.uint32))
    assert np.abs(np.array(f.phase[:]) - np.array(ref.f.phase[:])).max() <= 1e-5
    dev.close()


def test_vertices_all_into_device_memory(gpu):
    """sgz_vector_vertices_all / sgz_scope_vertices_all with DEVICE buffers: the vertex kernels write HBM themselves, one wait, the same
    bytes the host-buffer form returns (the device-resident hand-off bench.py times as ms_per_step_device_resident)"""
    import ctypes as C
    import torch
    L = api.lib()
    dev = api.Vector(sample_rate=SR, num_channels=4, window_size=3000, envelope_mode=1, lanes=8, fade_history=1, max_block=4096,
                     envelope_window=0.3, stereo_window=0.05, colours=[(1.0, 0.5, 0.25), (0.2, 0.9, 0.4)])
    x = synth.gen(12, SR, 9000, 4)
    for pos in range(0, x.shape[1], 700):
        _push(dev, x[:, pos:pos + 700])
    want_xyz, want_rgb = dev.vertices_all()
    d_xyz = torch.full((2, 3000, 3), float("nan"), dtype=torch.float32, device=gpu)
    d_rgb = torch.zeros((2, 3000, 3), dtype=torch.float32, device=gpu)
    cnt = C.c_uint32(3000)
    api.check(L.sgz_vector_vertices_all(dev.h, C.c_void_p(d_xyz.data_ptr()), C.c_void_p(d_rgb.data_ptr()), C.byref(cnt)))
    assert cnt.value == 3000
    assert np.array_equal(d_xyz.cpu().numpy().view(np.uint32), want_xyz.view(np.uint32))
    assert np.array_equal(d_rgb.cpu().numpy().view(np.uint32), want_rgb.view(np.uint32))
    # Oscilloscope: two strips
    sc = api.Scope(sample_rate=48000.0, window_size=2000.0, num_channels=2, trigger_mode=4, channel_mode=0, envelope_mode=0, interpolation=3,
                   max_block=4096, trigger_threshold=0.05, trigger_channel=1.0, envelope_window=0.3)
    y = synth.gen(13, 48000, 12000, 2)
    for pos in range(0, y.shape[1], 900):
        blk = np.ascontiguousarray(y[:, pos:pos + 900])
        while sc.push(blk) == api.SGZ_BUSY:
            pass
    v = api.ScopeView(2000.0, 0.0, 1.0, 1.0, 4001, 0)
    n = L.sgz_scope_vertex_count(sc.h, C.byref(v))
    host = [(np.zeros((n, 3), np.float32), np.zeros((n, 4), np.uint8)) for _ in (0, 1)]
    want = sc.vertices_all(v, (0, 1), (0, 0), host)
    dx = [torch.full((n, 3), float("nan"), dtype=torch.float32, device=gpu) for _ in (0, 1)]
    dc = [torch.zeros((n, 4), dtype=torch.uint8, device=gpu) for _ in (0, 1)]
    ev = (C.c_uint32 * 2)(0, 1); ch = (C.c_uint32 * 2)(0, 0)
    xs = (C.c_void_p * 2)(dx[0].data_ptr(), dx[1].data_ptr()); cs = (C.c_void_p * 2)(dc[0].data_ptr(), dc[1].data_ptr())
    cnts = (C.c_uint32 * 2)(n, n)
    api.check(L.sgz_scope_vertices_all(sc.h, C.byref(v), 2, ev, ch, xs, cs, cnts))
    for k in (0, 1):
        m = cnts[k]
        assert m == want[k][0].shape[0]
        assert np.array_equal(dx[k].cpu().numpy()[:m].view(np.uint32), want[k][0].view(np.uint32))
        assert np.array_equal(dc[k].cpu().numpy()[:m], want[k][1])
