"""The Oscilloscope real-time handle (sgz_scope_*, csrc/scope_stream.hip) against the oracle's restatement of the reference's
audio-thread state machine (oracle/scope_stream.c): StreamState::audioEntryPoint -> ZeroCrossingProcessor -> TriggeringProcessor::
processMutating -> swapBuffers, the RMS envelope, runPeakFilter in every channel mode, drawWavePlot's vertices.

Bars: integer / index work (trigger counters, window selection, ring contents and cursors) and every fp32 recurrence: bit-exact.
Lanczos vertices: <= 2e-6 (fp64 kernel weights evaluated in a different but cancellation-free order); Linear vertices: exact."""
import numpy as np
import pytest

from signalizer_amd import api, synth

pytestmark = pytest.mark.gpu

SR = 192000.0


def _signal(seed, n, channels, f0=441.7):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / SR
    x = np.zeros((channels, n), np.float32)
    for c in range(channels):
        x[c] = (0.6 * np.sin(2 * np.pi * f0 * (1 + 0.31 * c) * t + 0.4 * c) + 0.25 * np.sin(2 * np.pi * 5.3 * f0 * t)
                + 0.05 * rng.standard_normal(n)).astype(np.float32)
    return x


def _push(dev, blk):
    """push never waits: SGZ_BUSY means the block was not taken (the GPU is 8 blocks behind this loop) -- offer it again"""
    while True:
        st = dev.push(blk)
        if st == api.SGZ_OK:
            return
        assert st == api.SGZ_BUSY


def _feed(po, cfg, x, seed, max_block=3000, defer=False, park=False):
    """the same random block schedule into the HIP handle and the oracle stream"""
    dev = api.Scope(**cfg)
    if defer:
        dev.set_option(api.RT_OPT_DEFER_SUBMIT, 1)
    if park:
        dev.set_option(api.RT_OPT_PARK_PUSHES, 1)
    ref = po.ScopeStream(cfg["num_channels"], cfg["sample_rate"], cfg["window_size"], cfg["trigger_mode"], cfg["trigger_threshold"],
                         cfg["channel_mode"], cfg["trigger_channel"], cfg["envelope_mode"], cfg["envelope_window"])
    rng = np.random.default_rng(seed)
    pos = 0
    checks = 0
    while pos < x.shape[1]:
        n = int(rng.integers(1, max_block))
        blk = x[:, pos:pos + n]
        _push(dev, blk)
        ref.audio(blk)
        pos += blk.shape[1]
        if rng.random() < 0.15:                                       # spot checks in mid-stream
            assert dev.state() == ref.state()
            checks += 1
    assert checks > 0
    return dev, ref


def _cfg(**over):
    cfg = dict(sample_rate=SR, window_size=19200.0, num_channels=2, trigger_mode=4, channel_mode=0, envelope_mode=0, interpolation=3,
               max_block=4096, trigger_threshold=0.05, trigger_channel=1.0, envelope_window=0.3)
    cfg.update(over)
    return cfg


@pytest.mark.parametrize("over", [
    dict(),                                                            # BASELINE cfg3: stereo 192 kHz, 100 ms window, trigger on L
    dict(window_size=480.3, trigger_threshold=0.0),                    # fractional window, many triggers per block
    dict(window_size=1000.0, channel_mode=2, trigger_threshold=0.2),   # Mid trigger
    dict(window_size=777.0, channel_mode=3, num_channels=4, trigger_channel=2.0),     # Side trigger of the second pair
    dict(window_size=2048.0, channel_mode=4, num_channels=6, trigger_channel=5.0, envelope_mode=1),   # Separate + RMS, 6 channels
    dict(window_size=300.0, channel_mode=5, envelope_mode=1, trigger_threshold=0.3),  # MidSide + RMS
    dict(window_size=5000.5, channel_mode=1, envelope_mode=1),         # Right + RMS
    dict(window_size=64.0, trigger_threshold=5.0),                     # threshold above the peak: no trigger ever fires
    dict(window_size=4000.0, trigger_mode=0, envelope_mode=1, channel_mode=2),        # triggering off: straight into the front buffer
])
def test_stream_state_machine_is_bit_exact(gpu, oracle, over):
    po = oracle
    cfg = _cfg(**over)
    x = _signal(3, 120000, cfg["num_channels"])
    dev, ref = _feed(po, cfg, x, seed=11)
    assert dev.state() == ref.state()
    if cfg["trigger_mode"] == 4 and cfg["trigger_threshold"] < 1:
        assert ref.state()["swaps"] > 10
    for c in range(cfg["num_channels"]):
        got, gcur = dev.front(c)
        want, wcur = ref.front(c)
        assert gcur == wcur
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (c, int((got != want).sum()))
    gain, env = dev.gains()
    if cfg["envelope_mode"] == 1:
        assert gain == ref.envelope_gain
        assert np.array_equal(env[:2].view(np.uint32), ref.envelopes()[:2].view(np.uint32))


@pytest.mark.parametrize("over,max_block", [
    (dict(), 700),                                                     # BASELINE cfg3, a dozen callbacks per launch
    (dict(window_size=480.3, trigger_threshold=0.0), 300),             # many triggers per callback
    (dict(window_size=1000.0, channel_mode=2, trigger_threshold=0.2), 64),
    (dict(window_size=777.0, channel_mode=3, num_channels=4, trigger_channel=2.0), 1500),
    (dict(window_size=300.0, channel_mode=5, envelope_mode=1, trigger_threshold=0.3), 9),          # tiny callbacks: sixteen per launch
    (dict(window_size=2048.0, channel_mode=4, num_channels=6, trigger_channel=5.0, envelope_mode=1), 400),
])
def test_batched_launches_are_the_callback_walk(gpu, oracle, over, max_block, monkeypatch):
    """the ingest kernel takes every callback that waited in ONE launch, with the zero-crossing detector run once over their
    concatenation (scopeIngestKernel, "A for the whole batch"): SGZ_RT_OPT_DEFER_SUBMIT makes every launch a multi-callback one (blocks
    wait for a full batch or a reader), and the state machine, the rings and the gains are still the oracle's callback-by-callback walk"""
    po = oracle
    cfg = _cfg(**over)
    x = _signal(5, 60000, cfg["num_channels"])
    dev, ref = _feed(po, cfg, x, seed=23, max_block=max_block, defer=True)
    assert dev.state() == ref.state()
    if cfg["trigger_threshold"] < 1:
        assert ref.state()["swaps"] > 10
    for c in range(cfg["num_channels"]):
        got, gcur = dev.front(c)
        want, wcur = ref.front(c)
        assert gcur == wcur
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (c, int((got != want).sum()))
    gain, env = dev.gains()
    if cfg["envelope_mode"] == 1:
        assert gain == ref.envelope_gain
        assert np.array_equal(env[:2].view(np.uint32), ref.envelopes()[:2].view(np.uint32))


@pytest.mark.parametrize("over,max_block", [(dict(window_size=480.3, trigger_threshold=0.0), 300),
                                            (dict(window_size=2048.0, channel_mode=4, num_channels=6, trigger_channel=5.0, envelope_mode=1), 400)])
def test_parked_blocks_reach_the_gpu_with_the_next_read(gpu, oracle, over, max_block):
    """A push that finds the render thread submitting (or no staging slot free) parks its block in the handle's host FIFO.  Flush on read
    covers that FIFO: with SGZ_RT_OPT_PARK_PUSHES EVERY block goes that way, and the only thing that hands blocks to the GPU is a reader
    (the spot checks of _feed, the reads below) -- no flush, no push that finds the way free.  State machine, rings and gains are the
    oracle's callback walk, the blocks pushed after the last mid-stream read included (a stopped transport)."""
    po = oracle
    cfg = _cfg(**over)
    x = _signal(9, 40000, cfg["num_channels"])
    dev, ref = _feed(po, cfg, x, seed=41, max_block=max_block, park=True)
    gain, env = dev.gains()                                           # (first read after the last push: peak-filter-free path)
    assert dev.state() == ref.state()
    for c in range(cfg["num_channels"]):
        got, gcur = dev.front(c)
        want, wcur = ref.front(c)
        assert gcur == wcur
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (c, int((got != want).sum()))
    if cfg["envelope_mode"] == 1:
        assert gain == ref.envelope_gain
        assert np.array_equal(env[:2].view(np.uint32), ref.envelopes()[:2].view(np.uint32))
    # ... and through the peak filter's own submission (it rides on the ingest launch of whatever waits)
    more = _signal(77, 1500, cfg["num_channels"])
    _push(dev, more); ref.audio(more)
    dt = 1.0 / 60
    coeff = np.power(np.exp(-8.0 / (cfg["envelope_window"] * cfg["sample_rate"])), ref.size * dt)
    assert dev.peak_filter(dt, 8) == ref.peak_filter(8, float(coeff))
    assert dev.state() == ref.state()


@pytest.mark.parametrize("mode,channels", [(0, 2), (1, 2), (2, 2), (3, 2), (4, 6), (5, 4)])
def test_peak_filter_every_channel_mode(gpu, oracle, mode, channels):
    """Oscilloscope::runPeakFilter incl. the Mid / Side / MidSide mixes and Separate's running maximum (OscilloscopeDSP.inl:770-880)"""
    po = oracle
    cfg = _cfg(window_size=1001.0, channel_mode=mode, num_channels=channels, envelope_mode=2, trigger_threshold=0.1)
    x = _signal(5, 30000, channels) * np.linspace(0.2, 1.0, channels, dtype=np.float32)[:, None]
    dev, ref = _feed(po, cfg, x, seed=2)
    for frame in range(5):
        dt = 1.0 / 60
        power = ref.size * dt
        coeff = np.power(np.exp(-8.0 / (cfg["envelope_window"] * cfg["sample_rate"])), power)
        want = ref.peak_filter(8, float(coeff))
        got = dev.peak_filter(dt, 8)
        assert got == want, (frame, got, want)
        _, env = dev.gains()
        assert np.array_equal(env.view(np.uint32), ref.envelopes().view(np.uint32))
        more = _signal(50 + frame, 2000, channels)
        _push(dev, more); ref.audio(more)


@pytest.mark.parametrize("trigger_mode,interp,window,evaluator", [
    (4, 3, 19200.0, 0), (4, 3, 19200.0, 1), (4, 3, 480.3, 2), (4, 3, 481.0, 3), (0, 3, 4000.0, 0), (4, 2, 1000.0, 0), (0, 2, 777.0, 2),
    (4, 3, 19200.5, 0)])
def test_wave_plot_vertices(gpu, oracle, trigger_mode, interp, window, evaluator):
    """drawWavePlot's vertex stream from the handle against the oracle's, on the stream's own front buffers.  evaluator: OscChannels
    Left / Right / Mid / Side"""
    po = oracle
    cfg = _cfg(window_size=window, trigger_mode=trigger_mode, interpolation=interp, colours=[(10, 20, 30, 255), (200, 100, 50, 255)])
    x = _signal(7, 90000, 2)
    dev, ref = _feed(po, cfg, x, seed=4)
    width = 1920
    views = [(0.0, 1.0), (0.25, 0.5), (0.4, 0.41)]
    for left, right in views:
        v = api.ScopeView(window, left, right, 1.0, width, 0)
        vo = po.ScopeView(window, left, right, 1.0, width, 0)
        m0, cur = ref.front(0)
        m1, _ = ref.front(1)
        if evaluator == 0: a, b, em = m0, m0, 0
        elif evaluator == 1: a, b, em = m1, m1, 0
        else: a, b, em = m0, m1, evaluator - 1
        want = po.scope_wave_plot(vo, trigger_mode, interp, a, b, em, cur)
        got, colours = dev.vertices(v, evaluator, 0)
        assert got.shape == want.shape, (got.shape, want.shape)
        assert np.array_equal(got[:, 0], want[:, 0]) and np.all(got[:, 2] == 0)
        lanczos = interp == 3 and (width - 1) / (max(1.0, window - 1) * (right - left)) >= 1
        if lanczos:
            assert np.abs(got[:, 1] - want[:, 1]).max() <= 2e-6
        else:
            assert np.array_equal(got[:, 1].view(np.uint32), want[:, 1].view(np.uint32))
        key = (10, 20, 30, 255) if evaluator in (0, 2) else (200, 100, 50, 255)      # getDefaultKey(): Left / Mid -> ch 0, Right / Side -> ch 1
        assert (colours == np.array(key, np.uint8)).all()


def test_cfg3_full_shape(gpu, oracle):
    """BASELINE configs[2]: stereo 192 kHz, 1 s of audio in 512-sample callbacks, 19 200-sample window, threshold 0.05, trigger L,
    Lanczos 8x => ~153 600 points per channel"""
    po = oracle
    cfg = _cfg()
    x = synth.gen(3, 192000, 192000, 2)
    dev = api.Scope(**cfg)
    ref = po.ScopeStream(2, SR, 19200.0, 4, 0.05, 0, 1.0, 0, 0.3)
    for pos in range(0, x.shape[1], 512):
        _push(dev, x[:, pos:pos + 512])
        ref.audio(x[:, pos:pos + 512])
    assert dev.state() == ref.state()
    width = 19200 * 8 + 1
    v = api.ScopeView(19200.0, 0.0, 1.0, 1.0, width, 0)
    for ev in (0, 1):
        got, _ = dev.vertices(v, ev, 0)
        m, cur = ref.front(ev)
        want = po.scope_wave_plot(po.ScopeView(19200.0, 0.0, 1.0, 1.0, width, 0), 4, 3, m, m, 0, cur)
        assert got.shape == want.shape and 153590 <= got.shape[0] <= 153610      # 8 points per sample over the window
        assert np.array_equal(got[:, 0], want[:, 0])
        assert np.abs(got[:, 1] - want[:, 1]).max() <= 2e-6


def test_push_never_waits_and_rejects_bad_input(gpu):
    cfg = _cfg(window_size=1000.0, max_block=256)
    dev = api.Scope(**cfg)
    x = _signal(1, 257, 2)
    with pytest.raises(api.SgzError):
        dev.push(x)                                                    # longer than max_block
    with pytest.raises(api.SgzError):
        dev.push(_signal(1, 10, 4))                                    # wrong channel count
    # a burst far beyond the staging depth: every call returns at once with OK or BUSY, never an error, never a hang
    import time
    t0 = time.perf_counter()
    res = [dev.push(x[:, :256]) for _ in range(4000)]
    dt = time.perf_counter() - t0
    assert set(res) <= {api.SGZ_OK, api.SGZ_BUSY}
    assert res.count(api.SGZ_OK) >= 8
    assert dt < 5.0
    with pytest.raises(api.SgzError):
        api.Scope(**_cfg(trigger_mode=7))                              # no such TriggeringMode
    with pytest.raises(api.SgzError):
        api.Scope(**_cfg(interpolation=9))
    with pytest.raises(api.SgzError):
        api.Scope(**_cfg(num_channels=3))


# ---------------------------------------------------------------------------------------------------------------------------------
# frequency colouring and Spectral triggering

BANDS = [(1.0, 0.25, 0.1), (0.2, 1.0, 0.3), (0.15, 0.35, 1.0)]
KEYS = [(10, 20, 30, 255), (200, 100, 50, 255), (0, 255, 0, 128), (90, 90, 255, 255)]


def _colour_signal(seed, n, channels, sr):
    """tones in all three bands with slowly moving weights + noise: the band energies (hence the colours) keep changing"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    x = np.zeros((channels, n), np.float32)
    for c in range(channels):
        w = 0.5 + 0.5 * np.sin(2 * np.pi * (1.3 + c) * t[:, None] * np.array([0.7, 1.1, 1.9]) + c)
        x[c] = (w[:, 0] * np.sin(2 * np.pi * 97.0 * t + c) + w[:, 1] * np.sin(2 * np.pi * 1130.0 * t) * (0.8 if c & 1 else 1.0)
                + w[:, 2] * 0.7 * np.sin(2 * np.pi * 7900.0 * t + 0.3 * c) + 0.02 * rng.standard_normal(n)).astype(np.float32)
    return x


@pytest.mark.parametrize("over", [
    dict(trigger_mode=0, window_size=3000.0),                                      # straight into the front rings
    dict(trigger_mode=4, window_size=1500.5, trigger_threshold=0.1),               # colours travel through the back rings and the swaps
    dict(trigger_mode=4, window_size=700.0, num_channels=4, channel_mode=4, trigger_channel=3.0, frequency_colouring_blend=0.35),
    dict(trigger_mode=0, window_size=900.0, sample_rate=44100.0, colour_smoothing_ms=0.5, frequency_colouring_blend=0.0),
])
def test_frequency_colouring_is_bit_exact(gpu, oracle, over):
    """audioProcessing's colour path (OscilloscopeDSP.inl:445-517, :588-647): 3-band crossover, band-energy smoothers, accumulateColour,
    for left / right (colourData) and mid / side (auxColourData), in rings that move with the audio rings -- every fp32 operation in the
    oracle's order, so the RGBA8 rings must be identical"""
    po = oracle
    cfg = _cfg(colour_by_frequency=1, frequency_colouring_blend=0.8, colour_smoothing_ms=4.0, band_colours=BANDS, colours=KEYS)
    cfg.update(over)
    C = cfg["num_channels"]
    sr = cfg["sample_rate"]
    x = _colour_signal(2, 60000, C, sr)
    dev = api.Scope(**cfg)
    ref = po.ScopeStream(C, sr, cfg["window_size"], cfg["trigger_mode"], cfg["trigger_threshold"], cfg["channel_mode"],
                         cfg["trigger_channel"], cfg["envelope_mode"], cfg["envelope_window"])
    ref.enable_colours(BANDS, cfg["frequency_colouring_blend"], cfg["colour_smoothing_ms"], (KEYS * 16)[:C])
    rng = np.random.default_rng(8)
    pos = 0
    while pos < x.shape[1]:
        n = int(rng.integers(1, 2500))
        _push(dev, x[:, pos:pos + n]); ref.audio(x[:, pos:pos + n])
        pos += n
    assert dev.state() == ref.state()
    distinct = 0
    for c in range(C):
        a, cur = dev.front(c)
        b, wcur = ref.front(c)
        assert cur == wcur and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for aux in (False, True):
            got = dev.front_colours(c, aux)
            want, _ = ref.front_colours(c, aux)
            assert np.array_equal(got, want), (c, aux, int((got != want).sum()), got[:4], want[:4])
            distinct += len(np.unique(got))
    if cfg["frequency_colouring_blend"] > 0:
        assert distinct > 50 * C                                                   # the colours really vary along the ring
    else:
        assert distinct == 2 * C                                                   # blended out: every pixel is its channel's key

    # per-vertex colours of drawWavePlot (Linear: the sample's colour; Lanczos: lerp of the two newest kernel samples)
    W = cfg["window_size"]
    for interp, width in ((3, 4 * int(W) + 1), (2, 400)):
        dev.configure(interpolation=interp)
        for evaluator in range(4):
            v = api.ScopeView(W, 0.1, 0.9, 1.0, width, 0)
            vo = po.ScopeView(W, 0.1, 0.9, 1.0, width, 0)
            m0, cur = ref.front(0)
            m1, _ = ref.front(1)
            if evaluator == 0: a, b, em, cm = m0, m0, 0, ref.front_colours(0, False)[0]
            elif evaluator == 1: a, b, em, cm = m1, m1, 0, ref.front_colours(1, False)[0]
            elif evaluator == 2: a, b, em, cm = m0, m1, 1, ref.front_colours(0, True)[0]
            else: a, b, em, cm = m0, m1, 2, ref.front_colours(1, True)[0]
            want, wcol = po.scope_wave_plot_ex(vo, cfg["trigger_mode"], interp, a, b, em, cur, 0.0, 0.0, cm)
            got, gcol = dev.vertices(v, evaluator, 0)
            assert got.shape == want.shape
            assert np.abs(got[:, 1] - want[:, 1]).max() <= 2e-6
            # the lerp weight is a difference of fp64 sums computed in closed form on the device: a component may land on the other
            # side of an integer once in a long while
            diff = np.abs(gcol.astype(int) - wcol.astype(int))
            assert diff.max() <= 1 and (diff != 0).mean() < 1e-3, (interp, evaluator, diff.max(), (diff != 0).mean())


def _tone(n, sr, f0, seed, harmonics=(1.0, 0.5, 0.25)):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    x = sum(a * np.sin(2 * np.pi * f0 * (k + 1) * t + 0.7 * k) for k, a in enumerate(harmonics))
    return np.stack([x + 0.01 * rng.standard_normal(n), 0.6 * x + 0.01 * rng.standard_normal(n)]).astype(np.float32)


@pytest.mark.parametrize("sr,window,f0,evaluator,hyst", [
    (48000.0, 2000.0, 441.3, 0, 0.0), (48000.0, 12000.5, 97.0, 2, 0.3), (192000.0, 19200.0, 1234.5, 1, 0.1), (44100.0, 800.0, 60.0, 3, 0.0)])
def test_spectral_trigger_against_the_oracle(gpu, oracle, sr, window, f0, evaluator, hyst):
    """sgz_scope_analyse = calculateFundamentalPeriod + calculateTriggeringOffset on the device ring, frame after frame (the median of 8
    makes it stateful), then drawWavePlot with cycleSamples / sampleOffset inside the reference's ring of the moment.
    Integer decisions (winning bin, median record, ring size) identical; fp64 results to 1e-9 relative (the transform's butterfly
    order and the Goertzel sum's order differ), sampleOffset to 1e-6 samples."""
    po = oracle
    cfg = _cfg(sample_rate=sr, window_size=window, trigger_mode=1, trigger_threshold=0.02, trigger_hysteresis=hyst,
               trigger_phase_offset=30.0, interpolation=3, envelope_mode=2)
    x = _tone(int(sr * 1.2), sr, f0, seed=4)
    dev = api.Scope(**cfg)
    ref = po.ScopeStream(2, sr, window, 1, 0.02, 0, 1.0, 2, 0.3)
    ts = po.SpectralState()
    sz = max(int(0.5 + 0.0 + np.ceil(window)), 8192)
    em = {0: 0, 1: 0, 2: 1, 3: 2}[evaluator]
    pos = 0
    block = 1777
    for frame in range(12):
        for _ in range(6):
            _push(dev, x[:, pos:pos + block]); ref.audio(x[:, pos:pos + block]); pos += block
        mem = [ref.logical(c, sz) for c in (0, 1)]
        a, b = (mem[1], mem[1]) if evaluator == 1 else (mem[0], mem[1]) if em else (mem[0], mem[0])
        po.scope_analyse(ts, a, b, em, 0, window, sr, 0.02, hyst, 30.0)
        got = dev.analyse(evaluator, 0)
        assert got.record_index == ts.record.index, (frame, got.record_index, ts.record.index)
        assert abs(got.record_value - ts.record.value) <= 1e-9 * max(1.0, ts.record.value)
        assert abs(got.record_offset - ts.record.offset) <= 1e-8
        assert abs(got.fundamental - ts.fundamental) <= 1e-8 * ts.fundamental
        assert abs(got.cycle_samples - ts.cycle_samples) <= 1e-8 * ts.cycle_samples
        assert abs(got.sample_offset - ts.sample_offset) <= 1e-6, (frame, got.sample_offset, ts.sample_offset)
        sz = max(int(0.5 + ts.cycle_samples + np.ceil(window)), 8192)
        assert got.ring_size == sz
        if frame >= 5:
            assert abs(got.fundamental - f0) < 0.02 * f0 + 0.5
        if frame in (2, 7, 11):
            # the vertices of this frame, both interpolations (the device call uses its own fp64 cycleSamples / sampleOffset)
            mem = [ref.logical(c, sz) for c in (0, 1)]
            a, b = (mem[1], mem[1]) if evaluator == 1 else (mem[0], mem[1]) if em else (mem[0], mem[0])
            for interp, width in ((3, 2 * int(window) + 1), (2, 300)):
                dev.configure(interpolation=interp)
                v = api.ScopeView(window, 0.0, 1.0, 1.0, width, 0)
                vo = po.ScopeView(window, 0.0, 1.0, 1.0, width, 0)
                want, _ = po.scope_wave_plot_ex(vo, 1, interp, a, b, em, 0, got.cycle_samples, got.sample_offset)
                g, _ = dev.vertices(v, evaluator, 0)
                assert g.shape == want.shape, (g.shape, want.shape)
                assert np.array_equal(g[:, 0], want[:, 0])
                assert np.abs(g[:, 1] - want[:, 1]).max() <= (2e-6 if interp == 3 else 0.0)
    # runPeakFilter over the reference's ring of the moment (the newest ring_size samples; envelopes start at 0 on both sides)
    dt = 1 / 60
    coeff = float(np.power(np.exp(-8.0 / (0.3 * sr)), sz * dt))
    scratch = po.ScopeStream(2, sr, float(sz - 1), 0, 0.0, 0, 1.0, 2, 0.3)       # a ring of exactly sz samples, written once: cursor 0
    assert scratch.size == sz
    scratch.audio(np.stack([ref.logical(c, sz) for c in (0, 1)]))
    assert dev.peak_filter(dt, 8) == scratch.peak_filter(8, coeff)
    dev.close()


def test_spectral_rejects_bad_config(gpu):
    with pytest.raises(api.SgzError):
        api.Scope(**_cfg(trigger_mode=1, trigger_hysteresis=1.5))
    with pytest.raises(api.SgzError):
        api.Scope(**_cfg(trigger_mode=1, custom_trigger=1, custom_trigger_frequency=0.0))
    with pytest.raises(api.SgzError):
        api.Scope(**_cfg(colour_by_frequency=1, sample_rate=4000.0, band_colours=BANDS))


def test_vertices_into_device_buffers(gpu):
    """SURVEY 8(f) #1, vertex side: sgz_scope_vertices_device / sgz_vector_vertices_device write the streams the host calls return into
    caller-owned DEVICE memory (a mapped vertex buffer, or sgz_export_alloc memory exported as a dma-buf) without the D2H copy"""
    import ctypes as C
    import os
    import torch
    L = api.lib()
    cfg = _cfg(window_size=3000.0, colours=[(10, 20, 30, 255), (200, 100, 50, 255)], colour_by_frequency=1, frequency_colouring_blend=0.7,
               colour_smoothing_ms=3.0, band_colours=BANDS)
    dev = api.Scope(**cfg)
    x = _colour_signal(9, 20000, 2, SR)
    for pos in range(0, x.shape[1], 1000):
        _push(dev, x[:, pos:pos + 1000])
    v = api.ScopeView(3000.0, 0.0, 1.0, 1.0, 6001, 0)
    for evaluator in (0, 3):
        want_xyz, want_rgba = dev.vertices(v, evaluator, 0)
        n = want_xyz.shape[0]
        # pinned host buffers: written by the DMA engine directly (rt_common.hpp readBack), the same bytes as through the bounce buffer
        pin_xyz = torch.full((n, 3), float("nan"), dtype=torch.float32).pin_memory().numpy()
        pin_rgba = torch.zeros((n, 4), dtype=torch.uint8).pin_memory().numpy()
        got_xyz, got_rgba = dev.vertices(v, evaluator, 0, out=(pin_xyz, pin_rgba))
        assert np.array_equal(got_xyz.view(np.uint32), want_xyz.view(np.uint32)) and np.array_equal(got_rgba, want_rgba)
        # exported memory: one allocation for vertices + colours
        d_ptr, got_bytes, fd = C.c_void_p(), C.c_size_t(0), C.c_int(-1)
        api.check(L.sgz_export_alloc(n * 16, C.byref(d_ptr), C.byref(got_bytes), C.byref(fd)))
        try:
            assert fd.value >= 0 and got_bytes.value >= n * 16 and got_bytes.value % 4096 == 0
            os.fstat(fd.value)
            cnt = C.c_uint32(n)
            api.check(L.sgz_scope_vertices_device(dev.h, C.byref(v), evaluator, 0, d_ptr, C.c_void_p(d_ptr.value + n * 12), C.byref(cnt)))
            assert cnt.value == n
            host = np.zeros(n * 16, np.uint8)
            hip = C.CDLL("libamdhip64.so")
            assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), d_ptr, C.c_size_t(host.nbytes), 2) == 0
            assert np.array_equal(host[:n * 12].view(np.float32).reshape(n, 3), want_xyz)
            assert np.array_equal(host[n * 12:].reshape(n, 4), want_rgba)
            cnt = C.c_uint32(n - 1)
            assert L.sgz_scope_vertices_device(dev.h, C.byref(v), evaluator, 0, d_ptr, None, C.byref(cnt)) == api.SGZ_EINVAL and cnt.value == n
        finally:
            os.close(fd.value)
            L.sgz_export_free(d_ptr)
    # mixed destinations (round-5 advisor): vertices into DEVICE memory, colours into a pageable host array -- and the reverse.  Neither
    # pair is "all mapped", so the call stages in the handle's own buffers; the copy behind it must treat each destination on its own
    # (a host memcpy into a device pointer would fault)
    want_xyz, want_rgba = dev.vertices(v, 0, 0)
    n = want_xyz.shape[0]
    d_xyz = torch.full((n, 3), float("nan"), dtype=torch.float32, device=gpu)
    h_rgba = np.zeros((n, 4), np.uint8)
    cnt = C.c_uint32(n)
    api.check(L.sgz_scope_vertices(dev.h, C.byref(v), 0, 0, C.c_void_p(d_xyz.data_ptr()), C.c_void_p(h_rgba.ctypes.data), C.byref(cnt)))
    assert cnt.value == n and np.array_equal(d_xyz.cpu().numpy().view(np.uint32), want_xyz.view(np.uint32)) and np.array_equal(h_rgba, want_rgba)
    h_xyz = np.full((n, 3), np.nan, np.float32)
    d_rgba = torch.zeros((n, 4), dtype=torch.uint8, device=gpu)
    cnt = C.c_uint32(n)
    api.check(L.sgz_scope_vertices(dev.h, C.byref(v), 0, 0, C.c_void_p(h_xyz.ctypes.data), C.c_void_p(d_rgba.data_ptr()), C.byref(cnt)))
    assert cnt.value == n and np.array_equal(h_xyz.view(np.uint32), want_xyz.view(np.uint32)) and np.array_equal(d_rgba.cpu().numpy(), want_rgba)
    # several evaluators in one call (sgz_scope_vertices_all): the same strips, pinned buffers (one wait) and pageable ones (item by item)
    items = (0, 1, 2, 3)
    singles = [dev.vertices(v, e, 0) for e in items]
    n = singles[0][0].shape[0]
    for pinned in (True, False):
        mk = (lambda shape, dt: torch.zeros(shape, dtype=dt).pin_memory().numpy()) if pinned else (lambda shape, dt: torch.zeros(shape, dtype=dt).numpy())
        outs = [(mk((n, 3), torch.float32), mk((n, 4), torch.uint8)) for _ in items]
        got = dev.vertices_all(v, items, (0,) * len(items), outs)
        for (gx, gc), (wx, wc) in zip(got, singles):
            assert np.array_equal(gx.view(np.uint32), wx.view(np.uint32)) and np.array_equal(gc, wc)
    cnts = (C.c_uint32 * 2)(n, n - 1)
    xs = (C.c_void_p * 2)(outs[0][0].ctypes.data, outs[1][0].ctypes.data)
    ev = (C.c_uint32 * 2)(0, 1); ch = (C.c_uint32 * 2)(0, 0)
    assert L.sgz_scope_vertices_all(dev.h, C.byref(v), 2, ev, ch, xs, None, cnts) == api.SGZ_EINVAL and cnts[1] == n
    dev.close()

    vec = api.Vector(sample_rate=96000.0, num_channels=4, window_size=2000, envelope_mode=2, lanes=8, fade_history=1, max_block=512,
                     envelope_window=0.3, stereo_window=0.1, colours=[(1.0, 0.5, 0.25), (0.2, 0.9, 0.4)])
    y = _colour_signal(4, 5000, 4, 96000.0)
    for pos in range(0, y.shape[1], 500):
        while vec.push(y[:, pos:pos + 500]) == api.SGZ_BUSY:
            pass
    for pair in (0, 1):
        want_xyz, want_rgb = vec.vertices(pair)
        n = want_xyz.shape[0]
        t_xyz = torch.zeros((n, 3), dtype=torch.float32, device=gpu)
        t_rgb = torch.zeros((n, 3), dtype=torch.float32, device=gpu)
        torch.cuda.synchronize()                                       # (the fills run on torch's stream, the handle's kernels on its own)
        cnt = C.c_uint32(n)
        api.check(L.sgz_vector_vertices_device(vec.h, pair, t_xyz.data_ptr(), t_rgb.data_ptr(), C.byref(cnt)))
        torch.cuda.synchronize()
        assert cnt.value == n
        assert np.array_equal(t_xyz.cpu().numpy(), want_xyz) and np.array_equal(t_rgb.cpu().numpy(), want_rgb)
    vec.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_scope_configurations(gpu, oracle, seed):
    """seeded sweep over the handle's configuration space (channels, trigger / channel / envelope modes, window, threshold, colouring,
    sample rate) and block schedules (1 .. 3000 samples, incl. blocks shorter than one prefetch batch): trigger bookkeeping, rings, colour
    rings, envelopes -- everything bit for bit"""
    po = oracle
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.choice([2, 2, 4, 6]))
    sr = float(rng.choice([44100.0, 48000.0, 96000.0, 192000.0]))
    colours = bool(rng.integers(0, 2))
    cfg = _cfg(sample_rate=sr, window_size=float(np.round(rng.uniform(40, 4000), int(rng.integers(0, 3)))), num_channels=C,
               trigger_mode=int(rng.choice([0, 4, 4, 3, 2])), channel_mode=int(rng.integers(0, 6)), envelope_mode=int(rng.integers(0, 3)),
               trigger_threshold=float(rng.choice([0.0, 0.05, 0.3, 2.0])), trigger_channel=float(rng.integers(1, C + 1)),
               trigger_hysteresis=float(rng.choice([0.0, 0.1, 0.7])), interpolation=int(rng.integers(0, 4)),
               envelope_window=float(rng.uniform(0.01, 0.5)), colour_by_frequency=int(colours),
               frequency_colouring_blend=float(rng.choice([0.0, 0.25, 1.0])), colour_smoothing_ms=float(rng.uniform(0.2, 20.0)),
               band_colours=BANDS, colours=(KEYS * 16)[:C])
    n = int(rng.integers(8000, 40000))
    x = _colour_signal(seed, n, C, sr) * np.float32(rng.uniform(0.1, 1.5))
    dev = api.Scope(**cfg)
    ref = po.ScopeStream(C, sr, cfg["window_size"], cfg["trigger_mode"], cfg["trigger_threshold"], cfg["channel_mode"], cfg["trigger_channel"],
                         cfg["envelope_mode"], cfg["envelope_window"])
    ref.set_hysteresis(cfg["trigger_hysteresis"])
    if colours:
        ref.enable_colours(BANDS, cfg["frequency_colouring_blend"], cfg["colour_smoothing_ms"], (KEYS * 16)[:C])
    pos = 0
    while pos < n:
        m = int(rng.choice([rng.integers(1, 16), rng.integers(16, 600), rng.integers(600, 3000)]))
        _push(dev, x[:, pos:pos + m]); ref.audio(x[:, pos:pos + m])
        pos += m
    assert dev.state() == ref.state(), cfg
    for c in range(C):
        a, cur = dev.front(c)
        b, wcur = ref.front(c)
        assert cur == wcur and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (cfg, c)
        if colours:
            for aux in (False, True):
                assert np.array_equal(dev.front_colours(c, aux), ref.front_colours(c, aux)[0]), (cfg, c, aux)
    gain, env = dev.gains()
    if cfg["envelope_mode"] == 1:
        assert gain == ref.envelope_gain and np.array_equal(env[:2].view(np.uint32), ref.envelopes()[:2].view(np.uint32)), cfg
    if cfg["envelope_mode"] == 2:
        coeff = float(np.power(np.exp(-8.0 / (cfg["envelope_window"] * sr)), ref.size / 60))
        assert dev.peak_filter(1 / 60, 8) == ref.peak_filter(8, coeff), cfg
    # the vertices of one evaluator in the configured trigger mode / interpolation (Window: at a random transport position)
    transport = int(rng.integers(0, 1 << 40))
    dev.set_transport(transport)
    W, interp, tm = cfg["window_size"], cfg["interpolation"], cfg["trigger_mode"]
    width = int(rng.integers(200, 3000))
    left = float(rng.choice([0.0, rng.uniform(0, 0.5)])); right = float(min(1.0, left + rng.uniform(0.1, 1.0)))
    v = api.ScopeView(W, left, right, 1.0, width, 0)
    vo = po.ScopeView(W, left, right, 1.0, width, 0)
    m0, cur = ref.front(0)
    cm = ref.front_colours(0, False)[0] if colours else None
    key = int(np.array((KEYS * 16)[0], np.uint8).view(np.uint32)[0])
    want, wcol = po.scope_wave_plot_ex2(vo, tm, interp, m0, m0, 0, cur, transport_position=transport, key=key, colour_mem=cm)
    got, gcol = dev.vertices(v, 0, 0)
    assert got.shape == want.shape and np.array_equal(got[:, 0], want[:, 0]), (cfg, got.shape, want.shape)
    lanczos = interp == 3 and abs((width - 1) / (max(1.0, W - 1) * (right - left))) >= 1
    assert np.abs(got[:, 1] - want[:, 1]).max() <= (2e-6 if lanczos else 0.0), cfg
    if not lanczos:
        assert np.array_equal(gcol, wcol), cfg


# ---- the remaining trigger modes and interpolations (outside SURVEY 8's rows; VERDICT r2 "missing" #4) ---------------------------------

def _bursts(seed, n, channels):
    """decaying tone bursts: what an envelope-hold trigger is for"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / SR
    env = np.exp(-((t * 37.0) % 1.0) * 6.0)
    x = np.zeros((channels, n), np.float32)
    for c in range(channels):
        x[c] = (env * np.sin(2 * np.pi * 523.0 * (1 + 0.2 * c) * t) + 0.01 * rng.standard_normal(n)).astype(np.float32)
    return x


@pytest.mark.parametrize("over", [dict(window_size=3000.0, trigger_hysteresis=0.1, trigger_threshold=0.05),
                                  dict(window_size=480.3, trigger_hysteresis=0.0, trigger_threshold=0.0, channel_mode=2),
                                  dict(window_size=2048.0, trigger_hysteresis=0.5, trigger_threshold=0.3, channel_mode=4, num_channels=4,
                                       trigger_channel=3.0, envelope_mode=1)])
def test_envelope_hold_trigger_is_bit_exact(gpu, oracle, over):
    """TriggeringMode::EnvelopeHold: PeakHoldProcessor (StreamPreprocessing.h:270-313) feeds the same processMutating automaton"""
    po = oracle
    cfg = _cfg(trigger_mode=3, **over)
    x = _bursts(5, 150000, cfg["num_channels"])
    dev = api.Scope(**cfg)
    ref = po.ScopeStream(cfg["num_channels"], cfg["sample_rate"], cfg["window_size"], 3, cfg["trigger_threshold"], cfg["channel_mode"],
                         cfg["trigger_channel"], cfg["envelope_mode"], cfg["envelope_window"])
    ref.set_hysteresis(cfg["trigger_hysteresis"])
    rng = np.random.default_rng(21)
    pos = 0
    while pos < x.shape[1]:
        n = int(rng.integers(1, 3000))
        blk = x[:, pos:pos + n]
        _push(dev, blk); ref.audio(blk)
        pos += blk.shape[1]
        if rng.random() < 0.1:
            assert dev.state() == ref.state()
    assert dev.state() == ref.state() and ref.state()["swaps"] > 5
    for c in range(cfg["num_channels"]):
        got, gcur = dev.front(c)
        want, wcur = ref.front(c)
        assert gcur == wcur and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # drawn like ZeroCrossing (OscilloscopeRendering.cpp:593-597, :802-805)
    m0, cur = ref.front(0)
    for interp in (3, 2):
        dev.configure(interpolation=interp)
        v = api.ScopeView(cfg["window_size"], 0.0, 1.0, 1.0, 1600, 0)
        vo = po.ScopeView(cfg["window_size"], 0.0, 1.0, 1.0, 1600, 0)
        want, _ = po.scope_wave_plot_ex2(vo, 3, interp, m0, m0, 0, cur)
        got, _ = dev.vertices(v, 0, 0)
        assert got.shape == want.shape and np.array_equal(got[:, 0], want[:, 0])
        assert np.abs(got[:, 1] - want[:, 1]).max() <= (2e-6 if interp == 3 else 0.0)


@pytest.mark.parametrize("interp,window,width", [(0, 777.0, 1920), (1, 777.0, 1920), (1, 300.5, 4000), (1, 19200.0, 1920), (0, 19200.0, 1920)])
def test_none_and_rectangular_interpolation(gpu, oracle, interp, window, width):
    """SubSampleInterpolation::None = the Linear vertex list (drawn as points), Rectangular = two vertices per sample with the previous
    and the current sample's colour; Rectangular (not None) falls back to Linear below one pixel per sample (:575-578)"""
    po = oracle
    cols = [(10, 20, 30, 255), (200, 100, 50, 255)]
    for by_freq in (0, 1):
        cfg = _cfg(window_size=window, trigger_mode=0, interpolation=interp, colours=cols, colour_by_frequency=by_freq,
                   frequency_colouring_blend=0.3, colour_smoothing_ms=5.0,
                   band_colours=[(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)])
        x = _signal(9, 60000, 2)
        dev = api.Scope(**cfg)
        ref = po.ScopeStream(2, SR, window, 0, 0.05, 0, 1.0, 0, 0.3)
        if by_freq:
            ref.enable_colours([(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)], 0.3, 5.0, cols)
        for pos in range(0, x.shape[1], 2500):
            _push(dev, x[:, pos:pos + 2500]); ref.audio(x[:, pos:pos + 2500])
        m1, cur = ref.front(1)
        cm = ref.front_colours(1, False)[0] if by_freq else None
        key = int(np.array(cols[1], np.uint8).view(np.uint32)[0])
        for left, right in ((0.0, 1.0), (0.3, 0.35)):
            v = api.ScopeView(window, left, right, 1.0, width, 0)
            vo = po.ScopeView(window, left, right, 1.0, width, 0)
            want, wcol = po.scope_wave_plot_ex2(vo, 0, interp, m1, m1, 0, cur, key=key, colour_mem=cm)
            got, gcol = dev.vertices(v, 1, 0)
            assert got.shape == want.shape, (got.shape, want.shape)
            assert np.array_equal(got, want)
            assert np.array_equal(gcol, wcol)
        n_lin = max(2, int(np.ceil(window)))
        pps = abs((width - 1) / (max(1.0, window - 1) * (right - left)))            # of the last (zoomed) view
        assert got.shape[0] == (2 * n_lin if interp == 1 and pps >= 1 else n_lin)


@pytest.mark.parametrize("interp,window", [(2, 1000.0), (3, 1000.0), (3, 480.3), (1, 777.0)])
def test_window_trigger_follows_the_transport(gpu, oracle, interp, window):
    """TriggeringMode::Window: audio thread as None, the drawn window starts at fmod(transportPosition, window) (:588-592, :798-801)"""
    po = oracle
    cfg = _cfg(window_size=window, trigger_mode=2, interpolation=interp)
    x = _signal(13, 40000, 2)
    dev = api.Scope(**cfg)
    ref = po.ScopeStream(2, SR, window, 2, 0.05, 0, 1.0, 0, 0.3)
    for pos in range(0, x.shape[1], 1999):
        _push(dev, x[:, pos:pos + 1999]); ref.audio(x[:, pos:pos + 1999])
    m0, cur = ref.front(0)
    for transport in (0, 40000, 123456789, 40000 + 317):
        dev.set_transport(transport)
        v = api.ScopeView(window, 0.0, 1.0, 1.0, 2400, 0)
        vo = po.ScopeView(window, 0.0, 1.0, 1.0, 2400, 0)
        want, _ = po.scope_wave_plot_ex2(vo, 2, interp, m0, m0, 0, cur, transport_position=transport)
        got, _ = dev.vertices(v, 0, 0)
        assert got.shape == want.shape and np.array_equal(got[:, 0], want[:, 0]), (transport, got.shape, want.shape)
        assert np.abs(got[:, 1] - want[:, 1]).max() <= (2e-6 if interp == 3 else 0.0)


def test_custom_trigger_frequency(gpu, oracle):
    """state.customTrigger (OscilloscopeDSP.inl:71-81): Spectral triggering on a named frequency -- no transform, no median filter"""
    po = oracle
    sr, window, f0 = 48000.0, 2000.0, 330.0
    cfg = _cfg(sample_rate=sr, window_size=window, trigger_mode=1, trigger_threshold=0.02, trigger_hysteresis=0.1, trigger_phase_offset=15.0,
               interpolation=3, custom_trigger=1, custom_trigger_frequency=f0)
    t = np.arange(int(sr * 0.8)) / sr
    x = np.stack([np.sin(2 * np.pi * f0 * t + 0.3), 0.5 * np.sin(2 * np.pi * 2 * f0 * t)]).astype(np.float32)
    dev = api.Scope(**cfg)
    ref = po.ScopeStream(2, sr, window, 1, 0.02, 0, 1.0, 0, 0.3)
    ts = po.SpectralState()
    pos, block, sz = 0, 1500, 8192
    for frame in range(6):
        for _ in range(4):
            _push(dev, x[:, pos:pos + block]); ref.audio(x[:, pos:pos + block]); pos += block
        mem = ref.logical(0, sz)
        po.scope_analyse(ts, mem, mem, 0, 0, window, sr, 0.02, 0.1, 15.0, custom_frequency=f0)
        got = dev.analyse(0, 0)
        assert got.record_index == 0 and got.record_value == 1.0 and got.fundamental == f0
        assert abs(got.record_offset - ts.record.offset) <= 1e-12 and got.cycle_samples == ts.cycle_samples
        assert abs(got.sample_offset - ts.sample_offset) <= 1e-6, (frame, got.sample_offset, ts.sample_offset)
        sz = max(int(0.5 + ts.cycle_samples + np.ceil(window)), 8192)
        assert got.ring_size == sz


def test_audio_and_render_threads_run_concurrently(gpu, oracle):
    """a producer thread pushes 1500 callbacks flat out while the render thread runs its frame (peak filter -- riding on the waiting
    batch's launch when there is one --, both channels' strips, state reads): no call fails, no callback is lost or reordered -- the
    trigger state machine and the rings end where the oracle's callback-by-callback walk ends (they depend on the pushes alone)"""
    import threading
    po = oracle
    cfg = _cfg(window_size=3000.0, trigger_threshold=0.02, envelope_mode=2)
    x = _signal(8, 1500 * 160, 2)
    dev = api.Scope(**cfg)
    ref = po.ScopeStream(cfg["num_channels"], cfg["sample_rate"], cfg["window_size"], cfg["trigger_mode"], cfg["trigger_threshold"],
                         cfg["channel_mode"], cfg["trigger_channel"], cfg["envelope_mode"], cfg["envelope_window"])
    errors, frames = [], [0]
    done = threading.Event()

    def producer():
        try:
            for pos in range(0, x.shape[1], 160):
                blk = np.ascontiguousarray(x[:, pos:pos + 160])
                while True:
                    st = dev.push(blk)
                    if st == api.SGZ_OK:
                        break
                    if st != api.SGZ_BUSY:
                        errors.append(("push", st)); return
        finally:
            done.set()

    def render():
        import torch
        view = api.ScopeView(3000.0, 0.0, 1.0, 1.0, 1500, 0)
        n = api.lib().sgz_scope_vertex_count(dev.h, C.byref(view))
        outs = [(torch.zeros((n, 3), dtype=torch.float32).pin_memory().numpy(), torch.zeros((n, 4), dtype=torch.uint8).pin_memory().numpy()) for _ in (0, 1)]
        try:
            while not done.is_set() or frames[0] < 8:            # (at least eight frames however fast the producer is: the count is not a speed test)
                dev.peak_filter(1 / 60, 8)
                dev.vertices_all(view, (0, 1), (0, 0), outs)
                frames[0] += 1
        except Exception as e:                                     # noqa: BLE001
            errors.append(("render", repr(e)))

    import ctypes as C
    tp, tr = threading.Thread(target=producer), threading.Thread(target=render)
    tr.start(); tp.start(); tp.join(timeout=180); tr.join(timeout=180)
    assert not errors and not tp.is_alive() and not tr.is_alive(), errors[:3]
    assert frames[0] >= 8
    for pos in range(0, x.shape[1], 160):
        ref.audio(np.ascontiguousarray(x[:, pos:pos + 160]))
    assert dev.state() == ref.state()
    for c in range(2):
        got, gcur = dev.front(c)
        want, wcur = ref.front(c)
        assert gcur == wcur and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    dev.close()


@pytest.mark.parametrize("freq,block", [(10000.0, 512), (20000.0, 512), (15000.0, 4096), (7000.0, 1000)])
def test_dense_triggers_one_pass_copy(gpu, oracle, freq, block):
    """a tone whose zero crossings make a hundred and more swaps per callback (more than a wave has lanes, up to the 1024 the one-pass
    copy takes, and beyond: 15 kHz in 4096-sample callbacks is 1280): rings and state are the oracle's"""
    po = oracle
    sr = 48000.0
    cfg = _cfg(sample_rate=sr, window_size=4800.0, trigger_threshold=0.05, max_block=4096)
    t = np.arange(block * 40) / sr
    x = np.stack([0.5 * np.sin(2 * np.pi * freq * t), 0.4 * np.sin(2 * np.pi * freq * t + 0.3)]).astype(np.float32)
    dev = api.Scope(**cfg)
    ref = po.ScopeStream(cfg["num_channels"], cfg["sample_rate"], cfg["window_size"], cfg["trigger_mode"], cfg["trigger_threshold"],
                         cfg["channel_mode"], cfg["trigger_channel"], cfg["envelope_mode"], cfg["envelope_window"])
    for pos in range(0, x.shape[1], block):
        blk = np.ascontiguousarray(x[:, pos:pos + block])
        _push(dev, blk)
        ref.audio(blk)
    assert dev.state() == ref.state()
    assert ref.state()["swaps"] > 40 * 60
    for c in range(2):
        got, gcur = dev.front(c)
        want, wcur = ref.front(c)
        assert gcur == wcur and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (c, int((got != want).sum()))
    dev.close()
