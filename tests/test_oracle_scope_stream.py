"""Known answers for the oracle's restatement of the Oscilloscope's audio-thread state machine (oracle/scope_stream.c): the
reference ships no tests, so the restatement is pinned against what the code it follows must do by construction."""
import numpy as np
import pytest

SR = 48000.0


def _sine(n, f=440.3, amp=0.8):
    t = np.arange(n) / SR
    return np.stack([amp * np.sin(2 * np.pi * f * t), 0.5 * np.sin(2 * np.pi * 100 * t)]).astype(np.float32)


def _feed(s, x, seed=0, max_block=700):
    rng = np.random.default_rng(seed)
    pos = 0
    while pos < x.shape[1]:
        n = int(rng.integers(1, max_block))
        s.audio(x[:, pos:pos + n])
        pos += n


def test_triggering_off_front_buffer_is_the_newest_window(oracle):
    """KA: TriggeringMode::None -> audioProcessing writes straight into the front ring (OscilloscopeDSP.inl:415-418): in time order
    the ring holds the newest ceil(W + 1) samples"""
    po = oracle
    x = _sine(5000)
    s = po.ScopeStream(2, SR, 480.0, trigger_mode=po.TRIG_NONE)
    _feed(s, x)
    assert s.size == 481
    for c in range(2):
        assert np.array_equal(s.front_in_time_order(c), x[c, -481:])


def test_threshold_above_the_peak_never_swaps(oracle):
    """KA11: threshold > peak => no trigger => the front buffer is never written"""
    po = oracle
    s = po.ScopeStream(2, SR, 480.0, threshold=2.0)
    _feed(s, _sine(20000))
    st = s.state()
    assert st["swaps"] == 0 and st["peaks"] == 0
    assert not s.front(0)[0].any()
    assert st["bufferedSamples"] == 481 and st["frontOrigin"] + st["bufferedSamples"] == st["steadyClock"] == 20000


def test_front_buffer_is_contiguous_audio_ending_after_the_trigger(oracle):
    """processMutating always hands the OLDEST buffered samples to the front ring (swapBuffers(cappedSize, -bufferedSamples)), so in
    time order the ring is one contiguous slice of the input ending at frontOrigin, and the last trigger lies inside it with at
    least half a window behind it"""
    po = oracle
    x = _sine(30000)
    s = po.ScopeStream(2, SR, 480.0, threshold=0.1)
    _feed(s, x, seed=3)
    st = s.state()
    assert st["swaps"] > 200
    end = st["frontOrigin"]
    for c in range(2):
        assert np.array_equal(s.front_in_time_order(c), x[c, end - 481:end])
    assert end - 481 <= st["oldPeak"] - 240 and st["oldPeak"] < end
    # the trigger is an upward zero crossing of channel 0 followed by a sample above the threshold
    k = st["oldPeak"]
    assert x[0, k] > 0 and x[0, k - 1] < 0


def test_trigger_indices_match_the_stateless_detector(oracle):
    """the stream's detector (executeSamplingWindows on the block, state carried in TriggeringProcessor) fires where the stand-alone
    ZeroCrossingProcessor restatement does: count of swaps + pending == number of triggers"""
    po = oracle
    x = _sine(12000, f=997.0)
    s = po.ScopeStream(2, SR, 256.0, threshold=0.3)
    _feed(s, x, seed=5)
    zs = po.ZeroCrossingState()
    zs.threshold = 0.3
    trig = po.zero_crossing(zs, po.OSC_LEFT, x[0])
    st = s.state()
    assert st["swaps"] + st["peaks"] == len(trig)


@pytest.mark.parametrize("mode", range(6))
def test_rms_envelope_known_answer(oracle, mode):
    """KA13: y_n = x^2 + a (y_{n-1} - x^2) on a constant input converges to x^2; envelopeGain = 1 / sqrt(max env)"""
    po = oracle
    x = np.full((2, 40000), 0.5, np.float32)
    x[1] = 0.25
    s = po.ScopeStream(2, SR, 480.0, trigger_mode=po.TRIG_NONE, osc_mode=mode, env_mode=po.ENV_RMS, envelope_window=0.01)
    _feed(s, x)
    expect = {0: 0.5, 1: 0.25, 2: 0.375, 3: 0.125, 4: 0.5, 5: np.sqrt(0.5 * 0.75 ** 2)}[mode]
    assert abs(1.0 / s.envelope_gain - expect) < 1e-4
