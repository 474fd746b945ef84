"""Host-side bookkeeping for the stream-mode tests: WHICH W samples each frame of the reference's block-driven paths sees, derived
independently of oracle/spectrum_stream.c (numpy on the timeline, no ring) so that the two can be held against each other:

  strict_frames   TransformPair::audioEntryPoint as written (Source/Spectrum/TransformDSP.inl:1165-1211 with the block overload of
                  prepareTransform, :234-484): quirk Q1 (every frame of a callback = the history before the callback ++ the first
                  min(availableSamples, W) samples of the UN-offset block) and quirk Q2 (history longer than the window: the frame
                  loses the newest `history - W` history samples and is zero-padded)
  newest_windows  DisplayMode::LineGraph (SpectrumRendering.cpp:617-635, whole-ring prepareTransform :39-231): the W newest samples
                  at every render call

Laid end to end, such frames are a stream whose IDEAL framing at hop == W reproduces them one by one -- which is how the tests tie the
block-driven paths to the offline render (and, through tests/parity_chain.py, to the oracle)."""
import numpy as np


def strict_frames(blocks, W, hop, history=None):
    """blocks: list of [channels][n] float32.  Returns (frames: list of [channels][W], frames_per_block: list[int])."""
    nch = blocks[0].shape[0]
    H = W if not history else int(history)
    extra = H - W
    hist = np.zeros((nch, H), np.float32)                      # oldest first; the stream starts as silence
    since = 0
    frames, per_block = [], []
    for blk in blocks:
        n = blk.shape[1]
        rem, produced = n, 0
        while rem > 0:
            remaining = 0 if since > hop else hop - since
            avail = min(remaining, rem)
            since += avail
            if since >= hop:
                stop = min(avail, W)
                if stop + extra <= W:
                    take = hist[:, stop + extra:W]              # Q2: W - stop - extra samples; the newest `extra` are never reached
                else:
                    take = hist[:, stop + extra:]               # (sizeToStopAt wrapped around: everything that is left)
                f = np.zeros((nch, W), np.float32)
                f[:, :take.shape[1]] = take
                f[:, take.shape[1]:take.shape[1] + stop] = blk[:, :stop]      # Q1: the block's FIRST samples, whatever `offset` is
                frames.append(f)
                produced += 1
                since = 0
            rem -= avail
        hist = np.concatenate([hist, blk], axis=1)[:, -H:]
        per_block.append(produced)
    return frames, per_block


def newest_windows(blocks, W, render_after):
    """the W newest samples after block k for every k in render_after (a block index may repeat: several renders without new audio)"""
    nch = blocks[0].shape[0]
    hist = np.zeros((nch, W), np.float32)
    out = []
    for k, blk in enumerate(blocks):
        hist = np.concatenate([hist, blk], axis=1)[:, -W:]
        out += [hist.copy() for r in render_after if r == k]
    return out


def cut(x, sizes):
    """x [channels][S] cut into consecutive blocks of the given sizes (cycled) until S is used up"""
    blocks, at, i = [], 0, 0
    while at < x.shape[1]:
        n = min(sizes[i % len(sizes)], x.shape[1] - at)
        blocks.append(np.ascontiguousarray(x[:, at:at + n]))
        at += n
        i += 1
    return blocks
