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
    if timeout > 0:
        # the producer is idle while a test waits for columns: blocks that queued up behind busy staging slots are enqueued now
        # (on a live stream the next push does that)
        api.lib().sgz_spectrum_flush.argtypes = [C.c_void_p]
        api.check(api.lib().sgz_spectrum_flush(h))
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


def test_stalled_gpu_delays_blocks_instead_of_dropping_them(gpu):
    """push never waits, and a late GPU must not punch holes into the stream.  The handle's stream is stalled
    (it waits for an event behind a 0.3 s spin kernel on another stream) while 40 blocks arrive back to back -- 8 fit the staging slots, the
    rest wait in the host FIFO (rt_common.hpp Backlog) and are enqueued in order once the stream moves again.  The column stream must
    be the batch render's, byte for byte, and no push may have been refused."""
    import torch
    cfg = config.spectrum_config(window_size=4096, hop=4096, axis_points=200)
    W, hop, P, block, nblocks = 4096, 4096, 200, 512, 48
    x = synth.gen(18, 48000, nblocks * block, 2)
    c = api.config_from_dict(cfg)
    h = C.c_void_p()
    L = api.lib()
    api.check(L.sgz_spectrum_create(C.byref(c), C.byref(h)))
    L.sgz_spectrum_stream.restype = C.c_void_p
    L.sgz_spectrum_stream.argtypes = [C.c_void_p]
    L.sgz_spectrum_backlog.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    try:
        side = torch.cuda.Stream()
        ev = torch.cuda.Event()
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(0.3 * 2.0e9))                         # ~0.3 s of spinning
            ev.record(side)
        assert hip.hipStreamWaitEvent(L.sgz_spectrum_stream(h), C.c_void_p(ev.cuda_event), 0) == 0
        for b in range(nblocks - 8):
            blk = np.ascontiguousarray(x[:, b * block:(b + 1) * block])
            ptrs = (C.c_void_p * 2)(blk[0].ctypes.data, blk[1].ctypes.data)
            assert L.sgz_spectrum_push(h, ptrs, 2, block) == api.SGZ_OK      # accepted, every one of them, without waiting
        deferred, waiting = C.c_uint64(0), C.c_uint32(0)
        api.check(L.sgz_spectrum_backlog(h, C.byref(deferred), C.byref(waiting)))
        assert deferred.value >= nblocks - 8 - 8 - 1 and waiting.value > 0, (deferred.value, waiting.value)   # the stream really was stalled
        torch.cuda.synchronize()                                         # the spin kernel is over
        cols = []
        for b in range(nblocks - 8, nblocks):                            # the next pushes drain the FIFO, in order, then take their own blocks
            blk = np.ascontiguousarray(x[:, b * block:(b + 1) * block])
            ptrs = (C.c_void_p * 2)(blk[0].ctypes.data, blk[1].ctypes.data)
            assert L.sgz_spectrum_push(h, ptrs, 2, block) == api.SGZ_OK
            cols += _pop_all(h, P, 1, timeout=0.0)
        frames = nblocks * block // hop
        cols += _pop_all(h, P, frames - len(cols))
        api.check(L.sgz_spectrum_backlog(h, C.byref(deferred), C.byref(waiting)))
        assert waiting.value == 0
        dropped, refused = C.c_uint64(0), C.c_uint64(0)
        api.check(L.sgz_spectrum_stats(h, C.byref(dropped), C.byref(refused)))
        assert dropped.value == 0 and refused.value == 0
        padded = np.concatenate([np.zeros((2, W), np.float32), x], axis=1)[:, hop:]
        plan = api.Plan(cfg).upload()
        batch = plan.render(torch.from_numpy(np.ascontiguousarray(padded)).to(gpu)).cpu().numpy()[:frames]
        assert len(cols) == frames and np.array_equal(np.stack(cols), batch)
    finally:
        L.sgz_spectrum_destroy(h)


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


def _push_all(h, x, block):
    for pos in range(0, x.shape[1], block):
        blk = np.ascontiguousarray(x[:, pos:pos + block])
        ptrs = (C.c_void_p * blk.shape[0])(*[blk[c].ctypes.data for c in range(blk.shape[0])])
        while True:
            st = api.lib().sgz_spectrum_push(h, ptrs, blk.shape[0], blk.shape[1])
            if st != api.SGZ_BUSY:
                break
        api.check(st)
    # (blocks that queued up behind busy staging slots: enqueued now, the caller looks at the ring / the columns next)
    api.lib().sgz_spectrum_flush.argtypes = [C.c_void_p]
    api.check(api.lib().sgz_spectrum_flush(h))


def test_device_ring_history_across_wrap_around(gpu):
    """the audio history is a mirrored ring in HBM that K_A reads in place.  After far more samples than the ring
    holds (several wrap-arounds, odd block sizes), the window a frame would transform is exactly the W newest samples pushed."""
    cfg = config.spectrum_config(window_size=4096, hop=1024, axis_points=64)
    c = api.config_from_dict(cfg)
    h = C.c_void_p()
    api.check(api.lib().sgz_spectrum_create(C.byref(c), C.byref(h)))
    try:
        rng = np.random.default_rng(3)
        x = synth.gen(21, 48000, 200000, 2)                 # ring capacity: 4096 + 16384 samples
        pos = 0
        buf = np.zeros(4096, np.float32)
        while pos < x.shape[1]:
            n = int(rng.integers(1, 5000))
            _push_all(h, x[:, pos:pos + n], 5000)
            pos = min(pos + n, x.shape[1])
            if rng.random() < 0.2 and pos >= 4096:
                for ch in range(2):
                    api.check(api.lib().sgz_spectrum_history(h, ch, buf.ctypes.data_as(C.c_void_p)))
                    assert np.array_equal(buf, x[ch, pos - 4096:pos]), (pos, ch)
    finally:
        api.lib().sgz_spectrum_destroy(h)


def test_mix_matrix_routes_sources_additively(gpu, oracle):
    """MixGraphListener::deliver (MixGraphListener.cpp:247-334): destination = the sum of its routed source channels, added in
    ascending source order onto a cleared row.  Columns of the mixed stream == the batch render of the host-mixed audio."""
    import torch
    cfg = config.spectrum_config(window_size=4096, hop=1024, axis_points=200, num_pairs=2)
    c = api.config_from_dict(cfg)
    h = C.c_void_p()
    api.check(api.lib().sgz_spectrum_create(C.byref(c), C.byref(h)))
    try:
        nsrc = 6
        M = np.zeros((4, nsrc), np.uint8)
        M[0, [0, 3]] = 1; M[1, [1, 4, 5]] = 1; M[2, 2] = 1               # destination 3 gets nothing: "nothing" port, silence
        api.check(api.lib().sgz_spectrum_set_mix(h, nsrc, M.ctypes.data_as(C.c_void_p)))
        src = synth.gen(31, 48000, 1024 * 12, nsrc)
        _push_all(h, src, 480)
        cols = _pop_all(h, 200, 12)
        assert len(cols) == 10                                           # the queue holds frameQueue's 10
        mixed = np.zeros((4, src.shape[1]), np.float32)
        for d in range(4):
            for s_ in range(nsrc):
                if M[d, s_]:
                    mixed[d] = mixed[d] + src[s_]
        hist = np.zeros(4096, np.float32)
        for d in range(4):
            api.check(api.lib().sgz_spectrum_history(h, d, hist.ctypes.data_as(C.c_void_p)))
            assert np.array_equal(hist, mixed[d, -4096:])
        padded = np.concatenate([np.zeros((4, 4096), np.float32), mixed], axis=1)[:, 1024:]
        batch = api.Plan(cfg).upload().render(torch.from_numpy(np.ascontiguousarray(padded)).to(gpu)).cpu().numpy()
        assert np.array_equal(np.stack(cols), batch[:10])
    finally:
        api.lib().sgz_spectrum_destroy(h)


def test_push_never_waits(gpu):
    """a burst far beyond the staging depth: every push returns at once with OK or BUSY (block not taken), and the stream of
    columns is that of exactly the accepted blocks"""
    import time
    cfg = config.spectrum_config(window_size=32768, hop=8192, axis_points=1024)
    c = api.config_from_dict(cfg)
    h = C.c_void_p()
    api.check(api.lib().sgz_spectrum_create(C.byref(c), C.byref(h)))
    try:
        x = synth.gen(5, 48000, 8192, 2)
        ptrs = (C.c_void_p * 2)(x[0].ctypes.data, x[1].ctypes.data)
        worst = 0.0
        res = []
        for _ in range(300):
            t0 = time.perf_counter()
            res.append(api.lib().sgz_spectrum_push(h, ptrs, 2, 8192))
            worst = max(worst, time.perf_counter() - t0)
        assert set(res) <= {api.SGZ_OK, api.SGZ_BUSY}
        assert res.count(api.SGZ_OK) >= 8
        assert worst < 0.05, worst                                        # no hipMalloc, no event wait, no stream sync in push
        dropped, refused = C.c_uint64(0), C.c_uint64(0)
        api.check(api.lib().sgz_spectrum_stats(h, C.byref(dropped), C.byref(refused)))
        assert refused.value == res.count(api.SGZ_BUSY)
    finally:
        api.lib().sgz_spectrum_destroy(h)


# ---------------------------------------------------------------------------------------------------------------------------------
# the display hand-off without the host

def _flush_all(h, want, timeout=10.0):
    """(first, count) of every non-empty flush until `want` columns have been written"""
    out, total = [], 0
    t0 = time.time()
    first, cnt = C.c_uint32(0), C.c_uint32(0)
    while total < want and time.time() - t0 < timeout:
        st = api.lib().sgz_spectrum_flush_columns(h, C.byref(first), C.byref(cnt))
        if st == api.SGZ_OK:
            out.append((first.value, cnt.value)); total += cnt.value
        else:
            assert st == api.SGZ_EMPTY
            time.sleep(0.001)
    return out, total


_IMPORTER = r"""
# second process of test_columns_land_in_a_device_image: import the exported dma-buf fd as HIP external memory, read the image back
import ctypes as C, sys
import numpy as np
fd, nbytes, alloc, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
hip = C.CDLL("libamdhip64.so")
class Handle(C.Union):
    _fields_ = [("fd", C.c_int), ("win32", C.c_void_p * 2), ("nvSciBufObject", C.c_void_p)]
class HandleDesc(C.Structure):
    _fields_ = [("type", C.c_int), ("handle", Handle), ("size", C.c_ulonglong), ("flags", C.c_uint), ("reserved", C.c_uint * 16)]
class BufferDesc(C.Structure):
    _fields_ = [("offset", C.c_ulonglong), ("size", C.c_ulonglong), ("flags", C.c_uint), ("reserved", C.c_uint * 16)]
assert hip.hipSetDevice(0) == 0
hd = HandleDesc(); hd.type = 1; hd.handle.fd = fd; hd.size = alloc          # hipExternalMemoryHandleTypeOpaqueFd
ext = C.c_void_p()
e = hip.hipImportExternalMemory(C.byref(ext), C.byref(hd))
assert e == 0, ("hipImportExternalMemory", e)
bd = BufferDesc(); bd.offset = 0; bd.size = alloc
ptr = C.c_void_p()
e = hip.hipExternalMemoryGetMappedBuffer(C.byref(ptr), ext, C.byref(bd))
assert e == 0, ("hipExternalMemoryGetMappedBuffer", e)
host = np.zeros(nbytes, np.uint8)
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
assert hip.hipMemcpy(host.ctypes.data, ptr, nbytes, 2) == 0
np.save(out, host)
print("imported", nbytes)
"""


@pytest.mark.parametrize("own_image", [False, True])
def test_columns_land_in_a_device_image(gpu, oracle, own_image):
    """sgz_spectrum_flush_columns writes texel (x, y) = column[y] at x = framePixelPosition, wrapping at the image width -- the texels
    oglImage.updateSingleColumn would upload (SpectrumRendering.cpp:696-721) -- into caller-owned device memory (the mock of a mapped
    interop resource) or into the library's own image, which is also exported as a dma-buf fd."""
    import os
    import torch
    P, hop, W, columns = 200, 512, 4096, 7
    cfg = config.spectrum_config(window_size=W, hop=hop, axis_points=P)
    c = api.config_from_dict(cfg)
    x = synth.gen(5, 48000, hop * 19, 2)
    # reference columns: a second handle drained with pop_column
    h2 = C.c_void_p()
    api.check(api.lib().sgz_spectrum_create(C.byref(c), C.byref(h2)))
    want = []
    for pos in range(0, x.shape[1], hop):
        _push_all(h2, x[:, pos:pos + hop], hop)
        want += _pop_all(h2, P, 1)
    api.lib().sgz_spectrum_destroy(h2)
    want = np.stack(want).view(np.uint32)[:, :, 0]                                  # [frames][P]
    assert want.shape[0] == 19
    # ... and those columns are the ORACLE's, through the parity chain: the batch render of the same stream (history = W zeros) gives the
    # same bytes, and the batch render is held to the oracle link by link (tests/parity_chain.py)
    from parity_chain import check_render
    padded = np.ascontiguousarray(np.concatenate([np.zeros((2, W), np.float32), x], axis=1)[:, hop:])
    plan = api.Plan(cfg).upload()
    batch = plan.render(torch.from_numpy(padded).to(gpu)).cpu().numpy()[:19]
    assert np.array_equal(batch.view(np.uint32)[:, :, 0], want)
    problems, _ = check_render(oracle, plan, cfg, padded, gpu)
    assert not problems, problems

    h = C.c_void_p()
    api.check(api.lib().sgz_spectrum_create(C.byref(c), C.byref(h)))
    fd = -1
    try:
        assert api.lib().sgz_spectrum_flush_columns(h, None, None) == api.SGZ_EINVAL           # nothing bound
        if own_image:
            d_img, pitch, cfd = C.c_void_p(), C.c_size_t(0), C.c_int(-1)
            api.check(api.lib().sgz_spectrum_create_image(h, columns, C.byref(d_img), C.byref(pitch), C.byref(cfd)))
            fd = cfd.value
            assert fd >= 0 and os.fstat(fd).st_size >= 0                                       # a live file descriptor (the dma-buf)
            assert pitch.value >= 4 * columns and pitch.value % 256 == 0
            pitch_b = pitch.value
            img_t = None
        else:
            pitch_b = 4 * (columns + 3)                                                        # a pitch wider than the image
            img_t = torch.full((P, pitch_b // 4), 0x01020304, dtype=torch.int32, device=gpu)
            api.check(api.lib().sgz_spectrum_bind_image(h, img_t.data_ptr(), columns, pitch_b))

            def read_image():
                torch.cuda.synchronize()
                return img_t.cpu().numpy().view(np.uint32)
        written = 0
        for pos in range(0, x.shape[1], hop * 3):                                              # three frames per push: several columns per flush
            _push_all(h, x[:, pos:pos + hop * 3], hop * 3)
            k = min(3, 19 - written)
            ranges, total = _flush_all(h, k)
            assert total == k
            x0 = ranges[0][0]
            assert x0 == written % columns
            written += k
        if own_image:
            host = np.zeros((P, pitch_b // 4), np.uint32)
            hip = C.CDLL("libamdhip64.so")
            assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), d_img, C.c_size_t(host.nbytes), 2) == 0       # hipMemcpyDeviceToHost
            img = host
            # the consumer side of the hand-off: ANOTHER PROCESS imports the dma-buf fd (HIP external memory, the path a Vulkan / GL
            # interop or a compositor takes) and must read the same texels
            import subprocess, sys, tempfile
            alloc = (host.nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)        # the library exports whole 2 MiB blocks (sgz.h)
            with tempfile.TemporaryDirectory() as td:
                outp = os.path.join(td, "img.npy")
                r = subprocess.run([sys.executable, "-c", _IMPORTER, str(fd), str(host.nbytes), str(alloc), outp], pass_fds=(fd,),
                                   capture_output=True, text=True, timeout=300)
                assert r.returncode == 0 and "imported" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-1500:])
                other = np.load(outp).view(np.uint32).reshape(host.shape)
            assert np.array_equal(other, host)
        else:
            img = read_image()
            assert (img[:, columns:] == 0x01020304).all()                                      # texels beyond the image width are untouched
        # column x holds the newest frame f with f % columns == x
        for xcol in range(columns):
            f = max(f for f in range(19) if f % columns == xcol)
            assert np.array_equal(img[:, xcol], want[f]), (xcol, f)
        # rebinding resets framePixelPosition; unbinding makes flush an error again
        if not own_image:
            api.check(api.lib().sgz_spectrum_bind_image(h, None, 0, 0))
            assert api.lib().sgz_spectrum_flush_columns(h, None, None) == api.SGZ_EINVAL
            assert api.lib().sgz_spectrum_bind_image(h, img_t.data_ptr(), columns, 4 * columns - 4) == api.SGZ_EINVAL
    finally:
        api.lib().sgz_spectrum_destroy(h)
        if fd >= 0:
            os.close(fd)


def test_gl_buffer_binding_fails_cleanly_without_a_context(gpu):
    """an MI355X has no graphics engine and this box no display: hipGraphicsGLRegisterBuffer cannot succeed, and the call must say so
    with a status instead of taking the process down (run in a child so that a crash inside the GL loader shows up as a failure)"""
    import subprocess
    import sys
    code = (
        "import ctypes as C\n"
        "from signalizer_amd import api, config\n"
        "c = api.config_from_dict(config.spectrum_config(window_size=4096, hop=512, axis_points=64))\n"
        "h = C.c_void_p()\n"
        "api.check(api.lib().sgz_spectrum_create(C.byref(c), C.byref(h)))\n"
        "st = api.lib().sgz_spectrum_bind_gl_buffer(h, 1, 16, 64)\n"
        "assert st < 0, st\n"
        "assert api.lib().sgz_spectrum_flush_columns(h, None, None) == api.SGZ_EINVAL\n"
        "api.lib().sgz_spectrum_destroy(h)\n"
        "print('clean', st)\n")
    import os
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "clean" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])


_GL_ROUND_TRIP = r'''
import ctypes as C, ctypes.util, os, sys
import numpy as np
reasons = []
def load(name):
    try:
        return C.CDLL(name)
    except OSError as e:
        reasons.append(f"{name}: {e}")
        return None
ctx_ok = False
egl = load("libEGL.so.1") or load("libEGL.so")
getproc = None
if egl is not None:
    EGL_PLATFORM_SURFACELESS_MESA, EGL_OPENGL_API, EGL_NONE = 0x31DD, 0x30A2, 0x3038
    egl.eglGetProcAddress.restype = C.c_void_p; egl.eglGetProcAddress.argtypes = [C.c_char_p]
    gpd = egl.eglGetProcAddress(b"eglGetPlatformDisplayEXT")
    dpy = None
    if gpd:
        dpy = C.CFUNCTYPE(C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p)(gpd)(EGL_PLATFORM_SURFACELESS_MESA, None, None)
    if not dpy:
        egl.eglGetDisplay.restype = C.c_void_p; egl.eglGetDisplay.argtypes = [C.c_void_p]
        dpy = egl.eglGetDisplay(None)
    major, minor = C.c_int(), C.c_int()
    egl.eglInitialize.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    if not dpy or not egl.eglInitialize(dpy, C.byref(major), C.byref(minor)):
        reasons.append("EGL: no display could be initialised (surfaceless / default)")
    else:
        egl.eglBindAPI(EGL_OPENGL_API)
        cfg, n = C.c_void_p(), C.c_int()
        attrs = (C.c_int * 3)(0x3033, 0x0001, EGL_NONE)          # EGL_SURFACE_TYPE, EGL_PBUFFER_BIT
        egl.eglChooseConfig.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        egl.eglChooseConfig(dpy, attrs, C.byref(cfg), 1, C.byref(n))
        egl.eglCreateContext.restype = C.c_void_p; egl.eglCreateContext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        ctx = egl.eglCreateContext(dpy, cfg if n.value else None, None, None)
        egl.eglMakeCurrent.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if not ctx or not egl.eglMakeCurrent(dpy, None, None, ctx):
            reasons.append("EGL: no surfaceless OpenGL context (eglCreateContext / eglMakeCurrent failed)")
        else:
            ctx_ok = True
            getproc = lambda nm: egl.eglGetProcAddress(nm)
if not ctx_ok:
    x11, glx = load("libX11.so.6"), load("libGL.so.1")
    if x11 is not None and glx is not None:
        x11.XOpenDisplay.restype = C.c_void_p; x11.XOpenDisplay.argtypes = [C.c_char_p]
        xd = x11.XOpenDisplay(None)
        if not xd:
            reasons.append(f"GLX: XOpenDisplay(NULL) failed (DISPLAY={os.environ.get('DISPLAY')!r}: no X server on this box)")
        else:
            reasons.append("GLX: an X display exists but this test only drives EGL contexts")
if not ctx_ok:
    print("NO_GL " + "; ".join(reasons))
    sys.exit(0)
# ---- a context is current: GL buffer <- sgz_spectrum_flush_columns, read back, held to a second handle's popped columns -------------
from signalizer_amd import api, config, synth
def gl(name, restype, *argtypes):
    p = getproc(name)
    assert p, name
    return C.CFUNCTYPE(restype, *argtypes)(p)
glGenBuffers = gl(b"glGenBuffers", None, C.c_int, C.POINTER(C.c_uint))
glBindBuffer = gl(b"glBindBuffer", None, C.c_uint, C.c_uint)
glBufferData = gl(b"glBufferData", None, C.c_uint, C.c_ssize_t, C.c_void_p, C.c_uint)
glGetBufferSubData = gl(b"glGetBufferSubData", None, C.c_uint, C.c_ssize_t, C.c_ssize_t, C.c_void_p)
glFinish = gl(b"glFinish", None)
GL_PIXEL_UNPACK_BUFFER, GL_DYNAMIC_DRAW = 0x88EC, 0x88E8
P, columns, hop = 64, 16, 512
d = config.spectrum_config(window_size=4096, hop=hop, axis_points=P)
c = api.config_from_dict(d)
L = api.lib()
h, h2 = C.c_void_p(), C.c_void_p()
api.check(L.sgz_spectrum_create(C.byref(c), C.byref(h))); api.check(L.sgz_spectrum_create(C.byref(c), C.byref(h2)))
buf = C.c_uint(0)
glGenBuffers(1, C.byref(buf)); glBindBuffer(GL_PIXEL_UNPACK_BUFFER, buf.value)
pitch = columns * 4
glBufferData(GL_PIXEL_UNPACK_BUFFER, pitch * P, None, GL_DYNAMIC_DRAW); glFinish()
st = L.sgz_spectrum_bind_gl_buffer(h, buf.value, columns, pitch)
if st != 0:
    print("NO_GL hipGraphicsGLRegisterBuffer refused the buffer of this context: " + (L.sgz_last_error() or b"").decode(errors="replace")); sys.exit(0)
x = synth.gen(5, 48000, 4096 + hop * 11, 2)
for hh in (h, h2):
    ptrs = (C.c_void_p * 2)(x[0].ctypes.data, x[1].ctypes.data)
    for pos in range(0, x.shape[1], hop):
        n = min(hop, x.shape[1] - pos)
        ptrs = (C.c_void_p * 2)(x[0, pos:].ctypes.data, x[1, pos:].ctypes.data)
        api.check(L.sgz_spectrum_push(hh, ptrs, 2, n))
    api.check(L.sgz_spectrum_flush(hh))
import torch; torch.cuda.synchronize()
first, count = C.c_uint32(), C.c_uint32()
got_cols = 0
image = np.zeros((P, columns, 4), np.uint8)
want = np.zeros((P, columns, 4), np.uint8)
for _ in range(200):
    st = L.sgz_spectrum_flush_columns(h, C.byref(first), C.byref(count))
    if st == 0: got_cols += count.value
    if got_cols >= 10: break
col = np.zeros((P, 4), np.uint8); k = 0
while k < got_cols and L.sgz_spectrum_pop_column(h2, col.ctypes.data_as(C.c_void_p), None) == 0:
    want[:, k % columns] = col; k += 1
glFinish()
glGetBufferSubData(GL_PIXEL_UNPACK_BUFFER, 0, image.nbytes, image.ctypes.data_as(C.c_void_p))
assert got_cols >= 10 and k == got_cols and np.array_equal(image, want), (got_cols, k, int((image != want).sum()))
print("GL_OK", got_cols)
'''


def test_gl_buffer_round_trip(gpu):
    """SURVEY 8(f) #1 / SpectrumRendering.cpp:696-721, :742-744: the columns written straight into an OpenGL buffer object
    (sgz_spectrum_bind_gl_buffer -> hipGraphicsGLRegisterBuffer; flush_columns maps / unmaps around its writes) and read back with
    glGetBufferSubData equal the columns a second handle pops for the same audio.  Needs an OpenGL context on the SAME device: an EGL
    surfaceless context is tried (then GLX is probed); where the box offers none the test SKIPS with the loader's own error text -- which is
    the proof that the path cannot be executed there (MI355X: no display engine; this image ships libGL / GLX only, no libEGL, no X server)."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _GL_ROUND_TRIP], capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    no_gl = [ln for ln in r.stdout.splitlines() if ln.startswith("NO_GL")]
    if r.returncode == 0 and no_gl:
        pytest.skip("no OpenGL context can be created on this box -- " + no_gl[0][6:])
    assert r.returncode == 0 and "GL_OK" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-2000:])
