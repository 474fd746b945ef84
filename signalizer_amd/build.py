"""Build libsgz.so (HIP kernels + C-ABI shim) in-tree for gfx950 with hipcc."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsgz.so")
SOURCES = ["plan.cpp", "spectrum_fft.hip", "spectrum_real.hip", "spectrum_real16.hip", "spectrum_generic.hip", "spectrum_post.hip", "resonator.hip", "scope_vector.hip", "scope_stream.hip", "vector_stream.hip", "sharded.hip", "tracker.hip", "api.hip", "realtime.hip"]
DRIVER = os.path.join(HERE, "librtdriver.so")
DRIVER_SRC = os.path.join(os.path.dirname(HERE), "tools", "rt_driver.cpp")
HEADERS = ["plan.hpp", "rt_lockfree.hpp", "kernels.hpp", "fft_common.hpp", "fft_scalar.hpp", "chunk_map.hpp", "real_common.hpp", "late_fix.hpp", "stft_body.hpp", "complex_dc.hpp", "decay_body.hpp", "runtime.hpp", "rt_common.hpp", "trace.hpp", "fade_chain.hpp", os.path.join("..", "..", "include", "sgz.h")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__), DRIVER_SRC]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if s.endswith(".cpp"):   # pure host code: strict fp (tables must match the reference's fp64 expression order)
            cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "c++", "-c", s, "-o", o]
        else:
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fhip-fp32-correctly-rounded-divide-sqrt",
                   "-x", "hip", "-c", s, "-o", o] + os.environ.get("SGZ_EXTRA_HIPCC_FLAGS", "").split()
            if s.endswith(("spectrum_fft.hip", "spectrum_real.hip", "spectrum_real16.hip", "resonator.hip")):
                cmd.append("-fno-slp-vectorize")   # packed-f32 SLP adds v_mov shuffles around the butterflies (measured -2.5 %) and beside the MFMAs
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    subprocess.check_call(cmd)
    build_driver()
    return LIB



def build_driver() -> str:
    """tools/rt_driver.cpp: bench.py's frame loop for the real-time handles in C++ (plain host code against include/sgz.h; not part of
    the product)"""
    if os.path.exists(DRIVER_SRC):
        try:
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", DRIVER, DRIVER_SRC, "-L" + HERE, "-l:libsgz.so",
                                   "-Wl,-rpath,$ORIGIN"])
        except (OSError, subprocess.CalledProcessError) as e:      # (a bench convenience: bench.py falls back to its Python loop)
            sys.stderr.write(f"librtdriver.so not built: {e}\n")
            if os.path.exists(DRIVER):
                os.remove(DRIVER)
    return DRIVER


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
