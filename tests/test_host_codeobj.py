"""The built library's gfx950 code objects: the K_A / K_B kernels keep their working set in registers.  A spill in one of them is a
silent 15-20 % (round 2: a run-time channel-mix branch put 14 registers of stftRealKernel<5, ...> into scratch and cfg5 lost 18 %), so
the figure the compiler wrote into the code object is held here.  No GPU needed: the metadata note is read with llvm-readelf."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HOT = ("stftRealKernel", "stftMapKernel", "stftHalfKernel", "stftComplexKernel", "mapSideKernel", "decayColourFusedKernel",
       "decayLocalCarryKernel", "scopeIngestKernel", "scopeLanczosKernel", "vectorIngestKernel")


def test_hot_kernels_do_not_spill():
    import codeobj_report as cr
    lib = os.path.join(ROOT, "signalizer_amd", "libsgz.so")
    if not (os.path.exists(lib) and os.path.exists(f"{cr.LLVM}/llvm-readelf") and os.path.exists(f"{cr.LLVM}/llvm-objcopy")):
        pytest.skip("library or llvm tools not present")
    rows = cr.kernels(lib)
    hot = [r for r in rows if any(h in r["demangled"] for h in HOT)]
    assert len(hot) >= 60, len(hot)                        # 24 real-input + 18 whole-frame + 8 half + 4 complex + ...
    bad = [(r["demangled"], r.get("vgpr_spill_count"), r.get("private_segment_fixed_size")) for r in hot
           if r.get("vgpr_spill_count", 0) or r.get("private_segment_fixed_size", 0)]
    assert not bad, bad
    # launch bounds: 1024-thread kernels must fit four waves per SIMD (128 registers), or one workgroup no longer fits a CU
    for r in hot:
        if "stftRealKernel<5" in r["demangled"] or "stftMapKernel<5" in r["demangled"]:
            assert r["vgpr_count"] <= 128, r


def test_no_lds_store_in_flight_at_a_barrier():
    """Round 6's K_A race as a listing check (tools/barrier_audit.py): inline-asm LDS stores are invisible to the compiler's wait-count
    insertion, so a barrier behind them needs its own s_waitcnt lgkmcnt(0) -- fft_common.hpp ldsBarrier().  The library before that fix
    has 37 such barriers, all in stftRealKernel; no test on an idle device saw them in 20 000 fuzz cases."""
    import barrier_audit
    import codeobj_report as cr
    lib = os.path.join(ROOT, "signalizer_amd", "libsgz.so")
    if not (os.path.exists(lib) and os.path.exists(f"{cr.LLVM}/llvm-objdump") and os.path.exists(f"{cr.LLVM}/llvm-objcopy")):
        pytest.skip("library or llvm tools not present")
    bad, totals = barrier_audit.audit(lib)
    assert totals["kernels"] >= 80 and totals["barriers"] >= 500, totals             # (the walk found the kernels at all)
    assert not bad, bad[:5]
