#!/usr/bin/env python3
"""Every s_barrier of every kernel in the built library: is an LDS write (or atomic) still in flight when the wave reaches it?

Round 6 found one (K_A's exchange 1: inline-asm ds_write_b128 the compiler's wait-count pass does not see, behind a barrier whose
fence was for the local address space only): a wave could pass the barrier with its stores still queued, and on a busy device another
wave read the old contents in 1 of 1 000 launches.  No test on an idle device can see that class, the listing can.  The walk is linear
in listing order (loops once, every branch): the lgkm counter is modelled as a queue of the LDS / scalar-memory instructions issued since
the last wait, `s_waitcnt lgkmcnt(N)` keeps its N newest.  A barrier reached with an LDS write or atomic in the queue is reported; LDS
reads in flight at a barrier are counted only (their register results are waited for at the use; the LDS pipeline is in order per CU).

    python tools/barrier_audit.py [library]          # exit code 1 when a store is in flight at a barrier
tests/test_host_codeobj.py runs it on the built library."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codeobj_report as cr


def audit(lib=None):
    """[(kernel, barrier index, [instructions in flight])] for barriers passed with LDS stores queued, and the per-library totals"""
    lib = lib or os.path.join(cr.ROOT, "signalizer_amd", "libsgz.so")
    bad, totals = [], {"kernels": 0, "barriers": 0, "reads_in_flight": 0}
    for elf in cr.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            dis = subprocess.run([f"{cr.LLVM}/llvm-objdump", "-d", "--demangle", f.name], capture_output=True, text=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <([^\n]*)>:\n(.*?)(?=^[0-9a-f]+ <[^\n]*>:\n|\Z)", dis, re.S | re.M):
            name, body = m.group(1), m.group(2)
            if name.startswith("L") or "s_endpgm" not in body and "s_barrier" not in body:
                pass
            queue, nbar = [], 0
            seen_barrier = False
            for line in body.splitlines():
                t = line.split("//")[0].strip()
                if not t or t.endswith(":"):
                    continue
                op = t.split()[0]
                if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store"):
                    queue.append(t)
                elif op == "s_waitcnt":
                    w = re.search(r"lgkmcnt\((\d+)\)", t)
                    if w:
                        keep = int(w.group(1))
                        queue = queue[len(queue) - keep:] if keep else []
                elif op == "s_barrier":
                    seen_barrier = True
                    nbar += 1
                    totals["barriers"] += 1
                    stores = [q for q in queue if q.startswith("ds_") and not re.match(r"ds_(read|bpermute|permute|swizzle|consume|append)", q)]
                    reads = [q for q in queue if q.startswith("ds_read")]
                    totals["reads_in_flight"] += bool(reads)
                    if stores:
                        bad.append((name, nbar, stores))
            totals["kernels"] += seen_barrier
    return bad, totals


if __name__ == "__main__":
    bad, totals = audit(sys.argv[1] if len(sys.argv) > 1 else None)
    print(f"{totals['kernels']} kernels with barriers, {totals['barriers']} barriers; LDS reads in flight at {totals['reads_in_flight']} of them; "
          f"LDS stores / atomics in flight at {len(bad)}")
    for name, n, stores in bad:
        print(f"  {name[:110]}: barrier {n}: {stores[-1]}" + (f" (+{len(stores) - 1})" if len(stores) > 1 else ""))
    sys.exit(1 if bad else 0)
