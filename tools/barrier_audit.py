#!/usr/bin/env python3
"""Every s_barrier of every kernel in the built library: is an LDS write (or atomic) still in flight when the wave reaches it?

Round 6 found one (K_A's exchange 1: inline-asm ds_write_b128 the compiler's wait-count pass does not see, behind a barrier whose
fence was for the local address space only): a wave could pass the barrier with its stores still queued, and on a busy device another
wave read the old contents in 1 of 1 000 launches.  No test on an idle device can see that class, the listing can.  The walk follows the
control-flow graph (branch targets from the disassembly; back edges until nothing changes): the lgkm counter is modelled as a queue of
the LDS / scalar-memory instructions issued since the last wait, `s_waitcnt lgkmcnt(N)` keeps its N newest; where paths join, the worse
queue wins.  A barrier that can be reached with an LDS write or atomic in the queue is reported.

    python tools/barrier_audit.py [library]          # exit code 1 when a store is in flight at a barrier
tests/test_host_codeobj.py runs it on the built library."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codeobj_report as cr


STORE = re.compile(r"ds_(?!read|bpermute|permute|swizzle|consume|append|nop)")


def _worse(a, b):
    """is pending-queue a worse (more LDS stores in flight, then longer) than b"""
    sa, sb = sum(1 for q in a if STORE.match(q)), sum(1 for q in b if STORE.match(q))
    return (sa, len(a)) > (sb, len(b))


def audit_kernel(body):
    """[(barrier ordinal in listing order, stores in flight)] of one kernel's listing.  Data flow over the control-flow graph: the state
    is the queue of LDS / scalar-memory instructions issued since the last wait (`s_waitcnt lgkmcnt(N)` keeps the N newest); where paths
    join, the worse state wins; back edges are followed until nothing gets worse."""
    ins = []                                               # (address, opcode, text, branch target address or None)
    for line in body.splitlines():
        code, _, tail = line.partition("//")
        t = code.strip()
        m = re.match(r"\s*([0-9A-Fa-f]+):", tail)
        if not t or not m or not re.match(r"^[a-z_0-9]+", t):
            continue
        op = t.split()[0]
        target = None
        if op.startswith("s_cbranch") or op == "s_branch":
            off = re.search(r"\+0x([0-9a-fA-F]+)>\s*$", tail)
            target = ("rel", int(off.group(1), 16) if off else 0)
        ins.append([int(m.group(1), 16), op, t, target])
    if not ins:
        return [], 0
    base = ins[0][0]
    at = {a: i for i, (a, _, _, _) in enumerate(ins)}
    worst = [None] * len(ins)                             # worst state seen on entry to instruction i
    work = [(0, ())]
    hits = {}
    while work:
        i, state = work.pop()
        while i < len(ins):
            if worst[i] is not None and not _worse(state, worst[i]):
                break
            worst[i] = state
            a, op, t, target = ins[i]
            if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store"):
                state = (state + (t,))[-64:]
            elif op == "s_waitcnt":
                w = re.search(r"lgkmcnt\((\d+)\)", t)
                if w:
                    keep = int(w.group(1))
                    state = state[len(state) - keep:] if keep else ()
            elif op == "s_barrier":
                stores = [q for q in state if STORE.match(q)]
                if stores and len(stores) > len(hits.get(i, [])):
                    hits[i] = stores
            elif op == "s_endpgm":
                break
            if target is not None:
                j = at.get(base + target[1])
                if j is not None:
                    work.append((j, state))
                if op == "s_branch":
                    break
            i += 1
    order = [i for i, x in enumerate(ins) if x[1] == "s_barrier"]
    return [(order.index(i) + 1, hits[i]) for i in sorted(hits)], len(order)


def audit(lib=None):
    """[(kernel, barrier index, [instructions in flight])] for barriers that can be reached with LDS stores queued, and the library's totals"""
    lib = lib or os.path.join(cr.ROOT, "signalizer_amd", "libsgz.so")
    bad, totals = [], {"kernels": 0, "barriers": 0, "reads_in_flight": 0}
    for elf in cr.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            dis = subprocess.run([f"{cr.LLVM}/llvm-objdump", "-d", "--demangle", f.name], capture_output=True, text=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <([^\n]*)>:\n(.*?)(?=^[0-9a-f]+ <[^\n]*>:\n|\Z)", dis, re.S | re.M):
            name, body = m.group(1), m.group(2)
            found, nbar = audit_kernel(body)
            totals["kernels"] += bool(nbar)
            totals["barriers"] += nbar
            bad += [(name, n, stores) for n, stores in found]
    return bad, totals


if __name__ == "__main__":
    bad, totals = audit(sys.argv[1] if len(sys.argv) > 1 else None)
    print(f"{totals['kernels']} kernels with barriers, {totals['barriers']} barriers; LDS stores / atomics can be in flight at {len(bad)}")
    for name, n, stores in bad:
        print(f"  {name[:110]}: barrier {n}: {stores[-1]}" + (f" (+{len(stores) - 1})" if len(stores) > 1 else ""))
    sys.exit(1 if bad else 0)
