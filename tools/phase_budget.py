#!/usr/bin/env python3
"""Static instruction budget of the channel-split K_A per PHASE: the -DSGZ_DEBUG build of spectrum_real.hip carries a clock read (s_memtime)
at every phase boundary (RCLK / SGZ_MAPCLK, the markers tools/phase_clocks.py reads at run time); this tool disassembles that build, cuts the
kernel's instruction stream at the markers and counts, per phase, vector / scalar / LDS / memory / control instructions and the vector
issue clocks (the cost model of tools/isa_histogram.py: tools/ubench/valu2.hip).  The kernel is straight-line code up to the pixel map's
cold paths (literal replay of runs below 2^-62, rounds beyond 2 x T pixels per side), which are listed apart (instructions inside loops).
    tools/mkvariant.sh dbg spectrum_real.hip -DSGZ_DEBUG && python tools/phase_budget.py tools/ab/lib_dbg.so ['stftRealKernel<4, true, 0>']
Run-time counterpart: profiles/*/sq_counters.txt (per launch / 696 workgroups / 8 waves)."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codeobj_report as cr

lib = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "stftRealKernel<4, true, 0>"
PHASES = ["prologue (work-list index, stagger, priorities, addresses)", "sample + table requests, window", "pass 1 (radix R1 per column) + twiddles",
          "exchange 1", "pass 2 (radix 32) + table twiddles", "exchange 2", "pass 3 (radix 32)", "column 0, mirror exchange, recombination, |X|",
          "|X| -> LDS, pads", "low bins / test hook", "map: weights, chunk read, segmented scan, tile stores", "map: interpolated taps",
          "map: barrier", "map: arg-max pixels (+ cold paths), stores", "epilogue"]


def valu_cost(op, operands):
    tail = operands.split(",", 1)[1] if "," in operands else ""
    sgpr = bool(re.search(r"(^|[ ,])s\d+|s\[\d+|vcc|exec", tail))
    if op.startswith("v_pk_"):
        return 4.45
    if re.match(r"v_(sqrt|rcp|rsq|sin|cos|log|exp)", op):
        return 8.2
    if re.match(r"v_(cndmask|cmp|max3|min3|add3|bfe|lshl_add|lshl_or|mad_u|mad_i|mul_lo|mul_hi|cvt|readlane|readfirstlane|lshlrev|lshrrev|and_or|"
                r"or3|xad|perm|addc|add_co|subb|sub_co|alignbit)", op) or "dpp" in operands:
        return 4.2
    return 4.1 if sgpr else 2.3


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache")):
        return "smem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_branch", "s_nop", "s_sleep", "s_setprio", "s_endpgm", "s_sendmsg", "s_code_end")):
        return "ctrl"
    if op.startswith("s_"):
        return "salu"
    return "other"


body = None
for elf in cr.code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        dis = subprocess.run([f"{cr.LLVM}/llvm-objdump", "-d", "--demangle", f.name], capture_output=True, text=True).stdout
    m = re.search(r"^[0-9a-f]+ <(void )?sgz::" + re.escape(want) + r"[^>]*>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M)
    if m:
        body = m.group(2)
        break
if body is None:
    raise SystemExit(f"{want} not found in {lib}")
ins = []
for line in body.splitlines():
    mm = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
    if mm:
        ins.append((mm.group(1), mm.group(2), int(mm.group(3), 16)))
addr = [a for _, _, a in ins]
# instructions inside a loop: between the target of a backward branch and the branch
in_loop = [False] * len(ins)
for i, (op, operands, a) in enumerate(ins):
    if op.startswith(("s_cbranch", "s_branch")):
        mm = re.match(r"(-?\d+)", operands.strip())
        if mm:
            tgt = a + 4 + 4 * int(mm.group(1)) if int(mm.group(1)) < 32768 else a + 4 + 4 * (int(mm.group(1)) - 65536)
            if tgt <= a and a - tgt <= 1600:           # (short backward branches: loops; the long ones return from the markers' out-of-line blocks)
                for j in range(len(ins)):
                    if tgt <= addr[j] <= a:
                        in_loop[j] = True
marks = [i for i, (op, _, _) in enumerate(ins) if op == "s_memtime"]
cuts = [0] + marks + [len(ins)]
rows = []
tot = dict(valu=0, pk=0, clocks=0.0, salu=0, lds=0, vmem=0, smem=0, ctrl=0, cold=0)
for p in range(len(cuts) - 1):
    r = dict(valu=0, pk=0, clocks=0.0, salu=0, lds=0, vmem=0, smem=0, ctrl=0, cold=0)
    for i in range(cuts[p], cuts[p + 1]):
        op, operands, _ = ins[i]
        if in_loop[i]:
            r["cold"] += 1
            continue
        k = kind(op)
        if k == "valu":
            r["valu"] += 1
            r["pk"] += op.startswith("v_pk_")
            r["clocks"] += valu_cost(op, operands)
        elif k in r:
            r[k] += 1
    rows.append(r)
    for k in tot:
        tot[k] += r[k]
print(f"{want}  ({lib}; {len(ins)} instructions, {len(marks)} phase markers; per WAVE -- a workgroup is 8 waves)")
print(f"{'phase':78s} {'VALU':>5s} {'(pk)':>5s} {'issue clk':>9s} {'SALU':>5s} {'LDS':>4s} {'VMEM':>4s} {'SMEM':>4s} {'ctrl':>4s} {'in loops':>8s}")
for p, r in enumerate(rows):
    name = PHASES[p] if len(rows) == len(PHASES) else f"phase {p}"
    print(f"{name:78s} {r['valu']:5d} {r['pk']:5d} {r['clocks']:9.0f} {r['salu']:5d} {r['lds']:4d} {r['vmem']:4d} {r['smem']:4d} {r['ctrl']:4d} {r['cold']:8d}")
print(f"{'total (straight-line part)':78s} {tot['valu']:5d} {tot['pk']:5d} {tot['clocks']:9.0f} {tot['salu']:5d} {tot['lds']:4d} {tot['vmem']:4d} {tot['smem']:4d} {tot['ctrl']:4d} {tot['cold']:8d}")
