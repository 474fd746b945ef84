#!/usr/bin/env python3
"""Opcode histogram of one kernel of the built library (static count, weighted with the issue costs tools/ubench/valu2.hip measured on
gfx950): where the VALU cycles of a straight-line kernel go.    python tools/isa_histogram.py 'stftRealKernel<4, true, 0>' [--dump]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import codeobj_report as cr

want = sys.argv[1]
lib = os.path.join(cr.ROOT, "signalizer_amd", "libsgz.so")
for elf in cr.code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        dis = subprocess.run([f"{cr.LLVM}/llvm-objdump", "-d", "--demangle", f.name], capture_output=True, text=True).stdout
    m = re.search(r"^[0-9a-f]+ <(void )?sgz::" + re.escape(want) + r"[^>]*>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M)
    if not m:
        continue
    body = m.group(2)
    if "--dump" in sys.argv:
        print(body)
        sys.exit(0)
    def cost_of(op, operands):
        """issue clocks per wave-instruction per SIMD (tools/ubench/valu2.hip, valu3.hip): VGPR / literal / inline-constant sources run at full
        rate, an SGPR (or vcc / exec) source halves it; packed, 64-bit, conversion, lane and select / compare / bit-field opcodes are half rate"""
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

    ops = collections.Counter()
    cost = collections.Counter()
    phase = 0
    per_phase = collections.defaultdict(lambda: [0, 0.0, 0, 0])          # VALU count, VALU clocks, LDS, VMEM between two s_barrier lines
    for line in body.splitlines():
        t = line.strip().split()
        if not t or not re.match(r"^[a-z_0-9]+$", t[0]):
            continue
        op = t[0]
        operands = line.split("//")[0].strip()[len(op):]
        ops[op] += 1
        if op == "s_barrier":
            phase += 1
        elif op.startswith("v_"):
            c = cost_of(op, operands)
            cost[op] += c
            per_phase[phase][0] += 1
            per_phase[phase][1] += c
        elif op.startswith("ds_"):
            per_phase[phase][2] += 1
        elif op.startswith("global_") or op.startswith("buffer_"):
            per_phase[phase][3] += 1
    tot = sum(cost.values())
    print(f"{want}: {sum(ops.values())} instructions, VALU {sum(v for k, v in ops.items() if k.startswith('v_'))}, "
          f"weighted VALU clocks per wave (static: loops once, every branch) {tot:.0f}")
    for k, v in cost.most_common(30):
        print(f"  {k:40s} {v:9.0f} clk  {100 * v / tot:5.1f} %")
    print("between barriers, in listing order (VALU instr, VALU clk, LDS instr, VMEM instr):")
    for ph in sorted(per_phase):
        n, c, l, m = per_phase[ph]
        if n + l + m >= 8:
            print(f"  segment {ph:2d}: {n:5d} {c:8.0f} {l:5d} {m:4d}")
    print("non-VALU:", {k: v for k, v in ops.most_common() if not k.startswith("v_") and v >= 8})
    break
else:
    print("kernel not found")
