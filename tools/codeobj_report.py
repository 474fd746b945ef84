#!/usr/bin/env python3
"""Register / spill / LDS figures of every kernel in the built library, read from the gfx950 code objects embedded in libsgz.so
(the .hip_fatbin section: clang offload bundles, one per translation unit; the AMDGPU metadata note of each ELF).
    python tools/codeobj_report.py [--spills]          # all kernels, or only those that spill
tests/test_host_codeobj.py holds the hot kernels to zero spills with it."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """the gfx950 ELF images inside the library's .hip_fatbin section"""
    with tempfile.TemporaryDirectory() as tmp:
        sec = os.path.join(tmp, "fatbin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={sec}", lib, os.path.join(tmp, "copy")], check=True)
        blob = open(sec, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, pos)
            ident = blob[pos + 24:pos + 24 + idlen].decode()
            pos += 24 + idlen
            if "gfx950" in ident and size:
                out.append(blob[base + off:base + off + size])
    return out


def kernels(lib=None):
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, lds, scratch}] over every kernel of the library"""
    lib = lib or os.path.join(ROOT, "signalizer_amd", "libsgz.so")
    rows = []
    for elf in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f.name], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in notes.splitlines():
            s = line.strip()
            if s.startswith("- .agpr_count:") or (s.startswith("- .") and cur is not None and "args" not in s and s.startswith("- .agpr")):
                pass
            m = re.match(r"-?\s*\.(\w+):\s*(.*)$", s)
            if not m:
                continue
            key, val = m.group(1), m.group(2).strip().strip("'\"")
            if key == "agpr_count" and s.startswith("- "):
                cur = {"agpr": int(val)}
                rows.append(cur)
            elif cur is not None:
                if key == "name" and "name" not in cur and val.startswith("_Z"):
                    cur["name"] = val
                elif key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size",
                             "private_segment_fixed_size") and key not in cur:
                    cur[key] = int(val)
    names = [r.get("name", "?") for r in rows]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for r, d in zip(rows, dem):
        r["demangled"] = d
    return rows


if __name__ == "__main__":
    only = "--spills" in sys.argv
    for r in sorted(kernels(), key=lambda r: r["demangled"]):
        if only and not (r.get("vgpr_spill_count") or r.get("sgpr_spill_count")):
            continue
        print(f"{r['demangled'][:96]:96s} vgpr {r.get('vgpr_count', -1):3d} agpr {r['agpr']:3d} spill v{r.get('vgpr_spill_count', 0)} "
              f"s{r.get('sgpr_spill_count', 0)} lds {r.get('group_segment_fixed_size', 0)} scratch {r.get('private_segment_fixed_size', 0)}")
