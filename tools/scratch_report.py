#!/usr/bin/env python3
"""Per-kernel register / scratch figures of a built library, read from the code objects inside it (no compilation, no GPU):
    python tools/scratch_report.py [cmdiad_amd/libcmdiad_hip.so]
The .hip_fatbin section holds one clang offload bundle per translation unit; every gfx950 entry is an ELF whose AMDGPU metadata
note lists .vgpr_count / .private_segment_fixed_size (scratch bytes per lane) / spill counts per kernel."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def kernels(lib):
    out = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        cur = None
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(name|private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_spill_count|agpr_count):\s*(\S+)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == "name":
                cur = out.setdefault(v, {}) if v.startswith("_Z") or "kernel" in v else None
            elif cur is not None:
                cur[k] = int(v)
    return {k: v for k, v in out.items() if "vgpr_count" in v}


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool:
        return {n: n for n in names}
    r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, r))


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cmdiad_amd", "libcmdiad_hip.so")
    ks = kernels(lib)
    dm = demangle(list(ks))
    print(f"{len(ks)} kernels in {lib}")
    for name, v in sorted(ks.items(), key=lambda kv: -kv[1].get("private_segment_fixed_size", 0)):
        if v.get("private_segment_fixed_size", 0) or v.get("vgpr_spill_count", 0):
            short = re.sub(r"\(anonymous namespace\)::", "", dm[name]).split("(")[0]
            print(f"  scratch {v.get('private_segment_fixed_size', 0):5d} B/lane  vgpr {v['vgpr_count']:3d}  vgpr spills {v.get('vgpr_spill_count', 0):3d}  {short[:110]}")
