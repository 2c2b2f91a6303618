#!/usr/bin/env python3
"""Static check of the MFMA main loops in the compiled gfx950 ISA (no GPU needed).

The GEMM-family kernels stage the next K-step with LDS-DMA (`global_load_lds`) while the current one is
multiplied.  That overlap silently disappears if the compiler puts an `s_waitcnt vmcnt(0)` between the
DMA issue and the fragment `ds_read`s of the same iteration (seen when ordinary global loads were
speculated into the loop).  This script compiles a .hip file to assembly and reports, for every basic block
that holds >= 16 MFMAs, the waitcnts that precede its fragment reads.  Exit code 1 if any such block
waits on vmcnt(0) before reading.

    python tools/isa_lint.py cmdiad_amd/csrc/gemm.hip [cmdiad_amd/csrc/l2min.hip ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_to_asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    # -DCMDIAD_AB_VARIANTS: the test-only formulations are linted too (they are the A/B references of the production loops)
    # per-file flags as csrc/Makefile has them (NOSLP := l2min.o)
    extra = ["-fno-slp-vectorize"] if os.path.basename(src) == "l2min.hip" else []
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DCMDIAD_AB_VARIANTS", *extra, "-I", os.path.join(ROOT, "include"),
           "-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def hot_blocks(asm_path):
    """yield (kernel, label, [events]) for blocks with >= 16 MFMAs; events before the first MFMA only."""
    kernel, label, events, n_mfma = None, None, [], 0
    with open(asm_path) as f:
        for line in f:
            if re.match(r"^_Z\w+:", line):
                kernel, label, events, n_mfma = line.split(":")[0], None, [], 0
                continue
            m = re.match(r"^(\.LBB\w+):", line)
            if m:
                if n_mfma >= 16:
                    yield kernel, label, events
                label, events, n_mfma = m.group(1), [], 0
                continue
            t = line.split()
            if not t or not line.startswith("\t"):
                continue
            op = t[0]
            if op.startswith("scratch_") and "SCRATCH" not in events:
                events.append("SCRATCH")
            if op.startswith("v_mfma"):
                n_mfma += 1
            elif n_mfma == 0 and op.startswith("global_load_lds") and "READ" not in events:
                # only waits BETWEEN the DMA issue and the fragment reads matter (a fully unrolled kernel carries the previous
                # step's end-of-step wait in the same basic block)
                events = [e for e in events if e == "SCRATCH"]
            elif n_mfma == 0:
                if op == "s_waitcnt":
                    events.append(" ".join(t[1:]))
                elif op.startswith("ds_read") and "READ" not in events:
                    events.append("READ")
            if op == "s_endpgm" and n_mfma >= 16:
                yield kernel, label, events
                n_mfma = 0


def wide_kernel_violations(asm_path):
    """Kernels built on gemm_wide.h keep their accumulators in AGPRs by convention (named in asm text): the compiler
    must not generate AGPR writes of its own anywhere in them (VGPR spills must go to scratch, not to AGPRs)."""
    bad, kernel = [], None
    with open(asm_path) as f:
        for line in f:
            if re.match(r"^_Z\w+:", line):
                kernel = line.split(":")[0] if "_wide_" in line else None
                continue
            if kernel is None or not line.startswith("\t"):
                continue
            op = line.split()[0] if line.split() else ""
            if op.startswith("v_accvgpr_write") or op.startswith("v_accvgpr_mov"):
                bad.append((kernel, "-", [line.strip()]))
            if op == "s_endpgm":
                kernel = None
    return bad


def lint(src):
    bad = []
    asm = compile_to_asm(src)
    try:
        for kernel, label, events in hot_blocks(asm):
            before_read = events[: events.index("READ")] if "READ" in events else events
            if any(re.search(r"vmcnt\(0\)", e) for e in before_read) or "SCRATCH" in events:
                bad.append((kernel, label, events))
        bad += wide_kernel_violations(asm)
    finally:
        os.unlink(asm)
    return bad


def main():
    srcs = sys.argv[1:] or [os.path.join(ROOT, "cmdiad_amd", "csrc", f) for f in ("gemm.hip", "gemm_sk.hip", "l2min.hip", "conv.hip", "encoder_tail.hip")]
    rc = 0
    for src in srcs:
        bad = lint(src)
        for kernel, label, events in bad:
            print(f"{os.path.basename(src)}: {kernel} {label}: {events}")
            rc = 1
        if not bad:
            print(f"{os.path.basename(src)}: main loops clean")
    return rc


if __name__ == "__main__":
    sys.exit(main())
