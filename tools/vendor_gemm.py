#!/usr/bin/env python3
"""Reference point for the roofline fractions: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on this box for
the shapes of this package's GEMMs, on random and on all-zero operands (the chip trades clock for MFMA density: MI355X_MICROARCH.md).
Not part of the product path or of any test -- a measurement to read the package's own numbers against (profiles/r2_notes.md)."""
import sys, torch
sys.path.insert(0, "tools")
from microbench import timeit

def run(M, N, K, dtype, zeros):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    b = torch.randn(N, K, device="cuda", generator=g).to(dtype)
    if zeros: a.zero_(); b.zero_()
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    ms = timeit(lambda: torch.matmul(a, b.t(), out=out), iters=10, warm=3)
    print(f"torch.matmul {str(dtype)[6:]:8s} {M}x{N}x{K} {'zeros ' if zeros else 'random'}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)

for dt in (torch.float16, torch.bfloat16):
    for z in (False, True):
        run(8192, 8192, 8192, dt, z)
for z in (False, True):
    run(25088, 19129, 768, torch.float16, z)     # rgb distance shape (the library also writes the 0.96 GB of products)
    run(16384, 16384, 768, torch.float16, z)     # same K, square
for (N, K) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    run(25120, N, K, torch.bfloat16, False)      # ViT-B/8 products at B = 32, no epilogue
