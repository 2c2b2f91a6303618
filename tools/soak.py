#!/usr/bin/env python3
"""Soak of the bench's pipelined step: thousands of steps over the four rotating batches, every step's outputs compared bit for bit
with the first outputs of its batch (bench.run_steps), device and host memory sampled along the way.  A race between the three
streams / two buffer sets, a leak of events, graphs or pinned buffers, or a drifting result shows up here and nowhere else.
    python tools/soak.py [steps] [workload]"""
import importlib.util
import os
import resource
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("CMDIAD_ALLOW_RANDOM_INIT", "1")
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from cmdiad_amd.predictor import BatchPredictor  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
workload = sys.argv[2] if len(sys.argv) > 2 else "dino_pointmae"
dev = torch.device("cuda", 0)
st = bench.build_state(dev, workload)
pred = BatchPredictor(st["engine"], st["bank_xyz"], st["bank_second"], st["stats"], st["det"], st["seg"], batch=bench.BATCH,
                      n_max=bench.N_POINTS, workload=workload, halluc=st["halluc"])
host_fed = len(sys.argv) > 3 and sys.argv[3] == "host"      # pinned host batches: H2D inside the loop, staged two submits ahead
if host_fed:
    batches = bench.make_batches(0, workload, pinned=True)
else:
    batches = [(r.to(dev) if r is not None else None, p.to(dev)) for r, p in bench.make_batches(0, workload)]
first = bench.run_steps(pred, batches, 8)
torch.cuda.synchronize()
mem0, rss0 = torch.cuda.memory_allocated(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
t0, done = time.time(), 0
while done < steps:
    n = min(500, steps - done)
    bench.run_steps(pred, batches, n, first)
    done += n
    torch.cuda.synchronize()
    print(f"{workload}: {done} steps, {1e3 * (time.time() - t0) / done:.2f} ms/step, device memory {torch.cuda.memory_allocated() - mem0:+d} B, "
          f"reserved {torch.cuda.memory_reserved() >> 20} MiB, host peak RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - rss0:+d} KiB", flush=True)
assert torch.cuda.memory_allocated() - mem0 < (64 << 20), "device memory grew during the soak"
print("soak ok", workload, "host-fed" if host_fed else "resident", steps, flush=True)
