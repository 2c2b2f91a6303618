#!/usr/bin/env python3
"""What ANY speed-up of farthest-point sampling could buy the bench step (timing only, results are wrong on purpose): the same
pipelined steps with ops.fps replaced by a copy of one pre-computed (indices, centres) pair -- the step without the FPS kernel's
2.1 ms on 32 CUs.  python tools/fps_upper_bound.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from cmdiad_amd import ops  # noqa: E402
from cmdiad_amd.predictor import BatchPredictor  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
st = bench.build_state(dev)
batches = [(r.to(dev), p.to(dev)) for r, p in bench.make_batches(0, "dino_pointmae")]
real_fps = ops.fps


def run(tag):
    pred = BatchPredictor(st["engine"], st["bank_xyz"], st["bank_second"], st["stats"], st["det"], st["seg"], batch=bench.BATCH,
                          n_max=bench.N_POINTS)
    pending = []
    def go(n):
        for i in range(n):
            if len(pending) >= 3:
                pending.pop(0).wait()
            pending.append(pred.submit(*batches[i % 4]))
        while pending:
            pending.pop(0).wait()
    go(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{tag}: {dt * 1e3:.3f} ms per step = {bench.BATCH / dt:.1f} images/s", flush=True)


for rep in range(2):
    ops.fps = real_fps
    run("with FPS   ")
    cache = {}

    def fake_fps(xyz, G, n_valid=None):
        key = (xyz.shape, G)
        if key not in cache:
            cache[key] = real_fps(xyz, G, n_valid)
            torch.cuda.synchronize()
        idx, cen = cache[key]
        return idx.clone(), cen.clone()      # two small copies in the FPS kernel's place
    ops.fps = fake_fps
    run("without FPS")
ops.fps = real_fps
