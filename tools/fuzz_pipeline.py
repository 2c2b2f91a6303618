#!/usr/bin/env python3
"""Randomised EXACT invariants of the batched pipeline (predictor.BatchPredictor), no oracle involved -- what must hold bit for bit
whatever the batch looks like:

  * HIP-graph replay == eager launches; a second submit of the same batch (other buffer set) == the first;
  * the exact removal of repeated query rows in front of the searches (CMDIAD_DEDUP=0 searches every row) changes nothing;
  * a sample's scores do not depend on its neighbours in the batch (the batch reversed gives the reversed outputs; a batch of
    one gives the same outputs as that sample inside a larger batch);
  * the row-sharded search through RCCL with a world of one (ShardedSearch: compaction, sticky cap, overflow + repeat) == the plain one.

Clouds: foreground share 0.4 % ... 100 % of the image (a few hundred points ... no background patch at all), mixed inside one batch;
batches of 1-6; the 'mtfi' workload (hallucinated second modality) one case in four.
    python tools/fuzz_pipeline.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("CMDIAD_ALLOW_RANDOM_INIT", "1")
import importlib.util  # noqa: E402

from cmdiad_amd.predictor import BatchPredictor  # noqa: E402
from cmdiad_amd.synth import synth_cloud, synth_rgb  # noqa: E402

DEV = torch.device("cuda", 0)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + rs.randint(0, 400)), RANK="0", WORLD_SIZE="1")
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)                       # RCCL's banner
    td.init_process_group("nccl", device_id=DEV)
    group = td.group.WORLD
    warm = torch.ones(1, device=DEV)
    td.all_reduce(warm, group=group)       # the communicator is up before anything is captured (as bench.py does)
    torch.cuda.synchronize()
    os.dup2(saved, 1)
    states = {w: bench.build_state(DEV, w) for w in ("dino_pointmae", "mtfi")}
    t0, n, kinds = time.time(), 0, {}
    while time.time() - t0 < budget:
        wl = "mtfi" if rs.rand() < 0.25 else "dino_pointmae"
        st = states[wl]
        B = int(rs.randint(1, 7))
        fr = [float(rs.choice([0.004, 0.01, 0.03, 0.1, 0.3, 0.5, 0.8, 1.0, 1.6])) for _ in range(B)]
        seeds = [int(rs.randint(0, 10 ** 6)) for _ in range(B)]
        pcs = torch.cat([synth_cloud(s, f, texture=float(rs.choice([0.0, 0.004]))) for s, f in zip(seeds, fr)]).to(DEV)
        rgb = torch.cat([synth_rgb(s) for s in seeds]).to(DEV) if wl == "dino_pointmae" else None
        counts = (pcs != 0).all(1).flatten(1).sum(1)
        if int(counts.min()) < 128:
            continue
        n_max = int(counts.max()) + int(rs.choice([0, 1, 700, 20000]))
        mk = lambda **kw: BatchPredictor(st["engine"], st["bank_xyz"], st["bank_second"], st["stats"], st["det"], st["seg"],   # noqa: E731
                                         batch=kw.pop("batch", B), n_max=n_max, workload=wl, halluc=st["halluc"], **kw)
        tag = (wl, B, tuple(fr), n_max)
        graph = mk(use_graph=True)
        a = graph.predict_batch(rgb, pcs)
        assert same(a, graph.predict_batch(rgb, pcs)), ("second submit differs", tag)
        assert graph.use_graph, ("graph capture failed", tag)
        eager = mk(use_graph=False).predict_batch(rgb, pcs)
        assert same(a, eager), ("graph != eager", tag)
        os.environ["CMDIAD_DEDUP"] = "0"
        try:
            every = mk(use_graph=False)
        finally:
            del os.environ["CMDIAD_DEDUP"]
        assert same(a, every.predict_batch(rgb, pcs)), ("dedup changes the outputs", tag)
        rev = mk(use_graph=False).predict_batch(rgb.flip(0) if rgb is not None else None, pcs.flip(0))
        assert same(a, (rev[0][::-1], rev[1][::-1])), ("a sample's scores depend on its place in the batch", tag)
        if B > 1:
            i = int(rs.randint(0, B))
            one = mk(use_graph=False, batch=1).predict_batch(rgb[i:i + 1] if rgb is not None else None, pcs[i:i + 1])
            assert same((a[0][i:i + 1], a[1][i:i + 1]), one), ("a sample alone differs from the sample inside its batch", tag, i)
        sh = mk(use_graph=bool(rs.rand() < 0.5), group=group)
        for _ in range(2):                 # the second submit meets the sticky cap the first one set
            assert same(a, sh.predict_batch(rgb, pcs)), ("row-sharded search (world of one) differs", tag)
        n += 1
        kinds[wl] = kinds.get(wl, 0) + 1
    td.destroy_process_group()
    print("pipeline fuzz ok", n, kinds, flush=True)


if __name__ == "__main__":
    main()
