"""Worker of tests/test_gpu_world2.py: ONE of two real ranks (processes) of a row-sharded node, both on cuda:0, the collectives over
gloo (RCCL refuses two ranks on one device).  Everything a rank of an N > 1 node executes runs here with N = 2 REAL processes --
engine.Bank row shards, engine.ShardedSearch (compact, counts exchange, all-gather of the live rows, the segments launch over both
ranks' queries, all_reduce(MIN), expansion; sticky cap, overflow flag, regrow), the sharded re-weighting step with its four collectives
(Bank(replicate_f32=False)), the row-sharded coreset rounds, and BatchPredictor with the row-sharded search inside the pipeline -- and
every result is compared with the single-library computation of the same rank (no collective), bit for bit.
Launched as: python -m torch.distributed.run --nproc-per-node 2 tests/world2_worker.py"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as td

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("CMDIAD_ALLOW_RANDOM_INIT", "1")

from cmdiad_amd import coreset, engine as eng, ops  # noqa: E402

DEV = torch.device("cuda", 0)


def main():
    td.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    assert world == 2
    torch.cuda.set_device(0)
    group = td.group.WORLD
    done = []

    # ---- 1. ShardedSearch over two real ranks: every rank brings its OWN queries
    Q, D, Nb = 6000, 768, 7001
    lib = torch.randn(Nb, D, generator=torch.Generator().manual_seed(5)).to(DEV)            # the same library on both ranks
    whole, shard = eng.Bank(lib, 0, 1), eng.Bank(lib, rank, world)
    assert shard.row_offset == (0 if rank == 0 else 3584) and shard.shard_rows in (3584, 3417)

    def batch(bg_share, seed):
        gq = torch.Generator().manual_seed(seed + 1000 * rank)
        q = torch.randn(Q, D, generator=gq)
        q[torch.rand(Q, generator=gq) < bg_share] = -0.3
        q[:50] = lib[rank * 50:(rank + 1) * 50].cpu()                                      # exact matches, owned by either shard
        return ops.normalize_cast(q.to(DEV))

    def single(q16, qsq):
        return ops.l2_min_keys(q16, qsq, whole.bf16, whole.sqnorm, ops.new_keys(Q, DEV, runner=True))     # best + runner-up planes

    ss = eng.ShardedSearch(shard, group, cap_rows="auto", slack=0.02)
    for i in range(3):
        q16, _, qsq = batch(0.55 + 0.1 * rank, 100 + i)                                     # ragged live counts between the ranks
        keys = ss.gather(q16, qsq).gemm().reduce()
        assert torch.equal(keys, single(q16, qsq)) and not ss.overflowed(), f"rank {rank}: sharded keys differ (step {i})"
    assert ss.host_reads == 1
    q16, _, qsq = batch(0.05, 200)                                                          # more live rows than the sticky cap
    bad = ss.gather(q16, qsq).gemm().reduce()
    assert ss.overflowed()                                                                  # (the same flag on both ranks)
    ss.regrow()
    keys = ss.gather(q16, qsq).gemm().reduce()
    assert torch.equal(keys, single(q16, qsq)) and not ss.overflowed()
    keys_exact, _ = eng.sharded_min_keys(q16, qsq, shard, group)
    assert torch.equal(keys_exact, keys)
    done.append("sharded_search")

    # ---- 2. the sharded re-weighting step (fp32 rows sharded too): four collectives per scored batch.  SURVEY 8(e): the queries of
    # this mode are REPLICATED -- every rank scores the same batch, the owner of a row contributes its part
    B, Qp = 3, 784
    patch = torch.randn(B, Qp, D, generator=torch.Generator().manual_seed(77)).to(DEV)
    patch[0, :20] = lib[3570:3590]                                                           # winners on both sides of the shard boundary
    ref = eng.score_patches(patch, whole, (28, 28))
    got = eng.score_patches(patch, eng.Bank(lib, rank, world, replicate_f32=False), (28, 28), group=group)
    for k in ("min_val", "min_idx", "s_idx", "s_star", "s", "s_map_pre", "top3", "knn_d"):
        assert torch.equal(got[k], ref[k]), f"rank {rank}: sharded scoring differs in {k}"
    done.append("sharded_fp32_scoring")

    # ---- 3. row-sharded coreset rounds: one all_reduce(MAX) of 8 bytes per round
    z = torch.randn(9001, 200, generator=torch.Generator().manual_seed(3)).to(DEV)
    z[8000] = z[40]                                                                           # a duplicate in the other rank's rows
    assert torch.equal(coreset.greedy_coreset_sharded(z, 150, group).cpu(), coreset.greedy_coreset(z, 150).cpu())
    done.append("sharded_coreset")

    # ---- 4. BatchPredictor with the row-sharded search inside the pipeline: each rank scores its own batches
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from cmdiad_amd.predictor import BatchPredictor
    from cmdiad_amd.synth import synth_cloud, synth_rgb
    st = bench.build_state(DEV)
    Bp, n_max = 4, 34000
    bx, br = eng.Bank(st["bank_xyz"].f32, rank, world), eng.Bank(st["bank_second"].f32, rank, world)
    mk = lambda g, x, r: BatchPredictor(st["engine"], x, r, st["stats"], st["det"], st["seg"], batch=Bp, n_max=n_max, group=g)   # noqa: E731
    ref_p, sh_p = mk(None, st["bank_xyz"], st["bank_second"]), mk(group, bx, br)

    def pbatch(seed, frac):
        return (torch.cat([synth_rgb(seed + i) for i in range(Bp)]).to(DEV), torch.cat([synth_cloud(seed + i, frac) for i in range(Bp)]).to(DEV))

    order = [pbatch(10 + 100 * rank, 0.36), pbatch(30 + 100 * rank, 0.40), pbatch(50 + 100 * rank, 0.62), pbatch(10 + 100 * rank, 0.36)]
    want = [ref_p.predict_batch(*b) for b in order]
    pending, got = [], []
    for b in order:
        if len(pending) == 2:
            got.append(pending.pop(0).wait())
        pending.append(sh_p.submit(*b))
    got += [t.wait() for t in pending]
    bad = [(i, float(np.abs(gs - ws).max()), float(np.abs(gm - wm).max()), float(np.abs(wm).max()))
           for i, ((gs, gm), (ws, wm)) in enumerate(zip(got, want)) if not (np.array_equal(gs, ws) and np.array_equal(gm, wm))]
    if bad and os.environ.get("WORLD2_DIAG"):
        print(f"[diag] rank {rank}: graph pipeline: batches (index, max |d score|, max |d map|, max |map|) {bad}; redone {sh_p.redone}", file=sys.stderr, flush=True)
        for tag, env_post, graph in (("eager", "1", False), ("graph, searches in line", "0", True)):
            os.environ["CMDIAD_SEARCH_POST"] = env_post
            p2 = BatchPredictor(st["engine"], bx, br, st["stats"], st["det"], st["seg"], batch=Bp, n_max=n_max, group=group, use_graph=graph)
            g2 = [p2.predict_batch(*b) for b in order]
            b2 = [(i, float(np.abs(gs - ws).max()), float(np.abs(gm - wm).max())) for i, ((gs, gm), (ws, wm)) in enumerate(zip(g2, want))
                  if not (np.array_equal(gs, ws) and np.array_equal(gm, wm))]
            print(f"[diag] rank {rank}: {tag}: differing batches {b2}; redone {p2.redone}", file=sys.stderr, flush=True)
        os.environ.pop("CMDIAD_SEARCH_POST", None)
    assert not bad, f"rank {rank}: the row-sharded pipeline's outputs differ: {bad}"
    done.append(f"sharded_pipeline(redone={sh_p.redone})")

    # ---- 5. the same pipeline over libraries whose fp32 rows are sharded too: the ranks score the SAME batches (replicated queries);
    # ranks that submit different batches are refused on every rank
    fx, fr = (eng.Bank(st["bank_xyz"].f32, rank, world, replicate_f32=False), eng.Bank(st["bank_second"].f32, rank, world, replicate_f32=False))
    f_p = mk(group, fx, fr)
    same = [pbatch(10, 0.36), pbatch(50, 0.62), pbatch(30, 0.40)]
    want_same = [ref_p.predict_batch(*b) for b in same]
    for (gs, gm), (ws, wm) in zip([f_p.predict_batch(*b) for b in same], want_same):
        assert np.array_equal(gs, ws) and np.array_equal(gm, wm), f"rank {rank}: sharded-fp32 pipeline differs from the unsharded one"
    try:
        f_p.predict_batch(*pbatch(10 + 100 * rank, 0.36))
        raise AssertionError("different batches on the two ranks were accepted by a predictor over sharded fp32 rows")
    except ValueError as exc:
        assert "REPLICATED" in str(exc)
    done.append(f"sharded_fp32_pipeline(redone={f_p.redone})")

    ok = torch.tensor([1])
    td.all_reduce(ok, op=td.ReduceOp.MIN)
    td.barrier()
    td.destroy_process_group()
    if rank == 0:
        print(json.dumps({"world2": True, "ranks": world, "checked": done}), flush=True)


if __name__ == "__main__":
    main()
