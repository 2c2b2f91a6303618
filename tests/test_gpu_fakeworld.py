"""GPU: BASELINE configs[3] -- the patch-library search with the library's rows sharded W ways -- at its real shard shapes on ONE
device ("fake world"): the W shards are searched in turn through exactly what one rank of a W-rank node
executes (engine.Bank shards + cmdiad_l2_min_keys_segments over the W gathered, separately compacted query sets), the integer MIN
over the shards stands in for the all_reduce(MIN), and the result must equal the single-library keys bit for bit
(features.py:186-190,227 on one device).  Plus the segments launch against its definition on small ragged shapes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cmdiad_amd import engine as eng  # noqa: E402
from cmdiad_amd import ops  # noqa: E402

DEV = "cuda"


def _segments_case(counts, stride, Nb, D, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    W = len(counts)
    bank = torch.randn(Nb, D, generator=g).to(DEV)
    q = torch.randn(W * stride, D, generator=g).to(DEV)
    k = min(64, W * stride, Nb)
    q[:k] = bank[:k]                          # self-matches: exact zeros
    b16, _, bsq = ops.normalize_cast(bank, dtype=dtype)
    q16, _, qsq = ops.normalize_cast(q, dtype=dtype)
    cnt = torch.tensor(counts, dtype=torch.int32, device=DEV)
    # single plane at an arbitrary row offset; best + runner-up planes at a multiple of 64 (the runner-up's groups of 16 rows)
    got = ops.l2_min_keys_segments(q16, qsq, cnt, stride, b16, bsq, ops.new_keys(W * stride, DEV), 7)
    got2 = ops.l2_min_keys_segments(q16, qsq, cnt, stride, b16, bsq, ops.new_keys(W * stride, DEV, runner=True), 192)
    want, want2 = ops.new_keys(W * stride, DEV), ops.new_keys(W * stride, DEV, runner=True)
    for w, n in enumerate(counts):
        n = max(0, min(n, stride))
        if n:
            lo = w * stride
            ops.l2_min_keys(q16[lo:lo + n].contiguous(), qsq[lo:lo + n].contiguous(), b16, bsq, want[lo:lo + n], 7)
            part = ops.l2_min_keys(q16[lo:lo + n].contiguous(), qsq[lo:lo + n].contiguous(), b16, bsq, ops.new_keys(n, DEV, runner=True), 192)
            want2[:, lo:lo + n] = part
    assert torch.equal(got2, want2), "segments launch, best + runner-up planes"
    live = got != eng.KEY_EMPTY
    assert torch.equal(got2[0][live] - 185, got[live]) and bool((got2[:, ~live] == eng.KEY_EMPTY).all())   # same keys, rows shifted
    return got, want


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("counts,stride,Nb", [
    ((700, 0, 513, 1), 768, 1300),            # ragged, an empty segment, partial tiles, library with a partial last tile
    ((1024, 1024), 1024, 512),                # full segments, whole tiles
    ((300, 900, 2000), 1024, 777),            # a count ABOVE the stride is cut to the stride (rows beyond were not gathered)
    ((5,), 256, 256),                         # one short segment: the 128 x 128 kernel's path (Q < 512)
    ((0, 0), 512, 600),                       # nothing live: no key is touched
    ((260, 250, 255, 257, 1, 256, 511, 3), 512, 2048),   # eight segments
])
def test_segments_launch_equals_one_launch_per_segment(counts, stride, Nb, dtype):
    got, want = _segments_case(counts, stride, Nb, 768, dtype, seed=sum(counts) + Nb)
    assert torch.equal(got, want)
    for w, n in enumerate(counts):            # rows at and beyond a segment's live count are neither searched nor written
        assert bool((got[w * stride + min(n, stride):(w + 1) * stride] == eng.KEY_EMPTY).all())


def _query_sets(W, Q, D, bg_share, seed):
    """W batches of Q rows, a share of each the repeated background row -> their dedup plans (each rank compacts its own)."""
    plans, fulls = [], []
    for w in range(W):
        g = torch.Generator(device=DEV).manual_seed(seed + w)
        q = torch.randn(Q, D, generator=g, device=DEV)
        bg = torch.rand(Q, generator=g, device=DEV) < bg_share * (0.8 + 0.4 * w / max(W - 1, 1))   # ragged live counts
        q[bg] = -0.3
        q16, _, qsq = ops.normalize_cast(q)
        del q
        plans.append(ops.rows_dedup_plan(q16, qsq))
        fulls.append((q16, qsq))
    return plans, fulls


@pytest.mark.parametrize("cls_rows", [76518, 113209], ids=["bagel", "peach"])
@pytest.mark.parametrize("W", [2, 4, 8])
def test_fake_world_full_size_sharded_search_equals_single_library(W, cls_rows):
    """Full size: W ranks x 32 images (100 352 query rows each, ~46 % background), bagel (76 518 rows: 9 600-row shards of 38 tiles
    at W = 8) and peach (113 209 rows).  For every rank r: ONE segments launch of all W ranks' live rows against shard r; MIN
    over r; expansion per rank -- equal to that rank's de-duplicated search of the whole library, which the dedup tests pin to
    the search of every row."""
    Q, D = 32 * 3136, 768
    gb = torch.Generator(device=DEV).manual_seed(4321 + cls_rows)
    full = torch.randn(cls_rows, D, generator=gb, device=DEV)
    plans, _ = _query_sets(W, Q, D, 0.46, seed=1000 * W)
    counts = [int(p.count.item()) for p in plans]
    assert len(set(counts)) > 1                                             # ragged
    cap = min(Q, (max(counts) + 255) // 256 * 256)
    q_all = torch.cat([p.q16[:cap] for p in plans])
    s_all = torch.cat([p.q_sq[:cap] for p in plans])
    cnt = torch.tensor(counts, dtype=torch.int32, device=DEV)
    merged = ops.new_keys(W * cap, DEV, runner=True)             # best + runner-up planes (include/cmdiad_hip.h)
    covered = 0
    for r in range(W):
        bank = eng.Bank(full, r, W)
        assert bank.row_offset == covered and bank.bf16.shape[0] % 256 == 0
        covered += bank.shard_rows
        keys = eng._HipSearch.search_segments(q_all, s_all, cnt, cap, bank, ops.new_keys(W * cap, DEV, runner=True))
        merged = eng.merge_key_planes(merged, keys)                         # the two all_reduce(MIN) of engine.merge_shard_keys
        del bank, keys
    assert covered == cls_rows
    whole = eng.Bank(full, 0, 1)
    for w in range(W):
        p = plans[w]
        ref_c = ops.l2_min_keys_counted(p.q16, p.q_sq, p.count, whole.bf16, whole.sqnorm, ops.new_keys(Q, DEV, runner=True))
        seg = merged[:, w * cap:w * cap + counts[w]]
        assert torch.equal(seg, ref_c[:, :counts[w]]), f"rank {w}: sharded keys (best + runner-up) differ from the single-library keys"
        assert bool((merged[:, w * cap + counts[w]:(w + 1) * cap] == eng.KEY_EMPTY).all())
        out = ops.keys_expand(merged[:, w * cap:(w + 1) * cap].contiguous(), p.slot, torch.empty((2, Q), dtype=torch.int64, device=DEV)) \
            if cap == Q else None
        if out is not None:
            assert torch.equal(out, ops.keys_expand(ref_c, p.slot, torch.empty_like(out)))
    idx = (merged[:, :counts[0]] & 0xFFFFFFFF)
    assert int(idx.max()) < cls_rows                                        # no key names a pad row


def _world_of_one():
    import os
    import socket
    import torch.distributed as td
    if not td.is_initialized():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        td.init_process_group("nccl", device_id=torch.device("cuda", 0))
    return td.group.WORLD


def test_pipeline_repeats_a_step_whose_live_rows_exceed_the_sticky_cap():
    """BatchPredictor with the row-sharded search (RCCL, world of one): batches of sparse clouds set a small sticky gather cap;
    a batch of dense clouds then has more live query rows than the cap -- the step's flag reaches the ticket, wait() repeats the
    batch with a re-read cap, and every batch's scores and maps equal the unsharded predictor's, bit for bit."""
    import importlib.util
    import os
    import torch.distributed as td
    from cmdiad_amd.predictor import BatchPredictor
    from cmdiad_amd.synth import synth_cloud, synth_rgb
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    group = _world_of_one()
    try:
        st = bench.build_state(torch.device("cuda", 0))
        B, n_max = 8, 34000

        def batch(seed, frac):
            return (torch.cat([synth_rgb(seed + i) for i in range(B)]).to(DEV),
                    torch.cat([synth_cloud(seed + i, frac) for i in range(B)]).to(DEV))

        sparse_a, sparse_b, medium, dense = batch(10, 0.36), batch(30, 0.37), batch(70, 0.46), batch(50, 0.64)
        mk = lambda g: BatchPredictor(st["engine"], st["bank_xyz"], st["bank_second"], st["stats"], st["det"], st["seg"], batch=B,
                                      n_max=n_max, group=g)
        ref, sh = mk(None), mk(group)
        # medium then dense: two CONSECUTIVE steps overflow, the second by more -- when the first one's repeat has re-gathered
        # buffer set 0 the flags the searches hold no longer say that the second one overflowed (the repeat checks its own)
        order = [sparse_a, sparse_b, medium, dense, dense, sparse_b]
        want = [ref.predict_batch(*b) for b in order]
        got = []
        pending = []
        for b in order:                                   # pipelined: two tickets outstanding, as bench.run_steps drives it
            if len(pending) == 2:
                got.append(pending.pop(0).wait())
            pending.append(sh.submit(*b))
        got += [t.wait() for t in pending]
        assert sh.redone >= 2, "the medium and the dense batch must have overflowed the cap set by the sparse ones"
        assert sh.redone <= 3
        for (gs, gm), (ws, wm) in zip(got, want):
            assert np.array_equal(gs, ws) and np.array_equal(gm, wm)
    finally:
        td.destroy_process_group()


def test_host_batches_staged_two_submits_ahead_change_nothing():
    """BatchPredictor.submit(stage=): a pinned host batch copied on the post stream behind the tail of the step two submits earlier
    (what brings bench.py's `h2d_inclusive` towards the resident rate, profiles/r6_notes.md section 4) gives the outputs of the
    plain submits, bit for bit -- with three tickets outstanding as bench.run_steps keeps them; a staged batch that is NOT the one
    submitted later (the caller changed its mind) is simply copied again; resident batches ignore the hint."""
    import importlib.util
    import os
    from cmdiad_amd.predictor import BatchPredictor
    from cmdiad_amd.synth import synth_cloud, synth_rgb
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    st = bench.build_state(torch.device("cuda", 0))
    B, n_max = 4, 34000
    host = [(torch.cat([synth_rgb(900 + 10 * k + i) for i in range(B)]).pin_memory(),
             torch.cat([synth_cloud(900 + 10 * k + i, 0.36 + 0.03 * k) for i in range(B)]).pin_memory()) for k in range(5)]
    mk = lambda: BatchPredictor(st["engine"], st["bank_xyz"], st["bank_second"], st["stats"], st["det"], st["seg"], batch=B, n_max=n_max)   # noqa: E731
    plain = mk()
    want = [plain.predict_batch(*b) for b in host]
    order = [0, 1, 2, 3, 4, 0, 2, 1]

    def run(pred, hint):
        pending, got = [], []
        for n, k in enumerate(order):
            if len(pending) == 3:
                got.append(pending.pop(0).wait())
            pending.append(pred.submit(*host[k], stage=hint(n)))
        return got + [t.wait() for t in pending]

    right = lambda n: host[order[n + 2]] if n + 2 < len(order) else None      # noqa: E731  the batch submitted two submits later
    wrong = lambda n: host[(order[n] + 3) % 5]                                   # noqa: E731  some OTHER batch: must be ignored
    for name, hint in (("staged", right), ("wrongly staged", wrong), ("no hint", lambda n: None)):
        got = run(mk(), hint)
        for (gs, gm), k in zip(got, order):
            assert np.array_equal(gs, want[k][0]) and np.array_equal(gm, want[k][1]), (name, k)
    dev = [(r.to(DEV), p.to(DEV)) for r, p in host]
    pred = mk()
    got = [pred.submit(*dev[k], stage=dev[(k + 2) % 5]).wait() for k in range(5)]
    for (gs, gm), k in zip(got, range(5)):
        assert np.array_equal(gs, want[k][0]) and np.array_equal(gm, want[k][1]), ("resident", k)


def test_sharded_search_object_world_of_one_rccl_sticky_cap_and_overflow():
    """engine.ShardedSearch through RCCL with a world of one rank: the sticky cap ("auto") reads the counts on the host once,
    a later batch with MORE live rows than the cap raises the device flag, regrow() + the repeated step is exact again."""
    import torch.distributed as td
    _world_of_one()
    try:
        Q, D, Nb = 4096, 768, 3000
        g = torch.Generator().manual_seed(5)
        lib = torch.randn(Nb, D, generator=g).to(DEV)
        bank = eng.Bank(lib, 0, 1)

        def batch(bg_share, seed):
            gq = torch.Generator().manual_seed(seed)
            q = torch.randn(Q, D, generator=gq)
            q[torch.rand(Q, generator=gq) < bg_share] = -0.3
            return ops.normalize_cast(q.to(DEV))

        def single(q16, qsq):
            return ops.l2_min_keys(q16, qsq, bank.bf16, bank.sqnorm, ops.new_keys(Q, DEV, runner=True))

        ss = eng.ShardedSearch(bank, td.group.WORLD, cap_rows="auto", slack=0.02)
        for i in range(3):
            q16, _, qsq = batch(0.6, 100 + i)
            keys = ss.gather(q16, qsq).gemm().reduce()
            assert torch.equal(keys, single(q16, qsq)) and not ss.overflowed()
        assert ss.host_reads == 1 and ss.cap < Q
        q16, _, qsq = batch(0.1, 200)                                       # far more live rows than the cap
        bad = ss.gather(q16, qsq).gemm().reduce()
        assert ss.overflowed() and ss.host_reads == 1
        assert not torch.equal(bad, single(q16, qsq))                       # the flag is what tells: these keys are incomplete
        ss.regrow()
        keys = ss.gather(q16, qsq).gemm().reduce()
        assert torch.equal(keys, single(q16, qsq)) and not ss.overflowed() and ss.host_reads == 2
    finally:
        td.destroy_process_group()


def _lockstep(gens):
    """Drives W `_sharded_score_steps` generators (one per fake rank) in lock step on one device: the stand-in for the collectives."""
    W = len(gens)
    reqs = [next(g) for g in gens]
    out = [None] * W
    while any(o is None for o in out):
        kind = reqs[0][0]
        assert all(k == kind for k, _ in reqs)
        if kind == "sum":
            tot = reqs[0][1].clone()
            for _, t in reqs[1:]:
                tot += t
            res = [tot.clone() for _ in range(W)]
        else:
            stack = torch.stack([t for _, t in reqs])
            res = [stack] * W
        nxt = []
        for r, g in enumerate(gens):
            try:
                nxt.append(g.send(res[r]))
            except StopIteration as done:
                out[r] = done.value
        reqs = nxt
    return out


@pytest.mark.parametrize("W,rows", [(2, 20000), (4, 20000), (8, 9000), (4, 300)])
def test_fake_world_sharded_fp32_library_reweight_equals_single_library(W, rows):
    """SURVEY 8(e)'s re-weight step with the fp32 rows sharded as well (engine.Bank(replicate_f32=False)): exact re-score, s*,
    m_star hand-over, per-shard top-3 + merge, the two re-weighting distances -- W fake ranks in lock step on one device against
    the single-library `score_patches_from_keys` (features.py:225-290): every output identical, bit for bit, on every rank;
    duplicates of the winning row in DIFFERENT shards (ties -> lowest global row), and (W = 4, 300 rows) an EMPTY last shard."""
    B, Q, D = 6, 784, 768
    g = torch.Generator().manual_seed(100 + W)
    lib = torch.randn(rows, D, generator=g)
    lib[rows - 5] = lib[7]                                   # exact duplicates across shards
    lib[rows // 2 + 3] = lib[7]
    patch = lib[torch.randint(0, rows, (B * Q,), generator=g)] + 0.4 * torch.randn(B * Q, D, generator=g)
    patch[5] = lib[7]                                        # a query that IS the duplicated row
    patch = patch.view(B, Q, D).to(DEV)
    whole = eng.Bank(lib.to(DEV))
    q16, _, qsq = ops.normalize_cast(patch.reshape(B * Q, D))
    keys = ops.l2_min_keys(q16, qsq, whole.bf16, whole.sqnorm, ops.new_keys(B * Q, DEV, runner=True))
    want = eng.score_patches_from_keys(patch, keys, whole, (28, 28))
    banks = [eng.Bank(lib.to(DEV), r, W, replicate_f32=False) for r in range(W)]
    assert sum(b.f32_rows for b in banks) == rows and all(b.f32_sharded for b in banks)
    if rows == 300:
        assert banks[-1].f32_rows == 0                        # 128-row aligned shards: 300 rows leave the fourth rank nothing
    got = _lockstep([eng._sharded_score_steps(patch, keys, b, (28, 28), 224) for b in banks])
    for r, res in enumerate(got):
        for k in ("min_val", "min_idx", "s_idx", "s_star", "top3", "knn_d", "s", "s_map_pre"):
            assert torch.equal(res[k], want[k]), (r, k)
    assert int(want["min_idx"].view(-1)[5]) == 7              # the duplicated row: lowest global row wins


def test_pipeline_with_sharded_fp32_library_through_rccl_world_of_one():
    """BatchPredictor on libraries whose fp32 rows are sharded too (the scoring tail then runs eagerly with its four collectives),
    through RCCL with a world of one: scores and maps equal the replicated-library predictor's, bit for bit."""
    import importlib.util
    import os
    import torch.distributed as td
    from cmdiad_amd.predictor import BatchPredictor
    from cmdiad_amd.synth import synth_cloud_fixed_n, synth_rgb
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    group = _world_of_one()
    try:
        st = bench.build_state(torch.device("cuda", 0))
        B = 4
        rgb = torch.cat([synth_rgb(70 + i) for i in range(B)]).to(DEV)
        pcs = torch.cat([synth_cloud_fixed_n(70 + i, 24576) for i in range(B)]).to(DEV)
        ref = BatchPredictor(st["engine"], st["bank_xyz"], st["bank_second"], st["stats"], st["det"], st["seg"], batch=B, n_max=24576)
        want = ref.predict_batch(rgb, pcs)
        bx = eng.Bank(st["bank_xyz"].f32, 0, 1, replicate_f32=False)
        br = eng.Bank(st["bank_second"].f32, 0, 1, replicate_f32=False)
        sh = BatchPredictor(st["engine"], bx, br, st["stats"], st["det"], st["seg"], batch=B, n_max=24576, group=group)
        assert sh.stage2_eager
        for _ in range(3):                                    # both buffer sets
            got = sh.predict_batch(rgb, pcs)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        with pytest.raises(ValueError):
            BatchPredictor(st["engine"], bx, br, st["stats"], st["det"], st["seg"], batch=B, n_max=24576, group=None)
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("W", [2, 4, 8])
def test_fake_world_row_sharded_coreset_selection_equals_single_device(W):
    """SURVEY 8(e) fit-time sharding: every fake rank scans its 4-row aligned range of the projected library per round and proposes a
    packed (running minimum, row) key; the MAX over the ranks is the next pivot (cmdiad_coreset_prepare / _round / _decode).  The
    picks equal cmdiad_coreset_greedy's, pick for pick -- including duplicate rows in different shards (ties -> lowest row)."""
    from cmdiad_amd import coreset
    g = torch.Generator().manual_seed(7)
    n, d, n_sel = 20003, 334, 300
    z = torch.randn(n, d, generator=g)
    z[15000] = z[40]
    z[19999] = z[40]                                        # duplicates across shards
    z = z.to(DEV)
    want = coreset.greedy_coreset(z, n_sel).cpu()
    ranks = [coreset._HipRounds(z) for _ in range(W)]
    bounds = [coreset.shard_rows(n, r, W) for r in range(W)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n and all(b[0] % 4 == 0 for b in bounds)
    keys = torch.zeros((n_sel - 1,), dtype=torch.int64, device=DEV)
    for r in range(n_sel - 1):
        mine = torch.zeros((W,), dtype=torch.int64, device=DEV)
        for w in range(W):
            ranks[w].round(bounds[w][0], bounds[w][1], keys[r - 1:r] if r else None, mine[w:w + 1])
        keys[r] = mine.max()                                # the all_reduce(MAX)
    got = ranks[0].decode(keys, n_sel).cpu()
    assert torch.equal(got, want)


def test_row_sharded_coreset_through_rccl_world_of_one():
    import torch.distributed as td
    from cmdiad_amd import coreset
    group = _world_of_one()
    try:
        z = torch.randn(9001, 200, generator=torch.Generator().manual_seed(3)).to(DEV)
        assert torch.equal(coreset.greedy_coreset_sharded(z, 120, group).cpu(), coreset.greedy_coreset(z, 120).cpu())
    finally:
        td.destroy_process_group()


def test_pipeline_invariants_hold_on_random_batches_with_a_live_process_group():
    """tools/fuzz_pipeline.py for 12 s: random batches (1-6 clouds of a few hundred points ... no background at all, both workloads)
    through dozens of BatchPredictor instances beside a live RCCL process group -- graph == eager == second submit, the row
    de-duplication changes nothing, a sample's scores do not depend on its batch, the row-sharded search (world of one) equals the
    plain one, bit for bit.  Round 5: per-instance torch streams walked through torch's pool of 32 onto RCCL's own stream; a graph
    capture that forked onto it made RCCL's watchdog thread raise and abort the process (now: ops.shared_stream)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(repo, "tools", "fuzz_pipeline.py"), "12", "17"], capture_output=True, text=True,
                         timeout=600, env=env, cwd=repo)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("pipeline fuzz ok")][-1]
    assert int(line.split()[3]) >= 15, line
