"""configs[3]: the row-sharded library search -- over the ranks of a real process group (`sharded_search`) and replayed at the real
shard shapes on one GPU (`fake_world`)."""
import os
import time

from .common import BATCH, N_POINTS, PEAK_BF16_TFLOPS, class_rows

def sharded_search(dev, group, rank, world, rows_list, iters=10, warm=3):
    """configs[3]: the patch-library nearest-neighbour search with the library's ROWS sharded over the ranks
    (cmdiad_amd.engine.ShardedSearch).  Every rank brings the 16-bit queries of its own batch of 32 images (100 352 x 768, 45.8 %
    of the rows the repeated background row, as in the bench's clouds); one iteration = local de-duplication -> counts exchange
    -> all-gather of the LIVE rows only -> distance GEMM of all ranks' live rows against this rank's row shard -> ONE
    integer-MIN all-reduce of the packed keys (RCCL over xGMI) -> expansion to one key per original row.  Iteration i + 1's
    exchange is issued on a second stream under iteration i's GEMM.  Timed with a barrier on both sides, max over ranks;
    the serial split (gather / GEMM / reduce + expand, HIP events, un-overlapped) is measured in a separate pass."""
    import types
    import torch
    import torch.distributed as td
    from cmdiad_amd import engine as eng
    from cmdiad_amd import ops
    Q = BATCH * 3136
    g = torch.Generator(device=dev).manual_seed(977 + rank)
    q32 = torch.randn(Q, 768, generator=g, device=dev)
    bg = torch.rand(Q, generator=g, device=dev) < (1.0 - N_POINTS / 50176.0) * 0.9   # patches without a foreground pixel
    q32[bg] = -0.3
    q16, _, qsq = ops.normalize_cast(q32)
    del q32
    side = ops.shared_stream(dev, "bench.exchange")
    out = []
    for name, rows in rows_list:
        lo, hi = eng.shard_range(rows, rank, world)
        gb = torch.Generator(device=dev).manual_seed(4321 + rows)  # every rank draws the same library, keeps its rows
        full = torch.randn(rows, 768, generator=gb, device=dev)
        b16, _, bsq = ops.normalize_cast(full[lo:hi].contiguous())
        del full
        bank = types.SimpleNamespace(bf16=b16, sqnorm=bsq, row_offset=lo)
        stats = {}
        searches = [eng.ShardedSearch(bank, group, stats=stats) for _ in range(2)]
        cur = torch.cuda.current_stream()

        def gather_on_side(s, after):
            side.wait_event(after)        # NOT wait_stream(cur): the GEMM just queued on `cur` is what this exchange runs under
            with torch.cuda.stream(side):
                s.gather(q16, qsq)

        def mark():
            e = torch.cuda.Event()
            e.record(cur)
            return e

        def run(n):
            gather_on_side(searches[0], mark())
            keys = None
            for i in range(n):
                s = searches[i & 1]
                cur.wait_stream(side)                 # this iteration's exchange has landed
                before_gemm = mark()                  # everything up to the previous iteration's reduce: the other buffer set is free
                s.gemm()
                for t in (s.q_all, s.s_all):          # allocated on `side`, read on `cur`
                    t.record_stream(cur)
                if i + 1 < n:
                    gather_on_side(searches[(i + 1) & 1], before_gemm)   # the next exchange, under this GEMM; its collectives are
                keys = s.reduce()                                        # queued before this iteration's min-reduce
            return keys

        merged = run(warm)
        assert int((merged == eng.KEY_EMPTY).sum()) == 0           # every query found a row somewhere
        td.barrier(group)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(iters)
        torch.cuda.synchronize()
        td.barrier(group)
        dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        td.all_reduce(dt, op=td.ReduceOp.MAX, group=group)
        ms = float(dt.item()) / iters * 1e3
        # the serial split: the three stages one after the other on one stream, HIP events between them
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(3)]
        for e4 in ev:
            s = searches[0]
            e4[0].record(); s.gather(q16, qsq); e4[1].record(); s.gemm(); e4[2].record(); s.reduce(); e4[3].record()
        torch.cuda.synchronize()
        split = [sum(e4[k].elapsed_time(e4[k + 1]) for e4 in ev) / len(ev) for k in range(3)]
        live = sum(stats["live_rows"])
        flops = 2.0 * live * (hi - lo) * 768
        out.append(dict(cls=name, rows=rows, rows_this_rank=hi - lo, ms_per_search=round(ms, 3),
                        images_per_s=round(world * BATCH / (ms * 1e-3), 1),
                        serial_ms_rank0=dict(dedup_and_gather=round(split[0], 3), gemm=round(split[1], 3), reduce_and_expand=round(split[2], 3)),
                        gemm_tflops_rank0=round(flops / (split[1] * 1e-3) / 1e12, 1),
                        overlap_gain_ms=round(sum(split) - ms, 3),
                        live_rows_per_rank=stats["live_rows"], gathered_rows_per_rank=stats["gathered_rows_per_rank"],
                        gather_MB_received_per_rank=round(stats["gather_bytes_received"] / 1e6, 2),
                        gather_MB_received_without_compaction=round(stats["gather_bytes_received_without_compaction"] / 1e6, 2),
                        reduce_MB=round(stats["reduce_bytes"] / 1e6, 3)))
        del b16, bsq, searches
    return dict(what="row-sharded library search: local de-duplication of the repeated background row -> all-gather of the live 16-bit "
                     "query rows only -> per-shard distance GEMM -> one all_reduce(MIN) of packed int64 keys -> expansion; the next "
                     "iteration's exchange runs under the current GEMM; weak scaling, 32 images (100 352 query rows) per rank",
                rccl_ranks=td.get_world_size(group), backend=td.get_backend(group), classes=out)


def fake_world_leg(dev, classes=("bagel", "peach"), worlds=(1, 2, 4, 8), iters=4):
    """configs[3] at its REAL shard shapes, on one GPU ("fake world"): for W in `worlds` the library's rows are cut
    into the W shards `engine.Bank` makes (128-row aligned, search operand padded to whole tiles), W separately compacted query sets
    of 32 images each (100 352 rows, 45.8 % of them the repeated background row) are laid out as the gathered operand of
    `engine.ShardedSearch` (W segments of `cap` rows + the W live counts on the device), and EVERY shard's distance GEMM -- one
    `cmdiad_l2_min_keys_segments` launch, what one rank of a W-rank node executes per step -- is timed alone with HIP events.  The
    integer MIN over the W shards' keys is compared with the single-library keys (bit for bit).  The exchange is NOT measured here
    (one GPU): gather / reduce bytes are stated and a link model turns them into a predicted per-search time and rate."""
    import torch
    from cmdiad_amd import engine as eng
    from cmdiad_amd import ops
    Q, D = BATCH * 3136, 768
    LINK_GBS, LINK_EFF, COLL_LAT_US = 153.0, 0.8, 30.0      # xGMI: one link per peer, 153 GB/s per direction (MI355X_MICROARCH.md)
    g = torch.Generator(device=dev).manual_seed(977)
    q32 = torch.randn(Q, D, generator=g, device=dev)
    bg = torch.rand(Q, generator=g, device=dev) < (1.0 - N_POINTS / 50176.0) * 0.9   # patches without a foreground pixel
    q32[bg] = -0.3
    q16, _, qsq = ops.normalize_cast(q32)
    del q32
    wmax = max(worlds)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plans = []
    for w in range(wmax):                    # rank w's own batch: the same rows in another order (its own compaction)
        sh = (w * 9973) % Q
        plans.append(ops.rows_dedup_plan(torch.roll(q16, sh, 0).contiguous(), torch.roll(qsq, sh, 0).contiguous()))
    e1.record()
    torch.cuda.synchronize()
    dedup_ms = e0.elapsed_time(e1) / wmax    # incl. the roll; an upper bound of the plan's ~0.12 ms
    counts = [int(p.count.item()) for p in plans]
    cap = min(Q, (max(counts) + 255) // 256 * 256)
    row_bytes = D * 2 + 4
    out = []
    for name in classes:
        rows = class_rows(name)
        gb = torch.Generator(device=dev).manual_seed(4321 + rows)
        full = torch.randn(rows, D, generator=gb, device=dev)
        whole = eng.Bank(full, 0, 1)             # the single-library answer for rank 0's live rows (the counted launch of the pipeline)
        ref_keys = ops.l2_min_keys_counted(plans[0].q16, plans[0].q_sq, plans[0].count, whole.bf16, whole.sqnorm, ops.new_keys(Q, dev, runner=True))
        del whole
        for W in worlds:
            q_all = torch.cat([p.q16[:cap] for p in plans[:W]])
            s_all = torch.cat([p.q_sq[:cap] for p in plans[:W]])
            cnt = torch.tensor(counts[:W], dtype=torch.int32, device=dev)
            merged = None
            ms = []
            for r in range(W):
                bank = eng.Bank(full, r, W)
                keys = ops.new_keys(W * cap, dev, runner=True)
                ops.l2_min_keys_segments(q_all, s_all, cnt, cap, bank.bf16, bank.sqnorm, keys, bank.row_offset)   # warm + the checked result
                merged = keys if merged is None else eng.merge_key_planes(merged, keys)
                scratch = ops.new_keys(W * cap, dev, runner=True)
                t = 0.0
                for _ in range(iters):
                    scratch.fill_(eng.KEY_EMPTY)
                    e0.record()
                    ops.l2_min_keys_segments(q_all, s_all, cnt, cap, bank.bf16, bank.sqnorm, scratch, bank.row_offset)
                    e1.record()
                    torch.cuda.synchronize()
                    t += e0.elapsed_time(e1)
                ms.append(t / iters)
                shard_rows, shard_tiles = bank.shard_rows, bank.bf16.shape[0] // 256
                del bank, keys, scratch
            same = bool(torch.equal(merged[:, :counts[0]], ref_keys[:, :counts[0]]))     # best AND runner-up planes
            live = sum(counts[:W])
            per = ((rows + W - 1) // W + 127) // 128 * 128
            flops = 2.0 * live * min(per, rows) * D      # the largest (= every but the last) shard
            gemm = max(ms)
            gather_b = (W - 1) * cap * row_bytes
            reduce_b = W * cap * 8 * 2                   # two MIN all-reduces: best and runner-up planes
            t_gather = cap * row_bytes / (LINK_GBS * 1e9 * LINK_EFF) * 1e3 + COLL_LAT_US * 1e-3 if W > 1 else 0.0   # every peer's segment over its own link
            t_reduce = (2.0 * (W - 1) / W * reduce_b / (min(W - 1, 7) * LINK_GBS * 1e9 * LINK_EFF) * 1e3 + COLL_LAT_US * 1e-3) if W > 1 else 0.0
            t_search = dedup_ms + max(gemm, t_gather) + t_reduce
            out.append(dict(cls=name, rows=rows, world=W, rows_per_rank=min(per, rows), shard_tiles=shard_tiles, live_rows_per_rank=counts[:W],
                            gathered_rows_per_rank=cap, gemm_ms_slowest_rank=round(gemm, 3), gemm_ms_mean=round(sum(ms) / len(ms), 3),
                            gemm_tflops_per_rank=round(flops / (gemm * 1e-3) / 1e12, 1), gemm_frac_of_peak=round(flops / (gemm * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                            merged_keys_equal_single_library=same,
                            gather_MB_received_per_rank=round(gather_b / 1e6, 2), reduce_MB=round(reduce_b / 1e6, 3),
                            model=dict(gather_ms=round(t_gather, 3), reduce_ms=round(t_reduce, 3), search_ms=round(t_search, 3),
                                       images_per_s=round(W * BATCH / (t_search * 1e-3), 1))))
            assert same, f"fake world {name} W={W}: the MIN over the shards' keys differs from the single-library keys"
        del full
    return dict(what="compute side measured on 1 GPU, links not measured: per W the distance GEMM of ONE rank of a W-rank node (all W ranks' live "
                     "query rows against a 1/W row shard, one cmdiad_l2_min_keys_segments launch, HIP events, every shard timed in turn); "
                     "model = dedup + max(GEMM, all-gather) + all-reduce with one xGMI link per peer",
                link_model=dict(link_GBs_per_direction=LINK_GBS, efficiency=LINK_EFF, collective_latency_us=COLL_LAT_US),
                dedup_ms=round(dedup_ms, 3), shapes=out)
