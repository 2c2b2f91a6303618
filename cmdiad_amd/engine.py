"""Batched, device-resident hot path: extract (a1-a10) -> normalise (a11) -> patch-library nearest
neighbour + re-weighting (a12-a13) -> bilinear score map.  The drop-in classes in
feature_extractors/ call this with B = 1 exactly where the reference calls its torch ops; bench.py
calls it with B = 32.  Results per sample do not depend on B (tests/test_gpu_engine.py).

Nothing here touches the CPU oracle; every stage is a HIP kernel from libcmdiad_hip.so.
"""
import ctypes
import math
import os

import torch

from . import _native as nat
from . import ops


class Bank:
    """One patch library resident in HBM.

    * ``f32``  [Nb,D]  the reference's normalised library (exact re-score, re-weighting): replicated on every GPU
      (<= 350 MB for the largest MVTec-3D class); ``blk16`` is the same data in the MFMA-operand layout the
      re-weighting scan streams (one more copy: 288 GB of HBM);
    * ``bf16`` [rows,D] + ``sqnorm`` [rows]: the operand of the distance GEMM.  On a multi-GPU node this is the
      rank's ROW SHARD of the search (``row_offset`` = first global row): every rank searches its
      shard for the queries of ALL ranks and the per-shard minima are combined with one integer-MIN
      all-reduce of packed (distance, global row) keys over RCCL/xGMI."""

    def __init__(self, rows_f32, rank=0, world=1, replicate_f32=True):
        """replicate_f32=False: this rank keeps ONLY its row shard of the fp32 library (and of the block16 copy) as well -- the
        partitioning of SURVEY 8(e) to the letter.  The exact re-score and the re-weighting step (features.py:235-290) then need
        four small collectives per scored batch (score_patches_from_keys with `group`): who owns a row contributes it.  The default
        replicates the fp32 rows (<= 700 MB for the largest MVTec-3D class: 0.25 % of 288 GB) and needs none."""
        assert rows_f32.is_cuda and rows_f32.dtype == torch.float32
        n = rows_f32.shape[0]
        lo, hi = shard_range(n, rank, world)
        self.row_offset, self.rank, self.world = lo, rank, world
        self.shard_rows = hi - lo
        self.total_rows = n
        self.f32_sharded = not replicate_f32      # (a world of one takes the collective path too: that is how one GPU tests it)
        if hi > lo:
            b16, _, sq = ops.normalize_cast(rows_f32[lo:hi].contiguous())
        else:       # an empty shard (fewer rows than 128 x (world - 1)): nothing to search, its keys stay KEY_EMPTY
            b16 = torch.empty((0, rows_f32.shape[1]), dtype=ops.SEARCH_DTYPE, device=rows_f32.device)
            sq = torch.empty((0,), dtype=torch.float32, device=rows_f32.device)
        if self.f32_sharded:
            self.f32 = rows_f32[lo:hi].clone() if hi > lo else torch.zeros((1, rows_f32.shape[1]), dtype=torch.float32, device=rows_f32.device)
            self.f32_offset, self.f32_rows = lo, hi - lo          # (an empty shard keeps one dummy row that no index can name)
        else:
            self.f32 = rows_f32.contiguous()
            self.f32_offset, self.f32_rows = 0, n
        # The distance GEMM's production kernel takes whole 256-row library tiles and hands the last n % 256 rows to a second,
        # much less efficient launch (76 518 = 298 x 256 + 230).  Pad the SEARCH operand to a tile boundary with rows that cannot
        # win -- all-zero rows whose squared norm is +inf, so d2 = (|q|^2 + inf) - 2 * 0 = inf never passes `d2 < best` -- and the
        # whole library is one launch.  Keys never name a pad row; the fp32 library (re-score, re-weighting) is untouched.
        pad = (-self.shard_rows) % 256
        if pad and self.shard_rows >= 256:
            b16 = torch.cat([b16, torch.zeros((pad, b16.shape[1]), dtype=b16.dtype, device=b16.device)])
            sq = torch.cat([sq, torch.full((pad,), float("inf"), dtype=sq.dtype, device=sq.device)])
        self.bf16, self.sqnorm = b16.contiguous(), sq.contiguous()
        self._blk16 = None

    @property
    def blk16(self):
        """fp32 copy in the MFMA-operand layout of the re-weighting scan (cmdiad_bank_block16), built on first use."""
        if self._blk16 is None:
            self._blk16 = ops.bank_block16(self.f32)
        return self._blk16

    @property
    def rows(self):
        """rows of the WHOLE library (the fp32 copy holds all of them unless f32_sharded)"""
        return self.total_rows


def shard_range(n, rank, world):
    """Contiguous row range of shard `rank`; 128-row aligned so shards start on a GEMM tile boundary."""
    per = ((n + world - 1) // world + 127) // 128 * 128
    return min(rank * per, n), min((rank + 1) * per, n)


KEY_EMPTY = ops.KEY_EMPTY


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def merge_shard_keys(keys, group=None):
    """Combine per-shard packed keys: key = (fp32 bits of d2 >= 0) << 32 | global row.  The keys -- and the "no
    candidate" sentinel ops.KEY_EMPTY a rank with an EMPTY shard leaves behind -- are non-negative as int64, so
    signed MIN == unsigned MIN == (smallest distance, lowest global row).
    keys [2, n] (best + runner-up per query, ops.new_keys(runner=True)): the best plane as above; the runner-up of the whole
    library is the smallest key among every shard's best OTHER than the global best and every shard's runner-up -- a second
    MIN all-reduce over (own best == global best ? own runner-up : own best).  Both planes then equal what one device searching
    the whole library returns (the runner-up is defined on groups of 16 GLOBAL rows, and shards start on multiples of 128)."""
    if group is not None:
        import torch.distributed as td
        if keys.dim() == 2:
            own_best = keys[0].clone()
            td.all_reduce(keys[0], op=td.ReduceOp.MIN, group=group)
            cand = torch.where(own_best == keys[0], keys[1], own_best)
            td.all_reduce(cand, op=td.ReduceOp.MIN, group=group)
            keys[1].copy_(cand)
        else:
            td.all_reduce(keys, op=td.ReduceOp.MIN, group=group)
    return keys


def merge_key_planes(a, b):
    """The merge of merge_shard_keys for two key sets that live on ONE device (replayed shards: tests, bench.py `fake_world`):
    [n] keys -> element-wise minimum; [2, n] keys -> (smallest, second smallest) of the four candidates of every query."""
    if a.dim() == 1:
        return torch.minimum(a, b)
    return torch.stack([torch.minimum(a[0], b[0]), torch.minimum(torch.maximum(a[0], b[0]), torch.minimum(a[1], b[1]))])


def gather_queries(q16, qsq, group=None):
    """All-gather the bf16 queries (+ their squared norms) of every rank: [Q,D] -> [W*Q,D]."""
    if group is None:
        return q16, qsq
    import torch.distributed as td
    world = td.get_world_size(group)
    q_all = torch.empty((world * q16.shape[0], q16.shape[1]), dtype=q16.dtype, device=q16.device)
    s_all = torch.empty((world * qsq.shape[0],), dtype=qsq.dtype, device=qsq.device)
    td.all_gather_into_tensor(q_all.view(torch.uint8), q16.contiguous().view(torch.uint8), group=group)  # raw bytes: gloo has no bf16
    td.all_gather_into_tensor(s_all, qsq, group=group)
    return q_all, s_all


class _HipSearch:
    """The device-side pieces of the row-sharded search (HIP kernels); tests/test_host_cpu.py swaps in a torch stand-in to check
    the host-side composition -- who gathers what, which rows are searched, how the keys come back -- without a GPU."""
    RUNNER = True     # the kernels also return every query's runner-up (keys [2, n]); a stand-in without it says False

    @staticmethod
    def plan(q16, q_sq, reuse=None):
        return ops.rows_dedup_plan(q16, q_sq, reuse)

    @staticmethod
    def search(q16, q_sq, bank, keys):
        return ops.l2_min_keys(q16, q_sq, bank.bf16, bank.sqnorm, keys, bank.row_offset)

    @staticmethod
    def search_segments(q_all, s_all, counts, cap, bank, keys_all):
        """Every rank's live rows (segment w = rows [w * cap, w * cap + counts[w]), counts on the device) against this rank's
        shard: ONE launch over the live query tiles of all segments (cmdiad_l2_min_keys_segments)."""
        return ops.l2_min_keys_segments(q_all, s_all, counts, cap, bank.bf16, bank.sqnorm, keys_all, bank.row_offset)

    @staticmethod
    def expand(keys_compact, slot, out):
        return ops.keys_expand(keys_compact, slot, out)


class ShardedSearch:
    """Nearest-neighbour keys of this rank's queries against the WHOLE library when every rank holds a row shard of it
    (features.py:186-190,227 for one device).  Compact first, then gather -- in three stages, so that a caller
    can put step i + 1's exchange under step i's distance GEMM (bench.py `sharded_search`):

      gather(q16, q_sq): the repeated background row of the rank's own queries is removed locally (csrc/dedup.hip): `count`
              live rows; the ranks exchange their counts (W int32, they STAY on the device); all-gather of the first `cap` rows
              of every rank's compacted queries + norms: about half of the 154 MB per rank that gathering every row takes
              (45 % of a batch's patches have no foreground pixel);
      gemm(): ONE launch: the distance GEMM over the live rows of all W segments against this rank's shard (global rows via
              row_offset), the live counts read on the device -- no collective, no host read;
      reduce(): ONE integer-MIN all-reduce of the packed (distance, global row) keys, then the rank's own segment is expanded
              back to one key per original row -> keys [Q] int64.

    `cap_rows`: how many rows of every rank travel.
      "exact": every gather() reads the W counts on the host (one synchronisation per step) and takes cap = max(count) rounded up
               to 256 rows;
      "auto" (default): the FIRST gather() does that once and keeps cap = max(count) * (1 + slack) (rounded up to 256) for the
               following steps -- no host read in steady state.  A step whose live rows exceed the cap on ANY rank cannot be
               answered from what was gathered: `overflow` (a device flag, the same on every rank because it is computed from the
               all-gathered counts) says so, and the caller checks it (`overflowed()`: a host read) wherever it synchronises with the
               step's results anyway, then calls `regrow()` and repeats the step;
      an int:  that many rows (Q = no compaction, never overflows).

    `stats` (a dict) receives the bytes this rank received in the gather, the live counts (of the last host read) and cap."""

    def __init__(self, bank, group, impl=_HipSearch, stats=None, cap_rows="auto", slack=0.06):
        import torch.distributed as td
        self.bank, self.group, self.impl, self.stats = bank, group, impl, stats
        self.world, self.rank = td.get_world_size(group), td.get_rank(group)
        self.runner = bool(getattr(impl, "RUNNER", False))   # keys [2, n]: best + runner-up (exact fp32 decision downstream)
        self.plan = None
        if not (cap_rows in ("auto", "exact") or (isinstance(cap_rows, int) and cap_rows > 0)):
            raise ValueError(f"cap_rows must be 'auto', 'exact' or a positive row count, not {cap_rows!r}")
        self.cap_rows, self.slack = cap_rows, float(slack)
        self.cap = None              # rows of every rank that travel (sticky in "auto" mode)
        self.counts = None           # live rows per rank at the last host read
        self.host_reads = 0
        self.overflow = None
        self.floor = 0               # "auto": the largest cap this search has ever used (regrow)

    def regrow(self):
        """Let the next gather() read the live counts on the host again.  The sticky cap only ever GROWS ("auto"): a sparse batch
        right after an overflow must not lower it below what a dense batch has already been seen to need."""
        self.floor = max(self.floor, self.cap or 0)
        self.cap = None

    def overflowed(self):
        """True when the last gather()'s live rows did not fit the cap on some rank (host read; identical on every rank)."""
        return self.overflow is not None and bool(self.overflow.item())

    def gather(self, q16, q_sq):
        import torch.distributed as td
        Q, D = q16.shape
        self.Q = Q
        self.plan = self.impl.plan(q16, q_sq, self.plan)
        counts_t = self.counts_dev = torch.empty((self.world,), dtype=torch.int32, device=q16.device)
        td.all_gather_into_tensor(counts_t, self.plan.count.view(1), group=self.group)
        if isinstance(self.cap_rows, int):
            cap = min(Q, (self.cap_rows + 255) // 256 * 256)
        elif self.cap_rows == "exact" or self.cap is None or self.cap > Q:
            self.counts = [int(c) for c in counts_t.cpu().tolist()]        # the one host read (every step / the first step)
            self.host_reads += 1
            grow = 1.0 if self.cap_rows == "exact" else 1.0 + self.slack
            cap = min(Q, (int(math.ceil(max(self.counts) * grow)) + 255) // 256 * 256)
            if self.cap_rows == "auto":
                cap = min(Q, max(cap, self.floor))
        else:
            cap = self.cap
        self.cap = cap
        self.overflow = counts_t.max() > cap
        self.q_all = torch.empty((self.world * cap, D), dtype=q16.dtype, device=q16.device)
        self.s_all = torch.empty((self.world * cap,), dtype=torch.float32, device=q16.device)
        td.all_gather_into_tensor(self.q_all.view(torch.uint8), self.plan.q16[:cap].view(torch.uint8), group=self.group)   # raw bytes: gloo has no bf16
        td.all_gather_into_tensor(self.s_all, self.plan.q_sq[:cap], group=self.group)
        if self.stats is not None:
            row = D * q16.element_size() + 4
            self.stats.update(world=self.world, rows_per_rank=Q, live_rows=self.counts, gathered_rows_per_rank=cap,
                              host_reads=self.host_reads,
                              gather_bytes_received=(self.world - 1) * cap * row,
                              gather_bytes_received_without_compaction=(self.world - 1) * Q * row,
                              reduce_bytes=self.world * cap * 8 * (2 if self.runner else 1))
        return self

    def gemm(self, timer=None):
        cap = self.cap
        shape = (2, self.world * cap) if self.runner else (self.world * cap,)
        self.keys_all = torch.full(shape, KEY_EMPTY, dtype=torch.int64, device=self.q_all.device)
        with (timer if timer is not None else _Null()):
            self.impl.search_segments(self.q_all, self.s_all, self.counts_dev, cap, self.bank, self.keys_all)
        return self

    def reduce(self):
        cap = self.cap
        keys_all = merge_shard_keys(self.keys_all, self.group)
        mine = keys_all[..., self.rank * cap:(self.rank + 1) * cap]
        lead = keys_all.shape[:-1]
        if cap < self.Q:
            # slot[] names compacted rows up to count - 1, and count is known on the device only: in a step that overflowed the
            # cap the expansion must still read inside its buffer (such rows get "no candidate"; `overflow` tells the caller)
            kc = torch.full((*lead, self.Q), KEY_EMPTY, dtype=torch.int64, device=keys_all.device)
            kc[..., :cap] = mine
            mine = kc
        return self.impl.expand(mine.contiguous(), self.plan.slot, torch.empty((*lead, self.Q), dtype=torch.int64, device=keys_all.device))


def sharded_min_keys(q16, q_sq, bank, group, plan=None, timer=None, stats=None, impl=_HipSearch):
    """ShardedSearch's three stages in line, the cap read per call ("exact") -> (keys [Q] int64, plan)."""
    s = ShardedSearch(bank, group, impl, stats, cap_rows="exact")
    s.plan = plan
    keys = s.gather(q16, q_sq).gemm(timer).reduce()
    return keys, s.plan


def _call(name, *args):
    nat.check(getattr(nat.lib(), name)(*args), name)


def score_patches(patch32, bank, dims, gt_size=224, group=None):
    """features.py:225-294 for a batch.  patch32 [B,Q,D] f32 cuda, ALREADY normalised.
    Returns dict(min_val [B,Q], min_idx [B,Q] int64, s_idx [B], s_star [B], s [B], s_map_pre [B,gt,gt], ...).

    With `group` (torch.distributed over RCCL) the search is row-sharded: all-gather of the bf16 queries,
    per-shard distance GEMM, integer-MIN all-reduces of the packed keys (best plane, then runner-up plane); the exact fp32
    decision between the two candidates and the re-weighting use the replicated fp32 library, so no further collectives are needed."""
    B, Q, D = patch32.shape
    dev = patch32.device
    flat = patch32.reshape(B * Q, D)
    q16, _, qsq = ops.normalize_cast(flat)
    if group is not None and os.environ.get("CMDIAD_DEDUP", "1") != "0":
        keys, _ = sharded_min_keys(q16, qsq, bank, group)      # compact locally, gather the live rows only
        return score_patches_from_keys(patch32, keys, bank, dims, gt_size, group)
    q_all, s_all = gather_queries(q16, qsq, group)
    keys = ops.new_keys(q_all.shape[0], dev, runner=True)
    if os.environ.get("CMDIAD_DEDUP", "1") != "0":
        # the rows of the patches without a foreground pixel repeat one vector: searched once (csrc/dedup.hip), keys identical
        plan = ops.rows_dedup_plan(q_all, s_all)
        kc = ops.l2_min_keys_counted(plan.q16, plan.q_sq, plan.count, bank.bf16, bank.sqnorm, keys, bank.row_offset)
        keys = ops.keys_expand(kc, plan.slot, torch.empty_like(kc))
    else:
        ops.l2_min_keys(q_all, s_all, bank.bf16, bank.sqnorm, keys, bank.row_offset)
    keys = merge_shard_keys(keys, group)
    if group is not None:
        keys = keys[..., bank.rank * B * Q:(bank.rank + 1) * B * Q].contiguous()
    return score_patches_from_keys(patch32, keys, bank, dims, gt_size, group)


def _sharded_score_steps(patch32, keys, bank, dims, gt_size):
    """score_patches_from_keys for a library whose fp32 rows are SHARDED too (Bank(replicate_f32=False)); SURVEY 8(e)'s re-weight
    step, features.py:225-290 per shard.  PRECONDITION ("queries are replicated"): every rank passes the SAME patch32 and
    keys -- a rank contributes the parts of the rows it owns to sums over the ranks.  A generator: every `yield (kind, tensor)` is one collective over the ranks and receives its
    result -- "sum": element-wise sum (exactly one rank, the owner of the row in question, contributes a non-zero value, so the sum
    is exact); "gather": [W, *shape] of every rank's tensor.  Driven by real collectives (`_drive_collectives`) or, on one device,
    by a lock-step loop over W generators (tests/test_gpu_fakeworld.py).
      1. exact fp32 distance to the winning row (keys [2, .]: to the best and the runner-up): by the rank that owns it -> sum [B*Q] / [2, B*Q] f32
      2. s* = max over a sample's patches; m_star = the winning row of that patch: sent by its owner       -> sum   [B, D] f32
      3. re-weighting scan of the LOCAL rows: three smallest (distance, global row) keys per probe         -> gather [W, B, 3] keys,
         merged by integer order (ties -> lowest global row, as the single-library scan)
      4. || m_test - row || for the 2nd and 3rd of them: computed by the owners of those rows              -> sum   [B, 2] f32"""
    B, Q, D = patch32.shape
    dev = patch32.device
    flat = patch32.reshape(B * Q, D)
    off, nloc = bank.f32_offset, bank.f32_rows
    if keys.dim() == 2:
        # best + runner-up: every candidate's squared fp32 distance is computed by the rank that owns its row (one sum over the
        # ranks, [2, B*Q]); the decision -- the nearer one, of equal ones the lower row -- is then the same everywhere
        d2_pair = torch.zeros((2, B * Q), dtype=torch.float32, device=dev)
        if nloc:
            ops.l2_rescore_pair_d2(flat, bank.f32, keys, d2_pair, off)
        d2_pair = yield ("sum", d2_pair)
        min_val = torch.zeros((B * Q,), dtype=torch.float32, device=dev)
        min_idx = torch.full((B * Q,), -1, dtype=torch.int64, device=dev)
        ops.l2_choose(keys, d2_pair, min_val, min_idx)
    else:
        min_val = torch.zeros((B * Q,), dtype=torch.float32, device=dev)
        scratch_idx = torch.full((B * Q,), -1, dtype=torch.int64, device=dev)
        ops.l2_rescore(flat, bank.f32, keys, min_val, scratch_idx, off) if nloc else None
        min_val = yield ("sum", min_val)
        min_idx = torch.where(keys == KEY_EMPTY, torch.full_like(keys, -1), keys & 0xFFFFFFFF)       # global rows, known everywhere
    s_star = torch.empty((B,), dtype=torch.float32, device=dev)
    s_idx = torch.empty((B,), dtype=torch.int32, device=dev)
    m_test = torch.empty((B, D), dtype=torch.float32, device=dev)
    m_star = torch.zeros((B, D), dtype=torch.float32, device=dev)
    st = ops._stream()
    _call("cmdiad_score_head", ops._p(min_val), ops._p(min_idx), ops._p(flat), ops._p(bank.f32), B, Q, D, nloc, off,
          ops._p(s_star), ops._p(s_idx), ops._p(m_test), ops._p(m_star), st)
    m_star = yield ("sum", m_star)
    if nloc:
        top3_local = ops.reweight_scan(m_star, bank.f32, bank.blk16, row_offset=off)
    else:
        top3_local = torch.full((B, 3), KEY_EMPTY, dtype=torch.int64, device=dev)
    everyone = yield ("gather", top3_local)                                                      # [W, B, 3]
    top3 = everyone.permute(1, 0, 2).reshape(B, -1).sort(1).values[:, :3].contiguous()           # keys are non-negative int64
    knn_d = torch.zeros((B, 2), dtype=torch.float32, device=dev)
    _call("cmdiad_score_tail", ops._p(s_star), ops._p(m_test), ops._p(top3), ops._p(bank.f32), B, D, nloc, off, ops._p(knn_d), st)
    knn_d = yield ("sum", knn_d)
    s = torch.empty((B,), dtype=torch.float32, device=dev)
    _call("cmdiad_score_final", ops._p(s_star), ops._p(knn_d), B, D, ops._p(s), st)
    s_map = ops.bilinear_up(min_val.view(B, dims[0], dims[1]), gt_size)
    return dict(min_val=min_val.view(B, Q), min_idx=min_idx.view(B, Q), s_idx=s_idx, s_star=s_star, s=s,
                s_map_pre=s_map, top3=top3, knn_d=knn_d)


def _drive_collectives(gen, group):
    """Runs a `_sharded_score_steps` generator over a torch.distributed group (RCCL on the GPUs, gloo in the CPU tests)."""
    import torch.distributed as td
    world = td.get_world_size(group)
    try:
        kind, t = next(gen)
        while True:
            if kind == "sum":
                td.all_reduce(t, op=td.ReduceOp.SUM, group=group)
                res = t
            else:
                t = t.contiguous()
                flat = torch.empty((world * t.shape[0], *t.shape[1:]), dtype=t.dtype, device=t.device)   # (gloo wants the concatenated shape)
                td.all_gather_into_tensor(flat, t, group=group)
                res = flat.view(world, *t.shape)
            kind, t = gen.send(res)
    except StopIteration as done:
        return done.value


def score_patches_from_keys(patch32, keys, bank, dims, gt_size=224, group=None):
    """Everything after the (merged) nearest-neighbour keys: exact re-score, s*, re-weighting, score map.  `group` is needed only
    when the library's fp32 rows are sharded as well (Bank(replicate_f32=False))."""
    if getattr(bank, "f32_sharded", False):
        if group is None:
            raise ValueError("score_patches_from_keys: the library's fp32 rows are sharded -- the process group is needed")
        return _drive_collectives(_sharded_score_steps(patch32, keys, bank, dims, gt_size), group)
    B, Q, D = patch32.shape
    dev = patch32.device
    flat = patch32.reshape(B * Q, D)
    # a key that names no row of the library (only possible when EVERY shard was empty) leaves (0, -1) behind, not garbage
    min_val = torch.zeros((B * Q,), dtype=torch.float32, device=dev)
    min_idx = torch.full((B * Q,), -1, dtype=torch.int64, device=dev)
    ops.l2_rescore(flat, bank.f32, keys, min_val, min_idx, 0)
    s_star = torch.empty((B,), dtype=torch.float32, device=dev)
    s_idx = torch.empty((B,), dtype=torch.int32, device=dev)
    m_test = torch.empty((B, D), dtype=torch.float32, device=dev)
    m_star = torch.empty((B, D), dtype=torch.float32, device=dev)
    st = ops._stream()
    _call("cmdiad_score_head", ops._p(min_val), ops._p(min_idx), ops._p(flat), ops._p(bank.f32), B, Q, D, bank.rows,
          0, ops._p(s_star), ops._p(s_idx), ops._p(m_test), ops._p(m_star), st)
    top3 = ops.reweight_scan(m_star, bank.f32, bank.blk16)
    knn_d = torch.empty((B, 2), dtype=torch.float32, device=dev)
    _call("cmdiad_score_tail", ops._p(s_star), ops._p(m_test), ops._p(top3), ops._p(bank.f32), B, D, bank.rows, 0,
          ops._p(knn_d), st)
    s = torch.empty((B,), dtype=torch.float32, device=dev)
    _call("cmdiad_score_final", ops._p(s_star), ops._p(knn_d), B, D, ops._p(s), st)
    s_map = ops.bilinear_up(min_val.view(B, dims[0], dims[1]), gt_size)
    return dict(min_val=min_val.view(B, Q), min_idx=min_idx.view(B, Q), s_idx=s_idx, s_star=s_star, s=s,
                s_map_pre=s_map, top3=top3, knn_d=knn_d)


def score_patches_from_keys_pair(patch_a, keys_a, bank_a, dims_a, patch_b, keys_b, bank_b, dims_b, gt_size=224, group=None):
    """score_patches_from_keys for the TWO libraries of a scored batch (multiple_features.py:976-1003: xyz and rgb / fusion) with
    their re-weighting scans as ONE launch pair (ops.reweight_scan_pair: the small library's fixed cost runs beside the large
    library's stream); every output is what the two separate calls return (bit for bit, unless more than four of a probe's eight
    best approximate rows share one lane slot of the scan: csrc/scan.hip, cmdiad_reweight_scan_pair).  Falls back to them when a library's fp32
    rows are sharded, a batch exceeds 32 samples, or the feature widths differ."""
    B, Qa, D = patch_a.shape
    if (getattr(bank_a, "f32_sharded", False) or getattr(bank_b, "f32_sharded", False) or B > 32 or patch_b.shape[0] != B
            or patch_b.shape[2] != D or bank_a.rows == 0 or bank_b.rows == 0 or os.environ.get("CMDIAD_SCAN_PAIR", "1") == "0"):
        return (score_patches_from_keys(patch_a, keys_a, bank_a, dims_a, gt_size, group),
                score_patches_from_keys(patch_b, keys_b, bank_b, dims_b, gt_size, group))
    dev = patch_a.device
    st = ops._stream()
    heads = []
    for patch32, keys, bank in ((patch_a, keys_a, bank_a), (patch_b, keys_b, bank_b)):
        Q = patch32.shape[1]
        flat = patch32.reshape(B * Q, D)
        min_val = torch.zeros((B * Q,), dtype=torch.float32, device=dev)
        min_idx = torch.full((B * Q,), -1, dtype=torch.int64, device=dev)
        ops.l2_rescore(flat, bank.f32, keys, min_val, min_idx, 0)
        s_star = torch.empty((B,), dtype=torch.float32, device=dev)
        s_idx = torch.empty((B,), dtype=torch.int32, device=dev)
        m_test = torch.empty((B, D), dtype=torch.float32, device=dev)
        m_star = torch.empty((B, D), dtype=torch.float32, device=dev)
        _call("cmdiad_score_head", ops._p(min_val), ops._p(min_idx), ops._p(flat), ops._p(bank.f32), B, Q, D, bank.rows,
              0, ops._p(s_star), ops._p(s_idx), ops._p(m_test), ops._p(m_star), st)
        heads.append((min_val, min_idx, s_star, s_idx, m_test, m_star, Q))
    tops = ops.reweight_scan_pair(heads[0][5], bank_a.f32, bank_a.blk16, heads[1][5], bank_b.f32, bank_b.blk16)
    out = []
    for (min_val, min_idx, s_star, s_idx, m_test, m_star, Q), top3, bank, dims in zip(heads, tops, (bank_a, bank_b), (dims_a, dims_b)):
        knn_d = torch.empty((B, 2), dtype=torch.float32, device=dev)
        _call("cmdiad_score_tail", ops._p(s_star), ops._p(m_test), ops._p(top3), ops._p(bank.f32), B, D, bank.rows, 0,
              ops._p(knn_d), st)
        s = torch.empty((B,), dtype=torch.float32, device=dev)
        _call("cmdiad_score_final", ops._p(s_star), ops._p(knn_d), B, D, ops._p(s), st)
        s_map = ops.bilinear_up(min_val.view(B, dims[0], dims[1]), gt_size)
        out.append(dict(min_val=min_val.view(B, Q), min_idx=min_idx.view(B, Q), s_idx=s_idx, s_star=s_star, s=s,
                        s_map_pre=s_map, top3=top3, knn_d=knn_d))
    return out[0], out[1]


class Extraction:
    """Device-resident outputs of one extract() call (everything Features.__call__ returns, plus the
    interpolation indices/weights that replace the reference's 154 MB interpolated tensor)."""
    __slots__ = ("rgb_tokens", "xyz_feats", "center", "ori_idx", "center_idx", "idx3", "w3", "pix2pt", "n_valid",
                 "xyz", "nz", "size")


class Engine:
    def __init__(self, vit, pointmae, size=224):
        """vit: runtime.PackedViT (or a callable returning one); pointmae: runtime.PackedPointMAE."""
        self._vit, self._pm, self.size = vit, pointmae, size

    @property
    def vit(self):
        return self._vit() if callable(self._vit) else self._vit

    @property
    def pm(self):
        return self._pm() if callable(self._pm) else self._pm

    def extract(self, rgb, organized_pc=None, xyz=None, nz=None, want_rgb=True, want_xyz=True, n_max=None,
                side_stream=None, rgb_hook=None, xyz_hook=None):
        """rgb [B,3,S,S] f32 cuda; the cloud either organised ([B,3,S,S], zeros = background) or already
        unorganised (xyz [B,N,3] + nz [B,N] int32 pixel indices, B == 1 or equal N).
        With `side_stream` the point-cloud branch (unorganise, FPS, kNN-group, Point-MAE, 3-NN) runs on that
        HIP stream concurrently with the ViT on the current stream: FPS is a latency-bound chain on B
        workgroups (B of 256 CUs), the ViT GEMMs fill the rest of the chip.  `rgb_hook(ex)` (optional) runs on the current
        stream right after the ViT and BEFORE the point-cloud branch is joined: work that needs only the rgb tokens (e.g. the
        rgb library search) then runs beside the rest of that branch.  `xyz_hook(ex)` (optional) runs at the END of the
        point-cloud branch, on its stream: work that needs only that branch (patch pooling, the 16-bit queries, the row
        de-duplication plan) then runs beside the ViT's last layers and the rgb library search instead of after them; it returns
        the tensors it allocated (they are consumed on the current stream after the join)."""
        ex = Extraction()
        ex.size = self.size
        if not want_xyz:
            ex.rgb_tokens = self.vit.forward_tokens(rgb) if want_rgb else None
            return ex
        cur = torch.cuda.current_stream()
        ctx = torch.cuda.stream(side_stream) if side_stream is not None else _Null()
        if side_stream is not None:
            side_stream.wait_stream(cur)

        def cloud():
            if organized_pc is not None:
                ex.xyz, ex.nz, ex.pix2pt, ex.n_valid = ops.unorganize(organized_pc.contiguous(), n_max)
            else:
                B, N, _ = xyz.shape
                ex.xyz, ex.nz, ex.n_valid = xyz.contiguous(), nz, None
                ex.pix2pt = torch.full((B, self.size * self.size), -1, dtype=torch.int32, device=xyz.device)
                ex.pix2pt.scatter_(1, nz.long(), torch.arange(N, dtype=torch.int32, device=xyz.device).expand(B, N))

        def points(sampled=None):
            ex.xyz_feats, ex.center, ex.ori_idx, ex.center_idx = self.pm.forward(ex.xyz, ex.n_valid, sampled=sampled)
            ex.idx3, ex.w3 = ops.interp3nn(ex.xyz, ex.center, ex.n_valid)
            return list(xyz_hook(ex) or ()) if xyz_hook is not None else []

        # EAGER launches with both branches (the drop-in's micro-batches): the host queues one kernel at a time, so the order of
        # queueing is the order of starting.  Farthest-point sampling goes first -- a chain of 1 023 dependent rounds on one CU per
        # cloud that nothing can shorten -- then the ViT's ~90 launches (they fill the chip under it), then the rest of the
        # point-cloud branch.  Queued branch after branch (rounds 1-5) the ViT's first kernel reached the device when FPS had
        # finished: 83 % of a micro-batch ran with ONE kernel in flight (rocprofv3 trace, profiles/r6_notes.md section 11).
        # Under stream capture the order of queueing is irrelevant (the graph holds the dependencies) and stays as it was.
        fps_first = (side_stream is not None and want_rgb and not torch.cuda.is_current_stream_capturing()
                     and os.environ.get("CMDIAD_FPS_FIRST", "1") != "0")      # (=0: branch after branch, for A/B runs)
        if fps_first:
            with ctx:
                cloud()
                sampled = self.pm.sample(ex.xyz, ex.n_valid)
            ex.rgb_tokens = self.vit.forward_tokens(rgb)
            with ctx:
                hooked = points(sampled)
        else:
            with ctx:
                cloud()
                hooked = points()
            ex.rgb_tokens = self.vit.forward_tokens(rgb) if want_rgb else None
        if rgb_hook is not None and want_rgb:
            rgb_hook(ex)
        if side_stream is not None:
            cur.wait_stream(side_stream)
            for t in (ex.xyz, ex.nz, ex.pix2pt, ex.n_valid, ex.xyz_feats, ex.center, ex.ori_idx, ex.center_idx, ex.idx3, ex.w3, *hooked):
                if t is not None:
                    t.record_stream(cur)  # allocated on the side stream, consumed on the current one
        return ex

    def xyz_patch(self, ex, P=56, mean=0.0, inv_std=1.0, want_bf16=False):
        """[B, P*P, 768] f32 (features.py:169-184), optionally normalised on the fly."""
        return ops.xyz_patch_fused(ex.xyz_feats, ex.idx3, ex.w3, ex.pix2pt, ex.size, P, mean, inv_std,
                                   want_f32=True, want_bf16=want_bf16)[0]

    @staticmethod
    def rgb_patch(ex):
        """[B,784,768] view of the ViT tokens without cls (features.py:160-162)."""
        return ex.rgb_tokens[:, 1:]

    @staticmethod
    def rgb_patch56(ex):
        """features.py:165-166: AdaptiveAvgPool2d 28->56 == exact 2x nearest replication."""
        p = ex.rgb_tokens[:, 1:]
        B, T, C = p.shape
        s = int(math.isqrt(T))
        return p.reshape(B, s, 1, s, 1, C).expand(B, s, 2, s, 2, C).reshape(B, 4 * T, C)


def predict_batch(engine, rgb, pcs, bank_xyz, bank_second, stats, det, seg, **kw):
    """Batched `predict` (image scores [B] f64, pixel maps [B,gt,gt] f64): see cmdiad_amd.predictor.BatchPredictor, which
    also keeps the HIP graphs and buffer sets between calls."""
    from . import predictor
    return predictor.predict_batch(engine, rgb, pcs, bank_xyz, bank_second, stats, det, seg, **kw)


def normalize(x, mean, std):
    """(x - mean) / std with the scalar library statistics (multiple_features.py:976-977) -> f32 copy."""
    shape = x.shape
    _, out, _ = ops.normalize_cast(x.reshape(-1, shape[-1]).contiguous(), float(mean), 1.0 / float(std), want_f32=True,
                                   want_sq=False)
    return out.view(shape)
