"""Batched `predict` of the two-library method classes, device-resident end to end (`engine.predict_batch`).

What `DoubleRGBPointFeatures.predict` (multiple_features.py:929-1015) and `RGBorXYZWithOneHallucination.predict`
with main modality xyz (multiple_features.py:486-573) do for ONE sample -- extract, patch, normalise with the
(cross-wired) scalar statistics, nearest neighbour in both libraries, re-weighting, bilinear maps, 8-bit
Gaussian blur, lambda weights, the two linear one-class SVMs -- done for a batch of B samples per call, every
sample's result being what the B = 1 drop-in returns (tests/test_gpu_predictor.py).

One step = stage 1 (extraction + 16-bit queries; a HIP graph) -> search (distance GEMMs, eager so that the
collectives of the row-sharded mode and the HIP-event bracket of bench.py's `roofline` can sit there) -> stage 2
(exact re-score, re-weighting, maps, blur, SVM scores; a HIP graph) -> D2H of the final image scores and pixel maps
into a ring of pinned buffers.  Two complete buffer sets alternate: the scoring tail of step i runs on a second
stream beside the extraction of step i+1, and the inputs of step i+1 are copied (H2D from pinned host memory, or
D2D from resident batches) into the other set's static input buffers on a copy stream while step i computes.
"""
import os
import sys

import numpy as np
import torch

from . import engine as eng
from . import ops


class EventTimer:
    """HIP-event bracket on torch's current stream (the stream every cmdiad kernel is launched on)."""

    def __init__(self):
        self.pairs = []

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)
        self.e0.record()
        return self

    def __exit__(self, *a):
        self.e1.record()
        self.pairs.append((self.e0, self.e1))

    def mean_ms(self, skip=0):
        v = [a.elapsed_time(b) for a, b in self.pairs[skip:]]
        return sum(v) / max(len(v), 1)


class _SharedPlan:
    """The compacted 16-bit queries of one library search whose row plan (live count, slot map) is shared with another search."""
    __slots__ = ("q16", "q_sq", "count", "slot")

    def __init__(self, q16, q_sq, count, slot):
        self.q16, self.q_sq, self.count, self.slot = q16, q_sq, count, slot


class _NoTimer:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class Ticket:
    """One submitted batch: wait() -> (image scores [B] f64, pixel maps [B, gt, gt] f64) as numpy arrays.
    `flag` / `redo` (row-sharded search only): a pinned int32 that the step set to non-zero when some rank's live query rows did
    not fit the sticky gather cap (engine.ShardedSearch) -- the step's keys are then incomplete and wait() repeats the batch through
    `redo` (the same decision on every rank: the flag is computed from the all-gathered counts)."""

    def __init__(self, host_s, host_m, event, gt, flag=None, redo=None):
        self._s, self._m, self._ev, self._gt = host_s, host_m, event, gt
        self._flag, self._redo = flag, redo
        self._out = None

    @property
    def taken(self):
        """True once wait() has copied the results out of the pinned ring slot this ticket aliases."""
        return self._out is not None

    def wait(self):
        if self._out is None:
            self._ev.synchronize()
            B = self._s.shape[0]
            if self._flag is not None and int(self._flag.item()) != 0:
                self._out = self._redo()
            else:
                self._out = (self._s.numpy().reshape(B).copy(), self._m.numpy().reshape(B, self._gt, self._gt).copy())
            self._s = self._m = self._flag = self._redo = None
        return self._out


class BatchPredictor:
    """engine: eng.Engine.  bank_xyz / bank_second: eng.Bank (the main-modality library and the rgb -- or, for the
    'mtfi' workload, the hallucinated-feature -- library).  stats: dict(xyz_mean, xyz_std, rgb_mean, rgb_std) with the
    second library's statistics under the rgb_* keys.  det / seg: fitted sklearn SGDOneClassSVM (only coef_ / offset_
    are read; the fit stays on the host).  lambdas = (xyz_s, xyz_smap, second_s, second_smap)
    (main.py:114-125).  workload: 'dino_pointmae' (both modalities extracted) or 'mtfi' (Point-MAE extraction +
    `halluc`: runtime.PackedHallucination generating the second modality's features from the xyz patches).
    group: torch.distributed process group -> row-sharded library search (all-gather of the queries, per-shard
    distance GEMM, one integer-MIN all-reduce of packed keys); None -> every rank searches its own copy."""

    def __init__(self, engine, bank_xyz, bank_second, stats, det, seg, lambdas=(1.0, 1.0, 0.1, 0.1), batch=32, n_max=None,
                 workload="dino_pointmae", halluc=None, group=None, use_graph=True, timers=None, ring=3, size=224,
                 gt_size=224, blur_radius=4.0):
        if workload not in ("dino_pointmae", "mtfi"):
            raise ValueError(f"unknown workload {workload!r}")
        if workload == "mtfi" and halluc is None:
            raise ValueError("workload 'mtfi' needs the packed hallucination network")
        self.e, self.bank_xyz, self.bank_second, self.stats = engine, bank_xyz, bank_second, stats
        self.det, self.seg, self.lambdas = det, seg, tuple(float(v) for v in lambdas)
        self.B, self.n_max, self.workload, self.halluc, self.group = batch, n_max, workload, halluc, group
        self.size, self.gt, self.blur_radius = size, gt_size, float(blur_radius)
        self.timers = timers or {}
        self.use_graph = use_graph
        dev = bank_xyz.f32.device
        self.dev = dev
        # CMDIAD_STREAM_PRIO="side,post" (A/B runs): stream priorities of the point-cloud branch and of the search / scoring stage
        # (0 = default, -1 = high; the ViT branch runs on the caller's stream)
        prio = [int(v) for v in os.environ.get("CMDIAD_STREAM_PRIO", "0,0").split(",")]
        # HIP multiplexes the streams over a few hardware queues, and the copy stream shares the main stream's: a copy there runs
        # BETWEEN two steps, never under one (profiles/r6_notes.md, rocprofv3 kernel + memory-copy traces) -- which is why submit()
        # can put a later batch's H2D copy on the post stream instead (`stage`).  Giving the copy stream a priority (= a queue)
        # of its own, or more hardware queues (GPU_MAX_HW_QUEUES=8), changes how ALL streams share the queues and costs 12-18 %
        # of the step (measured, same notes): the default mapping stays.
        copy_prio = int(os.environ.get("CMDIAD_COPY_PRIO", "0"))
        self.side, self.post, self.copy = (ops.shared_stream(dev, "predictor.side", prio[0]), ops.shared_stream(dev, "predictor.post", prio[1]),
                                           ops.shared_stream(dev, "predictor.copy", copy_prio))
        # host ring: the step's FINAL outputs (image score, pixel map), f64 as sklearn's score_samples returns them
        self.ring = [(torch.empty((batch, 1), dtype=torch.float64, pin_memory=True),
                      torch.empty((batch, gt_size * gt_size), dtype=torch.float64, pin_memory=True)) for _ in range(ring)]
        # row-sharded search: one pinned flag per ring slot ("this step's live rows exceeded the gather cap on some rank")
        self.flag_ring = [torch.zeros((1,), dtype=torch.int32, pin_memory=True) for _ in range(ring)] if group is not None else None
        self.slot = 0
        self.tickets = [None] * ring     # the ticket that aliases each pinned slot (submit refuses to overwrite an unread one)
        self.step_no = 0
        self.sets = None
        self.static = {}
        # exact removal of the repeated background rows in front of the xyz search (csrc/dedup.hip); CMDIAD_DEDUP=0 searches every row
        self.dedup = os.environ.get("CMDIAD_DEDUP", "1") != "0"
        # row-sharded search: rows of every rank that travel per step -- "auto" (sticky cap, no host read in steady state) | "exact"
        self.shard_cap = os.environ.get("CMDIAD_SHARD_CAP", "auto")
        if self.shard_cap.strip().isdigit():
            self.shard_cap = int(self.shard_cap)      # a fixed row count: ShardedSearch takes it as an int
        self.stage2_eager = bool(getattr(bank_xyz, "f32_sharded", False) or getattr(bank_second, "f32_sharded", False))
        if self.stage2_eager and group is None:
            raise ValueError("BatchPredictor: a library with sharded fp32 rows (Bank(replicate_f32=False)) needs the process group")
        # Sharded fp32 rows = SURVEY 8(e)'s partitioning to the letter, whose queries are REPLICATED: the owner of a winning row adds
        # its part to a sum over the ranks, so every rank must be scoring the SAME batch (submit() checks a fingerprint of the
        # inputs across the ranks).  Ranks that score DIFFERENT images use the default (replicated fp32 rows, no collective in the tail).
        self._same_batch_check = self.stage2_eager and group is not None
        self.live_rows = torch.zeros((1,), dtype=torch.int64, device=dev)   # rows actually searched, summed over the xyz searches
        self.xyz_searches = 0
        self._raw_norm = None
        self.shard_stats = {}        # per library: what the last row-sharded search exchanged (engine.ShardedSearch)
        self.step_flags = []         # device flags of the current step's row-sharded searches ("live rows exceeded the gather cap")
        self.redone = 0              # steps repeated because of such a flag
        self.inputs = [self._new_inputs() for _ in range(2 if use_graph else 1)]

    def _new_inputs(self):
        want_rgb = self.workload == "dino_pointmae"
        return dict(rgb=torch.zeros((self.B, 3, self.size, self.size), dtype=torch.float32, device=self.dev) if want_rgb else None,
                    pcs=torch.zeros((self.B, 3, self.size, self.size), dtype=torch.float32, device=self.dev),
                    ready=None, free=None, staged=None)

    # ---- stage 1: everything up to the 16-bit queries of both libraries
    def stage1(self, inp):
        e, s = self.e, self.stats
        if self.workload == "mtfi":
            ex = e.extract(None, inp["pcs"], want_rgb=False, n_max=self.n_max)
            xyz_raw = e.xyz_patch(ex, 56)                                       # a9  [B,3136,768] f32
            B, Q, D = xyz_raw.shape
            xyz_q = eng.normalize(xyz_raw, s["xyz_mean"], s["xyz_std"])         # a11 (every row: the exact re-score reads them)
            if self.dedup and self.group is None and os.environ.get("CMDIAD_MTFI_ROWPLAN", "1") != "0":
                # (CMDIAD_MTFI_ROWPLAN=0: the MLP on every row and one plan per search, round 2's form, for A/B runs.)
                # Patches without a foreground pixel are ONE row of the raw xyz features (bit for bit), so their hallucinated
                # features are one row too: the rows are de-duplicated ONCE, on the fp32 bit patterns (the plan kernels compare
                # rows as opaque 16-bit words: an fp32 row is 2 D of them), the distillation MLP (a15) and both 16-bit query sets
                # are computed on the compacted rows only -- the GEMMs read the live row count on the device -- and the one plan
                # serves both library searches.  Outputs are bit-identical to running every row (CMDIAD_DEDUP=0).
                flat = xyz_raw.reshape(B * Q, D)
                if self._raw_norm is None or self._raw_norm.shape[0] != B * Q:
                    self._raw_norm = torch.zeros((B * Q,), dtype=torch.float32, device=flat.device)
                plan = ops.rows_dedup_plan(flat.view(torch.float16), self._raw_norm)
                raw_c = plan.q16.view(torch.float32)                            # [B*Q, D] f32, first plan.count rows live
                hall_c = self.halluc.generate(raw_c, "xyz", m_count=plan.count)  # a15 on the live rows
                out = {}
                for name, rows_c, mean, std in (("xyz", raw_c, s["xyz_mean"], s["xyz_std"]), ("rgb", hall_c, s["rgb_mean"], s["rgb_std"])):
                    q16_c, q32_c, qsq_c = ops.normalize_cast(rows_c, float(mean), 1.0 / float(std), want_f32=(name == "rgb"))
                    out[f"{name}_plan"] = _SharedPlan(q16_c, qsq_c, plan.count, plan.slot)
                    full = xyz_q if name == "xyz" else ops.rows_expand_f32(q32_c, plan.slot).view(B, Q, D)
                    out[name] = (full, None, None)
                return out
            hall = self.halluc.generate(xyz_raw, "xyz")                         # a15: hallucinated rgb features [B,3136,768]
            sec_q = eng.normalize(hall, s["rgb_mean"], s["rgb_std"])
            out = {}
            for name, q in (("xyz", xyz_q), ("rgb", sec_q)):
                q16, _, qsq = ops.normalize_cast(q.reshape(B * Q, D))
                out[name] = (q, q16, qsq)
            return out
        early = {}

        def rgb_branch(ex):
            # everything the rgb library needs depends on the ViT only: normalise, cast and SEARCH it here, beside the
            # rest of the point-cloud branch (the Point-MAE transformer leaves half of the chip's issue slots idle)
            # the patch rows of the [B, 785, C] tokens are read in place (cls skipped): normalised fp32 rows, 16-bit rows and norms
            # in ONE pass -- (x - mean) / std then the cast, the same two roundings as normalising first and casting after
            tok = ex.rgb_tokens
            B, Q, D = tok.shape[0], tok.shape[1] - 1, tok.shape[2]
            q16, rq, qsq = ops.normalize_cast(tok, float(s["rgb_mean"]), 1.0 / float(s["rgb_std"]), want_f32=True, skip_leading=1)
            rq = rq.view(B, Q, D)
            early["rgb"] = (rq, q16, qsq)
            if self.group is None:
                bank = self.bank_second
                k = ops.new_keys(B * Q, rq.device, runner=True)
                ops.l2_min_keys(q16, qsq, bank.bf16, bank.sqnorm, k, bank.row_offset)
                early["rgb_keys"] = k

        def xyz_branch(ex):
            # the tail of the point-cloud branch, on ITS stream: patch pooling, 16-bit queries and the row de-duplication plan need
            # nothing from the ViT and run beside its last layers and the rgb library search (0.65 ms that used to follow them)
            xyz_q = e.xyz_patch(ex, 56, s["xyz_mean"], 1.0 / s["xyz_std"])        # a9 + a11 fused
            B, Q, D = xyz_q.shape
            q16, _, qsq = ops.normalize_cast(xyz_q.reshape(B * Q, D))
            early["xyz"] = (xyz_q, q16, qsq)
            made = [xyz_q, q16, qsq]
            if self.dedup and self.group is None:
                plan = early["xyz_plan"] = ops.rows_dedup_plan(q16, qsq)
                made += [plan.slot, plan.rows, plan.count, plan.q16, plan.q_sq, plan.work]
            return made

        if os.environ.get("CMDIAD_XYZ_TAIL_SIDE", "1") != "0":
            e.extract(inp["rgb"], inp["pcs"], n_max=self.n_max, side_stream=self.side, rgb_hook=rgb_branch, xyz_hook=xyz_branch)
        else:   # A/B: the tail of the point-cloud branch after the join, behind the rgb search
            xyz_branch(e.extract(inp["rgb"], inp["pcs"], n_max=self.n_max, side_stream=self.side, rgb_hook=rgb_branch))
        out = {"xyz": early["xyz"], "rgb": early["rgb"]}
        for k in ("rgb_keys", "xyz_plan"):
            if k in early:
                out[k] = early[k]
        return out

    # ---- search: eager (HIP events around the distance GEMM; RCCL collectives when sharded)
    def search(self, qs, buf=0):
        keys = {}
        for name, bank in (("xyz", self.bank_xyz), ("rgb", self.bank_second)):
            if name == "rgb" and "rgb_keys" in qs:  # searched inside stage 1 already (beside the point-cloud branch)
                keys[name] = qs["rgb_keys"]
                continue
            q, q16, qsq = qs[name]
            B, Q, D = q.shape
            if self.group is not None and self.dedup:
                # row-sharded library: compact locally, all-gather the live rows only (engine.sharded_min_keys)
                # one ShardedSearch per (library, buffer set): its gather cap is sticky, so a steady-state step reads nothing on the host
                st = self.shard_stats.setdefault(name, {})
                ss = self.static.get(f"ss_{name}_{buf}")
                if ss is None:
                    ss = self.static[f"ss_{name}_{buf}"] = eng.ShardedSearch(bank, self.group, stats=st, cap_rows=self.shard_cap)
                k = ss.gather(q16, qsq).gemm(self.timers.get(name)).reduce()
                self.step_flags.append(ss.overflow)
                if name == "xyz":
                    self.live_rows += ss.counts_dev.sum()
                    self.xyz_searches += 1
                keys[name] = k
                continue
            plan = qs.get(f"{name}_plan") if self.dedup else None    # made in stage 1 already (unsharded library)
            q_all, s_all = (None, None) if plan is not None else eng.gather_queries(q16, qsq, self.group)
            n_rows = B * Q if q_all is None else q_all.shape[0]
            k = self.static.get(f"keys_{name}_{buf}")       # [2, rows]: best + runner-up (ops.new_keys(runner=True))
            if k is None or k.shape[1] != n_rows:
                k = self.static[f"keys_{name}_{buf}"] = torch.empty((2, n_rows), dtype=torch.int64, device=q.device)
            if self.dedup:
                # patches without a foreground pixel are one and the same row (and so are their hallucinated features in the MTFI
                # workload): searched once, the key copied to all
                if plan is None:
                    plan = self.static[f"plan_{name}_{buf}"] = ops.rows_dedup_plan(q_all, s_all, self.static.get(f"plan_{name}_{buf}"))
                kc = self.static.get(f"keysc_{name}_{buf}")
                if kc is None or kc.shape[1] != n_rows:
                    kc = self.static[f"keysc_{name}_{buf}"] = torch.empty((2, n_rows), dtype=torch.int64, device=q.device)
                kc.fill_(eng.KEY_EMPTY)
                with self.timers.get(name, _NoTimer()):
                    ops.l2_min_keys_counted(plan.q16, plan.q_sq, plan.count, bank.bf16, bank.sqnorm, kc, bank.row_offset)
                ops.keys_expand(kc, plan.slot, k)
                live = plan.count
            else:
                k.fill_(eng.KEY_EMPTY)
                with self.timers.get(name, _NoTimer()):
                    ops.l2_min_keys(q_all, s_all, bank.bf16, bank.sqnorm, k, bank.row_offset)
                live = q_all.shape[0]
            if name == "xyz":
                self.live_rows += live
                self.xyz_searches += 1
            k = eng.merge_shard_keys(k, self.group)
            keys[name] = k[..., bank.rank * B * Q:(bank.rank + 1) * B * Q] if self.group is not None else k
        return keys

    # ---- stage 2: exact re-score, re-weighting, bilinear maps, 8-bit blur (a14), lambda weights + one-class SVMs (a19)
    def stage2(self, qs, keys):
        lam = self.lambdas
        gt = self.gt
        side = (56, 56) if self.workload == "mtfi" else (28, 28)
        rx, rr = eng.score_patches_from_keys_pair(qs["xyz"][0], keys["xyz"].contiguous(), self.bank_xyz, (56, 56),
                                                  qs["rgb"][0], keys["rgb"].contiguous(), self.bank_second, side, gt, self.group)
        s = torch.stack([rx["s"], rr["s"]], 1)                                   # [B,2]
        maps = torch.stack([rx["s_map_pre"], rr["s_map_pre"]], 1).contiguous()   # [B,2,gt,gt]
        B = maps.shape[0]
        blurred = ops.blur8_maps(maps.view(B * 2, gt, gt), self.blur_radius).view(B, 2, gt * gt)
        pix = ops.ocsvm_score_maps(blurred, (lam[1], lam[3]), self.seg.coef_, self.seg.offset_)
        img = ops.ocsvm_score_maps(s.view(B, 2, 1).contiguous(), (lam[0], lam[2]), self.det.coef_, self.det.offset_)
        return img, pix

    def _capture(self):
        qs = self.stage1(self.inputs[0])  # one eager pass first: module loading / attribute setting must not happen in capture
        self.stage2(qs, self.search(qs, 0))
        torch.cuda.synchronize()
        # capture_error_mode="thread_local": with a process group alive, RCCL's watchdog THREAD polls the events of earlier
        # collectives (hipEventQuery); under the default "global" mode that call from another thread invalidates this thread's
        # capture (hipErrorStreamCaptureInvalidated -> eager fallback) and raises inside the watchdog, which aborts the process
        # (tools/fuzz_pipeline.py met both within seconds; profiles/r5_notes.md section 15)
        caller_stream = torch.cuda.current_stream()
        try:
            sets = []
            for s in range(2):  # two complete buffer sets: step i+1's extraction overlaps step i's scoring tail
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                    qs = self.stage1(self.inputs[s])
                g1.replay()
                keys = self.search(qs, s)
                k = {n: v.contiguous() for n, v in keys.items()}
                g2 = None
                if self.stage2_eager:       # fp32 rows sharded too: the scoring tail holds collectives and runs eagerly
                    out = self.stage2(qs, k)
                else:
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                        out = self.stage2(qs, k)
                torch.cuda.synchronize()
                sets.append(dict(g1=g1, g2=g2, qs=qs, k=k, out=out, done=None))
            self.sets = sets
            self.step_flags = []      # the rehearsal's row-sharded searches are not a step
        except Exception as exc:  # capture is an optimisation, never a requirement
            print(f"[cmdiad_amd.predictor] HIP graph capture unavailable ({type(exc).__name__}: {exc}); running eagerly",
                  file=sys.stderr)
            self.sets = None
            self.use_graph = False
            # torch.cuda.graph.__exit__ ends the capture BEFORE it restores the stream context: when the end itself raises (an
            # invalidated capture), the capture stream stays current and every later launch fails on it -- put the caller's back
            torch.cuda.set_stream(caller_stream)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass

    def _load_inputs(self, inp, rgb, pcs, wait=True):
        """Copies one batch into a set's static input buffers on the copy stream (H2D when the source is pinned host memory,
        D2D when it is already resident); the compute stream waits for the copy (wait=False: the caller does, later, on
        inp["ready"]), the copy waits until the previous user of these buffers (stage 1 of two steps ago) has finished with them."""
        cur = torch.cuda.current_stream()
        if inp["free"] is not None:
            self.copy.wait_event(inp["free"])
        else:
            self.copy.wait_stream(cur)
        if inp.get("ready") is not None:
            # a batch staged into these buffers (submit(stage=), on the post stream) that turned out not to be the one submitted:
            # its copy may still be pending and must not land AFTER this one
            self.copy.wait_event(inp["ready"])
        with torch.cuda.stream(self.copy):
            if inp["rgb"] is not None:
                inp["rgb"].copy_(rgb, non_blocking=True)
            inp["pcs"].copy_(pcs, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        for t in (rgb, pcs):
            if t is not None and t.is_cuda:
                t.record_stream(self.copy)
        inp["ready"] = ev
        inp["staged"] = (rgb, pcs)
        if wait:
            cur.wait_event(ev)

    def submit(self, rgb, pcs, stage=None):
        """rgb [B,3,S,S] f32 (None for 'mtfi'), pcs [B,3,S,S] f32 organised clouds; on the GPU or in (pinned) host memory.
        Returns a Ticket.  A ticket aliases one of the len(ring) pinned output slots until its wait() has copied the results
        out: submitting into a slot whose ticket has not been waited for raises instead of overwriting that batch's results.
        stage = (rgb, pcs) of the submit AFTER THE NEXT (optional; host tensors, the very objects that submit will pass): their
        H2D copy is enqueued on the post stream behind this step's search and scoring tail, into THIS step's input buffers (free
        again: stage 1 has read them before the tail starts, and the step in between uses the other set).  The post stream has
        ~10 ms of slack per step, so the copy costs nothing; on the copy stream it ran BETWEEN two steps whatever the order of
        enqueueing, because that stream shares a hardware queue with the main stream and HIP serves a queue in order
        (rocprofv3 kernel + memory-copy traces, profiles/r6_notes.md).  features.py:127-128 start from host tensors: this is what
        brings the PCIe-inclusive rate to the resident one."""
        if pcs.shape[0] != self.B:
            raise ValueError(f"batch of {pcs.shape[0]} given to a predictor built for {self.B}")
        if self._same_batch_check:
            import torch.distributed as td
            fp = torch.stack([pcs.double().sum(), (rgb.double().sum() if rgb is not None else pcs.double().abs().sum())]).to(self.dev)
            both = torch.cat([fp, -fp])                      # one collective: max(x) and max(-x) = -min(x)
            td.all_reduce(both, op=td.ReduceOp.MAX, group=self.group)
            if not torch.equal(both[:2], -both[2:]):
                raise ValueError("BatchPredictor: libraries with sharded fp32 rows (Bank(replicate_f32=False)) score REPLICATED queries -- "
                                 "every rank must submit the same batch; ranks that score different images use the default Bank")
        old = self.tickets[self.slot]
        if old is not None and not old.taken:
            raise RuntimeError(f"BatchPredictor.submit: {len(self.ring)} tickets are outstanding; wait() for the oldest one first "
                               "(its results live in the pinned output slot this submit would overwrite)")
        if self.use_graph and self.sets is None:
            for inp in self.inputs:  # capture (and its eager rehearsal) must see real clouds, not the zero-filled buffers
                if inp["rgb"] is not None:
                    inp["rgb"].copy_(rgb)
                inp["pcs"].copy_(pcs)
            torch.cuda.synchronize()
            self._capture()
        host_s, host_m = self.ring[self.slot]
        self.slot = (self.slot + 1) % len(self.ring)
        cur = torch.cuda.current_stream()
        if self.use_graph:
            # Software pipeline across batches: the scoring tail of step i (re-score, re-weighting scans, maps, blur, one-class
            # SVMs, D2H: ~2 ms of small bandwidth-bound kernels) runs on a second stream beside the extraction of step i+1,
            # which leaves most of the chip idle while farthest-point sampling walks its chain.  Two buffer sets alternate; a
            # set is reused only after its own tail has finished (event wait below).
            which = self.step_no & 1
            st, inp = self.sets[which], self.inputs[which]
            self.step_no += 1
            staged = inp.get("staged")
            if not (staged is not None and staged[0] is rgb and staged[1] is pcs and inp.get("ready") is not None):
                self._load_inputs(inp, rgb, pcs, wait=False)        # not staged two submits ago: copy now, on the copy stream
            ready = inp["ready"]
            inp["staged"] = inp["ready"] = None
            cur.wait_event(ready)
            if st["done"] is not None:
                cur.wait_event(st["done"])
            st["g1"].replay()
            inp["free"] = torch.cuda.Event()
            inp["free"].record()
            # The library searches go to the second stream as well and run beside the NEXT step's extraction.  MTFI workload: without
            # a ViT beside it the extraction leaves the chip idle while FPS walks its chain on 32 CUs (25.9 -> 23.5 ms).  With the
            # ViT in stage 1 the same move gains 2.5 % of the step (23.5 -> 22.9 ms); the distance GEMM then shares the chip, so its
            # duration inside the pipeline says little about the kernel -- bench.py times the same launch again, alone, for
            # `roofline.launch_ms`.  CMDIAD_SEARCH_POST=0 keeps the searches on the main stream.
            search_on_post = os.environ.get("CMDIAD_SEARCH_POST", "1") == "1"
            if not search_on_post:
                keys = self.search(st["qs"], which)
                for n, k in keys.items():
                    if k.data_ptr() != st["k"][n].data_ptr():
                        st["k"][n].copy_(k)
            self.post.wait_stream(cur)
            with torch.cuda.stream(self.post):
                if search_on_post:
                    keys = self.search(st["qs"], which)
                    for n, k in keys.items():
                        if k.data_ptr() != st["k"][n].data_ptr():
                            st["k"][n].copy_(k)
                if st["g2"] is not None:
                    st["g2"].replay()
                else:
                    st["out"] = self.stage2(st["qs"], st["k"])
                s_dev, maps_dev = st["out"]
                host_s.copy_(s_dev, non_blocking=True)
                host_m.copy_(maps_dev, non_blocking=True)
                flag = self._flag_to_host()
                ev = torch.cuda.Event()
                ev.record()
                if stage is not None and not stage[1].is_cuda:
                    # the batch of the submit after the next, into this step's input set, behind this step's tail (see the docstring).
                    # Host batches only: a resident batch's D2D copy takes 20 us at the step boundary.
                    if inp["rgb"] is not None:
                        inp["rgb"].copy_(stage[0], non_blocking=True)
                    inp["pcs"].copy_(stage[1], non_blocking=True)
                    inp["ready"] = torch.cuda.Event()
                    inp["ready"].record()
                    inp["staged"] = stage
            st["done"] = ev
            return self._ticket(host_s, host_m, ev, flag, rgb, pcs)
        inp = self.inputs[0]
        self.step_no += 1
        self._load_inputs(inp, rgb, pcs)
        qs = self.stage1(inp)
        inp["free"] = torch.cuda.Event()
        inp["free"].record()
        s_dev, maps_dev = self.stage2(qs, self.search(qs, 0))
        host_s.copy_(s_dev, non_blocking=True)
        host_m.copy_(maps_dev, non_blocking=True)
        flag = self._flag_to_host()
        ev = torch.cuda.Event()
        ev.record()
        return self._ticket(host_s, host_m, ev, flag, rgb, pcs)

    def _flag_to_host(self):
        """Row-sharded search: OR of this step's overflow flags -> the pinned flag of the step's ring slot (asynchronous)."""
        if self.flag_ring is None or not self.step_flags:
            self.step_flags = []
            return None
        dev_flag = torch.stack(self.step_flags).any().to(torch.int32).view(1)
        self.step_flags = []
        host = self.flag_ring[(self.slot - 1) % len(self.ring)]
        host.copy_(dev_flag, non_blocking=True)
        return host

    def _redo(self, rgb, pcs):
        """A step whose live query rows exceeded the sticky gather cap on some rank: its keys are incomplete.  Every rank sees the
        same flag, so every rank repeats the batch here, in the same place of its submit / wait sequence: drain the device, let the
        searches read the live counts again (a larger cap from now on), run the batch eagerly."""
        torch.cuda.synchronize()
        self.redone += 1
        # Grow only the LIBRARIES whose rows did not fit -- but every buffer set's search of such a library: the repeat runs on set 0
        # whichever set the step used.  The flags the searches hold now may belong to LATER steps (two tickets are outstanding and
        # an earlier repeat re-gathered set 0), so they only save a launch: the repeat itself checks its own flags and grows again.
        searches = {k: v for k, v in self.static.items() if k.startswith("ss_")}

        def grow(names):
            for k, v in searches.items():
                if k.split("_")[1] in names:
                    v.regrow()

        grow({k.split("_")[1] for k, v in searches.items() if v.overflow is None or v.overflowed()})
        inp = self.inputs[0]
        self._load_inputs(inp, rgb, pcs)
        qs = self.stage1(inp)
        inp["free"] = torch.cuda.Event()
        inp["free"].record()
        for _ in range(3):
            keys = self.search(qs, 0)
            self.step_flags = []
            over = {k.split("_")[1] for k, v in searches.items() if k.endswith("_0") and v.overflowed()}
            if not over:
                break
            grow(over)
        else:   # a fixed CMDIAD_SHARD_CAP below the live rows of this batch: nothing to re-read
            raise RuntimeError("BatchPredictor: the repeated batch still overflows the gather cap (CMDIAD_SHARD_CAP too small?); "
                               "its keys would be incomplete")
        s_dev, maps_dev = self.stage2(qs, keys)
        B = s_dev.shape[0]
        return s_dev.cpu().numpy().reshape(B).copy(), maps_dev.cpu().numpy().reshape(B, self.gt, self.gt).copy()

    def _ticket(self, host_s, host_m, ev, flag=None, rgb=None, pcs=None):
        t = Ticket(host_s, host_m, ev, self.gt, flag, (lambda: self._redo(rgb, pcs)) if flag is not None else None)
        self.tickets[(self.slot - 1) % len(self.ring)] = t
        return t

    def predict_batch(self, rgb, pcs):
        """-> (image scores [B] f64, pixel maps [B,gt,gt] f64): detect_fuser / seg_fuser.score_samples of every sample."""
        return self.submit(rgb, pcs).wait()


def predict_batch(engine, rgb, pcs, bank_xyz, bank_second, stats, det, seg, **kw):
    """One-shot convenience form (no graphs are kept): builds a BatchPredictor for this batch size and runs it once."""
    kw.setdefault("use_graph", False)
    p = BatchPredictor(engine, bank_xyz, bank_second, stats, det, seg, batch=pcs.shape[0], **kw)
    return p.predict_batch(rgb, pcs)
