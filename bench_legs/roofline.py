"""`roofline` of the JSON line: the dominant kernel's own duration (HIP events around the step's launch repeated alone) and the
fabric traffic of the committed PMC pass."""
import json
import os

from .common import REPO

def isolated_xyz_search_ms(pred, iters=12):
    """The xyz-library distance GEMM of the LAST step once more, alone on an idle chip (same operands: the step's compacted query
    rows and live count, the same library operand, the same launch), HIP events around each launch: the kernel's own duration.
    Inside the pipelined step the searches run on the second stream beside the next step's extraction, where the measured
    duration also contains the time the kernel spends sharing the CUs (`roofline.launch_ms_in_pipeline`)."""
    import torch
    from cmdiad_amd import engine as eng
    from cmdiad_amd import ops
    from cmdiad_amd.predictor import EventTimer
    torch.cuda.synchronize()
    bank = pred.bank_xyz
    t = EventTimer()
    ss = pred.static.get("ss_xyz_0")
    qs = pred.sets[0]["qs"] if pred.sets else None
    for _ in range(iters):
        if ss is not None:                       # row-sharded: the segments launch over the gathered live rows of all ranks
            ss.gemm(t)
        elif qs is not None and qs.get("xyz_plan") is not None:
            plan = qs["xyz_plan"]
            kc = ops.new_keys(plan.q16.shape[0], plan.q16.device, runner=True)       # as the pipeline launches it: best + runner-up
            with t:
                ops.l2_min_keys_counted(plan.q16, plan.q_sq, plan.count, bank.bf16, bank.sqnorm, kc, bank.row_offset)
        elif qs is not None:
            _, q16, qsq = qs["xyz"]
            k = ops.new_keys(q16.shape[0], q16.device, runner=True)
            with t:
                ops.l2_min_keys(q16, qsq, bank.bf16, bank.sqnorm, k, bank.row_offset)
        else:
            return None
        torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in t.pairs[2:])      # the first two launches follow the pipeline's last steps: skipped
    return v[len(v) // 2]                                        # median of ten: one launch beside a late D2H copy must not move it


# --------------------------------------------------------------------------------------------------------- secondary legs


def profiled_traffic(repo=None):
    """roofline.traffic: fabric-side bytes per launch of the dominant kernel (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md).  PMC
    counters cannot be read inside this process; the figure comes from the newest committed rocprofv3 --pmc pass
    (profiles/rN_pmc.json, tools/profile_round.sh) and is emitted ONLY while the kernel's source is byte-identical to the one that
    was profiled (profiles/rN_pmc_meta.json holds the sha256 of csrc/l2min.hip + gemm_core.h at that time): null as soon as the
    kernel changes."""
    import glob
    import hashlib
    import re
    here = repo or REPO
    metas = sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_meta.json")),
                   key=lambda f: int(re.search(r"r(\d+)_pmc_meta", f).group(1)), reverse=True)
    note = "no committed PMC pass"
    for meta_path in metas:
        tag = re.search(r"(r\d+)_pmc_meta", meta_path).group(1)
        try:
            meta = json.load(open(meta_path))
            h = hashlib.sha256()
            for f in meta["sources"]:
                h.update(open(os.path.join(here, f), "rb").read())
            if h.hexdigest() != meta["sha256"]:
                note = f"the distance GEMM's source changed since profiles/{tag}_pmc.json was taken: re-profile"
                continue
            if meta.get("standalone"):      # the launch ALONE on the chip: the regime `frac` / `launch_ms` are quoted in
                row = next(r for r in json.load(open(os.path.join(here, meta["standalone"]))) if r["shape"] == "bench")
                regime = "stand-alone launch of the bench's shape (tools/standalone_kernels.py l2), as `frac` / `launch_ms`"
                where = meta["standalone"]
            else:                           # the pipelined bench run (overlapped and isolated launches averaged)
                rows = [r for r in json.load(open(os.path.join(here, "profiles", f"{tag}_pmc.json"))) if r["kernel"].startswith("l2_min_pp3")]
                row = max(rows, key=lambda r: r["grid_threads"])
                regime = "inside the pipelined bench run"
                where = f"profiles/{tag}_pmc.md"
            return {"traffic": round(row["fetch_bytes"] + row["write_bytes"]), "traffic_regime": regime,
                    "traffic_note": f"fabric-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) from the committed PMC passes ({where}, profiles/{tag}_pmc.md, "
                                    f"commit {meta['commit']}; kernel source unchanged since: sha256 {meta['sha256'][:12]}); L2 hit {row['l2_hit']:.3f}"}
        except (OSError, KeyError, ValueError) as e:
            note = f"no usable committed PMC pass ({type(e).__name__})"
    return {"traffic": None, "traffic_note": note}
