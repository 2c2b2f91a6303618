"""Two variations of the headline loop itself, on the headline's own predictor state: `h2d_inclusive` (the same batches from pinned
host memory, H2D inside the loop -- the PCIe-inclusive rate, never `value`) and `every_row_searched` (no row de-duplication: all
100 352 xyz query rows go through the distance GEMM, as the reference's cdist does; outputs bit-identical)."""
import os
import time

from .common import BATCH, N_POINTS, run_steps


def h2d_leg(pred, host_batches, n, first):
    """PCIe-inclusive rate: pinned host batches, copied on the predictor's copy stream inside the loop (the copy of step i+1 is
    issued under step i); run_steps also checks that the H2D-fed outputs equal the resident-fed ones bit for bit."""
    import torch
    run_steps(pred, host_batches, 2, first)        # the host-fed path once before the clock (first-touch of the pinned pages)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run_steps(pred, host_batches, n, first)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    mb = sum(t.numel() * 4 for t in host_batches[0] if t is not None) / 1e6
    return dict(value=round(BATCH * n / dt, 2), unit="images/s per GPU", steps=n, ms_per_step=round(dt / n * 1e3, 3),
                h2d_MB_per_step=round(mb, 1),
                note="inputs in pinned host memory, copied inside the loop; outputs identical to the resident run")


def every_row_leg(st, pred, batches, n, first, workload):
    """The same steps with EVERY row of the patch grid searched (CMDIAD_DEDUP=0); outputs are checked bit for bit against the
    de-duplicated run's (run_steps compares with `first`)."""
    import torch
    from cmdiad_amd.predictor import BatchPredictor
    os.environ["CMDIAD_DEDUP"] = "0"
    try:
        pred_all = BatchPredictor(st["engine"], st["bank_xyz"], st["bank_second"], st["stats"], st["det"], st["seg"], batch=BATCH,
                                  n_max=N_POINTS, workload=workload, halluc=st["halluc"], group=None, use_graph=pred.use_graph)
    finally:
        del os.environ["CMDIAD_DEDUP"]
    run_steps(pred_all, batches, 3, first)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    run_steps(pred_all, batches, n, first)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t2
    return dict(value=round(BATCH * n / dt, 2), unit="images/s per GPU", steps=n, ms_per_step=round(dt / n * 1e3, 3),
                note="CMDIAD_DEDUP=0: all 100352 query rows per step go through the distance GEMM; "
                     "outputs bit-identical to the default run")
