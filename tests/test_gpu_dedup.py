"""GPU: exact removal of repeated query rows in front of the distance GEMM (csrc/dedup.hip, cmdiad_l2_min_keys_counted).
The reference searches the library for every row of the 56 x 56 patch grid (features.py:186-190); the rows of patches without a
foreground pixel are one repeated constant vector.  The plan is checked against a numpy restatement of its definition
(include/cmdiad_hip.h) and the compacted search + key expansion against the search of every row: identical keys."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cmdiad_amd import ops  # noqa: E402

DEV = "cuda"


def expected_plan(q16, qsq):
    """numpy restatement: first constant row = representative; a row repeats it iff constant with the same 16-bit value and the
    same squared-norm bits; the others keep their order."""
    q = q16.view(torch.int16).cpu().numpy()
    sq = qsq.cpu().numpy().view(np.uint32)
    const = (q == q[:, :1]).all(axis=1)
    Q = q.shape[0]
    reps = np.nonzero(const)[0]
    dup = np.zeros(Q, bool)
    if len(reps):
        r = reps[0]
        dup = const & (q[:, 0] == q[r, 0]) & (sq == sq[r])
        dup[r] = False
    rows = np.nonzero(~dup)[0]
    slot = np.empty(Q, np.int64)
    slot[rows] = np.arange(len(rows))
    if len(reps):
        slot[dup] = slot[reps[0]]
    return slot, rows


def make_queries(Q, D, dtype, n_const, seed, other_const=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(Q, D, generator=g)
    idx = torch.randperm(Q, generator=g)
    x[idx[:n_const]] = -0.37109375                 # the repeated background row (exact in both 16-bit types)
    if other_const and Q > n_const + 3:
        x[idx[n_const]] = 0.5                      # constant rows with another value: not repeats of the representative
        x[idx[n_const + 1]] = 0.5
        x[idx[n_const + 2], : D // 2] = -0.37109375   # half-constant: not constant
    q16, _, qsq = ops.normalize_cast(x.to(DEV), dtype=dtype)
    return q16, qsq


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Q,n_const", [(5000, 2300), (777, 0), (300, 299), (1024, 1024), (1, 1), (2049, 7)])
def test_plan_and_compacted_search_match_the_full_search(Q, n_const, dtype):
    D, Nb = 768, 1300
    q16, qsq = make_queries(Q, D, dtype, n_const, seed=Q + n_const)
    g = torch.Generator().manual_seed(99)
    b16, _, bsq = ops.normalize_cast(torch.randn(Nb, D, generator=g).to(DEV), dtype=dtype)
    plan = ops.rows_dedup_plan(q16, qsq)
    slot, rows = expected_plan(q16, qsq)
    n = int(plan.count.item())
    assert n == len(rows) and (n_const < 8 or n < Q)
    assert np.array_equal(plan.rows[:n].cpu().numpy(), rows)
    assert np.array_equal(plan.slot.cpu().numpy(), slot)
    assert torch.equal(plan.q16[:n], q16[torch.from_numpy(rows).to(DEV)])
    assert torch.equal(plan.q_sq[:n], qsq[torch.from_numpy(rows).to(DEV)])

    full = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, DEV))
    kc = ops.l2_min_keys_counted(plan.q16, plan.q_sq, plan.count, b16, bsq, ops.new_keys(Q, DEV))
    assert (kc[n:] == ops.KEY_EMPTY).all(), "rows beyond the live count must not be written"
    got = ops.keys_expand(kc, plan.slot, torch.empty_like(full))
    assert torch.equal(got, full)

    # the plan's buffers are reused by a second call with the same shape
    again = ops.rows_dedup_plan(q16, qsq, plan)
    assert again is plan and int(plan.count.item()) == n


def test_counted_search_with_zero_live_rows_writes_nothing():
    D = 768
    q16, qsq = make_queries(600, D, torch.float16, 0, seed=5)
    b16, _, bsq = ops.normalize_cast(torch.randn(512, D).to(DEV), dtype=torch.float16)
    keys = ops.new_keys(600, DEV)
    ops.l2_min_keys_counted(q16, qsq, torch.zeros(1, dtype=torch.int32, device=DEV), b16, bsq, keys)
    assert (keys == ops.KEY_EMPTY).all()
