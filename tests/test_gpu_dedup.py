"""GPU: exact removal of repeated query rows in front of the distance GEMM (csrc/dedup.hip, cmdiad_l2_min_keys_counted).
The reference searches the library for every row of the 56 x 56 patch grid (features.py:186-190); the rows of patches without a
foreground pixel are one repeated constant vector.  The plan is checked against its definition (include/cmdiad_hip.h) and the compacted search + key expansion against the search of every row: identical keys."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cmdiad_amd import ops  # noqa: E402

DEV = "cuda"


def check_plan(plan, q16, qsq, n_repeats):
    """The plan is valid -- the rows kept are the others in their order, every dropped row is bit for bit its representative (16-bit
    row and squared norm: everything the distance kernel reads) -- and it dropped the n_repeats - 1 later occurrences of the most
    repeated row (None: whatever it found; a pair among a few hundred rows can hide behind a hash-bucket tie, which costs nothing
    but the saving)."""
    q = q16.view(torch.int16).cpu().numpy()
    sq = qsq.cpu().numpy().view(np.uint32)
    Q = q.shape[0]
    n = int(plan.count.item())
    rows = plan.rows[:n].cpu().numpy()
    slot = plan.slot.cpu().numpy()
    assert 0 <= n <= Q and (np.diff(rows) > 0).all() and (n == 0 or (0 <= rows[0] and rows[-1] < Q))
    kept = np.zeros(Q, bool)
    kept[rows] = True
    assert np.array_equal(slot[rows], np.arange(n))
    dropped = np.nonzero(~kept)[0]
    if len(dropped):
        rep = rows[slot[dropped]]
        assert len(set(rep.tolist())) == 1 and (rep < dropped).all(), "one representative, the first occurrence"
        assert (q[dropped] == q[rep]).all() and (sq[dropped] == sq[rep]).all()
    if n_repeats is not None:
        assert len(dropped) == max(n_repeats - 1, 0), (len(dropped), n_repeats)
    idx = torch.from_numpy(rows).to(q16.device)
    assert torch.equal(plan.q16[:n], q16[idx]) and torch.equal(plan.q_sq[:n], qsq[idx])
    return n


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
@pytest.mark.parametrize("Q,n_const", [(5000, 2300), (777, 0), (300, 299), (1024, 1024), (1, 1), (2049, 16)])
def test_plan_and_compacted_search_match_the_full_search(Q, n_const, dtype):
    D, Nb = 768, 1300
    q16, qsq = make_queries(Q, D, dtype, n_const, seed=Q + n_const)
    g = torch.Generator().manual_seed(99)
    b16, _, bsq = ops.normalize_cast(torch.randn(Nb, D, generator=g).to(DEV), dtype=dtype)
    plan = ops.rows_dedup_plan(q16, qsq)
    n = check_plan(plan, q16, qsq, n_const if n_const >= 16 else (0 if Q == 1 else None))

    full = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, DEV))
    kc = ops.l2_min_keys_counted(plan.q16, plan.q_sq, plan.count, b16, bsq, ops.new_keys(Q, DEV))
    assert (kc[n:] == ops.KEY_EMPTY).all(), "rows beyond the live count must not be written"
    got = ops.keys_expand(kc, plan.slot, torch.empty_like(full))
    assert torch.equal(got, full)

    # the plan's buffers are reused by a second call with the same shape
    again = ops.rows_dedup_plan(q16, qsq, plan)
    assert again is plan and int(plan.count.item()) == n


def test_repeated_row_need_not_be_constant():
    """The hallucinated features of the background patches (multiple_features.py:351) repeat a NON-constant row."""
    Q, D = 3000, 768
    g = torch.Generator().manual_seed(3)
    x = torch.randn(Q, D, generator=g)
    idx = torch.randperm(Q, generator=g)[:1400]
    x[idx] = torch.randn(D, generator=g)
    q16, _, qsq = ops.normalize_cast(x.to(DEV), dtype=torch.float16)
    b16, _, bsq = ops.normalize_cast(torch.randn(1024, D, generator=g).to(DEV), dtype=torch.float16)
    plan = ops.rows_dedup_plan(q16, qsq)
    assert check_plan(plan, q16, qsq, 1400) == Q - 1400 + 1
    full = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, DEV))
    kc = ops.l2_min_keys_counted(plan.q16, plan.q_sq, plan.count, b16, bsq, ops.new_keys(Q, DEV))
    assert torch.equal(ops.keys_expand(kc, plan.slot, torch.empty_like(full)), full)


def test_counted_search_with_zero_live_rows_writes_nothing():
    D = 768
    q16, qsq = make_queries(600, D, torch.float16, 0, seed=5)
    b16, _, bsq = ops.normalize_cast(torch.randn(512, D).to(DEV), dtype=torch.float16)
    keys = ops.new_keys(600, DEV)
    ops.l2_min_keys_counted(q16, qsq, torch.zeros(1, dtype=torch.int32, device=DEV), b16, bsq, keys)
    assert (keys == ops.KEY_EMPTY).all()
