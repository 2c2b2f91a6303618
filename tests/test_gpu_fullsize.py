"""GPU: BASELINE-size inputs (batch 32, 24 576-point clouds, bagel-sized libraries), checked through properties that do
not need the CPU oracle at that size: batch invariance against single-cloud runs, sortedness, self-matches, sampled
brute-force agreement, bounds."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cmdiad_amd import ops  # noqa: E402
from cmdiad_amd.synth import synth_bank, synth_cloud_fixed_n  # noqa: E402

DEV = "cuda"
B, N, G, K = 32, 24576, 1024, 128


@pytest.fixture(scope="module")
def clouds():
    pcs = torch.cat([synth_cloud_fixed_n(500 + i, N) for i in range(B)], 0).to(DEV)
    xyz, nz, pix2pt, nv = ops.unorganize(pcs, N)
    assert bool((nv == N).all())
    return xyz


def test_fps_full_batch_properties(clouds):
    idx, cen = ops.fps(clouds, G)
    assert idx.shape == (B, G) and bool((idx[:, 0] == 0).all())
    srt = idx.sort(1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())                      # distinct points -> no index repeats
    for b in (0, 13, 31):                                               # batch invariance: the same cloud alone
        i1, c1 = ops.fps(clouds[b:b + 1].contiguous(), G)
        assert torch.equal(i1[0], idx[b]) and torch.equal(c1[0], cen[b])
    assert torch.equal(cen, torch.gather(clouds, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)))


def test_knn_group_full_batch_properties(clouds):
    idx_c, cen = ops.fps(clouds, G)
    idx, nb = ops.knn_group(clouds, cen, K)
    assert idx.shape == (B, G, K) and int(idx.min()) >= 0 and int(idx.max()) < N
    d2 = (nb[..., 0] * nb[..., 0] + nb[..., 1] * nb[..., 1]) + nb[..., 2] * nb[..., 2]   # the kernel's evaluation order
    assert bool((d2[:, :, 1:] >= d2[:, :, :-1]).all())                  # ascending distances
    assert torch.equal(idx[:, :, 0], idx_c.long()) and bool((d2[:, :, 0] == 0).all())   # a centre is its own nearest point
    srt = idx.sort(-1).values
    assert bool((srt[..., 1:] != srt[..., :-1]).all())                  # K distinct neighbours
    gathered = torch.gather(clouds, 1, idx.reshape(B, -1, 1).expand(-1, -1, 3)).reshape(B, G, K, 3) - cen.unsqueeze(2)
    assert torch.equal(gathered, nb)


def test_l2_search_full_size_sampled_brute_force():
    Q, Nb, D = B * 3136, 76518, 768
    g = torch.Generator().manual_seed(77)
    bank = synth_bank(Nb, D, 4321)
    # 8 000 library rows get a near-duplicate somewhere else (1e-2 noise: ~0.28 apart, far inside what 16-bit operands resolve) and
    # half of the queries sit next to such a pair: the 16-bit search alone picks the twin about every other time
    perm = torch.randperm(Nb, generator=g)
    pa, pb = perm[:8000], perm[8000:16000]
    bank[pb] = bank[pa] + 1e-2 * torch.randn(8000, D, generator=g)
    bank = bank.to(DEV)
    src = torch.randint(0, Nb, (Q,), generator=g)
    src[::2] = pa[torch.randint(0, 8000, ((Q + 1) // 2,), generator=g)]
    q = bank[src.to(DEV)] + 0.5 * torch.randn(Q, D, generator=g).to(DEV)
    b16, b32, bsq = ops.normalize_cast(bank, want_f32=True)
    q16, q32, qsq = ops.normalize_cast(q, want_f32=True)
    keys = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, DEV, runner=True))      # best + runner-up (include/cmdiad_hip.h)
    mv, mi = ops.l2_rescore(q32, b32, keys)
    _, mi_winner_only = ops.l2_rescore(q32, b32, keys[0].contiguous())
    assert int(mi.min()) >= 0 and int(mi.max()) < Nb and bool((mv >= 0).all())
    sel = torch.randperm(Q, generator=g)[:4096].to(DEV)
    # brute force over the fp32 rows (features.py:186-190,227), evaluated in float64 so that the reference has no ties of its own
    b64 = b32.double()
    ri = torch.cat([torch.cdist(q32[sel[i:i + 256]].double(), b64).argmin(1) for i in range(0, 4096, 256)])
    agree = (mi[sel] == ri)
    print(f"[full size: argmin == brute force on {agree.float().mean().item():.5f} of 4096 sampled rows "
          f"(winner only: {(mi_winner_only[sel] == ri).float().mean().item():.4f})]")
    assert (mi_winner_only[sel] == ri).float().mean().item() < 0.98, "the planted near-ties must defeat the 16-bit search alone"
    rv = (q32[sel].double() - b64[ri]).pow(2).sum(1).sqrt()
    got = (q32[sel].double() - b64[mi[sel]]).pow(2).sum(1).sqrt()
    # >= 99.99 % identical rows; a different row is admissible only as a tie at fp32 resolution of the distance itself
    assert agree.float().mean().item() >= 0.9995 and bool(((got - rv) <= 2e-7 * rv)[~agree].all()), (agree.float().mean().item(), (got - rv)[~agree])
    np.testing.assert_allclose(mv[sel].cpu().numpy(), rv.float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    # a query that IS a bank row finds itself at distance ~0
    keys2 = ops.l2_min_keys(b16[:4096].contiguous(), bsq[:4096].contiguous(), b16, bsq, ops.new_keys(4096, DEV))
    mv2, mi2 = ops.l2_rescore(b32[:4096].contiguous(), b32, keys2)
    assert torch.equal(mi2.cpu(), torch.arange(4096)) and float(mv2.max()) == 0.0


def test_deduplicated_search_full_size_equals_the_search_of_every_row():
    """Bench-size query set with the bench's structure (46 % of the rows are one repeated background row, csrc/dedup.hip): the
    compacted search + key expansion gives the keys of the search over all 100 352 rows, bit for bit."""
    Q, Nb, D = B * 3136, 76518, 768
    g = torch.Generator().manual_seed(78)
    bank = synth_bank(Nb, D, 4321).to(DEV)
    q = torch.randn(Q, D, generator=g)
    back = torch.rand(Q, generator=g) < 0.46
    q[back] = (0.0 - 0.013) / 0.21                      # (0 - mean) / std in every column
    n_back = int(back.sum())
    b16, _, bsq = ops.normalize_cast(bank)
    q16, _, qsq = ops.normalize_cast(q.to(DEV))
    full = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, DEV))
    plan = ops.rows_dedup_plan(q16, qsq)
    assert int(plan.count.item()) == Q - n_back + 1
    first = int(torch.nonzero(back)[0])
    assert int(plan.slot[first]) == first and bool((plan.slot[back.to(DEV)] == first).all())
    kc = ops.l2_min_keys_counted(plan.q16, plan.q_sq, plan.count, b16, bsq, ops.new_keys(Q, DEV))
    assert torch.equal(ops.keys_expand(kc, plan.slot, torch.empty_like(full)), full)


def test_blur_properties_full_batch():
    maps = torch.rand(64, 224, 224, device=DEV) * 5.0
    out = ops.blur8_maps(maps, 4.0)
    mx = maps.amax(dim=(1, 2), keepdim=True)
    assert bool((out >= 0).all()) and bool((out <= mx).all())            # a blur of [0,255] levels stays inside the range
    flat = torch.full((2, 224, 224), 3.25, device=DEV)
    assert torch.equal(ops.blur8_maps(flat, 4.0), flat)                  # constant map: 255 -> 255 -> x max, unchanged
    once = ops.blur8_maps(maps[:4], 4.0)
    assert torch.equal(once, out[:4])                                    # deterministic / batch independent
