#!/usr/bin/env python3
"""Randomised parity sweep of the kernels against the CPU oracle / float64 references (development QA, not part of the
pytest suites): random shapes including ragged tiles, tiny and odd sizes.  `python tools/fuzz_gpu.py [seconds] [seed]`."""
import os as _os
# A/B tool: needs the test-only build with the superseded kernel formulations (make -C cmdiad_amd/csrc ab)
_os.environ.setdefault("CMDIAD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cmdiad_amd", "libcmdiad_hip_ab.so"))
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd import engine as eng
from cmdiad_amd import ops  # noqa: E402
from oracle import kernels as ok  # noqa: E402

DEV = "cuda"


def cloud(rs, n):
    p = rs.rand(n, 3).astype(np.float32) * 0.2
    p[:, 2] += 0.5
    return p


def case_fps_knn(rs):
    B = int(rs.randint(1, 4))
    ns = [int(rs.randint(130, 6000)) for _ in range(B)]
    N = max(ns)
    xyz = np.zeros((B, N, 3), np.float32)
    for b, n in enumerate(ns):
        xyz[b, :n] = cloud(rs, n)
    G, K = int(rs.randint(1, 200)), int(rs.randint(1, 129))
    nv = torch.tensor(ns, dtype=torch.int32, device=DEV)
    idx, cen = ops.fps(torch.from_numpy(xyz).to(DEV), G, n_valid=nv)
    gi, nb = ops.knn_group(torch.from_numpy(xyz).to(DEV), cen, K, n_valid=nv)
    for b, n in enumerate(ns):
        ir, cr = ok.fps(xyz[b:b + 1, :n], G)
        assert np.array_equal(idx[b].cpu().numpy(), ir[0]), ("fps", ns, G)
        kr, nr = ok.knn_group(xyz[b:b + 1, :n], cr, K)
        assert np.array_equal(gi[b].cpu().numpy(), kr[0]) and np.array_equal(nb[b].cpu().numpy(), nr[0]), ("knn", ns, G, K)
    r = float(rs.uniform(0.005, 0.08)); nsamp = int(rs.randint(1, 40))
    bq = ops.ball_query(r, nsamp, torch.from_numpy(xyz).to(DEV), cen, n_valid=nv)
    for b, n in enumerate(ns):
        assert np.array_equal(bq[b].cpu().numpy(), ok.ball_query(r, nsamp, xyz[b:b + 1, :n], cen[b:b + 1].cpu().numpy())[0]), ("ball", ns, r)


def case_fps_big(rs):
    """large single clouds: every FPS kernel configuration (register-resident 1024x{4,8,16}, 512x{40,48,56}, memory fallback)"""
    n = int(rs.choice([int(rs.randint(3000, 9000)), int(rs.randint(15000, 28672)), int(rs.randint(28673, 50177))]))
    G = int(rs.randint(64, 1025))
    xyz = cloud(rs, n)[None]
    idx, cen = ops.fps(torch.from_numpy(xyz).to(DEV), G)
    ir, cr = ok.fps(xyz, G)
    assert np.array_equal(idx.cpu().numpy(), ir) and np.array_equal(cen.cpu().numpy(), cr), ("fps_big", n, G)
    K = int(rs.choice([32, 64, 128]))
    Gk = min(G, 96)
    gi, nb = ops.knn_group(torch.from_numpy(xyz).to(DEV), cen[:, :Gk].contiguous(), K)
    kr, nr = ok.knn_group(xyz, cr[:, :Gk], K)
    assert np.array_equal(gi.cpu().numpy(), kr) and np.array_equal(nb.cpu().numpy(), nr), ("knn_big", n, Gk, K)


def case_l2_big(rs):
    """production-sized query counts (the 4-wave wide kernel is selected automatically from Q >= 16 384)"""
    Q, Nb, D = int(rs.randint(16384, 40000)), int(rs.randint(300, 9000)), 64 * int(rs.choice([2, 6, 12]))
    g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
    bank = torch.randn(Nb, D, generator=g).to(DEV)
    q = bank[torch.randint(0, Nb, (Q,), generator=g).to(DEV)] + 0.4 * torch.randn(Q, D, generator=g).to(DEV)
    b16, b32, bsq = ops.normalize_cast(bank, want_f32=True)
    q16, q32, qsq = ops.normalize_cast(q, want_f32=True)
    keys = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, DEV))
    mv, mi = ops.l2_rescore(q32, b32, keys)
    sel = torch.randint(0, Q, (512,), generator=g).to(DEV)
    rv, ri = torch.cdist(q32[sel].double(), b32.double()).min(1)
    agree = mi[sel] == ri
    assert agree.float().mean() > 0.97, ("l2_big agree", Q, Nb, D, agree.float().mean().item())
    assert torch.allclose(mv[sel][agree].double(), rv[agree], rtol=1e-4, atol=1e-4), ("l2_big val", Q, Nb, D)
    assert torch.allclose(mv[sel][~agree].double(), rv[~agree], rtol=5e-3, atol=1e-3), ("l2_big near", Q, Nb, D)


def case_gemm(rs):
    M, N, K = int(rs.randint(1, 1500)), 4 * int(rs.randint(1, 300)), 64 * int(rs.randint(1, 9))
    A = torch.from_numpy(rs.randn(M, K).astype(np.float32)).bfloat16()
    W = torch.from_numpy((rs.randn(N, K) / np.sqrt(K)).astype(np.float32)).bfloat16()
    bias = torch.from_numpy(rs.randn(N).astype(np.float32))
    for env in (None, "4", "8"):
        if env:
            os.environ["CMDIAD_GEMM_WIDE"] = env
        else:
            os.environ.pop("CMDIAD_GEMM_WIDE", None)
        act = [ops.ACT_NONE, ops.ACT_RELU, ops.ACT_GELU][int(rs.randint(3))]
        o32, o16 = ops.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), act=act, want_f32=True)
        ref = A.double() @ W.double().T + bias.double()
        ref = ref if act == ops.ACT_NONE else (torch.relu(ref) if act == ops.ACT_RELU else torch.nn.functional.gelu(ref))
        assert torch.allclose(o32.cpu().double(), ref, rtol=2e-4, atol=2e-4), ("gemm", M, N, K, env, act)
        assert torch.allclose(o16.cpu().double(), ref, rtol=1e-2, atol=1e-2), ("gemm16", M, N, K, env, act)
    os.environ.pop("CMDIAD_GEMM_WIDE", None)


def case_l2(rs):
    Q, Nb, D = int(rs.randint(1, 3000)), int(rs.randint(1, 4000)), 64 * int(rs.randint(1, 13))
    bank = torch.from_numpy(rs.randn(Nb, D).astype(np.float32))
    q = bank[torch.from_numpy(rs.randint(0, Nb, Q))] + 0.4 * torch.from_numpy(rs.randn(Q, D).astype(np.float32))
    b16, b32, bsq = ops.normalize_cast(bank.to(DEV), want_f32=True)
    q16, q32, qsq = ops.normalize_cast(q.to(DEV), want_f32=True)
    d = torch.cdist(q.double(), bank.double())
    rv, ri = d.min(1)
    for tile in (None, "2", "5"):
        if tile:
            os.environ["CMDIAD_L2_TILE"] = tile
        else:
            os.environ.pop("CMDIAD_L2_TILE", None)
        keys = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, DEV))
        mv, mi = ops.l2_rescore(q32, b32, keys)
        agree = mi.cpu() == ri
        assert agree.float().mean() > 0.97, ("l2 agree", Q, Nb, D, tile, agree.float().mean())
        assert torch.allclose(mv.cpu()[agree].double(), rv[agree], rtol=1e-4, atol=1e-4), ("l2 val", Q, Nb, D, tile)
        assert torch.allclose(mv.cpu()[~agree].double(), rv[~agree], rtol=5e-3, atol=1e-3), ("l2 near", Q, Nb, D, tile)
    os.environ.pop("CMDIAD_L2_TILE", None)


def case_l2_identity(rs):
    """The key of a (query, row) pair must not depend on the formulation, the launch geometry or the shard the row is in (RowMin,
    csrc/l2min.hip): the 128 x 128 kernel, the lock-step 256 x 256 kernel (test build) and the two-group kernel on random shapes with
    ragged query / library tiles, both operand types, duplicates planted in the library; the MIN over a random two- or three-way row
    split; the counted launch on a random live count."""
    Q, Nb, D = int(rs.randint(1, 2500)), int(rs.randint(1, 5000)), 64 * int(rs.randint(3, 13))
    dt = torch.float16 if rs.rand() < 0.5 else torch.bfloat16
    bank = torch.from_numpy(rs.randn(Nb, D).astype(np.float32))
    if Nb > 20:      # repeated rows: ties must go to the lowest row in every formulation
        src = torch.from_numpy(rs.randint(0, Nb, 10))
        bank[torch.from_numpy(rs.randint(0, Nb, 10))] = bank[src]
    q = bank[torch.from_numpy(rs.randint(0, Nb, Q))] + float(rs.choice([0.0, 0.3])) * torch.from_numpy(rs.randn(Q, D).astype(np.float32))
    b16, _, bsq = ops.normalize_cast(bank.to(DEV), dtype=dt)
    q16, _, qsq = ops.normalize_cast(q.to(DEV), dtype=dt)
    keys = {}
    for tile in ("0", "2", "5"):
        os.environ["CMDIAD_L2_TILE"] = tile
        keys[tile] = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, DEV, runner=True)).clone()     # best + runner-up planes
    os.environ.pop("CMDIAD_L2_TILE", None)
    assert torch.equal(keys["0"], keys["2"]) and torch.equal(keys["0"], keys["5"]), ("l2 identity", Q, Nb, D, dt)
    one = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, DEV))
    assert torch.equal(one, keys["0"][0]), ("l2 best plane == single-plane launch", Q, Nb, D, dt)
    _, idx = ops.unpack_keys(keys["0"])
    has2 = keys["0"][1] != ops.KEY_EMPTY
    assert bool(((idx[0] >> 6 != idx[1] >> 6) | ((idx[0] >> 2) & 3 != (idx[1] >> 2) & 3))[has2].all()), ("runner-up outside the winner's group", Q, Nb, D)
    assert bool(has2.all()) or Nb <= 4, ("runner-up present (rows 0..3 are one group)", Q, Nb, D)
    cuts = sorted(set([0, Nb] + [int(c) // 64 * 64 for c in rs.randint(0, Nb + 1, int(rs.randint(1, 3)))]))
    merged = ops.new_keys(Q, DEV, runner=True)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        if hi > lo:
            k = ops.l2_min_keys(q16, qsq, b16[lo:hi].contiguous(), bsq[lo:hi].contiguous(), ops.new_keys(Q, DEV, runner=True), row_offset=lo)
            merged = eng.merge_key_planes(merged, k)
    assert torch.equal(merged, keys["0"]), ("l2 shards", Q, Nb, D, dt, cuts)
    live = int(rs.randint(1, Q + 1))
    cnt = torch.tensor([live], dtype=torch.int32, device=DEV)
    kc = ops.l2_min_keys_counted(q16, qsq, cnt, b16, bsq, ops.new_keys(Q, DEV, runner=True))
    assert torch.equal(kc[:, :live], keys["0"][:, :live]) and bool((kc[:, live:] == ops.KEY_EMPTY).all()), ("l2 counted", Q, Nb, D, live)


def case_reweight_pair(rs):
    """cmdiad_reweight_scan_pair == two cmdiad_reweight_scan calls == the exact float64 top-3, random library sizes and probe counts."""
    D = 128 * int(rs.randint(1, 7))
    n0, n1 = int(rs.randint(1, 9000)), int(rs.randint(1, 9000))
    r0, r1 = int(rs.randint(1, 33)), int(rs.randint(1, 33))
    banks = [torch.from_numpy(rs.randn(n, D).astype(np.float32)) for n in (n0, n1)]
    probes = [torch.cat([b[torch.from_numpy(rs.randint(0, b.shape[0], r - r // 2))], torch.from_numpy(rs.randn(r // 2, D).astype(np.float32))])
              for b, r in zip(banks, (r0, r1))]
    db, dp = [b.to(DEV) for b in banks], [p.to(DEV) for p in probes]
    blk = [ops.bank_block16(b) for b in db]
    t0, t1 = ops.reweight_scan_pair(dp[0], db[0], blk[0], dp[1], db[1], blk[1])
    for t, p, b, k16, pc, bc in ((t0, dp[0], db[0], blk[0], probes[0], banks[0]), (t1, dp[1], db[1], blk[1], probes[1], banks[1])):
        assert torch.equal(t, ops.reweight_scan(p, b, k16)), ("reweight pair", n0, n1, r0, r1, D)
        d = torch.stack([(bc.double() - x.double()).pow(2).sum(1) for x in pc])
        ri = torch.topk(d, min(3, bc.shape[0]), largest=False).indices
        assert torch.equal(ops.unpack_keys(t)[1][:, :ri.shape[1]].cpu(), ri), ("reweight exact", n0, n1, r0, r1, D)


def case_attention(rs):
    B, H, T = int(rs.randint(1, 4)), int(rs.randint(1, 7)), int(rs.randint(1, 1100))
    Tp = (T + 63) // 64 * 64
    q = torch.zeros(B, H, Tp, 64); k = torch.zeros(B, H, Tp, 64); v = torch.zeros(B, H, Tp, 64)
    q[:, :, :T] = torch.from_numpy(rs.randn(B, H, T, 64).astype(np.float32))
    k[:, :, :T] = torch.from_numpy(rs.randn(B, H, T, 64).astype(np.float32))
    v[:, :, :T] = torch.from_numpy(rs.randn(B, H, T, 64).astype(np.float32))
    q *= float(rs.choice([0.05, 0.18, 0.5, 1.0, 2.0]))          # flat ... nearly one-hot softmax: the lazily moved reference never / often moves
    if rs.rand() < 0.4:                                          # scores that climb (or fall) along the keys for some queries
        k[:, :, :T, 7] = float(rs.uniform(-0.2, 0.2)) * torch.arange(T, dtype=torch.float32)
        q[:, :, ::int(rs.randint(1, 40)), 7] = 1.0
    qb, kb, vb = q.bfloat16(), k.bfloat16(), v.bfloat16()
    out = ops.attention(qb.to(DEV), kb.to(DEV), vb.transpose(2, 3).contiguous().to(DEV), B, H, T).float().cpu()
    p = torch.softmax(qb[:, :, :T].double() @ kb[:, :, :T].double().transpose(2, 3) * np.log(2.0), -1)
    ref = (p @ vb[:, :, :T].double()).permute(0, 2, 1, 3).reshape(B * T, H * 64)
    assert torch.allclose(out.double(), ref, rtol=3e-2, atol=3e-2), ("attention", B, H, T)


def case_blur(rs):
    n, H, W = int(rs.randint(1, 5)), int(rs.randint(12, 257)), int(rs.randint(12, 257))
    radius = float(rs.uniform(0.6, 4.5))
    maps = torch.from_numpy(rs.rand(n, H, W).astype(np.float32)) * float(rs.uniform(0.1, 30))
    try:
        out = ops.blur8_maps(maps.to(DEV), radius).cpu()
    except Exception as e:  # short lines are rejected by design
        assert "shorter" in str(e), e
        return
    for i in range(n):
        mx = maps[i].max()
        u8 = (maps[i] / mx).mul(255).byte().numpy()
        ref = torch.from_numpy(ok.pil_gaussian_blur_u8(u8, radius)).float().div(255) * mx
        assert torch.equal(out[i], ref), ("blur", n, H, W, radius)


_ENC = {}


def case_encoder(rs):
    """Point-MAE encoder stages at random group counts / sizes: the persistent first stage and the persistent tail against the
    one-block-per-tile kernels of the test-only build (identical outputs), one to several row tiles per persistent block."""
    from cmdiad_amd.runtime import fold_pointmae_encoder
    from oracle import nets
    if "w" not in _ENC:
        _ENC["w"] = fold_pointmae_encoder(nets.synth_state_dict("pointmae", 21), "encoder.", DEV)
    w = _ENC["w"]
    Mg = int(rs.choice([32, 64, 128]))
    groups = int(rs.choice([1, 3, 7, 24, 130, 257, 300, 515, 1030])) if rs.rand() < 0.7 else int(rs.randint(1, 1200))
    nb = torch.from_numpy((0.05 * rs.randn(groups * Mg, 3)).astype(np.float32)).to(DEV)
    outs = {}
    for env in ("1", "0"):
        os.environ["CMDIAD_STAGE1_PERSIST"] = env
        outs[env] = [t.clone() for t in ops.encoder_stage1(nb, w["w1b1"], w["W2"], w["b2"], groups, Mg)]
    os.environ.pop("CMDIAD_STAGE1_PERSIST")
    assert all(torch.equal(a, b) for a, b in zip(outs["1"], outs["0"])), ("stage1", groups, Mg)
    h2 = outs["1"][0]
    gb = torch.from_numpy(rs.randn(groups, 512).astype(np.float32)).to(DEV)
    toks = {}
    for env in ("", "1", "0"):
        os.environ["CMDIAD_TAIL_PP"] = env
        toks[env] = ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, Mg).clone()
    os.environ.pop("CMDIAD_TAIL_PP")
    assert torch.equal(toks[""], toks["0"]) and torch.equal(toks["1"], toks["0"]), ("tail", groups, Mg)


def case_ocsvm(rs):
    """cmdiad_ocsvm_fit against scikit-learn on random score-like rows: identical coefficients, offset and epoch count."""
    import warnings
    from sklearn import linear_model
    from cmdiad_amd.ocsvm import DeviceSGDOneClassSVM
    n, F = int(rs.randint(2, 30000)), int(rs.randint(1, 5))
    nu = float(rs.choice([0.5, 0.5, 0.1, 0.9]))
    X = np.abs(rs.normal(1.0, 0.3, size=(n, F))).astype(np.float32)
    seed = int(rs.randint(0, 1000))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = linear_model.SGDOneClassSVM(random_state=seed, nu=nu, max_iter=50).fit(X)
    dev = DeviceSGDOneClassSVM(random_state=seed, nu=nu, max_iter=50).fit(torch.from_numpy(X).to(DEV))
    assert dev.n_iter_ == ref.n_iter_ and np.array_equal(dev.coef_, ref.coef_) and np.array_equal(dev.offset_, ref.offset_), ("ocsvm", n, F, nu, seed)


def case_unorganize(rs):
    """Random image sizes (chunk counts around the look-ahead of 7, ragged last chunks), fill fractions, zero coordinates in single
    planes, n_max below / at / above the count: compaction order, truncation, pixel -> point map."""
    B, H, W = int(rs.randint(1, 5)), int(rs.randint(1, 140)), int(rs.randint(1, 140))
    HW = H * W
    pc = rs.randn(B, 3, H, W).astype(np.float32)
    keep = rs.rand(B, 1, H, W) < rs.rand()
    pc = pc * keep
    pc[:, int(rs.randint(3))][rs.rand(B, H, W) < 0.05] = 0.0          # one zero coordinate drops the pixel (multiple_features.py:16)
    valid = np.all(pc.reshape(B, 3, HW) != 0, axis=1)
    counts = valid.sum(1)
    n_max = int(rs.choice([max(1, counts.max() // 2), max(1, counts.max()), HW]))
    xyz, nz, pix2pt, nv = ops.unorganize(torch.from_numpy(pc).to(DEV), n_max)
    for b in range(B):
        idx = np.nonzero(valid[b])[0]
        n = min(len(idx), n_max)
        assert int(nv[b]) == n, ("unorganize count", B, H, W, n_max)
        assert np.array_equal(nz[b, :n].cpu().numpy(), idx[:n]), ("unorganize order", B, H, W, n_max)
        assert np.array_equal(xyz[b, :n].cpu().numpy(), pc[b].reshape(3, HW).T[idx[:n]]), ("unorganize xyz", B, H, W, n_max)
        want = np.full(HW, -1, np.int32)
        want[idx[:n]] = np.arange(n)
        assert np.array_equal(pix2pt[b].cpu().numpy(), want), ("unorganize pix2pt", B, H, W, n_max)


def case_dedup(rs):
    """Random sizes, repeat fractions, repeated-row kinds (constant / arbitrary / none / everything) and both 16-bit types: the
    compacted search + key expansion equals the search of every row, the plan is valid."""
    Q, Nb, D = int(rs.randint(1, 6000)), int(rs.randint(1, 3000)), 64 * int(rs.randint(1, 13))
    dtype = torch.float16 if rs.rand() < 0.5 else torch.bfloat16
    x = torch.from_numpy(rs.randn(Q, D).astype(np.float32))
    kind = int(rs.randint(4))
    frac = float(rs.rand())
    sel = torch.from_numpy(rs.rand(Q) < frac)
    if kind == 0:
        x[sel] = float(rs.randn())
    elif kind == 1:
        x[sel] = torch.from_numpy(rs.randn(D).astype(np.float32))
    elif kind == 2:
        x[:] = 0.25
    second = kind != 3 and rs.rand() < 0.5 and Q > 8
    if second:      # a second, smaller group of repeats: searched row by row
        x[torch.from_numpy(rs.randint(0, Q, 3))] = -1.5
    q16, _, qsq = ops.normalize_cast(x.to(DEV), dtype=dtype)
    b16, _, bsq = ops.normalize_cast(torch.from_numpy(rs.randn(Nb, D).astype(np.float32)).to(DEV), dtype=dtype)
    full = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, DEV))
    plan = ops.rows_dedup_plan(q16, qsq)
    n = int(plan.count.item())
    rows, slot = plan.rows[:n].cpu().numpy(), plan.slot.cpu().numpy()
    assert 1 <= n <= Q and (np.diff(rows) > 0).all() and np.array_equal(slot[rows], np.arange(n)), ("dedup plan", Q, D, kind)
    qi = q16.view(torch.int16).cpu().numpy()
    assert (qi == qi[rows[slot]]).all() and (qsq.cpu().numpy().view(np.uint32) == qsq.cpu().numpy().view(np.uint32)[rows[slot]]).all()
    kc = ops.l2_min_keys_counted(plan.q16, plan.q_sq, plan.count, b16, bsq, ops.new_keys(Q, DEV))
    assert torch.equal(ops.keys_expand(kc, plan.slot, torch.empty_like(full)), full), ("dedup keys", Q, Nb, D, kind, str(dtype))
    if kind == 2:
        assert n == 1 if not second else n <= 4


def case_knn_production_grid(rs):
    """kNN grouping on a grid that selects the four-centres-per-wave instantiation (B * ceil(G / 16) >= 512), ragged clouds, and the
    3-nearest-centre interpolation weights on the same clouds."""
    B = int(rs.randint(4, 9))
    G = int(-(-512 // B) * 16 + rs.randint(0, 40))
    ns = [int(rs.randint(G + 200, 3500)) for _ in range(B)]
    N = max(ns)
    xyz = np.zeros((B, N, 3), np.float32)
    for b, n in enumerate(ns):
        xyz[b, :n] = cloud(rs, n)
    K = int(rs.choice([16, 64, 100, 128]))
    nv = torch.tensor(ns, dtype=torch.int32, device=DEV)
    x = torch.from_numpy(xyz).to(DEV)
    idx, cen = ops.fps(x, G, n_valid=nv)
    gi, nb = ops.knn_group(x, cen, K, n_valid=nv)
    i3, w3 = ops.interp3nn(x, cen, nv)
    for b in rs.choice(B, 2, replace=False):
        n = ns[b]
        c = cen[b:b + 1].cpu().numpy()
        ir, nr = ok.knn_group(xyz[b:b + 1, :n], c, K)
        assert np.array_equal(gi[b].cpu().numpy(), ir[0]), ("knn production grid", ns, G, K)
        assert np.array_equal(nb[b].cpu().numpy(), nr[0]), ("knn production grid neighbourhoods", ns, G, K)
        _, i3r, w3r = ok.interp3nn(xyz[b, :n], c[0], np.zeros((G, 4), np.float32), want_out=False)
        assert np.array_equal(i3[b, :n].cpu().numpy(), i3r) and np.array_equal(w3[b, :n].cpu().numpy(), w3r), ("interp3nn", ns, G)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rs = np.random.RandomState(seed)
    cases = [case_fps_knn, case_gemm, case_l2, case_attention, case_blur, case_fps_big, case_l2_big, case_encoder, case_ocsvm, case_dedup, case_unorganize, case_knn_production_grid,
             case_l2_identity, case_reweight_pair]
    only = [c for c in os.environ.get("FUZZ_ONLY", "").split(",") if c]        # e.g. FUZZ_ONLY=l2_identity,reweight_pair
    if only:
        cases = [c for c in cases if c.__name__[5:] in only]
    counts = {c.__name__: 0 for c in cases}
    t0 = time.time()
    while time.time() - t0 < budget:
        c = cases[int(rs.randint(len(cases)))]
        c(rs)
        counts[c.__name__] += 1
    print("fuzz ok", counts)


if __name__ == "__main__":
    main()
