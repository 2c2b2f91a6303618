"""CPU model of the round-6 neighbourhood searches (cmdiad_knn_group_ws: csrc/knn_group.hip knn_grid_*; cmdiad_interp3nn_ws:
csrc/interp_pool.hip interp3nn_bin / _grid): the SAME rules in numpy float32 -- grid on the two widest axes, cell coordinate
clamp((a - min) * inv_h), rings scanned innermost first, stop when the K-th best is certified against ((m - 0.01) h)^2 (minus the
formula's rounding bound for the 3-NN), next radius from the current K-th best -- checked against brute force over the whole set.
The GPU tests prove the kernels equal the streaming / full kernels and the C oracle bit for bit; this file checks the exactness
ARGUMENT itself (the 0.01-cell slack, clamping at the grid border, centres outside the bounding box, degenerate grids, the rounding
bound at large coordinates) on geometries chosen against it, where no GPU is needed."""
import numpy as np
import pytest

F = np.float32


def _grid(points, side):
    """-> (axes A < B, min on A, min on B, inv_h, h, cell index of every point) with the kernels' float32 arithmetic."""
    mn, mx = points.min(0), points.max(0)
    e = (mx - mn).astype(F)
    if e[0] >= e[1] and e[0] >= e[2]:
        A, B = 0, (1 if e[1] >= e[2] else 2)
    elif e[1] >= e[2]:
        A, B = 1, (0 if e[0] >= e[2] else 2)
    else:
        A, B = 2, (0 if e[0] >= e[1] else 1)
    A, B = min(A, B), max(A, B)
    ext = max(e[A], e[B])
    h = F(ext * F(1.0 / side)) if 0 < ext < np.inf else F(0)
    inv_h = F(1) / h if h > 0 else F(0)

    def coord(a, m):
        return np.clip(((a - m).astype(F) * inv_h).astype(F), F(0), F(side - 1)).astype(np.int64)
    return A, B, mn[A], mn[B], inv_h, h, coord


def knn_model(points, centre, K, side=64):
    """Indices of the K nearest points of `centre` in ascending (d2, index) order, by ring scanning + certificate."""
    points = points.astype(F)
    A, B, mnA, mnB, inv_h, h, coord = _grid(points, side)
    ca, cb = coord(points[:, A], mnA), coord(points[:, B], mnB)
    ia, ib = int(coord(np.array([centre[A]], F), mnA)[0]), int(coord(np.array([centre[B]], F), mnB)[0])
    d = (points - centre.astype(F)).astype(F)
    d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(F) + d[:, 2] * d[:, 2]).astype(F)
    keys = (d2.view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.arange(len(points), dtype=np.uint64)
    cheb = np.maximum(np.abs(ca - ia), np.abs(cb - ib))
    m, rounds, evaluated = (2 if h > 0 else side), 0, 0
    while True:
        m = min(m, side)
        seen = np.sort(keys[cheb <= m])
        evaluated = int((cheb <= m).sum())
        rounds += 1
        if m >= side:
            break
        if len(seen) >= K and np.isfinite(d2[int(seen[K - 1] & np.uint64(0xFFFFFFFF))]):
            d2k = d2[int(seen[K - 1] & np.uint64(0xFFFFFFFF))]
            r = F((F(m) - F(0.01)) * h)
            if d2k < r * r:
                break
            m = max(m + 1, int(np.sqrt(d2k) * inv_h) + 2)
        else:
            m *= 2
    return (seen[:K] & np.uint64(0xFFFFFFFF)).astype(np.int64), rounds, evaluated


def knn_brute(points, centre, K):
    d = (points.astype(F) - centre.astype(F)).astype(F)
    d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(F) + d[:, 2] * d[:, 2]).astype(F)
    return np.lexsort((np.arange(len(points)), d2))[:K]


def nn3_formula(points, centres):
    """d[i, s] = -2 a.b + |a|^2 + |b|^2 in the kernels' float32 operation order."""
    a, c = points.astype(F), centres.astype(F)
    n1 = ((a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]).astype(F) + a[:, 2] * a[:, 2]).astype(F)
    cw = ((c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]).astype(F) + c[:, 2] * c[:, 2]).astype(F)
    dot = ((a[:, None, 0] * c[None, :, 0] + a[:, None, 1] * c[None, :, 1]).astype(F) + a[:, None, 2] * c[None, :, 2]).astype(F)
    d = (F(-2) * dot).astype(F)
    d = (d + n1[:, None]).astype(F)
    return (d + cw[None, :]).astype(F), n1, cw


def nn3_model(points, centres, side=16):
    """Three smallest formula values per point, (d, index) order, by ring scanning with the rounding bound -> idx [N, 3]."""
    centres = centres.astype(F)
    A, B, mnA, mnB, inv_h, h, coord = _grid(centres, side)
    ca, cb = coord(centres[:, A], mnA), coord(centres[:, B], mnB)
    d, n1, cw = nn3_formula(points, centres)
    cwmax = cw.max()
    out = np.zeros((len(points), 3), np.int64)
    widened = 0
    for i, p in enumerate(points.astype(F)):
        ia, ib = int(coord(np.array([p[A]], F), mnA)[0]), int(coord(np.array([p[B]], F), mnB)[0])
        cheb = np.maximum(np.abs(ca - ia), np.abs(cb - ib))
        E = F(F(4e-6) * (n1[i] + cwmax))
        m = 1 if h > 0 else side
        while True:
            m = min(m, side)
            cand = np.nonzero(cheb <= m)[0]
            order = cand[np.lexsort((cand, d[i, cand]))][:3]
            if m >= side:
                break
            d2 = d[i, order[2]] if len(order) == 3 else F(np.inf)
            r = F((F(m) - F(0.01)) * h)
            if d2 < F(r * r - E):
                break
            widened += 1
            m = max(m + 1, int(np.sqrt(max(F(d2 + E), F(0))) * inv_h) + 2) if np.isfinite(d2) else m * 2
        out[i] = order
    return out, widened


def _geometries(rs):
    u = rs.rand(3000, 2).astype(F)
    sheet = np.stack([u[:, 0], u[:, 1], 0.05 * np.sin(6 * u[:, 0]) * np.cos(5 * u[:, 1])], 1).astype(F)
    return dict(sheet=sheet, wall=sheet[:, [0, 2, 1]].copy(), blob=rs.rand(2500, 3).astype(F),
                two_clusters=np.concatenate([rs.randn(1200, 3) * 0.01, rs.randn(1200, 3) * 0.01 + 5.0]).astype(F),
                line=np.stack([np.linspace(0, 1, 2200), np.zeros(2200), np.zeros(2200)], 1).astype(F),
                identical=np.tile(np.array([[0.3, -0.2, 0.9]], F), (2100, 1)), duplicates=np.concatenate([sheet[:1200], sheet[:1200]]))


@pytest.mark.parametrize("name", ["sheet", "wall", "blob", "two_clusters", "line", "identical", "duplicates"])
def test_knn_ring_certificate_is_exact(name):
    rs = np.random.RandomState(11)
    pts = _geometries(rs)[name]
    centres = [pts[i] for i in rs.randint(0, len(pts), 10)] + [pts[3] + F(0.04), pts.min(0) - F(0.3), pts.max(0) + F(2.0), pts.mean(0)]
    few = 0
    for K in (1, 37, 128):
        for c in centres:
            got, rounds, evaluated = knn_model(pts, np.asarray(c, F), K)
            np.testing.assert_array_equal(got, knn_brute(pts, np.asarray(c, F), K), err_msg=f"{name} K={K} centre {c}")
            few += evaluated < len(pts) // 2
    if name in ("sheet", "wall"):
        assert few >= 30, "on a sheet the search must certify long before it has seen the cloud (else the model tests nothing)"


@pytest.mark.parametrize("name", ["sheet", "wall", "blob", "two_clusters", "line", "duplicates"])
@pytest.mark.parametrize("scale", [1.0, 700.0], ids=["metres", "large_coordinates"])
def test_3nn_ring_certificate_with_the_rounding_bound_is_exact(name, scale):
    rs = np.random.RandomState(12)
    pts = (_geometries(rs)[name] * F(scale) + F(0 if scale == 1.0 else 250.0)).astype(F)
    cen = pts[rs.randint(0, len(pts), 200)].copy()
    cen[::9] += (rs.randn(len(cen[::9]), 3) * 0.03 * scale).astype(F)
    cen[7] = cen[3]
    cen[11] = pts.max(0) + F(5.0 * scale)
    q = pts[rs.randint(0, len(pts), 300)]
    d, _, _ = nn3_formula(q, cen)
    want = np.stack([np.lexsort((np.arange(cen.shape[0]), d[i]))[:3] for i in range(len(q))])
    got, widened = nn3_model(q, cen)
    np.testing.assert_array_equal(got, want)


def test_3nn_far_from_the_origin_never_certifies_early_and_stays_exact():
    """A unit sheet 3 000 units from the origin: |a|^2 ~ 2.7e7, so the formula's rounding (and the bound E ~ 200) dwarfs the squared
    centre spacing (~5e-3) -- the reference's own selection is rounding noise there.  The search may not certify on geometry alone:
    it must widen to the whole grid, and then equals the brute-force selection on the SAME noisy formula values."""
    rs = np.random.RandomState(13)
    pts = (_geometries(rs)["sheet"] + F(3000.0)).astype(F)
    cen = pts[rs.randint(0, len(pts), 200)].copy()
    q = pts[rs.randint(0, len(pts), 200)]
    d, _, _ = nn3_formula(q, cen)
    want = np.stack([np.lexsort((np.arange(cen.shape[0]), d[i]))[:3] for i in range(len(q))])
    got, widened = nn3_model(q, cen)
    np.testing.assert_array_equal(got, want)
    assert widened >= len(q), "every point must have widened its search at least once"
    true = np.stack([np.argsort(((q[i].astype(np.float64) - cen.astype(np.float64)) ** 2).sum(1), kind="stable")[:3] for i in range(len(q))])
    assert (got != true).any(), "(the formula's selection differs from the geometric one here: that is the regime this test is about)"
