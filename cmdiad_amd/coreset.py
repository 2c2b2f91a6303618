"""Greedy coreset selection on the GPU (reference features.py:372-425)."""
import torch

from . import _native as nat
from . import ops


def greedy_coreset(z, n_select, coreset_dtype="FP16"):
    """z [n,d] f32 cuda (randomly projected library) -> selected row indices [n_select] int64 (cuda)."""
    if coreset_dtype not in ("FP16", "TF32"):
        raise NotImplementedError(f"coreset_dtype {coreset_dtype!r}: the reference knows 'FP16' and 'TF32' (features.py:386-393)")
    z = z.float().contiguous()
    n, d = z.shape
    if coreset_dtype == "TF32":      # the fp32 scan (allow_tf32 changes matrix products only: the loop has none)
        out = torch.empty((n_select,), dtype=torch.int64, device=z.device)
        wsb = nat.lib().cmdiad_coreset_f32_workspace_bytes(n, d, n_select)
        ws = torch.empty((wsb + 7) // 8, dtype=torch.int64, device=z.device)
        nat.check(nat.lib().cmdiad_coreset_greedy_f32(ops._p(z), n, d, n_select, 0, ops._p(out), ops._p(ws), wsb, ops._stream()),
                  "cmdiad_coreset_greedy_f32")
        return out
    if d % 2:
        z = torch.nn.functional.pad(z, (0, 1))
        d += 1
    out = torch.empty((n_select,), dtype=torch.int64, device=z.device)
    wsb = nat.lib().cmdiad_coreset_workspace_bytes(n, d, n_select)
    ws = torch.empty((wsb + 7) // 8, dtype=torch.int64, device=z.device)
    nat.check(nat.lib().cmdiad_coreset_greedy(ops._p(z), n, d, n_select, 0, ops._p(out), ops._p(ws), wsb, ops._stream()),
              "cmdiad_coreset_greedy")
    return out


class _HipRounds:
    """The device side of the row-sharded selection (HIP kernels); tests/test_host_cpu.py swaps in a torch stand-in to run the
    exchange on a gloo group without a GPU."""

    def __init__(self, z):
        z = z.float().contiguous()
        if z.shape[1] % 2:
            z = torch.nn.functional.pad(z, (0, 1))
        self.n, self.d = z.shape
        wsb = nat.lib().cmdiad_coreset_workspace_bytes(self.n, self.d, 1)
        self.ws = torch.empty((wsb + 7) // 8, dtype=torch.int64, device=z.device)
        nat.check(nat.lib().cmdiad_coreset_prepare(ops._p(z), self.n, self.d, 0, ops._p(self.ws), wsb, ops._stream()), "cmdiad_coreset_prepare")

    def round(self, lo, hi, pivot_key, out_key):
        nat.check(nat.lib().cmdiad_coreset_round(ops._p(self.ws), self.n, self.d, lo, hi, ops._p(pivot_key), 0, ops._p(out_key), ops._stream()),
                  "cmdiad_coreset_round")

    def decode(self, keys, n_select):
        out = torch.empty((n_select,), dtype=torch.int64, device=keys.device)
        nat.check(nat.lib().cmdiad_coreset_decode(ops._p(keys), n_select, 0, ops._p(out), ops._stream()), "cmdiad_coreset_decode")
        return out


def shard_rows(n, rank, world):
    """4-row aligned contiguous row range of `rank` (the scan kernel owns groups of four rows)."""
    per = ((n + world - 1) // world + 3) // 4 * 4
    lo, hi = min(rank * per, n), min((rank + 1) * per, n)
    if lo >= hi:          # a rank beyond the last row: an EMPTY range at an aligned row (n itself need not be a multiple of 4)
        lo = hi = n // 4 * 4
    return lo, hi


def greedy_coreset_sharded(z, n_select, group, impl=_HipRounds):
    """features.py:372-425 with the SCAN of every round split over the ranks of `group` (fit-time sharding): z [n,d] f32,
    the same projected library on every rank; each rank scans its row range and proposes its (running minimum, row) winner as a packed
    key; ONE all_reduce(MAX) of 8 bytes per round makes the global winner, the next round's pivot, known to all.  Returns the
    selected rows [n_select] int64 -- identical on every rank and identical to `greedy_coreset(z, n_select)` (the keys are the ones the
    single-device loop chains internally; tests/test_gpu_fakeworld.py, tests/test_host_cpu.py).
    Cost model: a round is (scan of n / W rows) + one 8-byte collective (~20-30 us over xGMI) + two host calls; at 765 184 x 334 the
    scan is 90 us on one GPU, so the split pays from W = 4 (23 + ~40 us) and does not at W = 2 -- it exists for the class whose fit is
    the makespan of a class-sharded run, not as a default."""
    import torch.distributed as td
    rank, world = td.get_rank(group), td.get_world_size(group)
    rounds = impl(z)
    lo, hi = shard_rows(rounds.n, rank, world)
    keys = torch.zeros((max(n_select - 1, 1),), dtype=torch.int64, device=z.device)
    for r in range(n_select - 1):
        if hi > lo:       # (a rank with no rows proposes nothing: its key stays 0 and loses the MAX)
            rounds.round(lo, hi, keys[r - 1:r] if r else None, keys[r:r + 1])
        td.all_reduce(keys[r:r + 1], op=td.ReduceOp.MAX, group=group)      # keys are non-negative as int64: signed MAX == unsigned MAX
    return rounds.decode(keys, n_select)


def sparse_random_projection(z_lib, eps=0.9, random_state=None):
    """z_lib [n,d] f32 cuda -> SparseRandomProjection(eps=eps, random_state=random_state).fit_transform(z_lib) [n, n_comp] f32, on the
    device and bit-identical to the host's (features.py:360-371).  scikit-learn FITS the transformer -- the Johnson-Lindenstrauss
    dimension for n samples and the random sparse matrix, from its own generator -- on a zero-strided stand-in with the library's
    shape (the fit never looks at values); the TRANSFORM is cmdiad_sparse_project_f32.  Raises ValueError as scikit-learn does when
    eps asks for more components than there are features."""
    import numpy as np
    from sklearn import random_projection
    n, d = z_lib.shape
    tr = random_projection.SparseRandomProjection(eps=eps, random_state=random_state)
    tr.fit(np.lib.stride_tricks.as_strided(np.zeros((1,), dtype=np.float32), shape=(n, d), strides=(0, 0), writeable=False))
    comp = tr.components_.tocsr()
    comp.sort_indices()
    if comp.dtype != np.float32:
        raise RuntimeError(f"SparseRandomProjection.components_ is {comp.dtype}: the float32 arithmetic of the device transform "
                           "would not match this scikit-learn")
    dev = z_lib.device
    indptr = torch.from_numpy(comp.indptr.astype(np.int32)).to(dev)
    indices = torch.from_numpy(comp.indices.astype(np.int32)).to(dev)
    data = torch.from_numpy(comp.data.astype(np.float32)).to(dev)
    z = z_lib.float().contiguous()
    out = torch.empty((n, comp.shape[0]), dtype=torch.float32, device=dev)
    nat.check(nat.lib().cmdiad_sparse_project_f32(ops._p(z), n, d, ops._p(indptr), ops._p(indices), ops._p(data), comp.shape[0],
                                                  ops._p(out), ops._stream()), "cmdiad_sparse_project_f32")
    return out
