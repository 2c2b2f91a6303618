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
