"""Greedy coreset selection on the GPU (reference features.py:372-425)."""
import torch

from . import _native as nat
from . import ops


def greedy_coreset(z, n_select, coreset_dtype="FP16"):
    """z [n,d] f32 cuda (randomly projected library) -> selected row indices [n_select] int64 (cuda)."""
    if coreset_dtype != "FP16":
        raise NotImplementedError("cmdiad_amd implements coreset_dtype='FP16' (the reference default)")
    z = z.float().contiguous()
    n, d = z.shape
    if d % 2:
        z = torch.nn.functional.pad(z, (0, 1))
        d += 1
    out = torch.empty((n_select,), dtype=torch.int64, device=z.device)
    wsb = nat.lib().cmdiad_coreset_workspace_bytes(n, d, n_select)
    ws = torch.empty((wsb + 7) // 8, dtype=torch.int64, device=z.device)
    nat.check(nat.lib().cmdiad_coreset_greedy(ops._p(z), n, d, n_select, 0, ops._p(out), ops._p(ws), wsb, ops._stream()),
              "cmdiad_coreset_greedy")
    return out
