"""One-class SVM fit on the device: ``SGDOneClassSVM.fit`` of the reference's late fusion
(feature_extractors/features.py:352-358 -- ``detect_fuser.fit(s_lib)``, ``seg_fuser.fit(s_map_lib)``).

``DeviceSGDOneClassSVM`` has the constructor arguments and fitted attributes the reference touches (``coef_``, ``offset_``,
``n_iter_``, ``t_``; ``score_samples`` / ``decision_function``) and runs scikit-learn's float32 SGD with the same update order
in ``cmdiad_ocsvm_fit`` (cmdiad_amd/csrc/ocsvm.hip): coefficients, offset and epoch count equal scikit-learn's bit for bit
(tests/test_gpu_ocsvm.py).  The recurrence is strictly sequential, so the device is SLOWER than one host core at it
(docs/history.md section 7 has the numbers); the drop-in classes keep scikit-learn's host fit unless ``CMDIAD_OCSVM_DEVICE=1``.
The per-image side (``score_samples`` over 50 176 x k rows) is ``ops.ocsvm_score_maps`` either way.
"""
import ctypes

import numpy as np
import torch

from . import _native

MAX_INT32 = np.iinfo(np.int32).max


def xorshift32_step(s):
    """utils/_random.pxd our_rand_r: one state update of the 32-bit xorshift generator (13, 17, 5)."""
    s ^= (s << 13) & 0xFFFFFFFF
    s ^= s >> 17
    s ^= (s << 5) & 0xFFFFFFFF
    return s


def xorshift32_pow2_table():
    """[32, 32] uint32: row e, entry b = M^(2^e) applied to the unit vector of bit b, M = one generator step (linear over GF(2)):
    M^k s = XOR over the set bits b of s of column b, so a thread reaches step k in <= 32 table look-ups per set bit of k."""
    tab = np.zeros((32, 32), dtype=np.uint32)
    for b in range(32):
        tab[0, b] = xorshift32_step(1 << b)
    for e in range(1, 32):
        for b in range(32):
            s, y = int(tab[e - 1, b]), 0
            for c in range(32):
                if (s >> c) & 1:
                    y ^= int(tab[e - 1, c])
            tab[e, b] = y
    return tab


def fisher_yates_permutation(n, seed):
    """P with order_after[i] = order_before[P[i]] for ``ArrayDataset.shuffle(seed)`` (utils/_seq_dataset.pyx.tp:137-145), by the
    parallel construction of ocsvm.hip restated in numpy (host logic test; the device kernel is compared with it)."""
    seed = seed or 1
    j = np.empty(max(n - 1, 0), dtype=np.int64)
    s = seed
    for i in range(n - 1):
        s = xorshift32_step(s)
        j[i] = i + (s & 0x7FFFFFFF) % (n - i)
    keys = np.sort((j << 32) | np.arange(n - 1, dtype=np.int64))
    P = np.empty(n, dtype=np.int64)
    for i in range(n):
        pos, t = (int(j[i]) if i < n - 1 else n - 1), i
        while True:
            lo = int(np.searchsorted(keys, (pos << 32) | t, side="left"))
            if lo == 0 or (int(keys[lo - 1]) >> 32) != pos:
                break
            pos = t = int(keys[lo - 1]) & 0xFFFFFFFF
        P[i] = pos
    return P


class DeviceSGDOneClassSVM:
    """Subset of sklearn.linear_model.SGDOneClassSVM used by the reference (features.py:164-168, 352-358,
    multiple_features.py ``score_samples``): default hyper-parameters except ``nu``, ``max_iter``, ``random_state``."""

    def __init__(self, nu=0.5, max_iter=1000, tol=1e-3, random_state=None, n_iter_no_change=5):
        self.nu, self.max_iter, self.tol, self.random_state, self.n_iter_no_change = nu, max_iter, tol, random_state, n_iter_no_change

    def _seed(self):
        # _fit_one_class: seed = check_random_state(self.random_state).randint(0, np.iinfo(np.int32).max)
        from sklearn.utils import check_random_state
        return int(check_random_state(self.random_state).randint(0, MAX_INT32))

    def fit(self, X, y=None):
        if not torch.is_tensor(X):
            X = torch.as_tensor(np.asarray(X))
        if X.dtype != torch.float32:
            raise NotImplementedError("DeviceSGDOneClassSVM: float32 inputs (scikit-learn's float64 path, _plain_sgd64, rounds differently; "
                                      "the reference fits float32 score maps)")
        if not X.is_cuda:
            X = X.cuda()
        X = X.contiguous()
        n, F = X.shape
        L = _native.lib()
        ws = torch.empty(L.cmdiad_ocsvm_fit_workspace_bytes(n, F), dtype=torch.uint8, device=X.device)
        tab = np.ascontiguousarray(xorshift32_pow2_table())
        coef = np.zeros(F, dtype=np.float32)
        offset, n_iter = ctypes.c_double(0.0), ctypes.c_int(0)
        rc = L.cmdiad_ocsvm_fit(X.data_ptr(), n, F, float(self.nu), int(self.max_iter), float(self.tol if self.tol is not None else -np.inf),
                                int(self.n_iter_no_change), self._seed(), tab.ctypes.data_as(ctypes.c_void_p), coef.ctypes.data_as(ctypes.c_void_p),
                                ctypes.byref(offset), ctypes.byref(n_iter), ws.data_ptr(), ws.numel(), torch.cuda.current_stream(X.device).cuda_stream)
        _native.check(rc, "cmdiad_ocsvm_fit")
        self.coef_ = coef
        self.offset_ = np.array([offset.value])
        self.n_iter_ = n_iter.value
        self.t_ = 1.0 + self.n_iter_ * n
        self.n_features_in_ = F
        return self

    def decision_function(self, X):
        X = np.asarray(X.cpu() if torch.is_tensor(X) else X)
        return (X @ self.coef_.reshape(1, -1).T - self.offset_).ravel()   # safe_sparse_dot(X, coef_.T) - offset_

    def score_samples(self, X):
        return self.decision_function(X) + self.offset_
