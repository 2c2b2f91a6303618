"""Host logic of the device one-class-SVM fit (cmdiad_amd/ocsvm.py, csrc/ocsvm.hip): the xorshift32 jump table and the
parallel construction of the Fisher-Yates permutation, against scikit-learn's own dataset shuffling
(sklearn/utils/_seq_dataset.pyx.tp:137-145, utils/_random.pxd our_rand_r) -- the dependency the reference calls at
feature_extractors/features.py:352-358."""
import numpy as np
import pytest

from cmdiad_amd.ocsvm import fisher_yates_permutation, xorshift32_pow2_table, xorshift32_step


def test_xorshift_jump_table_matches_stepping():
    tab = xorshift32_pow2_table()
    rng = np.random.default_rng(0)
    for seed in [1, 42, 0x7FFFFFFF, 0xDEADBEEF] + [int(x) for x in rng.integers(1, 2**32, 6)]:
        for k in [0, 1, 2, 63, 64, 1000, 12345, 2**20 + 7]:
            s = seed
            for _ in range(k):
                s = xorshift32_step(s)
            j = seed
            for e in range(32):
                if (k >> e) & 1:
                    y = 0
                    for b in range(32):
                        if (j >> b) & 1:
                            y ^= int(tab[e, b])
                    j = y
            assert j == s, (seed, k)


@pytest.mark.parametrize("n,seed", [(2, 1), (3, 7), (37, 5), (1000, 1608637542), (4096, 42), (5001, 0)])
def test_parallel_fisher_yates_equals_sklearn_shuffle(n, seed):
    """ArrayDataset32.shuffle(seed) applied twice (the SGD re-shuffles the already permuted index array with the same seed every
    epoch) equals composing the permutation P built without executing the swaps."""
    from sklearn.utils._seq_dataset import ArrayDataset32
    X = np.arange(n, dtype=np.float32).reshape(n, 1)
    ds = ArrayDataset32(X, np.ones(n, dtype=np.float32), np.ones(n, dtype=np.float32), seed=1)

    def visit():
        return np.array([int(ds._next_py()[3]) for _ in range(n)])

    P = fisher_yates_permutation(n, seed)
    assert sorted(P.tolist()) == list(range(n))
    order = np.arange(n)
    for _ in range(3):
        ds._shuffle_py(seed)
        order = order[P]
        np.testing.assert_array_equal(visit(), order)
