"""cmdiad_ocsvm_fit (device fit of the late-fusion one-class SVMs, reference feature_extractors/features.py:352-358) against
scikit-learn's SGDOneClassSVM itself -- the dependency the reference calls -- on the same float32 data: identical coefficients,
offset and number of epochs (the device kernel keeps scikit-learn's sample order and arithmetic types)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _score_like(n, F, seed):
    """Rows shaped like the reference's s_lib / s_map_lib: positive anomaly scores of F memory banks, a few outliers."""
    g = np.random.default_rng(seed)
    x = np.abs(g.normal(1.0, 0.25, size=(n, F))).astype(np.float32)
    x[g.integers(0, n, max(1, n // 50))] *= 3.0
    return x


@pytest.mark.parametrize("n,F,nu,max_iter", [(2, 2, 0.5, 1000), (1000, 2, 0.5, 1000), (50176, 2, 0.5, 1000), (5000, 3, 0.5, 1000), (777, 1, 0.3, 1000),
                                             (20000, 4, 0.5, 3), (300000, 2, 0.5, 1000)])
def test_device_fit_equals_sklearn(n, F, nu, max_iter):
    from sklearn import linear_model
    from cmdiad_amd.ocsvm import DeviceSGDOneClassSVM
    X = _score_like(n, F, n + F)
    ref = linear_model.SGDOneClassSVM(random_state=42, nu=nu, max_iter=max_iter)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # ConvergenceWarning at max_iter = 3
        ref.fit(X)
    dev = DeviceSGDOneClassSVM(random_state=42, nu=nu, max_iter=max_iter).fit(torch.from_numpy(X).cuda())
    assert dev.n_iter_ == ref.n_iter_
    assert ref.coef_.dtype == np.float32
    np.testing.assert_array_equal(dev.coef_, ref.coef_)
    np.testing.assert_array_equal(dev.offset_, ref.offset_)
    assert dev.t_ == ref.t_
    q = _score_like(64, F, 3)
    np.testing.assert_array_equal(dev.score_samples(q), ref.score_samples(q))


def test_device_permutation_equals_host_construction():
    """The device kernels behind the shuffle (xorshift jump, sorted (target, step) pairs, chase) against the numpy restatement that
    tests/test_ocsvm_cpu.py pins to scikit-learn's ArrayDataset.shuffle: a fit of ONE epoch on data whose coordinates are the row
    numbers visits the rows in that order -- checked through max_iter = 1 results of two data sets that differ in one row."""
    from sklearn import linear_model
    from cmdiad_amd.ocsvm import DeviceSGDOneClassSVM
    n = 4097
    X = _score_like(n, 2, 9)
    for row in (0, 1234, n - 1):
        Y = X.copy()
        Y[row] *= 5.0
        ref = linear_model.SGDOneClassSVM(random_state=7, nu=0.5, max_iter=1, tol=None).fit(Y)
        dev = DeviceSGDOneClassSVM(random_state=7, nu=0.5, max_iter=1, tol=None).fit(torch.from_numpy(Y).cuda())
        np.testing.assert_array_equal(dev.coef_, ref.coef_)
        np.testing.assert_array_equal(dev.offset_, ref.offset_)
