import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
# every test loads seeded synthetic weights after construction (no checkpoints offline): explicit opt-in, see models.Model
os.environ.setdefault("CMDIAD_ALLOW_RANDOM_INIT", "1")
# CMDIAD_TEST_AB=1: load the test-only build (make -C cmdiad_amd/csrc ab) that also contains the superseded kernel
# formulations, so the variant parity tests run instead of skipping.  Must be set before cmdiad_amd is imported.
_AB = os.path.join(REPO, "cmdiad_amd", "libcmdiad_hip_ab.so")
if os.environ.get("CMDIAD_TEST_AB") == "1" and os.path.exists(_AB):
    os.environ["CMDIAD_HIP_LIB"] = _AB


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "oracle_prefetch(*names): module-level zero-argument functions of the test's module that compute "
                                       "its CPU-oracle reference; started on host threads as soon as the session's tests are collected")
    config.addinivalue_line("markers", "rehearsal: a further whole-bench.py subprocess run beyond the two the default suite keeps; "
                                       "runs only with CMDIAD_TEST_FULL=1")
    config.addinivalue_line("markers", "slow: a GPU test of a minute or more (a whole bench.py run); selected by -m gpu like the rest")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def need_ab_variants(what):
    """Skip unless the loaded library is the test-only build with the A/B kernel variants (CMDIAD_TEST_AB=1)."""
    from cmdiad_amd import _native as nat
    if not nat.lib().cmdiad_has_ab_variants():
        pytest.skip(f"{what}: A/B variant, only in the test build (make -C cmdiad_amd/csrc ab; CMDIAD_TEST_AB=1)")


def pmap(fn, items, workers=4, total=None):
    """fn over items on `workers` host threads, results in order.  For the CPU-oracle loops of the GPU parity tests: one oracle
    `predict` is ~2 s on 32 host threads and does not scale beyond that (bench.py cpu_baseline), while the GPU box has 128 -- the
    torch CPU operators and the ctypes calls into the C oracle release the GIL, so `workers` samples run side by side, each on its
    share of the cores (`total`: samples in flight over all nested pmap levels; at most 32 threads each).  (Test infrastructure
    only; the oracle's arithmetic is untouched: every sample is still computed alone.)"""
    import concurrent.futures as cf
    import torch
    items = list(items)
    if workers <= 1 or len(items) <= 1:
        return [fn(it) for it in items]
    before = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or workers) // (total or workers))))
    try:
        with cf.ThreadPoolExecutor(max_workers=workers) as ex:
            return list(ex.map(fn, items))
    finally:
        torch.set_num_threads(before)


# ---- CPU-oracle references in the background -----------------------------------------------------------------------------------
# The GPU parity tests spend most of their wall-clock in the CPU oracle (oracle/pipeline.py: seconds per sample), one test after
# the other, while the GPU box's 128 host threads idle during every other test.  A test marks the module-level functions that
# compute its oracle side with @pytest.mark.oracle_prefetch("fn"); they start on a small host thread pool when collection ends
# and the test picks the result up with prefetched(fn).  Nothing about the references changes: the same function, the same
# arguments, computed once -- only earlier.  (Not selected / not started: prefetched() computes in line.)
_PREFETCH = {}
_PREFETCH_POOL = None


def _prefetch_start(key, fn):
    global _PREFETCH_POOL
    if key in _PREFETCH:
        return
    if _PREFETCH_POOL is None:
        import concurrent.futures as cf
        _PREFETCH_POOL = cf.ThreadPoolExecutor(max_workers=int(os.environ.get("CMDIAD_TEST_PREFETCH_WORKERS", "2")))
    _PREFETCH[key] = _PREFETCH_POOL.submit(fn)


def prefetched(fn):
    """Result of the zero-argument oracle function `fn` (computed in the background since collection, or now)."""
    key = (fn.__module__, fn.__name__)
    if key not in _PREFETCH:
        import concurrent.futures as cf
        fut = cf.Future()
        _PREFETCH[key] = fut
        try:
            fut.set_result(fn())
        except BaseException as e:      # noqa: BLE001 -- re-raised by result() in every test that asks
            fut.set_exception(e)
    return _PREFETCH[key].result()


def pytest_collection_finish(session):
    if os.environ.get("CMDIAD_TEST_PREFETCH", "1") == "0" or session.config.option.collectonly:
        return
    import torch
    if torch.cuda.device_count() == 0:      # (counting devices does not initialise the GPU): the marked tests will not run here
        return
    for item in session.items:
        m = item.get_closest_marker("oracle_prefetch")
        if m is not None:
            for name in m.args:
                fn = getattr(item.module, name)
                _prefetch_start((fn.__module__, fn.__name__), fn)


def pytest_collection_modifyitems(config, items):
    if os.environ.get("CMDIAD_TEST_FULL") == "1":
        return
    skip = pytest.mark.skip(reason="a further bench.py rehearsal: CMDIAD_TEST_FULL=1 runs it (the default suite keeps one default run and one --gpus 2 rehearsal)")
    for item in items:
        if item.get_closest_marker("rehearsal") is not None:
            item.add_marker(skip)
