"""GPU: the variant parity tests in the TEST-ONLY build.  The superseded kernel formulations (csrc/ab/*.inc) exist only in
libcmdiad_hip_ab.so (make ab); the tests that compare them with the production kernels skip in a process that has loaded the
production library.  This test runs exactly those tests in a child process with CMDIAD_TEST_AB=1, so that they are exercised
wherever `pytest -m gpu` runs (the parent keeps the production library: what it loads is what ships)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_variant_parity_tests_run_in_the_test_build():
    ab = os.path.join(REPO, "cmdiad_amd", "libcmdiad_hip_ab.so")
    if not os.path.exists(ab):
        pytest.skip("libcmdiad_hip_ab.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    files = [f for f in ("tests/test_gpu_kernels.py", "tests/test_gpu_dedup.py", "tests/test_gpu_fullsize.py", "tests/test_gpu_nets.py")
             if "need_ab_variants(" in open(os.path.join(REPO, f)).read().replace("import need_ab_variants", "")]
    assert files
    env = dict(os.environ, CMDIAD_TEST_AB="1")
    env.pop("CMDIAD_HIP_LIB", None)
    # only the tests that HAVE a variant parametrisation (FPS "reg", kNN "block", the 4-wave GEMM shapes, the lock-step distance GEMM):
    # the rest of those files ran in the parent against the production library already
    select = "fps_ or knn_group_bit_exact or knn_group_ties or gemm_epilogues or pointmae_encoder_stages or l2_min"
    out = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", select], cwd=REPO, env=env,
                         capture_output=True, text=True, timeout=1500)
    tail = out.stdout[-1500:]
    assert out.returncode == 0, tail + out.stderr[-1500:]
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 40, tail
    assert "skipped" not in tail.splitlines()[-1], tail          # nothing needed a variant that the test build does not have
