"""CPU-only sanitizer runs -- never on the GPU box (GPU AddressSanitizer is not available
on this pool, and the device code of the sanitized library is compiled without it):

  * the C oracle (oracle/cmdiad_oracle.c) rebuilt with -fsanitize=address,undefined (`make -C oracle asan`) runs the golden-vector and
    property tests: every restated algorithm on every fixture without an out-of-bounds access, a signed overflow or a misaligned load;
  * the host side of the C ABI (argument validation, launch plumbing, error channel of every entry point of include/cmdiad_hip.h)
    rebuilt with hipcc's host AddressSanitizer + UBSan (`make -C cmdiad_amd/csrc asan`) is walked with invalid arguments
    (tests/capi_nullfuzz_worker.py): rejected with a message, never a crash.
Both run in subprocesses with the matching sanitizer runtime preloaded (python itself is not instrumented)."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAD = ("ERROR: AddressSanitizer", "runtime error:", "ERROR: LeakSanitizer", "AddressSanitizer:DEADLYSIGNAL")


def _clean(out):
    text = out.stdout + out.stderr
    assert not any(b in text for b in BAD), text[-4000:]
    assert out.returncode == 0, text[-4000:]
    return text


def test_c_oracle_under_asan_ubsan():
    if shutil.which("gcc") is None:
        pytest.skip("gcc not on PATH")
    rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("gcc's libasan.so not found")
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               CMDIAD_ORACLE_SANITIZE="1")
    # (the two "through reference glue" tests are a minute of torch-CPU network arithmetic around a handful of C calls that the
    # other tests make as well: left out of the sanitized run)
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-k", "not through_reference_glue",
                          os.path.join(REPO, "tests", "test_oracle_golden.py"), os.path.join(REPO, "tests", "test_oracle_properties.py")],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=REPO)
    text = _clean(out)
    assert " passed" in text and "failed" not in text, text[-2000:]
    # the sanitized object was the one in use
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from oracle import kernels as k; k.lib(); "
                            "print(open('/proc/self/maps').read().count('libcmdiad_oracle_asan.so') > 0)" % REPO],
                           capture_output=True, text=True, timeout=300, env=env)
    assert probe.stdout.strip().endswith("True"), probe.stdout + probe.stderr


def test_c_abi_host_side_under_asan_ubsan():
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rts:
        pytest.skip("hipcc's AddressSanitizer runtime not found")
    res = subprocess.run(["make", "-C", os.path.join(REPO, "cmdiad_amd", "csrc"), "-j8", "asan"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lib = os.path.join(REPO, "cmdiad_amd", "libcmdiad_hip_asan.so")
    env = dict(os.environ, LD_PRELOAD=rts[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:protect_shadow_gap=0",
               UBSAN_OPTIONS="print_stacktrace=1", CMDIAD_HIP_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "capi_nullfuzz_worker.py")], capture_output=True, text=True,
                         timeout=900, env=env)
    text = _clean(out)
    assert "capi null-fuzz ok" in text, text[-2000:]
    # and the C-ABI surface test of the ordinary suite against the sanitized object (symbols, version, signatures)
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(REPO, "tests", "test_host_cpu.py"),
                          "-k", "abi or symbol or fallback"], capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    _clean(out)
