"""Host-only walk over the C ABI (include/cmdiad_hip.h) with invalid arguments -- run by tests/test_sanitizers_cpu.py in a subprocess,
normally against the ASan + UBSan host build of the library (make -C cmdiad_amd/csrc asan).  No GPU is needed or touched by a call
that is REJECTED; a call that passes validation without a device fails in HIP and must come back as an error code too.
Every entry point is called with (1) all pointers NULL and all sizes 0, (2) all pointers NULL and sizes 1 / -1 / a large value:
whatever the combination, the library must return (0 for "nothing to do", or a negative cmdiad_status), never crash, and a rejected
call must leave a message in cmdiad_last_error()."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd import _native as nat  # noqa: E402

L = nat.lib()
P, I, U32, SZ, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_float
calls = rejected = 0
for name, args in sorted(nat.SIGNATURES.items()):
    fn = getattr(L, name)
    for ints in (0, 1, -1, 1 << 20):
        vals = []
        for a in args:
            if a is P:
                vals.append(None)
            elif a in (I, ctypes.c_int64, ctypes.c_long):
                vals.append(ints)
            elif a in (U32, SZ, ctypes.c_uint64, ctypes.c_ulonglong):
                vals.append(max(ints, 0))
            elif a in (F, ctypes.c_double):
                vals.append(1.0)
            else:
                vals.append(None)
        rc = fn(*vals)
        calls += 1
        assert rc <= 0, (name, ints, rc)
        if rc < 0:
            rejected += 1
            msg = L.cmdiad_last_error()
            assert msg and len(msg) > 3, (name, ints, rc, msg)
for name, args in sorted(nat.SIZE_QUERIES.items()):
    for ints in (0, 1, 1 << 20):
        getattr(L, name)(*[ints for _ in args])
        calls += 1
assert L.cmdiad_abi_version() >= 6
print(f"capi null-fuzz ok: {calls} calls, {rejected} rejected with a message")
