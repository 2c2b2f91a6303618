"""ctypes front-end of oracle/cmdiad_oracle.c (CPU restatement; TEST INFRASTRUCTURE ONLY).

Builds the shared object on first use with oracle/Makefile (gcc, -ffp-contract=off).
Reference anchors are in the C file next to each function.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcmdiad_oracle.so")
_lib = None


_SO_ASAN = os.path.join(_HERE, "_build", "libcmdiad_oracle_asan.so")


def build(force=False):
    """-> path of the shared object.  CMDIAD_ORACLE_SANITIZE=1: the AddressSanitizer + UBSan build (`make asan`); the python
    process must then have been started with LD_PRELOAD=<gcc's libasan.so> (tests/test_sanitizers_cpu.py)."""
    src = os.path.join(_HERE, "cmdiad_oracle.c")
    so, target = (_SO_ASAN, ["asan"]) if os.environ.get("CMDIAD_ORACLE_SANITIZE") == "1" else (_SO, [])
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + target, stdout=subprocess.DEVNULL)
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def fps(xyz, G):
    """xyz [B,N,3] -> (idx [B,G] int32, centers [B,G,3])."""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    idx = np.empty((B, G), np.int32)
    cen = np.empty((B, G, 3), np.float32)
    lib().orc_fps(_p(xyz), B, N, G, _p(idx, ctypes.c_int32), _p(cen))
    return idx, cen


def knn_group(xyz, center, K):
    """xyz [B,N,3], center [B,G,3] -> (idx [B,G,K] int64, neighborhood [B,G,K,3])."""
    xyz, center = _f32(xyz), _f32(center)
    B, N, _ = xyz.shape
    G = center.shape[1]
    idx = np.empty((B, G, K), np.int64)
    nb = np.empty((B, G, K, 3), np.float32)
    lib().orc_knn_group(_p(xyz), _p(center), B, N, G, K, _p(idx, ctypes.c_int64), _p(nb))
    return idx, nb


def interp3nn(xyz1, xyz2, feat, want_out=True):
    """xyz1 [N,3], xyz2 [S,3], feat [S,D] -> (out [N,D] | None, idx3 [N,3] int32, w3 [N,3])."""
    xyz1, xyz2, feat = _f32(xyz1), _f32(xyz2), _f32(feat)
    N, S, D = xyz1.shape[0], xyz2.shape[0], feat.shape[1]
    out = np.empty((N, D), np.float32) if want_out else None
    idx3 = np.empty((N, 3), np.int32)
    w3 = np.empty((N, 3), np.float32)
    lib().orc_interp3nn(_p(xyz1), _p(xyz2), _p(feat), N, S, D, _p(out), _p(idx3, ctypes.c_int32), _p(w3))
    return out, idx3, w3


def xyz_patch(interp, nz, S, P):
    """interp [N,D] point-major, nz [N] int64 -> [P*P, D]."""
    interp = _f32(interp)
    nz = np.ascontiguousarray(nz, dtype=np.int64)
    N, D = interp.shape
    out = np.empty((P * P, D), np.float32)
    lib().orc_xyz_patch(_p(interp), _p(nz, ctypes.c_int64), N, D, S, P, _p(out))
    return out


def l2_min_argmin(q, bank):
    q, bank = _f32(q), _f32(bank)
    Q, D = q.shape
    Nb = bank.shape[0]
    mv = np.empty(Q, np.float32)
    mi = np.empty(Q, np.int64)
    lib().orc_l2_min_argmin(_p(q), _p(bank), Q, Nb, D, _p(mv), _p(mi, ctypes.c_int64))
    return mv, mi


def bilinear_up(img, H):
    img = _f32(img)
    h = img.shape[0]
    out = np.empty((H, H), np.float32)
    lib().orc_bilinear_up(_p(img), h, H, _p(out))
    return out


def pil_gaussian_blur_u8(img, radius=4.0):
    """Pillow's ImageFilter.GaussianBlur on an 8-bit image (restated: oracle/cmdiad_oracle.c orc_pil_gaussian_blur_u8)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty_like(img)
    f = lib().orc_pil_gaussian_blur_u8
    f.restype = ctypes.c_int
    rc = f(_p(img), ctypes.c_int(img.shape[0]), ctypes.c_int(img.shape[1]), ctypes.c_float(radius), _p(out))
    if rc != 0:
        raise ValueError("image side shorter than the box window")
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    """pointnet2_ops ball_query restated (oracle/cmdiad_oracle.c orc_ball_query): xyz [B,N,3], new_xyz [B,M,3] -> idx [B,M,nsample] int32."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.empty((B, M, nsample), np.int32)
    lib().orc_ball_query(_p(xyz), _p(new_xyz), B, N, M, ctypes.c_float(radius), nsample, _p(idx, ctypes.c_int32))
    return idx
