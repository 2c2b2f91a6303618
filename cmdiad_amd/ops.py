"""Tensor-level entry points: torch tensors in (device memory + stream plumbing only),
libcmdiad_hip.so kernels underneath.  Every op raises if the tensors are not on the GPU or the
native library is missing -- there is no eager / CPU fallback here by design.
"""
import ctypes
import os

import torch

from . import _native as nat

ACT_NONE, ACT_GELU, ACT_RELU, ACT_RELU_POST = 0, 1, 2, 3
# "no candidate yet" key: the largest NON-NEGATIVE int64.  Every real key is (fp32 bits of d2 >= +0) << 32 | row, i.e. has bit 63
# clear, so this sentinel is >= every real key under the kernels' unsigned atomicMin AND under the signed MIN all-reduce of the
# row-sharded search (a rank whose shard is empty contributes only sentinels and can never win the reduce).
KEY_EMPTY = 0x7FFFFFFFFFFFFFFF


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Raw HIP stream of torch's current stream.  torch.cuda.current_stream() costs ~100 us per call (availability checks,
    Stream object construction) -- 20 ms per image over the ~600 launches of a B = 1 predict; the raw getters cost < 1 us."""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_SHARED_STREAMS = {}


def shared_stream(device, role, priority=0):
    """One HIP stream per (device, role, priority) for the whole process.  torch hands out streams from a pool of 32 per device
    and priority, round robin: objects that each create their own (a predictor per class, a method object per run) walk
    through the pool and end up on the SAME underlying stream as somebody else's -- RCCL's communicator stream, the graph-capture
    stream -- and a capture that forks onto such a stream puts RCCL's events "in a capturing stream": its watchdog thread then
    raises and takes the process down (tools/fuzz_pipeline.py, profiles/r5_notes.md section 15).  Roles are few and fixed."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, role, int(priority))
    st = _SHARED_STREAMS.get(key)
    if st is None:
        st = _SHARED_STREAMS[key] = torch.cuda.Stream(dev, priority=int(priority))
    return st


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise nat.NativeError(f"{name}: tensor must live on the GPU (cmdiad_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")


def _call(name, *args):
    nat.check(getattr(nat.lib(), name)(*args), name)


# ------------------------------------------------------------------------------------ point cloud
def fps(xyz, G, n_valid=None):
    """xyz [B,N,3] f32 -> (idx [B,G] int32, centers [B,G,3] f32).  models/models.py:70-78."""
    _chk(xyz, torch.float32, "fps.xyz"); _chk(n_valid, torch.int32, "fps.n_valid")
    B, N, _ = xyz.shape
    idx = torch.empty((B, G), dtype=torch.int32, device=xyz.device)
    cen = torch.empty((B, G, 3), dtype=torch.float32, device=xyz.device)
    wsb = nat.lib().cmdiad_fps_workspace_bytes(B, N)
    ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=xyz.device) if wsb else None
    _call("cmdiad_fps", _p(xyz), _p(n_valid), B, N, G, _p(idx), _p(cen), _p(ws), wsb, _stream())
    return idx, cen


def knn_group(xyz, center, K, n_valid=None, want_idx=True):
    """-> (idx [B,G,K] int64, neighborhood [B,G,K,3] f32).  models/models.py:88-113."""
    _chk(xyz, torch.float32, "knn.xyz"); _chk(center, torch.float32, "knn.center")
    B, N, _ = xyz.shape
    G = center.shape[1]
    idx = torch.empty((B, G, K), dtype=torch.int64, device=xyz.device) if want_idx else None
    nb = torch.empty((B, G, K, 3), dtype=torch.float32, device=xyz.device)
    wsb = nat.lib().cmdiad_knn_workspace_bytes(B, N)     # the binned clouds of the neighbourhood search (clouds >= 2 048 points)
    ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=xyz.device)
    _call("cmdiad_knn_group_ws", _p(xyz), _p(n_valid), _p(center), B, N, G, K, _p(idx), _p(nb), _p(ws), wsb, _stream())
    return idx, nb


def unorganize(organized_pc, n_max=None):
    """organized_pc [B,3,H,W] f32 -> (xyz [B,Nmax,3], nz [B,Nmax] i32, pix2pt [B,HW] i32, n_valid [B] i32).
    multiple_features.py:10-25."""
    _chk(organized_pc, torch.float32, "unorganize.pc")
    B, _, H, W = organized_pc.shape
    HW = H * W
    n_max = n_max or HW
    dev = organized_pc.device
    xyz = torch.zeros((B, n_max, 3), dtype=torch.float32, device=dev)
    nz = torch.zeros((B, n_max), dtype=torch.int32, device=dev)
    pix2pt = torch.empty((B, HW), dtype=torch.int32, device=dev)
    n_valid = torch.empty((B,), dtype=torch.int32, device=dev)
    _call("cmdiad_unorganize", _p(organized_pc), B, HW, n_max, _p(xyz), _p(nz), _p(pix2pt), _p(n_valid), _stream())
    return xyz, nz, pix2pt, n_valid


def ball_query(radius, nsample, xyz, new_xyz, n_valid=None):
    """pointnet2_ops ball_query: xyz [B,N,3], new_xyz [B,M,3] -> idx [B,M,nsample] int32."""
    _chk(xyz, torch.float32, "ball_query.xyz"); _chk(new_xyz, torch.float32, "ball_query.new_xyz")
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=xyz.device)
    _call("cmdiad_ball_query", _p(xyz), _p(n_valid), _p(new_xyz), B, N, M, float(radius), int(nsample), _p(idx), _stream())
    return idx


def gather_points(feat, idx):
    """feat [B,C,N] f32, idx [B,...] int32 -> [B,C,...] (gather_operation / grouping_operation)."""
    _chk(feat, torch.float32, "gather.feat"); _chk(idx, torch.int32, "gather.idx")
    B, C, N = feat.shape
    J = idx[0].numel()
    out = torch.empty((B, C, *idx.shape[1:]), dtype=torch.float32, device=feat.device)
    _call("cmdiad_gather_points", _p(feat), _p(idx), B, C, N, J, _p(out), _stream())
    return out


def interp3nn(xyz, center, n_valid=None):
    """-> (idx3 [B,N,3] i32, w3 [B,N,3] f32).  models/pointnet2_utils.py:45-71."""
    _chk(xyz, torch.float32, "interp3nn.xyz"); _chk(center, torch.float32, "interp3nn.center")
    B, N, _ = xyz.shape
    idx3 = torch.zeros((B, N, 3), dtype=torch.int32, device=xyz.device)
    w3 = torch.zeros((B, N, 3), dtype=torch.float32, device=xyz.device)
    S = center.shape[1]
    wsb = nat.lib().cmdiad_interp3nn_workspace_bytes(B, S)      # the binned centres of the neighbourhood search (S >= 64)
    ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=xyz.device)
    _call("cmdiad_interp3nn_ws", _p(xyz), _p(n_valid), _p(center), B, N, S, _p(idx3), _p(w3), _p(ws), wsb, _stream())
    return idx3, w3


def interp_gather(feat, idx3, w3, n_valid=None):
    """feat [B,S,D] f32 -> [B,N,D] f32 (pointnet2_utils.py:72)."""
    _chk(feat, torch.float32, "interp_gather.feat")
    B, S, D = feat.shape
    N = idx3.shape[1]
    out = torch.zeros((B, N, D), dtype=torch.float32, device=feat.device)
    _call("cmdiad_interp_gather", _p(feat), _p(idx3), _p(w3), _p(n_valid), B, N, S, D, _p(out), _stream())
    return out


def xyz_patch_fused(feat, idx3, w3, pix2pt, size=224, P=56, mean=0.0, inv_std=1.0, want_f32=True, want_bf16=False):
    """features.py:169-184 fused with the interpolation gather.  -> (patch_f32 [B,P*P,D] | None, bf16 | None)."""
    _chk(feat, torch.float32, "xyz_patch.feat")
    B, S, D = feat.shape
    N = idx3.shape[1]
    o32 = torch.empty((B, P * P, D), dtype=torch.float32, device=feat.device) if want_f32 else None
    o16 = torch.empty((B, P * P, D), dtype=torch.bfloat16, device=feat.device) if want_bf16 else None
    _call("cmdiad_xyz_patch_fused", _p(feat), _p(idx3), _p(w3), _p(pix2pt), B, N, S, D, size, P, float(mean),
          float(inv_std), _p(o32), _p(o16), _stream())
    return o32, o16


# ------------------------------------------------------------------------------------ dense blocks
def gemm(A, W, bias=None, act=ACT_NONE, residual=None, group_bias=None, group_rows=1, out_f32=None, out_bf16=None,
         want_f32=False, want_bf16=True, out_pre_bf16=None, dact_of=None, split_k=1, m_count=None,
         row_scale=None, ln_xb=None, ln_part=None, add2=None):
    """epilogue(A[M,K] . W[N,K]^T); A, W bf16.  Returns (out_f32 | None, out_bf16 | None).  m_count (device int32 [1]): only the
    first min(M, m_count) rows are computed and stored (a compacted row set whose size is known on the device only).
    LayerNorm fold (cmdiad_gemm_args, ABI 3): row_scale [>= M rounded up to 256] f32 multiplies the accumulator per row (consumer);
    ln_xb [M,N] bf16 + ln_part [N/64, M, 2] f32 (+ add2 [M,N] f32) are the producer's extra outputs of the in-place residual form."""
    _chk(A, torch.bfloat16, "gemm.A"); _chk(W, torch.bfloat16, "gemm.W")
    _chk(row_scale, torch.float32, "gemm.row_scale"); _chk(ln_xb, torch.bfloat16, "gemm.ln_xb")
    _chk(ln_part, torch.float32, "gemm.ln_part"); _chk(add2, torch.float32, "gemm.add2")
    M, K = A.shape
    N = W.shape[0]
    if row_scale is not None and row_scale.numel() < (M + 255) // 256 * 256:
        raise ValueError(f"gemm.row_scale: {row_scale.numel()} values, need M rounded up to 256 = {(M + 255) // 256 * 256}")
    if ln_part is not None and ln_part.numel() < (N // 64) * M * 2:
        raise ValueError(f"gemm.ln_part: {ln_part.numel()} values, need (N/64) * M * 2 = {(N // 64) * M * 2}")
    if out_f32 is None and want_f32:
        out_f32 = torch.empty((split_k, M, N) if split_k > 1 else (M, N), dtype=torch.float32, device=A.device)
    if out_bf16 is None and want_bf16:
        out_bf16 = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    a = nat.GemmArgs(_p(A), K, _p(W), K, M, N, K, _p(bias), _p(group_bias), group_rows, act,
                     _p(residual), N, _p(out_f32), N, _p(out_bf16), N, _p(out_pre_bf16), _p(dact_of), split_k, _p(m_count),
                     _p(row_scale), _p(ln_xb), N, _p(ln_part), _p(add2), N)
    _call("cmdiad_gemm_bf16", ctypes.byref(a), _stream())
    return out_f32, out_bf16


_streamk_ws = {}


def gemm_streamk_eligible(M, N, K):
    return bool(nat.lib().cmdiad_gemm_streamk_eligible(M, N, K))


def gemm_streamk(A, W, bias, residual, out_f32=None):
    """out_f32 = A[M,K] . W[N,K]^T + bias + residual (in place when out_f32 is residual) on the stream-K kernel
    (cmdiad_gemm_streamk_bf16; bit-identical to gemm()).  The workspace (64 MiB of hand-over slots + counters, zeroed once) is kept
    per device; launches are serialised per device (see below)."""
    _chk(A, torch.bfloat16, "gemm_streamk.A"); _chk(W, torch.bfloat16, "gemm_streamk.W")
    _chk(bias, torch.float32, "gemm_streamk.bias"); _chk(residual, torch.float32, "gemm_streamk.residual")
    M, K = A.shape
    N = W.shape[0]
    if out_f32 is None:
        out_f32 = torch.empty((M, N), dtype=torch.float32, device=A.device)
    # ONE workspace and ONE launch in flight per device: block b of the kernel spins on block b - 1's counter, so two launches
    # that overlap (different streams) could fill every CU with waiting blocks whose predecessors are not resident yet -- a launch on
    # another stream first waits for the previous one's event
    key = A.device.index
    ent = _streamk_ws.get(key)
    if ent is None:
        ent = _streamk_ws[key] = {"ws": torch.zeros((nat.lib().cmdiad_gemm_streamk_workspace_bytes(),), dtype=torch.uint8, device=A.device),
                                  "done": None, "stream": None}
    ws = ent["ws"]
    cur = torch.cuda.current_stream(A.device)
    # Not under stream capture: the cross-stream serialisation is an event recorded OUTSIDE the graph (waiting on it from inside a
    # capture, or keeping one that was recorded inside, invalidates the capture or raises), and replays of a captured stream-K
    # launch could overlap an eager one.  The plain tile kernel returns the same bits.
    if torch.cuda.is_current_stream_capturing():
        return gemm(A, W, bias=bias, residual=residual, out_f32=out_f32, want_f32=True, want_bf16=False)[0]
    if ent["done"] is not None and ent["stream"] != cur.cuda_stream:
        cur.wait_event(ent["done"])
    a = nat.GemmArgs(_p(A), K, _p(W), K, M, N, K, _p(bias), None, 1, ACT_NONE, _p(residual), N, _p(out_f32), N, None, 0, None, None, 1,
                     None, None, None, 0, None, None, 0)
    _call("cmdiad_gemm_streamk_bf16", ctypes.byref(a), _p(ws), ws.numel(), _stream())
    ent["done"] = torch.cuda.Event()
    ent["done"].record(cur)
    ent["stream"] = cur.cuda_stream
    return out_f32


def ln_stats_finalize(part, M, chunks, eps, rstd=None, want_mean=False):
    """part [chunks, M, 2] f32 (gemm(..., ln_part=)) -> rstd [M rounded up to 256] f32 (first M valid) (, mean [M])."""
    _chk(part, torch.float32, "ln_stats.part")
    if rstd is None:
        rstd = torch.empty(((M + 255) // 256 * 256,), dtype=torch.float32, device=part.device)
    mean = torch.empty((M,), dtype=torch.float32, device=part.device) if want_mean else None
    _call("cmdiad_ln_stats_finalize", _p(part), M, chunks, float(eps), _p(rstd), _p(mean), _stream())
    return (rstd, mean) if want_mean else rstd


def gemm_tn(P, Q, split_k=1, want_colsum=False):
    """sum_m P[m,:]^T Q[m,:]: P [M,N1], Q [M,N2] bf16 row-major -> f32 [N1,N2] (split_k == 1) or slabs [split_k,N1,N2];
    with want_colsum also the column sums of P ([N1] or [split_k,N1]) -> (out, colsum)."""
    _chk(P, torch.bfloat16, "gemm_tn.P"); _chk(Q, torch.bfloat16, "gemm_tn.Q")
    M, N1 = P.shape
    N2 = Q.shape[1]
    out = torch.empty((split_k, N1, N2) if split_k > 1 else (N1, N2), dtype=torch.float32, device=P.device)
    cs = torch.empty((split_k, N1) if split_k > 1 else (N1,), dtype=torch.float32, device=P.device) if want_colsum else None
    _call("cmdiad_gemm_tn_bf16", _p(P), N1, _p(Q), N2, M, N1, N2, split_k, _p(out), N2, _p(cs), _stream())
    return (out, cs) if want_colsum else out


def gemm_qkv(A, W, bias, B, T, q, k, vt, row_scale=None):
    """A [B*T,C] bf16 -> q,k [B,H,Tp,64], vt [B,H,64,Tp] (pre-allocated, zero-initialised padding).  row_scale [B*T] f32:
    the LayerNorm-folded form (qkv = row_scale[m] * (A . W^T) + bias)."""
    _chk(A, torch.bfloat16, "qkv.A"); _chk(W, torch.bfloat16, "qkv.W"); _chk(row_scale, torch.float32, "qkv.row_scale")
    C = A.shape[1]
    _call("cmdiad_gemm_qkv", _p(A), _p(W), _p(bias), _p(row_scale), B, T, C, _p(q), _p(k), _p(vt), _stream())


def attention(q, k, vt, B, H, T, out=None):
    """-> out [B*T, H*64] bf16."""
    if out is None:
        out = torch.empty((B * T, H * 64), dtype=torch.bfloat16, device=q.device)
    _call("cmdiad_attention", _p(q), _p(k), _p(vt), B, H, T, _p(out), _stream())
    return out


def encoder_tail(h2, gb, W3b, W4, b4, groups, Mg):
    """h2 [groups*Mg,256] bf16, gb [groups,512] f32 -> tokens [groups,384] f32 (cmdiad_encoder_tail: h3 never leaves LDS)."""
    _chk(h2, torch.bfloat16, "tail.h2"); _chk(gb, torch.float32, "tail.gb")
    tok = torch.empty((groups, 384), dtype=torch.float32, device=h2.device)
    _call("cmdiad_encoder_tail", _p(h2), _p(gb), _p(W3b), _p(W4), _p(b4), groups, Mg, _p(tok), _stream())
    return tok


def conv2d_nhwc(x, W, N, ksize=3, stride=1, bias=None, act=ACT_NONE, residual=None, out_f32=None, out_bf16=None,
                want_f32=False, want_bf16=True):
    """x [B,H,W,C] bf16 NHWC, W [N, ksize*ksize*C] bf16 (tap-major) -> (out_f32 | None, out_bf16 | None), each [B,Ho,Wo,ld]
    with ld = the given buffer's last dimension (>= N; extra columns are left untouched).  cmdiad_conv2d_nhwc_bf16."""
    _chk(x, torch.bfloat16, "conv.x"); _chk(W, torch.bfloat16, "conv.W"); _chk(residual, torch.float32, "conv.residual")
    B, H, Wd, C = x.shape
    pad = 1 if ksize == 3 else 0
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (Wd + 2 * pad - ksize) // stride + 1
    if out_f32 is None and want_f32:
        out_f32 = torch.empty((B, Ho, Wo, N), dtype=torch.float32, device=x.device)
    if out_bf16 is None and want_bf16:
        out_bf16 = torch.empty((B, Ho, Wo, N), dtype=torch.bfloat16, device=x.device)
    a = nat.ConvArgs(_p(x), B, H, Wd, C, _p(W), N, ksize, stride, _p(bias), act,
                     _p(residual), residual.shape[-1] if residual is not None else 0,
                     _p(out_f32), out_f32.shape[-1] if out_f32 is not None else 0,
                     _p(out_bf16), out_bf16.shape[-1] if out_bf16 is not None else 0)
    _call("cmdiad_conv2d_nhwc_bf16", ctypes.byref(a), _stream())
    return out_f32, out_bf16


def im2col3x3(img, stride=2, ld=64):
    """img [B,C,H,W] f32 -> [B*Ho*Wo, ld] bf16, column c * 9 + ky * 3 + kx (padding 1; zero beyond 9 C): cmdiad_im2col3x3_bf16."""
    _chk(img, torch.float32, "im2col3x3.img")
    B, C, H, W = img.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    cols = torch.empty((B * Ho * Wo, ld), dtype=torch.bfloat16, device=img.device)
    _call("cmdiad_im2col3x3_bf16", _p(img), B, C, H, W, stride, ld, _p(cols), _stream())
    return cols


def conv_stem(x, w, bias, stride=2):
    """x [B,Cin<=4,H,W] f32 NCHW, w [Cout,Cin,3,3] f32 (BatchNorm folded), bias [Cout] -> ReLU(conv) as bf16 NHWC."""
    _chk(x, torch.float32, "stem.x"); _chk(w, torch.float32, "stem.w"); _chk(bias, torch.float32, "stem.bias")
    B, Cin, H, Wd = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (Wd - 1) // stride + 1
    out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=x.device)
    _call("cmdiad_conv_stem", _p(x), _p(w), _p(bias), B, Cin, H, Wd, Cout, stride, _p(out), _stream())
    return out


def upsample_bicubic(x, C, H, W, out_bf16=None, nchw=False):
    """x [B,h,w,ld] f32 NHWC (first C channels used) -> bf16 NHWC [B,H,W,ldo] (given or allocated with ldo = C) or, with
    nchw=True, f32 [B,C,H,W].  torch's bicubic, align_corners=False (cmdiad_upsample_bicubic)."""
    _chk(x, torch.float32, "bicubic.x")
    B, h, w, ld = x.shape
    if nchw:
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        _call("cmdiad_upsample_bicubic", _p(x), B, h, w, C, ld, H, W, None, 0, _p(out), _stream())
        return out
    if out_bf16 is None:
        out_bf16 = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=x.device)
    _call("cmdiad_upsample_bicubic", _p(x), B, h, w, C, ld, H, W, _p(out_bf16), out_bf16.shape[-1], None, _stream())
    return out_bf16


def transformer_block_workspace_bytes(M, C, hidden):
    return int(nat.lib().cmdiad_transformer_block_workspace_bytes(M, C, hidden))


BLOCK_LN1_READY, BLOCK_PREP_NEXT = 1, 2   # include/cmdiad_hip.h CMDIAD_BLOCK_*
_BLOCK_FIELDS = ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                 "qkv_wf", "qkv_bf", "fc1_wf", "fc1_bf")


def transformer_block(x, pos, blk, B, T, H, eps, q, k, vt, workspace, flags=0):
    """x [B*T, C] f32 updated in place by one whole pre-LN block (cmdiad_transformer_block_fwd).  blk: dict of packed
    weights (runtime._pack_block); its ctypes struct is built once and cached in the dict.  flags: BLOCK_LN1_READY /
    BLOCK_PREP_NEXT chain the LayerNorm fold across consecutive calls on one workspace (needs blk's folded weights)."""
    _chk(x, torch.float32, "block.x"); _chk(pos, torch.float32, "block.pos")
    w = blk.get("_struct")
    if w is None:
        w = nat.BlockWeights(*[blk[n].data_ptr() if blk.get(n) is not None else None for n in _BLOCK_FIELDS])
        blk["_struct"] = w
    C = x.shape[1]
    _call("cmdiad_transformer_block_fwd", _p(x), _p(pos), ctypes.byref(w), B, T, C, H, blk["fc1_w"].shape[0], float(eps), int(flags),
          _p(q), _p(k), _p(vt), _p(workspace), workspace.numel(), _stream())
    return x


def layernorm(x, gamma, beta, eps, add=None, out_bf16=None, out_f32=None, want_bf16=True, stats=None):
    """x [M,C] f32 (updated in place to x+add when add is given) -> LN(x) as bf16 and/or f32."""
    _chk(x, torch.float32, "ln.x"); _chk(add, torch.float32, "ln.add")
    M, C = x.shape
    if out_bf16 is None and want_bf16:
        out_bf16 = torch.empty((M, C), dtype=torch.bfloat16, device=x.device)
    ld = out_f32.stride(0) if out_f32 is not None else 0
    mean_o, rstd_o = stats if stats is not None else (None, None)
    _call("cmdiad_layernorm", _p(x), _p(add), _p(gamma), _p(beta), float(eps), M, C, _p(out_bf16), _p(out_f32), ld,
          _p(mean_o), _p(rstd_o), _stream())
    return out_bf16


def col_moments(x):
    """x [rows, C] f32 -> (mean [C], biased variance [C]) in float64 (cmdiad_col_moments)."""
    _chk(x, torch.float32, "col_moments.x")
    rows, C = x.shape
    acc = torch.zeros((2, C), dtype=torch.float64, device=x.device)
    _call("cmdiad_col_moments", _p(x), rows, C, x.stride(0), _p(acc[0]), _p(acc[1]), _stream())
    mean = acc[0] / rows
    return mean, acc[1] / rows - mean * mean


def bn_relu_fwd(z, scale, shift, residual=None, relu=True, want_bf16=True, want_f32=False):
    """z [M,C] f32 -> z * scale + shift (+ residual) (ReLU) as bf16 [M,C] (and / or f32): batch-statistics BatchNorm2d + ReLU
    (cmdiad_bn_relu_fwd).  Returns the bf16 tensor, or (bf16 | None, f32) when want_f32."""
    _chk(z, torch.float32, "bn_relu.z"); _chk(scale, torch.float32, "bn_relu.scale"); _chk(shift, torch.float32, "bn_relu.shift")
    _chk(residual, torch.float32, "bn_relu.residual")
    M, C = z.shape
    y = torch.empty((M, C), dtype=torch.bfloat16, device=z.device) if want_bf16 else None
    y32 = torch.empty((M, C), dtype=torch.float32, device=z.device) if want_f32 else None
    _call("cmdiad_bn_relu_fwd", _p(z), _p(scale), _p(shift), _p(residual), 1 if relu else 0, M, C, _p(y), _p(y32), _stream())
    return (y, y32) if want_f32 else y


def bn_relu_bwd(dy, z, scale, shift, mean, rstd, chunks=None, masked=True):
    """Backward of bn_relu_fwd: dy, z [M,C] f32; per-channel f32 vectors -> (dz bf16 [M,C], dgamma [C], dbeta [C]).  masked=False:
    the layer had no ReLU of its own (dy already carries the mask of the ReLU after the residual sum).
    (cmdiad_bn_relu_bwd_reduce -> cmdiad_bn_partials_sum -> cmdiad_bn_relu_bwd_apply)."""
    _chk(dy, torch.float32, "bn_bwd.dy"); _chk(z, torch.float32, "bn_bwd.z")
    M, C = z.shape
    if chunks is None:   # ~128 rows per workgroup (64 columns x 4 row lanes each), at most 256 row ranges
        chunks = max(16, min(256, (M + 127) // 128))
    mk = 1 if masked else 0
    p1 = torch.empty((chunks, C), dtype=torch.float32, device=z.device)
    p2 = torch.empty((chunks, C), dtype=torch.float32, device=z.device)
    _call("cmdiad_bn_relu_bwd_reduce", _p(dy), _p(z), _p(scale), _p(shift), _p(mean), _p(rstd), mk, M, C, chunks, _p(p1), _p(p2), _stream())
    dbeta = torch.empty((C,), dtype=torch.float32, device=z.device)
    dgamma = torch.empty((C,), dtype=torch.float32, device=z.device)
    _call("cmdiad_bn_partials_sum", _p(p1), _p(p2), chunks, C, _p(dbeta), _p(dgamma), _stream())
    dz = torch.empty((M, C), dtype=torch.bfloat16, device=z.device)
    _call("cmdiad_bn_relu_bwd_apply", _p(dy), _p(z), _p(scale), _p(shift), _p(mean), _p(rstd), _p(dbeta), _p(dgamma), mk, M, C, _p(dz),
          _stream())
    return dz, dgamma, dbeta


def relu_bwd(dx, y, want_bf16=True, want_f32=False):
    """dx f32, y bf16 (the ReLU output), same shape -> dx where y > 0 else 0, as bf16 (and / or f32) (cmdiad_relu_bwd_bf16)."""
    _chk(dx, torch.float32, "relu_bwd.dx"); _chk(y, torch.bfloat16, "relu_bwd.y")
    assert dx.shape == y.shape
    dz = torch.empty(y.shape, dtype=torch.bfloat16, device=y.device) if want_bf16 else None
    dz32 = torch.empty(y.shape, dtype=torch.float32, device=y.device) if want_f32 else None
    _call("cmdiad_relu_bwd_bf16", _p(dx), _p(y), dx.numel(), _p(dz), _p(dz32), _stream())
    return (dz, dz32) if want_f32 else dz


def upsample_bicubic_bwd(grad_out, h, w):
    """grad_out [B,H,W,C] f32 NHWC -> gradient of the [B,h,w,C] input of upsample_bicubic (cmdiad_upsample_bicubic_bwd)."""
    _chk(grad_out, torch.float32, "bicubic_bwd.grad_out")
    B, H, W, C = grad_out.shape
    tmp = torch.empty((B, H, w, C), dtype=torch.float32, device=grad_out.device)
    out = torch.empty((B, h, w, C), dtype=torch.float32, device=grad_out.device)
    _call("cmdiad_upsample_bicubic_bwd", _p(grad_out), B, H, W, C, h, w, _p(tmp), _p(out), _stream())
    return out


def pad_nhwc(x, rows_multiple=64):
    """x [B,H,W,C] bf16 -> (buffer [guard + rows + guard, C] bf16, guard, rows): the zero-bordered copy [B,H+2,W+2,C] flattened to
    rows (zero-padded up to a multiple of rows_multiple), between two zero guards of W+3 rows so that every 3x3 tap's row
    offset stays inside the buffer (cmdiad_pad_nhwc_bf16)."""
    _chk(x, torch.bfloat16, "pad.x")
    B, H, W, C = x.shape
    guard = W + 3
    rows = (B * (H + 2) * (W + 2) + rows_multiple - 1) // rows_multiple * rows_multiple
    buf = torch.zeros((guard + rows + guard, C), dtype=torch.bfloat16, device=x.device)
    _call("cmdiad_pad_nhwc_bf16", _p(x), B, H, W, C, _p(buf[guard:]), _stream())
    return buf, guard, rows


def moments3(xyz):
    """xyz [rows, 3] f32 -> (mean [3], covariance [3,3]) in float64, biased (cmdiad_moments3)."""
    _chk(xyz, torch.float32, "moments3.xyz")
    rows = xyz.shape[0]
    acc = torch.zeros((9,), dtype=torch.float64, device=xyz.device)
    _call("cmdiad_moments3", _p(xyz), rows, _p(acc), _stream())
    mean = acc[:3] / rows
    m2 = torch.stack([acc[[3, 4, 5]], acc[[4, 6, 7]], acc[[5, 7, 8]]]) / rows
    return mean, m2 - torch.outer(mean, mean)


def encoder_stage1(neigh, w1b1, W2, b2, groups, Mg):
    """-> (h2 [groups*Mg,256] bf16, gmax [groups,256] f32, gmax_bf16)."""
    dev = neigh.device
    h2 = torch.empty((groups * Mg, 256), dtype=torch.bfloat16, device=dev)
    g32 = torch.empty((groups, 256), dtype=torch.float32, device=dev)
    g16 = torch.empty((groups, 256), dtype=torch.bfloat16, device=dev)
    _call("cmdiad_encoder_stage1", _p(neigh), _p(w1b1), _p(W2), _p(b2), groups, Mg, _p(h2), _p(g32), _p(g16), _stream())
    return h2, g32, g16


def gemm_groupmax(A, W, bias, groups, Mg, want_bf16=False):
    N, K = W.shape
    o32 = torch.empty((groups, N), dtype=torch.float32, device=A.device)
    o16 = torch.empty((groups, N), dtype=torch.bfloat16, device=A.device) if want_bf16 else None
    _call("cmdiad_gemm_groupmax", _p(A), _p(W), _p(bias), groups, Mg, N, K, _p(o32), _p(o16), _stream())
    return o32, o16


# ------------------------------------------------------------------------------------ scoring
# 16-bit operand type of the patch-library distance GEMM.  bfloat16 since round 5: the same kernel runs 6.5 % faster on bf16 operands
# than on fp16 ones (5.05-5.16 against 5.38-5.57 ms, same box: fewer mantissa bits toggle, the chip holds a higher clock --
# profiles/r5_notes.md section 6), and what the three mantissa bits buy is invisible behind the extractor's own 16-bit noise: the
# operand rounding moves a distance of 4-35 by ~0.03, the bf16 feature chain by ~0.5 (tests/test_gpu_predictor.py: image scores,
# pixel maps and AUROCs identical to the last printed digit on both types); the winner is re-scored in fp32 either way.
# CMDIAD_SEARCH_DTYPE=fp16 restores IEEE half operands.
SEARCH_DTYPE = torch.float16 if os.environ.get("CMDIAD_SEARCH_DTYPE", "bf16").lower() in ("fp16", "float16", "half") else torch.bfloat16


def normalize_cast(x, mean=0.0, inv_std=1.0, want_f32=False, want_sq=True, dtype=None, skip_leading=0):
    """x [rows,D] f32 -> (16-bit [rows,D] (the search operand type: bf16 by default, or fp16), f32 normalised | None,
    row |.|^2 of the ROUNDED rows | None).
    skip_leading = s > 0: x is [G, s + R, D] and the first s rows of every group are not taken (the cls token of the ViT's
    [B, 785, C] tokens): outputs have G * R rows, read in place -- no gather copy of the patch rows in between."""
    _chk(x, torch.float32, "normalize_cast.x")
    dtype = dtype or SEARCH_DTYPE
    if skip_leading:
        G, T, D = x.shape
        group = T - skip_leading
        rows = G * group
    else:
        rows, D = x.shape
        group = 0
    o16 = torch.empty((rows, D), dtype=dtype, device=x.device)
    o32 = torch.empty((rows, D), dtype=torch.float32, device=x.device) if want_f32 else None
    sq = torch.empty((rows,), dtype=torch.float32, device=x.device) if want_sq else None
    _call("cmdiad_normalize_cast_rows", _p(x), rows, D, group, int(skip_leading), float(mean), float(inv_std), _p(o16), _p(o32), _p(sq),
          1 if dtype == torch.float16 else 0, _stream())
    return o16, o32, sq


def new_keys(Q, device, runner=False):
    """Packed nearest-neighbour keys, "no candidate" everywhere.  runner=True: [2, Q] -- plane 0 the best row of every query,
    plane 1 its runner-up (the nearest row outside the winner's group of 16 rows, include/cmdiad_hip.h) for the exact fp32
    decision of l2_rescore; every function below that takes `keys` accepts either shape."""
    return torch.full((2, Q) if runner else (Q,), KEY_EMPTY, dtype=torch.int64, device=device)


def _key_planes(keys, n, name):
    """-> (pointer of the best plane, pointer of the runner-up plane or None) of a [n] or [2, n] int64 key tensor."""
    _chk(keys, torch.int64, name)
    if keys.dim() == 2:
        if keys.shape[0] != 2 or keys.shape[1] < n:
            raise ValueError(f"{name}: expected [2, >={n}] keys, got {tuple(keys.shape)}")
        return _p(keys[0]), _p(keys[1])
    if keys.shape[0] < n:
        raise ValueError(f"{name}: {keys.shape[0]} keys for {n} query rows")
    return _p(keys), None


def l2_min_keys(q16, q_sq, bank16, bank_sq, keys, row_offset=0):
    Q, D = q16.shape
    if q16.dtype != bank16.dtype:
        raise TypeError("l2_min_keys: queries and bank must share the 16-bit dtype")
    k1, k2 = _key_planes(keys, Q, "l2_min_keys.keys")
    _call("cmdiad_l2_min_keys", _p(q16), _p(q_sq), _p(bank16), _p(bank_sq), Q, bank16.shape[0], D, row_offset,
          k1, k2, 1 if q16.dtype == torch.float16 else 0, _stream())
    return keys


class DedupPlan:
    """Device-resident plan of cmdiad_rows_dedup_plan: slot [Q] i32 (compacted row answering for q), rows [Q] i32, count [1] i32,
    q16 [Q,D] / q_sq [Q] (first count rows live).  Buffers are reused when one is passed back in."""
    __slots__ = ("slot", "rows", "count", "q16", "q_sq", "work")

    def __init__(self, Q, D, dtype, device):
        self.slot = torch.empty((Q,), dtype=torch.int32, device=device)
        self.rows = torch.empty((Q,), dtype=torch.int32, device=device)
        self.count = torch.zeros((1,), dtype=torch.int32, device=device)
        self.q16 = torch.empty((Q, D), dtype=dtype, device=device)
        self.q_sq = torch.empty((Q,), dtype=torch.float32, device=device)
        self.work = torch.empty((max(int(nat.lib().cmdiad_rows_dedup_workspace_bytes(Q)), 4),), dtype=torch.uint8, device=device)


def rows_dedup_plan(q16, q_sq, plan=None):
    """Exact removal of the repeated constant rows (patches without a foreground pixel) of a 16-bit query set: include/cmdiad_hip.h."""
    Q, D = q16.shape
    if q16.dtype not in (torch.float16, torch.bfloat16) or not q16.is_contiguous():
        raise TypeError("rows_dedup_plan: contiguous 16-bit queries")
    _chk(q_sq, torch.float32, "rows_dedup_plan.q_sq")
    if plan is None or plan.q16.shape != q16.shape or plan.q16.dtype != q16.dtype:
        plan = DedupPlan(Q, D, q16.dtype, q16.device)
    _call("cmdiad_rows_dedup_plan", _p(q16), _p(q_sq), Q, D, _p(plan.work), _p(plan.slot), _p(plan.rows), _p(plan.count),
          _p(plan.q16), _p(plan.q_sq), _stream())
    return plan


def l2_min_keys_counted(q16, q_sq, count, bank16, bank_sq, keys, row_offset=0):
    """l2_min_keys over the first count[0] (device int32) rows of q16."""
    Q, D = q16.shape
    if q16.dtype != bank16.dtype:
        raise TypeError("l2_min_keys_counted: queries and bank must share the 16-bit dtype")
    _chk(count, torch.int32, "l2_min_keys_counted.count")
    k1, k2 = _key_planes(keys, Q, "l2_min_keys_counted.keys")
    _call("cmdiad_l2_min_keys_counted", _p(q16), _p(q_sq), _p(count), Q, _p(bank16), _p(bank_sq), bank16.shape[0], D, row_offset,
          k1, k2, 1 if q16.dtype == torch.float16 else 0, _stream())
    return keys


def l2_min_keys_segments(q16, q_sq, seg_counts, seg_stride, bank16, bank_sq, keys, row_offset=0):
    """l2_min_keys over the segments [w * seg_stride, w * seg_stride + seg_counts[w]) of q16 (seg_counts: device int32 [n_seg]):
    ONE launch over the live query tiles of all segments (include/cmdiad_hip.h)."""
    Q, D = q16.shape
    if q16.dtype != bank16.dtype:
        raise TypeError("l2_min_keys_segments: queries and bank must share the 16-bit dtype")
    _chk(seg_counts, torch.int32, "l2_min_keys_segments.seg_counts")
    n_seg = seg_counts.shape[0]
    if n_seg * seg_stride != Q or keys.shape[-1] < Q:
        raise ValueError(f"l2_min_keys_segments: {n_seg} segments of {seg_stride} rows != {Q} query rows (keys: {keys.shape[-1]})")
    k1, k2 = _key_planes(keys, Q, "l2_min_keys_segments.keys")
    _call("cmdiad_l2_min_keys_segments", _p(q16), _p(q_sq), _p(seg_counts), n_seg, seg_stride, _p(bank16), _p(bank_sq),
          bank16.shape[0], D, row_offset, k1, k2, 1 if q16.dtype == torch.float16 else 0, _stream())
    return keys


def rows_expand_f32(rows_compact, slot, out=None):
    """out[q] = rows_compact[slot[q]] (f32 rows): per-row results computed on the compacted rows, back on every original row."""
    _chk(rows_compact, torch.float32, "rows_expand.rows"); _chk(slot, torch.int32, "rows_expand.slot")
    Q, D = slot.shape[0], rows_compact.shape[1]
    if out is None:
        out = torch.empty((Q, D), dtype=torch.float32, device=rows_compact.device)
    _call("cmdiad_rows_expand_f32", _p(rows_compact), _p(slot), Q, D, _p(out), _stream())
    return out


def keys_expand(keys_compact, slot, keys):
    """keys[..., q] = keys_compact[..., slot[q]] ([n] or [2, n] key tensors: both planes)."""
    _chk(slot, torch.int32, "keys_expand.slot")
    if keys_compact.dim() != keys.dim():
        raise ValueError("keys_expand: compact and expanded keys must have the same number of planes")
    if keys.dim() == 2:
        for pl in range(2):
            _call("cmdiad_keys_expand", _p(keys_compact[pl]), _p(slot), slot.shape[0], _p(keys[pl]), _stream())
        return keys
    _call("cmdiad_keys_expand", _p(keys_compact), _p(slot), slot.shape[0], _p(keys), _stream())
    return keys


def l2_rescore(q32, bank32, keys, min_val=None, min_idx=None, row_offset=0):
    """Exact fp32 distance and row of every query's nearest library row.  keys [Q]: the search's winner is taken as it is;
    keys [2, Q] (best + runner-up): both are measured in fp32 and the nearer one wins (ties: the lower row) -- torch.min on the
    fp32 distance matrix (features.py:227)."""
    Q, D = q32.shape
    if min_val is None:
        min_val = torch.zeros((Q,), dtype=torch.float32, device=q32.device)
        min_idx = torch.zeros((Q,), dtype=torch.int64, device=q32.device)
    if keys.dim() == 2:
        _call("cmdiad_l2_rescore2", _p(q32), _p(bank32), _p(keys[0]), _p(keys[1]), Q, bank32.shape[0], D, row_offset, None,
              _p(min_val), _p(min_idx), _stream())
        return min_val, min_idx
    _call("cmdiad_l2_rescore", _p(q32), _p(bank32), _p(keys), Q, bank32.shape[0], D, row_offset, _p(min_val),
          _p(min_idx), _stream())
    return min_val, min_idx


def l2_rescore_pair_d2(q32, bank32, keys, d2_pair, row_offset=0):
    """Squared fp32 distances [2, Q] of the candidates of keys [2, Q] whose rows lie in this shard's [row_offset, +rows); the other
    entries are left as they are (zero-filled by the caller, summed over the shards, then l2_choose)."""
    Q, D = q32.shape
    _call("cmdiad_l2_rescore2", _p(q32), _p(bank32), _p(keys[0]), _p(keys[1]), Q, bank32.shape[0], D, row_offset, _p(d2_pair),
          None, None, _stream())
    return d2_pair


def l2_choose(keys, d2_pair, min_val, min_idx):
    """The decision of l2_rescore on [2, Q] keys from squared distances that were summed over the shards."""
    _call("cmdiad_l2_choose", _p(keys[0]), _p(keys[1]), _p(d2_pair), keys.shape[1], _p(min_val), _p(min_idx), _stream())
    return min_val, min_idx


def bank_block16(bank32):
    """[Nb,D] f32 -> the library copy laid out for the fp32 matrix cores (cmdiad_bank_block16), a flat f32 tensor."""
    _chk(bank32, torch.float32, "bank_block16.bank")
    Nb, D = bank32.shape
    out = torch.empty((int(nat.lib().cmdiad_bank_block16_floats(Nb, D)),), dtype=torch.float32, device=bank32.device)
    _call("cmdiad_bank_block16", _p(bank32), Nb, D, _p(out), _stream())
    return out


def reweight_scan(probes, bank32, blk16=None, top3=None, row_offset=0):
    """probes [R,D] f32 -> top3 [R,3] packed keys (int64 view of u64; exact fp32 d2).  One pass over the library per 32
    probes.  blk16: ops.bank_block16(bank32) (built on the fly when absent -- engine.Bank keeps one)."""
    _chk(probes, torch.float32, "reweight.probes"); _chk(bank32, torch.float32, "reweight.bank")
    R, D = probes.shape
    Nb = bank32.shape[0]
    if blk16 is None:
        blk16 = bank_block16(bank32)
    if top3 is None:
        top3 = torch.full((R, 3), KEY_EMPTY, dtype=torch.int64, device=probes.device)
    for lo in range(0, R, 32):
        r = min(32, R - lo)
        wsb = nat.lib().cmdiad_reweight_workspace_bytes(r, Nb)
        ws = torch.empty(max(wsb // 8, 1), dtype=torch.int64, device=probes.device)
        _call("cmdiad_reweight_scan", _p(probes[lo:lo + r]), _p(bank32), _p(blk16), r, Nb, D, row_offset, _p(top3[lo:lo + r]),
              _p(ws), wsb, _stream())
    return top3


def reweight_scan_pair(probes0, bank0, blk0, probes1, bank1, blk1, row_offset0=0, row_offset1=0):
    """reweight_scan for the TWO libraries of a scored batch in one launch pair (cmdiad_reweight_scan_pair): -> (top3_0, top3_1),
    each [R,3] packed keys, identical to two separate reweight_scan calls.  R <= 32 per library (a batch of 32 images)."""
    for t, n in ((probes0, "probes0"), (bank0, "bank0"), (probes1, "probes1"), (bank1, "bank1")):
        _chk(t, torch.float32, "reweight_pair." + n)
    (R0, D), R1 = probes0.shape, probes1.shape[0]
    Nb0, Nb1 = bank0.shape[0], bank1.shape[0]
    if R0 > 32 or R1 > 32 or Nb0 == 0 or Nb1 == 0 or probes1.shape[1] != D:
        raise ValueError("reweight_scan_pair: R <= 32 per library, both libraries non-empty, equal D")
    dev = probes0.device
    top0 = torch.full((R0, 3), KEY_EMPTY, dtype=torch.int64, device=dev)
    top1 = torch.full((R1, 3), KEY_EMPTY, dtype=torch.int64, device=dev)
    wsb = nat.lib().cmdiad_reweight_pair_workspace_bytes(Nb0, Nb1)
    ws = torch.empty(max(wsb // 8, 1), dtype=torch.int64, device=dev)
    _call("cmdiad_reweight_scan_pair", _p(probes0), _p(bank0), _p(blk0), R0, Nb0, row_offset0, _p(top0),
          _p(probes1), _p(bank1), _p(blk1), R1, Nb1, row_offset1, _p(top1), D, _p(ws), wsb, _stream())
    return top0, top1


def l2_dist_matrix(q32, bank32):
    """Exact fp32 [Q,Nb] matrix of L2 distances (features.py:186-190 materialised; API compatibility only)."""
    _chk(q32, torch.float32, "dist_matrix.q"); _chk(bank32, torch.float32, "dist_matrix.bank")
    Q, D = q32.shape
    out = torch.empty((Q, bank32.shape[0]), dtype=torch.float32, device=q32.device)
    _call("cmdiad_l2_dist_matrix", _p(q32), _p(bank32), Q, bank32.shape[0], D, _p(out), _stream())
    return out


def unpack_keys(keys):
    """int64 view of packed u64 keys -> (value f32, index int64)."""
    idx = keys & 0xFFFFFFFF
    val = (keys >> 32).to(torch.int32).view(torch.float32) if keys.numel() else keys.float()
    return val, idx


# ------------------------------------------------------------------------------------ small ops
def im2col_patch8(rgb):
    _chk(rgb, torch.float32, "im2col.rgb")
    B, _, S, _ = rgb.shape
    out = torch.empty((B * (S // 8) ** 2, 192), dtype=torch.bfloat16, device=rgb.device)
    _call("cmdiad_im2col_patch8", _p(rgb), B, S, _p(out), _stream())
    return out


def vit_assemble(patch_out, cls, pos, B, P, C):
    tokens = torch.empty((B * (P + 1), C), dtype=torch.float32, device=patch_out.device)
    _call("cmdiad_vit_assemble", _p(patch_out), _p(cls), _p(pos), B, P, C, _p(tokens), _stream())
    return tokens


def bilinear_up(x, H):
    _chk(x, torch.float32, "bilinear.x")
    B, h, _ = x.shape
    out = torch.empty((B, H, H), dtype=torch.float32, device=x.device)
    _call("cmdiad_bilinear_up", _p(x), B, h, H, _p(out), _stream())
    return out


def blur8_maps(maps, radius=4.0):
    """KNNGaussianBlur (utils/utils.py:71-83) on device: maps [n,H,W] f32 -> [n,H,W] f32, bit-exact with Pillow's 8-bit path."""
    _chk(maps, torch.float32, "blur8.maps")
    n, H, W = maps.shape
    out = torch.empty_like(maps)
    _call("cmdiad_blur8_maps", _p(maps), n, H, W, float(radius), _p(out), _stream())
    return out


def ocsvm_score_maps(maps, lambdas, coef, offset):
    """seg_fuser.score_samples over the lambda-weighted map stack: maps [B,K,HW] f32 -> [B,HW] f64
    (multiple_features.py:985-992; coef / offset from a host-fitted sklearn SGDOneClassSVM)."""
    import ctypes
    import numpy as np
    _chk(maps, torch.float32, "ocsvm.maps")
    B, K, HW = maps.shape
    lam = np.ascontiguousarray(lambdas, dtype=np.float32)
    cf = np.ascontiguousarray(np.asarray(coef).reshape(-1), dtype=np.float64)
    assert lam.shape == (K,) and cf.shape == (K,)
    out = torch.empty((B, HW), dtype=torch.float64, device=maps.device)
    _call("cmdiad_ocsvm_score_maps", _p(maps), B, K, HW, lam.ctypes.data_as(ctypes.c_void_p), cf.ctypes.data_as(ctypes.c_void_p),
          float(np.asarray(offset).reshape(-1)[0]), _p(out), _stream())
    return out


def linear3(x, wb, act=ACT_NONE):
    """x [M,3] f32, wb [N,4] f32 -> act(W x + b) as bf16 [M,N]."""
    _chk(x, torch.float32, "linear3.x")
    M, N = x.shape[0], wb.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    _call("cmdiad_linear3", _p(x), _p(wb), M, N, act, _p(out), _stream())
    return out


def cast_bf16(x):
    _chk(x, torch.float32, "cast.x")
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _call("cmdiad_cast_bf16", _p(x), x.numel(), _p(out), _stream())
    return out


def transpose_bf16(x):
    _chk(x, torch.bfloat16, "transpose.x")
    r, c = x.shape
    out = torch.empty((c, r), dtype=torch.bfloat16, device=x.device)
    _call("cmdiad_transpose_bf16", _p(x), r, c, _p(out), _stream())
    return out
