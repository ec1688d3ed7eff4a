"""Thin tensor->pointer wrappers over the C ABI (include/thinktwice_hip.h).

All functions take device tensors and launch on torch's current stream.  No function here
computes anything in PyTorch: allocation and stream handling only.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (TT_BF16, TT_F16, TT_F32, TTError, check, clear_device_faults, cur_stream, device_faults, lib, ptr,  # noqa: F401
                   raise_on_device_fault, require_cuda)

_c = ctypes.c_int


def dtype_code(t):
    if t.dtype == torch.float32:
        return TT_F32
    if t.dtype == torch.bfloat16:
        return TT_BF16
    if t.dtype == torch.float16:
        return TT_F16
    raise _lib.TTError(f"unsupported dtype {t.dtype}")


def frustum_voxel_index(frustum, mats, voxel_lo, voxel_size, batch_size, num_cams, want_f32=False):
    """frustum [D,fH,fW,4] f32 (device); mats [B*ncam,2,4,4] f32 (device): inv(ida), s2e@inv(K).
    voxel_lo / voxel_size: 3 python floats each.  -> geom_xyz int32 [B, ncam*D*fH*fW, 3]."""
    require_cuda(frustum, mats)
    assert frustum.is_contiguous() and mats.is_contiguous()
    D, fH, fW, _ = frustum.shape
    geom = torch.empty(batch_size, num_cams * D * fH * fW, 3, dtype=torch.int32, device=frustum.device)
    gf = torch.empty(geom.shape, dtype=torch.float32, device=frustum.device) if want_f32 else None
    lo = (ctypes.c_float * 3)(*[float(v) for v in voxel_lo])
    sz = (ctypes.c_float * 3)(*[float(v) for v in voxel_size])
    rc = lib().tt_frustum_voxel_index(_c(batch_size), _c(num_cams), _c(D), _c(fH), _c(fW),
                                      ptr(frustum), ptr(mats), lo, sz, ptr(geom), ptr(gf),
                                      cur_stream(frustum.device))
    check(rc, "tt_frustum_voxel_index")
    return (geom, gf) if want_f32 else geom


_LIFT_SPLAT_ATOMIC = os.environ.get("TT_LIFT_SPLAT_ATOMIC", "0") == "1"


def lift_splat(depth_logits, context, geom_xyz, voxel_num, batch_size, num_cams, out=None,
               out_coff=0, rot_flip=False, record=True):
    """depth_logits [B*ncam,fH,fW,D], context [B*ncam,fH,fW,C] (channel-last, f32 or bf16),
    geom_xyz int32 [B, ncam*D*fH*fW, 3] -> out f32 [B, Y, X, Ctot] (accumulated into `out`)."""
    require_cuda(depth_logits, context, geom_xyz)
    assert depth_logits.is_contiguous() and context.is_contiguous() and geom_xyz.is_contiguous()
    BN, fH, fW, D = depth_logits.shape
    C = context.shape[-1]
    vx, vy, vz = (int(v) for v in voxel_num)
    if out is None:
        oh, ow = (vx, vy) if rot_flip else (vy, vx)
        out = torch.zeros(batch_size, oh, ow, C, dtype=torch.float32, device=context.device)
    # workspace form: (strip, cell) partial rows + an ordered per-cell reduce instead of f32 atomics (bit-reproducible);
    # TT_LIFT_SPLAT_ATOMIC=1 selects the single-kernel atomic form for A/B
    ws, ws_bytes = None, 0
    if not _LIFT_SPLAT_ATOMIC:
        ws_bytes = int(lib().tt_lift_splat_workspace_bytes(_c(batch_size), _c(num_cams), _c(D), _c(fH), _c(fW), _c(C),
                                                           _c(vx), _c(vy)))
        if ws_bytes > 0:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=context.device)
    rc = lib().tt_lift_splat_fwd_ws(_c(batch_size), _c(num_cams), _c(D), _c(fH), _c(fW), _c(C), _c(vx),
                                    _c(vy), _c(vz), ptr(depth_logits), ptr(context),
                                    _c(dtype_code(context)), ptr(geom_xyz), ptr(out),
                                    _c(out.shape[-1]), _c(out_coff), _c(1 if rot_flip else 0),
                                    ptr(ws) if ws is not None else None, ctypes.c_longlong(ws_bytes),
                                    cur_stream(context.device))
    check(rc, "tt_lift_splat_fwd_ws")
    if record:
        from . import autodiff
        if autodiff.TAPE is not None:
            autodiff.TAPE.lift_splat(depth_logits, context, geom_xyz, (vx, vy, vz), batch_size, num_cams, out, out_coff,
                                     rot_flip)
    return out


def lift_splat_bwd(depth_logits, context, geom_xyz, voxel_num, batch_size, num_cams, gout, out_coff, gdepth, gctx):
    """gdepth += , gctx += backward of lift_splat (f32; gout: the BEV gradient buffer [B, Y, X, Ctot])."""
    require_cuda(depth_logits, context, geom_xyz, gout, gdepth, gctx)
    BN, fH, fW, D = depth_logits.shape
    C = context.shape[-1]
    vx, vy, vz = voxel_num
    assert all(t.is_contiguous() and t.dtype == torch.float32 for t in (depth_logits, context, gout, gdepth, gctx))
    check(lib().tt_lift_splat_bwd(_c(batch_size), _c(num_cams), _c(D), _c(fH), _c(fW), _c(C), _c(vx), _c(vy), _c(vz),
                                  ptr(depth_logits), ptr(context), ptr(geom_xyz), ptr(gout), _c(gout.shape[-1]),
                                  _c(out_coff), ptr(gdepth), ptr(gctx), cur_stream(context.device)), "tt_lift_splat_bwd")


# ----------------------------------------------------------------------------- conv / linear
_AUTO_SPLITK = os.environ.get("TT_CONV_AUTO_SPLITK", "1") == "1"
CONV_PROFILE = None   # bench.py sets this to a list to collect (flops, start, end, shape) per launch
CONV_KERNELS = None   # same order as CONV_PROFILE: tt_conv_last_kernel() of the launch
CONV_BYTES = None     # same order as CONV_PROFILE: compulsory HBM bytes of the launch (each operand moved once);
                      # dense: int; sparse: (bytes per live output row, fixed bytes)

class _ConvDesc(ctypes.Structure):
    _fields_ = [
        ("in_", ctypes.c_void_p), ("N", _c), ("H", _c), ("W", _c), ("Cin", _c), ("in_cstride", _c),
        ("in_coff", _c), ("in_nstride", ctypes.c_longlong),
        ("weight", ctypes.c_void_p), ("Cout", _c), ("KH", _c), ("KW", _c), ("stride", _c),
        ("pad", _c), ("dil", _c),
        ("out", ctypes.c_void_p), ("OH", _c), ("OW", _c), ("out_cstride", _c), ("out_coff", _c),
        ("out_nstride", ctypes.c_longlong),
        ("pixel_shuffle2", _c),
        ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
        ("shift_n", ctypes.c_void_p), ("shift_n_mod", _c),
        ("res1", ctypes.c_void_p), ("res1_cstride", _c), ("res1_coff", _c),
        ("res2", ctypes.c_void_p), ("res2_cstride", _c), ("res2_coff", _c),
        ("act", _c), ("dtype", _c), ("out_dtype", _c),
        ("gather_idx", ctypes.c_void_p), ("m_dev", ctypes.c_void_p), ("splitk_ws", ctypes.c_void_p),
        ("weight_x3", ctypes.c_void_p), ("row_perm", ctypes.c_void_p), ("row_mask", ctypes.c_void_p),
        ("splitk_slices", _c), ("in_pair", _c), ("out_pair", _c),
        ("weight_h2", ctypes.c_void_p), ("out2", ctypes.c_void_p), ("out2_cstride", _c), ("out2_coff", _c),
        ("res1_up_h", _c), ("res1_up_w", _c), ("res1_f32", _c),
    ]


def _dp(t):
    return None if t is None else t.data_ptr()


def _last_conv_kernel():
    L = lib()
    L.tt_conv_last_kernel.restype = ctypes.c_char_p
    return L.tt_conv_last_kernel().decode()


SPARSE_PAIRS = None   # bench.py: device int64 counter of existing (row, tap) pairs, filled by sp_tile_plan


def sp_tile_plan(nbr, m_dev):
    """Tile plan of a rulebook (tt_sp_tile_plan): (row_perm int32 [M], row_mask int32-typed uint32 [M]) -- the output
    rows sorted by tap-occupancy mask, so that a 256-row tile of the gathered GEMM only visits the union of its taps."""
    require_cuda(nbr, m_dev)
    M, KV = nbr.shape
    ws_bytes = int(lib().tt_sp_tile_plan_workspace_bytes(_ll(M)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=nbr.device)
    perm = torch.empty(M, dtype=torch.int32, device=nbr.device)
    mask = torch.empty(M, dtype=torch.int32, device=nbr.device)
    check(lib().tt_sp_tile_plan(ptr(nbr), ptr(m_dev), _ll(M), _c(KV), ptr(ws), _ll(ws_bytes), ptr(perm), ptr(mask),
                                ptr(SPARSE_PAIRS), cur_stream(nbr.device)), "tt_sp_tile_plan")
    return perm, mask


def gather_conv(feats, nbr, m_dev, w, *, scale=None, shift=None, act=0, res=None, out_dtype=None, w_x3=None, plan=None,
                out=None, in_rows=None, stride=1, _no_tape=False, bn_raw=False):
    """Sparse convolution as a gathered GEMM on MFMA: feats [R_in, C] rows, nbr int32 [M, taps]
    (rulebook, -1 = no input), m_dev device int (live output rows), w [Cout,1,taps,C] -> [M, Cout].  `stride`: the spatial
    stride of the sparse conv the rulebook came from (a dispatch hint: stride-1 rulebooks of cell-ordered rows take the
    run-staged kernel, csrc/sp_conv_runs.hip; the arithmetic does not depend on it)."""
    require_cuda(feats, nbr, w)
    M, taps = nbr.shape
    Cout, _, KW, Cin = w.shape
    assert KW == taps and feats.shape[1] == Cin and feats.is_contiguous() and nbr.is_contiguous()
    if out is None:
        out = torch.empty(M, Cout, dtype=out_dtype or feats.dtype, device=feats.device)
    assert tuple(out.shape) == (M, Cout) and out.is_contiguous()
    d = _ConvDesc()
    d.in_ = feats.data_ptr(); d.N = M; d.H = 1; d.W = 1; d.Cin = Cin; d.in_cstride = Cin; d.in_coff = 0
    d.in_nstride = 0
    d.weight = w.data_ptr(); d.Cout = Cout; d.KH = 1; d.KW = KW; d.stride = stride; d.pad = 0; d.dil = 1
    d.out = out.data_ptr(); d.OH = 1; d.OW = 1; d.out_cstride = Cout; d.out_coff = 0; d.out_nstride = 0
    d.scale = _dp(scale); d.shift = _dp(shift)
    d.res1 = _dp(res); d.res1_cstride = 0 if res is None else res.shape[-1]
    d.act = act; d.dtype = dtype_code(feats); d.out_dtype = dtype_code(out)
    d.gather_idx = nbr.data_ptr(); d.m_dev = _dp(m_dev)
    d.weight_x3 = _dp(w_x3)
    if plan is not None:
        d.row_perm, d.row_mask = plan[0].data_ptr(), plan[1].data_ptr()
    if CONV_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib().tt_conv2d_fwd(ctypes.byref(d), cur_stream(feats.device)), "tt_conv2d_fwd(gather)")
    if CONV_PROFILE is not None:
        e1.record()
        # FLOP accounting over the EXISTING (row, tap) pairs: per-row tap counts of the rulebook (measurement only)
        CONV_PROFILE.append((2.0 * Cout * KW * Cin, e0, e1, f"sparse M<={M} N={Cout} K={KW * Cin}", m_dev, M,
                             ((nbr >= 0).sum(1), 2.0 * Cout * Cin)))
        if CONV_KERNELS is not None:
            CONV_KERNELS.append(_last_conv_kernel())
        if CONV_BYTES is not None:      # one input row and one output row (+ residual row) per live output row, rulebook row
            esz, osz = feats.element_size(), out.element_size()
            CONV_BYTES.append((Cin * esz + Cout * osz + (Cout * esz if res is not None else 0) + KW * 4,
                               w.numel() * w.element_size()))
    if not _no_tape:
        from . import autodiff
        if autodiff.TAPE is not None:
            autodiff.TAPE.gather_conv(feats, nbr, m_dev, w, scale, shift, act, res, out, in_rows, bn_raw)
    return out


def gather_conv_wgrad(feats, nbr, m_dev, dy, taps, cin_pad=None):
    """Weight gradient of gather_conv: feats [R_in, Cin] f32, nbr int32 [M, taps], dy [M, Cout] f32 -> [Cout,1,taps,cin_pad]."""
    require_cuda(feats, nbr, dy)
    M, Cout = dy.shape
    Cin = feats.shape[1]
    cin_pad = cin_pad or Cin
    out = torch.empty(Cout, 1, taps, cin_pad, dtype=torch.float32, device=dy.device)
    L = lib()
    L.tt_gather_conv_wgrad_workspace_bytes.restype = ctypes.c_longlong
    nb = int(L.tt_gather_conv_wgrad_workspace_bytes(_ll(M), _c(Cout), _c(Cin), _c(cin_pad), _c(taps)))
    ws = torch.empty(nb, dtype=torch.uint8, device=dy.device)
    assert feats.is_contiguous() and dy.is_contiguous() and nbr.is_contiguous() and feats.dtype == torch.float32
    check(L.tt_gather_conv_wgrad(ptr(feats), _c(Cin), _c(Cin), ptr(nbr), ptr(m_dev), _ll(M), _c(taps), ptr(dy), _c(Cout),
                                 _c(Cout), _c(cin_pad), _c(0), ptr(out), ptr(ws), _ll(nb), _st(dy)), "tt_gather_conv_wgrad")
    return out


def sp_inverse_rulebook(nbr, m_dev, rows_in):
    """inv int32 [rows_in, taps]: inv[j][t] = m with nbr[m][t] == j, else -1 (transposed rulebook of a strided sparse conv)."""
    M, taps = nbr.shape
    inv = torch.full((rows_in, taps), -1, dtype=torch.int32, device=nbr.device)
    check(lib().tt_sp_inverse_rulebook(ptr(nbr), ptr(m_dev), _ll(M), _c(taps), ptr(inv), _st(nbr)), "tt_sp_inverse_rulebook")
    return inv


def sp_to_dense(x, coords, rows, max_rows, dims, batch_size):
    """rows [R, C] at coords (b, z, y, x) -> dense channel-last (B, H, W, C*D) (== spatial_features.view(N, C*D, H, W))."""
    D, H, W = dims
    C = x.shape[1]
    dense = torch.zeros(batch_size, H, W, C * D, dtype=x.dtype, device=x.device)
    dc = (ctypes.c_int * 3)(*dims)
    check(lib().tt_sp_to_dense(ptr(x), ptr(coords), ptr(rows), _ll(max_rows), _c(C), dc, ptr(dense), _c(dtype_code(x)),
                               cur_stream(x.device)), "tt_sp_to_dense")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.sp_to_dense(x, coords, rows, max_rows, dims, dense)
    return dense


def sp_from_dense(gdense, coords, rows, max_rows, dims, grows):
    """grows += the dense gradient gathered back to the rows (backward of sp_to_dense)."""
    D, H, W = dims
    C = grows.shape[1]
    check(lib().tt_sp_from_dense(ptr(gdense), ptr(coords), ptr(rows), _ll(max_rows), _c(C), _c(D), _c(H), _c(W),
                                 ptr(grows), cur_stream(grows.device)), "tt_sp_from_dense")


def conv2d(x, w, *, stride=1, pad=0, dil=1, scale=None, shift=None, act=0, res1=None, res1_coff=0,
           res2=None, res2_coff=0, out=None, out_coff=0, in_coff=0, cin=None, pixel_shuffle2=False,
           shift_n=None, shift_n_mod=1, out_dtype=None, out_nstride=0, out_hw=None, splitk_ws=None,
           in_cstride=None, w_x3=None, _no_tape=False, stop_grad=False, bn_raw=False, in_pair=False, out_pair=False,
           w_h2=None, out2=None, out2_coff=0, res1_up=False):
    """Channel-last implicit-GEMM convolution on MFMA (tt_conv2d_fwd).

    x   [N,H,W,Cs]  (f32 or bf16); channels [in_coff, in_coff+cin) are convolved
    w   [Cout,KH,KW,cin] same dtype (for pixel_shuffle2: [4*Cout_real,1,1,cin])
    out [N,OH,OW,Ct] written at channel offset out_coff (allocated if None)
    res1_up: res1 is a coarser [N, h, w, C] map added through nearest upsampling to the output size (PAFPN top-down path)
    w_h2: (x, w half) the f16 (hi, lo) weight pair of weights.split_pairs_h2 -- the layer runs the two-MFMA "h2" product
    out2 / out2_coff: optional second, f32, copy of the output rows at channel offset out2_coff of a [.., Ct2] tensor
    in_pair / out_pair (bf16x3 layers only): x is / out becomes a PAIR-format tensor (an f32-typed container holding, per 16
    channels, the bf16 hi and lo halves the kernel's operand split would produce: tt_conv_desc.in_pair).  Only for tensors whose
    every reader is a bf16x3 convolution (`pair_ok(rows)` says whether a layer of that many output rows takes one).
    in_cstride / out_hw: "row-run" form -- the kernel reads `cin` CONTIGUOUS elements starting at pixel (ih, iw) of a
    tensor whose pixels are only Cs < cin elements apart (a run of cin/Cs pixels along W), with the output size given
    explicitly; used for the 7x7/2 stem (lss.py).
    """
    require_cuda(x, w)
    assert w.is_contiguous() and x.dim() == 4 and w.dim() == 4
    N, H, W, Cs = x.shape
    # inner (H, W, C) block must be dense; the image (batch) stride may be anything
    assert x.stride(3) == 1 and (W == 1 or x.stride(2) == Cs) and (H == 1 or x.stride(1) == W * Cs), x.stride()
    Cout, KH, KW, Cin = w.shape
    if cin is None:
        cin = Cin
    assert cin == Cin, (cin, Cin)
    OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    if out_hw is not None:
        OH, OW = out_hw
    cr = Cout // 4 if pixel_shuffle2 else Cout
    if out is None:
        odt = out_dtype or x.dtype
        oh, ow = (2 * OH, 2 * OW) if pixel_shuffle2 else (OH, OW)
        out = torch.empty(N, oh, ow, cr, dtype=odt, device=x.device)
    d = _ConvDesc()
    d.in_ = x.data_ptr(); d.N = N; d.H = H; d.W = W; d.Cin = Cin; d.in_cstride = in_cstride or Cs; d.in_coff = in_coff
    d.in_nstride = x.stride(0) if N > 1 else 0
    d.weight = w.data_ptr(); d.Cout = Cout; d.KH = KH; d.KW = KW; d.stride = stride; d.pad = pad; d.dil = dil
    d.out = out.data_ptr(); d.OH = OH; d.OW = OW
    d.out_cstride = out.shape[-1]; d.out_coff = out_coff
    d.out_nstride = out_nstride or (out.stride(0) if (out.dim() == 4 and N > 1) else 0)
    d.pixel_shuffle2 = 1 if pixel_shuffle2 else 0
    d.scale = _dp(scale); d.shift = _dp(shift); d.shift_n = _dp(shift_n); d.shift_n_mod = shift_n_mod
    d.res1 = _dp(res1); d.res1_cstride = 0 if res1 is None else res1.shape[-1]; d.res1_coff = res1_coff
    if res1 is not None and res1.dtype == torch.float32 and x.dtype != torch.float32:
        d.res1_f32 = 1                  # an f32 sum chain beside half conv inputs (mixed mode's PAFPN)
    if res1_up:
        assert res1 is not None and res1.dim() == 4 and res1.is_contiguous() and res1.shape[0] == N and splitk_ws is None
        d.res1_up_h, d.res1_up_w = res1.shape[1], res1.shape[2]
    d.res2 = _dp(res2); d.res2_cstride = 0 if res2 is None else res2.shape[-1]; d.res2_coff = res2_coff
    d.act = act; d.dtype = dtype_code(x); d.out_dtype = dtype_code(out)
    from . import autodiff
    if w_h2 is not None:
        assert x.dtype == torch.float16 and w.dtype == torch.float16 and w_h2.dtype == torch.float16 and w_h2.is_contiguous() \
            and w_h2.numel() == 2 * w.numel() and w_x3 is None and splitk_ws is None
        if autodiff.TAPE is not None:
            raise _lib.TTError("conv2d: half-storage (h2) layers have no backward; train in dtype torch.float32 or 'f32x3'")
        d.weight_h2 = w_h2.data_ptr()
    if out2 is not None:
        assert out2.dtype == torch.float32 and out2.numel() // out2.shape[-1] == N * OH * OW and out2.stride(-1) == 1
        d.out2 = out2.data_ptr(); d.out2_cstride = out2.shape[-1]; d.out2_coff = out2_coff
    if in_pair or out_pair:
        assert w_x3 is not None and x.dtype == torch.float32 and out.dtype == torch.float32 and splitk_ws is None
        if autodiff.TAPE is not None:
            raise _lib.TTError("conv2d: pair-format activations are an inference-path layout (the tape reads f32 tensors)")
        d.in_pair, d.out_pair = int(bool(in_pair)), int(bool(out_pair))
    if w_x3 is not None and Cout < 64 and (autodiff.TAPE is not None or _no_tape):
        # training step (taped forward, recomputations, input-gradient convolutions): the layers with fewer than 64 output channels
        # keep the exact-f32 kernels the gradient goldens were taken with (inference runs them on the 256 x 32 bf16x3 tile)
        w_x3 = None
    if w_x3 is not None:       # (before the split-K query: with a bf16x3 operand the query answers for the bf16x3 split-K tile)
        assert x.dtype == torch.float32 and w_x3.shape == w.shape and w_x3.is_contiguous()
        # the training step (forward under the tape, and the backward's recomputations / input-gradient convolutions, which
        # pass _no_tape) keeps the exact-f32 split-K kernel for the few-row long-K layers -- the gradient goldens were taken
        # with it: the query must not see the bf16x3 operand there
        x3_splitk = autodiff.TAPE is None and not _no_tape
        if x3_splitk:
            d.weight_x3 = w_x3.data_ptr()
    if _AUTO_SPLITK and splitk_ws is None and not (in_pair or out_pair) and w_h2 is None and not res1_up and N * OH * OW <= 4096 and KH * KW * Cin >= 2048:
        # few rows, very long K (BEV-update conv K=18720, flatten MLPs): cross-workgroup split-K with an f32 workspace
        # beats conv_small.hip's in-workgroup split there (277 vs 416 us on the BEV-update conv: the direct 32 B/row
        # operand loads of the small kernel waste L2 sectors on a 10 MB weight matrix).  TT_CONV_AUTO_SPLITK=0 disables.
        # ordered form (no atomics, no zero fill): one [M][Cout] slice per K split, added in index order by the finalize kernel.
        # With a bf16x3 operand the layer runs the 64-wide bf16x3 tile with the K tiles dealt over workgroups
        # (conv_igemm_glds.hip, x3_splitk_plan) instead of the exact-f32 register-staged kernel
        slices = int(lib().tt_conv2d_splitk_slices(ctypes.byref(d)))
        if slices > 0:
            splitk_ws = torch.empty(slices, N * OH * OW, Cout, dtype=torch.float32, device=x.device)
            d.splitk_slices = slices
    if splitk_ws is not None:
        d.splitk_ws = splitk_ws.data_ptr()
    elif w_x3 is not None:
        d.weight_x3 = w_x3.data_ptr()
    if CONV_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib().tt_conv2d_fwd(ctypes.byref(d), cur_stream(x.device))
    check(rc, "tt_conv2d_fwd")
    if CONV_PROFILE is not None:
        e1.record()
        CONV_PROFILE.append((2.0 * N * OH * OW * Cout * KH * KW * Cin, e0, e1,
                             f"M={N * OH * OW} N={Cout} K={KH * KW * Cin} k{KH}x{KW}s{stride}"))
        if CONV_KERNELS is not None:
            CONV_KERNELS.append(_last_conv_kernel())
        if CONV_BYTES is not None:
            esz, osz = x.element_size(), out.element_size()
            m_out = N * OH * OW
            in_px = min(N * H * W, m_out * KH * KW)           # a strided 1x1 layer only touches the pixels it samples
            wsz = w.numel() * w.element_size() * (2 if w_h2 is not None else 1)
            CONV_BYTES.append(in_px * (in_cstride or Cin) * esz + m_out * Cout * osz + wsz +
                              m_out * Cout * esz * ((res1 is not None) + (res2 is not None)) +
                              (m_out * Cout * 4 if out2 is not None else 0))
    if not _no_tape:
        from . import autodiff
        if autodiff.TAPE is not None:
            autodiff.TAPE.conv(x, w, out, stride, pad, dil, scale, shift, act, in_coff, cin, out_coff, res1, res1_coff,
                               res2, res2_coff, pixel_shuffle2, in_cstride, shift_n, shift_n_mod, stop_grad, bn_raw)
    return out


# ----------------------------------------------------------------------------- glue kernels
_ll = ctypes.c_longlong
_f = ctypes.c_float


def _st(t):
    return cur_stream(t.device)


# ----------------------------------------------------------------------------- train-mode BatchNorm / dropout
BN_SYNC = True          # SyncBN (configs/thinktwice.py:39): all-reduce the batch statistics when torch.distributed runs > 1 rank
DROPOUT_MASKS = None    # tests: iterator of host uint8 keep-masks (the reference's torch masks) consumed by dropout()
DROPOUT_SEED = [0x5EED, 0]    # [seed, calls so far]: the product's counter-based generator


def _all_reduce_sum_(t):
    """In-place SUM over the ranks (no-op on a single rank): the SyncBN statistic exchange."""
    import torch.distributed as dist
    if not (BN_SYNC and dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    if t.is_cuda and dist.get_backend() == "gloo":        # CPU test rigs: stage through the host
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class BNSpec:
    """One BatchNorm layer's tensors for the train-mode path (device f32): affine, running statistics (updated in place),
    eps / momentum, and the state_dict prefix the parameter gradients are filed under."""

    def __init__(self, name, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1):
        self.name, self.gamma, self.beta, self.running_mean, self.running_var = name, gamma, beta, running_mean, running_var
        self.eps, self.momentum = float(eps), float(momentum)


def _bn_ws(C, groups, dev):
    L = lib()
    L.tt_bn_workspace_bytes.restype = ctypes.c_longlong
    nb = int(L.tt_bn_workspace_bytes(_c(C), _c(groups)))
    return torch.empty(nb, dtype=torch.uint8, device=dev), nb


def batchnorm_train(z, spec, act=0, res1=None, res1_coff=0, res2=None, res2_coff=0, out=None, out_coff=0, m_dev=None,
                    groups=1, update_running=True):
    """y = act(BN_batch(z) + res1 + res2): z [..., C] dense f32 rows (the raw conv output, bias included); out: row-linear
    buffer [..., Ct] written at channel offset out_coff (allocated if None).  `groups` equal row groups get their own
    statistics; m_dev: device count of live rows (sparse levels).  SyncBN: one all-reduce of the statistics block."""
    require_cuda(z)
    assert z.dtype == torch.float32 and z.is_contiguous()
    C = z.shape[-1]
    M = z.numel() // C
    dev = z.device
    if out is None:
        out = torch.empty_like(z)
    assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() // out.shape[-1] == M, (out.shape, z.shape)
    for r in (res1, res2):
        assert r is None or (r.is_contiguous() and r.dtype == torch.float32 and r.numel() // r.shape[-1] == M)
    stats = torch.empty(groups, 2 * C + 2, dtype=torch.float64, device=dev)
    ws, nb = _bn_ws(C, groups, dev)
    st = _st(z)
    L = lib()
    check(L.tt_bn_stats(ptr(z), _ll(M), _c(C), _c(C), _c(0), ptr(m_dev), _c(groups), ptr(stats), ptr(ws), _ll(nb), st),
          "tt_bn_stats")
    _all_reduce_sum_(stats)
    scale, shift, mean, invstd = (torch.empty(groups, C, dtype=torch.float32, device=dev) for _ in range(4))
    check(L.tt_bn_finalize(ptr(stats), _c(C), _c(groups), ptr(spec.gamma), ptr(spec.beta), _f(spec.eps), _f(spec.momentum),
                           ptr(spec.running_mean if update_running else None),
                           ptr(spec.running_var if update_running else None), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                           st), "tt_bn_finalize")

    def cs(t):
        return 0 if t is None else t.shape[-1]
    check(L.tt_bn_apply(ptr(z), _ll(M), _c(C), _c(C), _c(0), ptr(m_dev), _c(groups), ptr(scale), ptr(shift), ptr(mean), ptr(res1),
                        _c(cs(res1)), _c(res1_coff), ptr(res2), _c(cs(res2)), _c(res2_coff), _c(act), ptr(out),
                        _c(out.shape[-1]), _c(out_coff), st), "tt_bn_apply")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.bn_train(z, out, out_coff, spec, act, res1, res1_coff, res2, res2_coff, m_dev, groups, stats, scale,
                               mean, invstd)
    return out


def batchnorm_train_bwd(dy, dy_coff, y, y_coff, z, mean, invstd, scale, stats, act, dres1, dres1_coff, dres2, dres2_coff, dz,
                        m_dev=None, groups=1):
    """Backward of batchnorm_train: dy (the gradient buffer of `out`, its channel window is overwritten with g), -> dz
    written, dres* accumulated; returns (dgamma [C], dbeta [C]) from the LOCAL sums."""
    C = z.shape[-1]
    M = z.numel() // C
    dev = z.device
    sums = torch.empty(groups, 2 * C, dtype=torch.float64, device=dev)
    ws, nb = _bn_ws(C, groups, dev)
    st = _st(z)
    L = lib()

    def cs(t):
        return 0 if t is None else t.shape[-1]
    check(L.tt_bn_bwd_reduce(ptr(dy), _c(dy.shape[-1]), _c(dy_coff), ptr(y), _c(y.shape[-1]), _c(y_coff), ptr(z), _c(C), _c(0),
                             _ll(M), _c(C), ptr(m_dev), _c(groups), ptr(mean), ptr(invstd), _c(act), ptr(dres1), _c(cs(dres1)),
                             _c(dres1_coff), ptr(dres2), _c(cs(dres2)), _c(dres2_coff), ptr(sums), ptr(ws), _ll(nb), st),
          "tt_bn_bwd_reduce")
    local = sums.sum(0).to(torch.float32)
    _all_reduce_sum_(sums)
    check(L.tt_bn_bwd_apply(ptr(dy), _c(dy.shape[-1]), _c(dy_coff), ptr(z), _c(C), _c(0), _ll(M), _c(C), ptr(m_dev), _c(groups),
                            ptr(sums), ptr(stats), ptr(scale), ptr(mean), ptr(invstd), ptr(dz), _c(dz.shape[-1]), _c(0), st),
          "tt_bn_bwd_apply")
    return local[C:].contiguous(), local[:C].contiguous()


def dropout(x, p=0.5):
    """nn.Dropout(p) in train mode (lss.py:91,110).  Keep-mask from the counter-based device generator, or -- tests -- the next
    host mask of DROPOUT_MASKS (the reference's own torch mask, in x's channel-last layout)."""
    require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty_like(x)
    mask = torch.empty(x.numel(), dtype=torch.uint8, device=x.device)
    mask_in = None
    if DROPOUT_MASKS is not None:
        mask_in = next(DROPOUT_MASKS).to(device=x.device, dtype=torch.uint8).contiguous()
        assert mask_in.numel() == x.numel()
    DROPOUT_SEED[1] += 1
    seed = (DROPOUT_SEED[0] * 0x100000001B3 + DROPOUT_SEED[1] * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    check(lib().tt_dropout_fwd(ptr(x), ptr(out), _ll(x.numel()), _f(p), ctypes.c_ulonglong(seed), ptr(mask_in), ptr(mask),
                               _st(x)), "tt_dropout_fwd")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.dropout(x, out, mask, p)
    return out


def dropout_bwd(dout, mask, dx, p):
    assert dout.is_contiguous() and dx.is_contiguous()
    check(lib().tt_dropout_bwd(ptr(dout), ptr(mask), ptr(dx), _ll(dout.numel()), _f(p), _st(dout)), "tt_dropout_bwd")


def nchw_to_nhwc_border(x, out, top, left):
    """(N,C,H,W) f32 -> interior of out (N,Hp,Wp,Cp) at pixel offset (top, left); border untouched."""
    require_cuda(x, out)
    assert x.is_contiguous() and x.dtype == torch.float32 and out.is_contiguous()
    N, C, H, W = x.shape
    _, Hp, Wp, Cp = out.shape
    check(lib().tt_nchw_to_nhwc_border(ptr(x), ptr(out), _c(N), _c(C), _c(H), _c(W), _c(Cp), _c(Hp), _c(Wp),
                                       _c(top), _c(left), _c(dtype_code(out)), _st(x)), "tt_nchw_to_nhwc_border")
    return out


def nchw_to_nhwc_pad(x, dtype, c_pad):
    """(N,C,H,W) f32 -> (N,H,W,c_pad) `dtype`, zero-padded channels."""
    require_cuda(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    N, C, H, W = x.shape
    out = torch.empty(N, H, W, c_pad, dtype=dtype, device=x.device)
    check(lib().tt_nchw_to_nhwc_pad(ptr(x), ptr(out), _c(N), _c(C), _c(H), _c(W), _c(c_pad),
                                    _c(dtype_code(out)), _st(x)), "tt_nchw_to_nhwc_pad")
    return out


def nhwc_to_nchw(x, C=None, coff=0):
    """(N,H,W,Cs) -> (N,C,H,W) f32."""
    require_cuda(x)
    N, H, W, Cs = x.shape
    C = C or Cs
    out = torch.empty(N, C, H, W, dtype=torch.float32, device=x.device)
    check(lib().tt_nhwc_to_nchw(ptr(x), ptr(out), _c(N), _c(C), _c(H), _c(W), _c(Cs), _c(coff),
                                _c(dtype_code(x)), _st(x)), "tt_nhwc_to_nchw")
    return out


def maxpool3x3s2(x):
    N, H, W, C = x.shape
    out = torch.empty(N, (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1, C, dtype=x.dtype, device=x.device)
    check(lib().tt_maxpool3x3s2(ptr(x), ptr(out), _c(N), _c(H), _c(W), _c(C), _c(dtype_code(x)), _st(x)),
          "tt_maxpool3x3s2")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.maxpool3x3s2(x, out)
    return out


def upsample_nearest_add_(dst, src):
    N, H, W, C = dst.shape
    check(lib().tt_upsample_nearest_add(ptr(dst), ptr(src), _c(N), _c(H), _c(W), _c(C), _c(src.shape[1]),
                                        _c(src.shape[2]), _c(dtype_code(dst)), _st(dst)),
          "tt_upsample_nearest_add")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.upsample_nearest_add_(dst, src)
    return dst


PAIR = os.environ.get("TT_X3_PAIR", "1") != "0"     # A/B knob: 0 = every activation tensor stays plain f32


def pair_ok(rows, cin=32, cout=64):
    """Whether a bf16x3 convolution with this many output rows / these channel counts accepts a pair-format input
    (tt_conv_desc.in_pair: the LDS-DMA kernel's contract; up to 4096 rows a layer runs the exact-f32 latency kernel, which a
    pair-format layer could not take).  Producer and consumer of a tensor ask with the CONSUMER's numbers."""
    from . import autodiff
    return PAIR and autodiff.TAPE is None and rows > 4096 and cin % 32 == 0 and (8 <= cout <= 32 or cout >= 64)


def bilinear_up2(x, out_pair=False):
    N, H, W, C = x.shape
    out = torch.empty(N, 2 * H, 2 * W, C, dtype=x.dtype, device=x.device)
    if out_pair:      # f32 -> bf16x3 pair format for a consumer that is a bf16x3 convolution (conv2d(in_pair=True))
        assert x.dtype == torch.float32 and x.is_contiguous() and C % 16 == 0
        check(lib().tt_bilinear_up2_pair(ptr(x), ptr(out), _c(N), _c(H), _c(W), _c(C), _st(x)), "tt_bilinear_up2_pair")
        return out
    check(lib().tt_bilinear_up2(ptr(x), ptr(out), _c(N), _c(H), _c(W), _c(C), _c(dtype_code(x)), _st(x)),
          "tt_bilinear_up2")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.bilinear_up2(x, out)
    return out


def spatial_pool(x, mode, C=None, coff=0):
    """(N,H,W,Cs) -> f32 (N,C): mode 0 mean, mode 1 (mean+max)/2."""
    N, H, W, Cs = x.shape
    C = C or Cs
    out = torch.empty(N, C, dtype=torch.float32, device=x.device)
    check(lib().tt_spatial_pool(ptr(x), ptr(out), _c(N), _c(H * W), _c(C), _c(Cs), _c(coff), _c(mode),
                                _c(dtype_code(x)), _st(x)), "tt_spatial_pool")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.spatial_pool(x, out, mode, C, coff)
    return out


def channel_gate(x, gate, res=None, gate_act=_lib.ACT_SIGMOID, out_act=_lib.ACT_NONE, out=None):
    N, H, W, C = x.shape
    assert gate.dtype == torch.float32 and gate.shape == (N, C) and gate.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    check(lib().tt_channel_gate(ptr(x), ptr(gate), ptr(res), ptr(out), _c(N), _c(H * W), _c(C), _c(gate_act),
                                _c(out_act), _c(dtype_code(x)), _st(x)), "tt_channel_gate")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.channel_gate(x, gate, res, out, gate_act, out_act)
    return out


def affine_rows(x, scale, shift, act=0, out=None):
    """x (R, C) possibly a row-strided 2-D view."""
    R, C = x.shape
    out = torch.empty(R, C, dtype=x.dtype, device=x.device) if out is None else out
    check(lib().tt_affine_rows(ptr(x), ptr(scale), ptr(shift), ptr(out), _ll(R), _c(C), _c(x.stride(0)),
                               _c(out.stride(0)), _c(act), _c(dtype_code(x)), _st(x)), "tt_affine_rows")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.affine_rows(x, scale, shift, act, out)
    return out


def layernorm_rows(x, gamma, beta, eps=1e-5, out=None, D=None):
    """LayerNorm over the first D columns of each row of the 2-D (row-strided) x."""
    R = x.shape[0]
    D = D or x.shape[1]
    out = torch.zeros(R, x.shape[1], dtype=x.dtype, device=x.device) if out is None else out
    check(lib().tt_layernorm_rows(ptr(x), ptr(gamma), ptr(beta), ptr(out), _ll(R), _c(D), _c(x.stride(0)),
                                  _c(out.stride(0)), _f(eps), _c(dtype_code(x)), _st(x)), "tt_layernorm_rows")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.layernorm_rows(x, gamma, beta, out, D, eps)
    return out


def layernorm_rows_bwd(x, gamma, dout, dx, dgamma, dbeta, D, eps=1e-5):
    """dx[:, :D] += , dgamma += , dbeta += backward of layernorm_rows (2-D row-strided f32 views)."""
    R = x.shape[0]
    L = lib()
    L.tt_layernorm_rows_bwd_workspace_bytes.restype = ctypes.c_longlong
    nb = int(L.tt_layernorm_rows_bwd_workspace_bytes(_ll(R), _c(D)))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    check(L.tt_layernorm_rows_bwd(ptr(x), ptr(gamma), ptr(dout), ptr(dx), ptr(dgamma), ptr(dbeta), _ll(R), _c(D),
                                  _c(x.stride(0)), _c(dout.stride(0)), _c(dx.stride(0)), _f(eps), ptr(ws), _ll(nb), _st(x)),
          "tt_layernorm_rows_bwd")


def concat_piece_bwd(dout, coff, C, div, mod, dsrc):
    """dsrc[:, :C] += the gradient of one concat_rows piece (dout / dsrc 2-D row-strided f32 views)."""
    R = dout.shape[0]
    check(lib().tt_concat_piece_bwd(ptr(dout), _c(dout.stride(0)), _c(coff), _ll(R), _c(C), _c(div), _c(mod), ptr(dsrc),
                                    _c(dsrc.stride(0)), _c(dsrc.shape[0]), _st(dout)), "tt_concat_piece_bwd")


def copy_nhwc(x, out, C=None, in_coff=0, out_coff=0, rot_flip=False):
    N, H, W, Cs = x.shape
    C = C or Cs
    check(lib().tt_copy_nhwc(ptr(x), ptr(out), _c(N), _c(H), _c(W), _c(C), _c(Cs), _c(in_coff),
                             _c(out.shape[-1]), _c(out_coff), _c(1 if rot_flip else 0), _c(dtype_code(x)),
                             _c(dtype_code(out)), _st(x)), "tt_copy_nhwc")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.copy_nhwc(x, out, C, in_coff, out_coff, rot_flip)
    return out


def broadcast_rows(v, out, out_coff=0):
    """v (N, C) -> out (N,H,W,Ct)[..., out_coff:out_coff+C] = v[n, :]."""
    N, C = v.shape
    _, H, W, Ct = out.shape
    check(lib().tt_broadcast_rows(ptr(v), ptr(out), _c(N), _c(H * W), _c(C), _c(v.stride(0)), _c(Ct),
                                  _c(out_coff), _c(dtype_code(out)), _st(out)), "tt_broadcast_rows")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.broadcast_rows(v, out, out_coff)
    return out


def ew(op, a, b=None, g=None, out=None, C=None, a_coff=0, b_coff=0, g_coff=0, out_coff=0, act=0):
    """Row-wise elementwise op on 2-D (row-strided) views; see tt_ew."""
    a2 = a.reshape(-1, a.shape[-1])
    R = a2.shape[0]
    C = C or a2.shape[1]
    if out is None:
        out = torch.empty(R, C, dtype=a.dtype, device=a.device)
    o2 = out.reshape(-1, out.shape[-1])
    assert o2.data_ptr() == out.data_ptr() and o2.stride(1) == 1, "ew: `out` must be viewable as rows"
    b2 = None if b is None else b.reshape(-1, b.shape[-1])
    g2 = None if g is None else g.reshape(-1, g.shape[-1])
    check(lib().tt_ew(ptr(a2), ptr(b2), ptr(g2), ptr(o2), _ll(R), _c(C), _c(a2.stride(0)), _c(a_coff),
                      _c(0 if b2 is None else b2.stride(0)), _c(b_coff), _c(0 if g2 is None else g2.stride(0)),
                      _c(g_coff), _c(o2.stride(0)), _c(out_coff), _c(op), _c(act), _c(dtype_code(a)), _st(a)),
          "tt_ew")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.ew(op, act, R, C, a2, a_coff, b2, b_coff, g2, g_coff, o2, out_coff)
    return out


def ew_bwd(op, act, R, C, a2, a_coff, b2, b_coff, g2, g_coff, o2, out_coff, dout, da, db, dg):
    """Backward of `ew` on the same 2-D row views (d* are the gradient views of a2 / b2 / g2 / o2, or None)."""
    def st(t):
        return 0 if t is None else t.stride(0)
    check(lib().tt_ew_bwd(_c(op), _c(act), _ll(R), _c(C), ptr(a2), _c(st(a2)), _c(a_coff), ptr(b2), _c(st(b2)), _c(b_coff),
                          ptr(g2), _c(st(g2)), _c(g_coff), ptr(o2), _c(st(o2)), _c(out_coff), ptr(dout), _c(st(dout)),
                          _c(out_coff), ptr(da), _c(st(da)), _c(a_coff), ptr(db), _c(st(db)), _c(b_coff), ptr(dg),
                          _c(st(dg)), _c(g_coff), _st(a2)), "tt_ew_bwd")


def concat_rows(out, pieces, coff=0):
    """out (R, Ct) f32; pieces = [(src or None, C, div, mod), ...] laid out left to right from column `coff`:
    out[r, ...] = src[(r // div) % mod if mod else r // div, :C]  (None: zeros).  One launch (tt_concat_rows)."""
    o2 = out.reshape(-1, out.shape[-1])
    assert o2.data_ptr() == out.data_ptr() and o2.dtype == torch.float32 and 1 <= len(pieces) <= 8
    assert o2.stride(1) == 1, "concat_rows: `out` rows must be element-contiguous"
    n = len(pieces)
    srcs = (ctypes.c_void_p * n)()
    strides, widths, coffs, divs, mods = ((ctypes.c_int * n)() for _ in range(5))
    c = coff
    for i, (src, C, div, mod) in enumerate(pieces):
        if src is not None:
            s2 = src.reshape(-1, src.shape[-1])
            assert s2.dtype == torch.float32 and s2.stride(1) == 1 and s2.shape[1] >= C
            need = mod if mod else -(-o2.shape[0] // div)          # source rows the mapping reaches
            assert s2.shape[0] >= need, f"concat_rows: piece {i} has {s2.shape[0]} rows, mapping reads {need}"
            srcs[i], strides[i] = s2.data_ptr(), s2.stride(0)
        else:
            srcs[i], strides[i] = None, 0
        widths[i], coffs[i], divs[i], mods[i] = C, c, div, mod
        c += C
    check(lib().tt_concat_rows(ptr(o2), _ll(o2.shape[0]), _c(o2.stride(0)), _c(n), srcs, strides, widths, coffs, divs,
                               mods, _st(out)), "tt_concat_rows")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.concat_rows(o2, pieces, coff)
    return out


def deform_im2col3x3(x, offsets, pad=1):
    """x (N,H,W,C), offsets f32 (N,H,W,>=18) -> cols (N*H*W, 1, 9, C) (a [M,1,9,C] "image" for conv2d)."""
    N, H, W, C = x.shape
    assert offsets.dtype == torch.float32 and offsets.is_contiguous()
    cols = torch.empty(N * H * W, 1, 9, C, dtype=x.dtype, device=x.device)
    check(lib().tt_deform_im2col3x3(ptr(x), ptr(offsets), ptr(cols), _c(N), _c(H), _c(W), _c(C),
                                    _c(offsets.shape[-1]), _c(pad), _c(dtype_code(x)), _st(x)),
          "tt_deform_im2col3x3")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.deform_im2col3x3(x, offsets, cols, pad)
    return cols


def repeat_rows(t, times):
    """t (R, C) -> (times * R, C), the rows repeated block-wise (torch's t.repeat(times, 1)); a copy the training tape
    knows how to differentiate (the gradient is the sum over the repeats)."""
    assert t.dim() == 2 and t.is_contiguous()
    out = torch.empty(times * t.shape[0], t.shape[1], dtype=t.dtype, device=t.device)
    nb = t.numel() * t.element_size()
    for k in range(times):          # `times` contiguous block copies (tt_copy_bytes): no torch kernel in the forward
        check(lib().tt_copy_bytes(ctypes.c_void_p(out.data_ptr() + k * nb), ptr(t), _ll(nb), _st(t)), "tt_copy_bytes")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.repeat_rows(t, out, times)
    return out


# ----------------------------------------------------------------------------- look module
def look_project_pack(wp, lidar2img, ida_mat, img_hw):
    B = wp.shape[0]
    dev = wp.device
    ref = torch.empty(B, 4, 120, 2, dtype=torch.float32, device=dev)
    qos = torch.empty(B, 4, 120, dtype=torch.int32, device=dev)
    count = torch.empty(B, 4, dtype=torch.int32, device=dev)
    max_len = torch.empty(1, dtype=torch.int32, device=dev)
    check(lib().tt_look_project_pack(_c(B), ptr(wp), ptr(lidar2img), ptr(ida_mat), _f(img_hw[0]), _f(img_hw[1]),
                                     ptr(ref), ptr(qos), ptr(count), ptr(max_len), _st(wp)), "tt_look_project_pack")
    return ref, qos, count, max_len


def _level_args(maps):
    arr = (ctypes.c_void_p * 4)(*[m.data_ptr() for m in maps])
    hw = (ctypes.c_int * 8)(*[v for m in maps for v in (m.shape[1], m.shape[2])])
    return arr, hw


def look_gather_query(qos, ref, wp, ctrl_sp, temporal, static, meas, flat, maps, row_stride=1544):
    B = wp.shape[0]
    out = torch.empty(B * 4 * 120, row_stride, dtype=torch.float32, device=wp.device)
    arr, hw = _level_args(maps)
    check(lib().tt_look_gather_query(_c(B), ptr(qos), ptr(ref), ptr(wp), ptr(ctrl_sp), ptr(temporal), ptr(static),
                                     ptr(meas), ptr(flat), arr, hw, _c(dtype_code(maps[0])), ptr(out),
                                     _c(row_stride), _st(wp)), "tt_look_gather_query")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.look_gather_query(qos, ref, out, temporal, static, meas, flat, maps, row_stride)
    return out


def look_gather_query_bwd(B, qos, ref, dout, row_stride, dtemporal, dstatic, dmeas, dflat, dmaps):
    arr, hw = _level_args(dmaps)
    check(lib().tt_look_gather_query_bwd(_c(B), ptr(qos), ptr(ref), ptr(dout), _c(row_stride), ptr(dtemporal), ptr(dstatic),
                                         ptr(dmeas), ptr(dflat), arr, hw, _st(dout)), "tt_look_gather_query_bwd")


def msda_sample_bwd(B, value, coff, offsets, logits, ref, level_hw, dout, dvalue, doffsets, dlogits):
    hw = (ctypes.c_int * 8)(*[v for pair in level_hw for v in pair])
    assert value.dtype == torch.float32
    check(lib().tt_msda_sample_bwd(_c(B), ptr(value), _c(value.shape[-1]), _c(coff), ptr(offsets), ptr(logits), ptr(ref), hw,
                                   ptr(dout), ptr(dvalue), ptr(doffsets), ptr(dlogits), _st(dout)), "tt_msda_sample_bwd")


def sca_reduce_bwd(B, dout, max_len, dx):
    check(lib().tt_sca_reduce_bwd(_c(B), ptr(dout), ptr(max_len), ptr(dx), _st(dout)), "tt_sca_reduce_bwd")


def msda_sample(value, offsets, logits, ref, level_hw, B, coff=0):
    """value (B*4, S, Cv) with Cv >= 256: samples channels [coff, coff+256)."""
    out = torch.empty(B * 4 * 120, 256, dtype=torch.float32, device=value.device)
    hw = (ctypes.c_int * 8)(*[v for pair in level_hw for v in pair])
    check(lib().tt_msda_sample_strided(_c(B), ptr(value), _c(dtype_code(value)), _c(value.shape[-1]), _c(coff),
                                       ptr(offsets), ptr(logits), ptr(ref), hw, ptr(out), _st(value)),
          "tt_msda_sample")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.msda_sample(value, offsets, logits, ref, level_hw, B, coff, out)
    return out


def sca_reduce(x, max_len, B):
    out = torch.empty(B, 1024, dtype=torch.float32, device=x.device)
    check(lib().tt_sca_reduce(_c(B), ptr(x), ptr(max_len), ptr(out), _st(x)), "tt_sca_reduce")
    from . import autodiff
    if autodiff.TAPE is not None:
        autodiff.TAPE.sca_reduce(x, max_len, B, out)
    return out


# ----------------------------------------------------------------------------- decoder row chains (tt_mlp_chain)
class _ChainStage(ctypes.Structure):
    _fields_ = [
        ("w", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("K", _c), ("Kp", _c), ("N", _c), ("act", _c), ("in_sel", _c),
        ("res", ctypes.c_void_p), ("res_stride", _c), ("res_coff", _c),
        ("side", ctypes.c_void_p), ("side_w", ctypes.c_void_p), ("side_stride", _c), ("side_k", _c),
        ("out", ctypes.c_void_p), ("out_stride", _c), ("out_coff", _c),
    ]


class ChainLinear:
    """One nn.Linear prepared for tt_mlp_chain: weight [N, K] (+ optional leading `side_k` columns that multiply a
    separate few-column input, e.g. cat([wp, h]) @ W^T = W[:, :2] wp + W[:, 2:] h) -> pair-format bf16x3 operand."""

    def __init__(self, w, bias, act=0, side_k=0, device="cuda"):
        from . import weights
        w = w.detach().to(device=device, dtype=torch.float32)
        self.side_k = side_k
        self.side_w = w[:, :side_k].contiguous() if side_k else None
        w = w[:, side_k:]
        self.N, self.K = w.shape
        assert self.K % 4 == 0, "ChainLinear: K must be a multiple of 4"
        self.Kp = (self.K + 15) // 16 * 16
        wp = torch.zeros((self.N + 31) // 32 * 32, self.Kp, dtype=torch.float32, device=device)
        wp[: self.N, : self.K] = w
        self.w = weights.split_pairs_frag(wp)
        self.bias = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
        self.act = act


CHAIN_WIDE = os.environ.get("TT_CHAIN_WIDE", "1") != "0"   # A/B knob: 0 = always the one-workgroup-per-32-rows kernel
CHAIN_WIDE_MAX_ROW_BLOCKS = 16                               # beyond this the rows alone fill enough workgroups


def chain_is_wide(R):
    """Whether mlp_chain runs R rows on tt_mlp_chain_wide (callers order their stages for the form that runs)."""
    return CHAIN_WIDE and (R + 31) // 32 <= CHAIN_WIDE_MAX_ROW_BLOCKS


def chain_faults():
    """Blocking (device synchronise): non-zero if a tt_mlp_chain_wide launch on the current device gave up waiting at its
    barrier since the last `clear_device_faults()` (tests assert 0)."""
    return int(lib().tt_mlp_chain_wide_faults())


_wide_cap = {}


def chain_wide_max_workgroups(device):
    """Workgroups one tt_mlp_chain_wide launch may use on `device` (co-resident capacity / 2 concurrent launches)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _wide_cap:
        with torch.cuda.device(idx):
            cap = int(lib().tt_mlp_chain_wide_max_workgroups())
        if cap < 0:
            raise TTError(f"tt_mlp_chain_wide_max_workgroups failed: {lib().tt_last_error().decode()}")
        _wide_cap[idx] = cap
    return _wide_cap[idx]


def mlp_chain(x, stages, n_split=1, groups=None, wide=None):
    """x (R, >= K0) f32 rows (row-strided ok).  stages: list of dicts {lin: ChainLinear, src: -1 | earlier stage index,
    res: (tensor (R, *), coff) | None, side: tensor (R, >= side_k) | None, out: (tensor (R, *), coff) | None}.
    Few rows (<= 512) run tt_mlp_chain_wide: the columns of every stage over `groups` workgroups per 32 rows (default: the
    widest stage's 32-column blocks, at most 16; capped so that all workgroups are co-resident); otherwise tt_mlp_chain
    (`n_split`: the single-stage column split of that kernel).  `wide`: force one form (tests)."""
    require_cuda(x)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    R = x.shape[0]
    n = len(stages)
    arr = (_ChainStage * n)()
    for i, st in enumerate(stages):
        lin = st["lin"]
        d = arr[i]
        d.w = lin.w.data_ptr(); d.bias = _dp(lin.bias)
        d.K, d.Kp, d.N, d.act, d.in_sel = lin.K, lin.Kp, lin.N, lin.act, st.get("src", i - 1)
        res = st.get("res")
        if res is not None:
            t, coff = res
            assert t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[0] >= R and t.shape[1] >= coff + lin.N
            d.res, d.res_stride, d.res_coff = t.data_ptr(), t.stride(0), coff
        side = st.get("side")
        if lin.side_k:
            assert side is not None and side.dtype == torch.float32 and side.dim() == 2 and side.stride(1) == 1
            assert side.shape[0] >= R and side.shape[1] >= lin.side_k
            d.side, d.side_w, d.side_stride, d.side_k = side.data_ptr(), lin.side_w.data_ptr(), side.stride(0), lin.side_k
        out = st.get("out")
        if out is not None:
            t, coff = out
            assert t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[0] >= R and t.shape[1] >= coff + lin.N
            d.out, d.out_stride, d.out_coff = t.data_ptr(), t.stride(0), coff
    row_blocks = (R + 31) // 32
    use_wide = chain_is_wide(R) if wide is None else wide
    if use_wide:
        row_groups = 1 if row_blocks <= 2 else (row_blocks + 3) // 4       # workgroups take 1, 2 or 4 row blocks
        cap = min(128, chain_wide_max_workgroups(x.device))     # co-resident with room to spare (the library enforces cap)
        if cap < row_groups:
            if wide:
                raise TTError(f"tt_mlp_chain_wide: {row_groups} row groups cannot be co-resident on this device ({cap})")
            use_wide = False                                       # a small / partitioned device: the per-row-block kernel
    if use_wide:
        L = lib()
        if groups is None:
            groups = min(16, max((st["lin"].N + 31) // 32 for st in stages))
        groups = max(1, min(groups, cap // row_groups, 64))
        L.tt_mlp_chain_wide_workspace_bytes.restype = ctypes.c_longlong
        nbytes = int(L.tt_mlp_chain_wide_workspace_bytes(_ll(R), _c(n), arr))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        rc = L.tt_mlp_chain_wide(ptr(x), _ll(R), _c(x.stride(0)), _c(n), arr, _c(groups), ptr(ws), _ll(nbytes), _st(x))
        if rc != -4 or wide:
            check(rc, "tt_mlp_chain_wide")
            return
        # -4: the ticket slots of graph-captured launches are used up (a process that re-captures over and over): the
        # one-workgroup-per-row-block chain computes the same stages without a cross-workgroup barrier
    check(lib().tt_mlp_chain(ptr(x), _ll(R), _c(x.stride(0)), _c(n), arr, _c(n_split), _st(x)), "tt_mlp_chain")


# ----------------------------------------------------------------------------- convolution backward (training step)
def conv2d_wgrad(x, dy, kh, kw, stride=1, pad=0, dil=1, cin=None, in_coff=0, cout=None, dy_coff=0, cin_pad=None,
                 out=None, accumulate=False, x3=False):
    """Weight gradient of `conv2d` (tt_conv2d_wgrad): x [N,H,W,Cs] f32, dy [N,OH,OW,Cd] f32 (the gradient w.r.t. the
    convolution's raw output) -> dw [cout][kh][kw][cin_pad] f32 in the weight layout (added to `out` if accumulate)."""
    require_cuda(x, dy)
    assert x.dtype == torch.float32 and dy.dtype == torch.float32 and x.is_contiguous() and dy.is_contiguous()
    N, H, W, Cs = x.shape
    N2, OH, OW, Cd = dy.shape
    assert N2 == N
    cin = cin or (Cs - in_coff)
    cout = cout or (Cd - dy_coff)
    cin_pad = cin_pad or cin
    if out is None:
        assert not accumulate
        out = torch.empty(cout, kh, kw, cin_pad, dtype=torch.float32, device=x.device)
    assert tuple(out.shape) == (cout, kh, kw, cin_pad) and out.is_contiguous()
    L = lib()
    L.tt_conv2d_wgrad_workspace_bytes.restype = ctypes.c_longlong
    nb = int(L.tt_conv2d_wgrad_workspace_bytes(_c(N), _c(OH), _c(cout), _c(cin), _c(cin_pad), _c(kh), _c(kw)))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    if CONV_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fn = L.tt_conv2d_wgrad_x3 if x3 else L.tt_conv2d_wgrad      # x3: the forward's bf16x3 arithmetic (wide layers: LDS-staged kernel)
    check(fn(ptr(x), _c(N), _c(H), _c(W), _c(cin), _c(Cs), _c(in_coff), ptr(dy), _c(OH), _c(OW), _c(cout),
                            _c(Cd), _c(dy_coff), _c(kh), _c(kw), _c(stride), _c(pad), _c(dil), _c(cin_pad),
                            _c(1 if accumulate else 0), ptr(out), ptr(ws), _ll(nb), _st(x)), "tt_conv2d_wgrad")
    if CONV_PROFILE is not None:
        e1.record()
        CONV_PROFILE.append((2.0 * N * OH * OW * cout * kh * kw * cin, e0, e1,
                             f"wgrad M={N * OH * OW} N={cout} K={kh * kw * cin} k{kh}x{kw}s{stride}"))
        if CONV_KERNELS is not None:
            CONV_KERNELS.append("conv_wgrad_kernel")
        if CONV_BYTES is not None:
            CONV_BYTES.append((x.numel() + dy.numel() + out.numel()) * 4)
    return out


def dgrad_weight(w):
    """[Cout][KH][KW][Cin] -> [Cin][KH][KW][Cout] rotated by 180 degrees: the weights whose forward convolution over dy is
    the input gradient (layout plumbing; re-derived whenever the weights change)."""
    return w.flip(1, 2).permute(3, 1, 2, 0).contiguous()


def conv2d_dgrad(dy, w, in_hw, stride=1, pad=0, dil=1, x3=True, out=None, out_coff=0):
    """Input gradient of `conv2d`: dy [N,OH,OW,Cout] f32, w [Cout][KH][KW][Cin] f32 -> dx [N,H,W,Cin] f32.
    = tt_conv2d_fwd of dy (zero-inserted for stride > 1) with dgrad_weight(w) and padding dil*(K-1) - pad; `x3` runs it
    in bf16x3 like the forward of the headline mode (else exact f32).  With `out` the result is ADDED to the channel
    window [out_coff, out_coff + Cin) of that gradient buffer (the conv's residual input is the buffer itself)."""
    from . import weights
    require_cuda(dy, w)
    assert dy.dtype == torch.float32 and w.dtype == torch.float32
    Cout, KH, KW, Cin = w.shape
    H, W = in_hw
    N, OH, OW, Cd = dy.shape
    assert Cd == Cout
    fh, fw = H + 2 * pad - dil * (KH - 1), W + 2 * pad - dil * (KW - 1)     # stride-1 output size of the forward conv
    if stride > 1:
        z = torch.zeros(N, fh, fw, Cout, dtype=torch.float32, device=dy.device)
        z[:, :(OH - 1) * stride + 1:stride, :(OW - 1) * stride + 1:stride] = dy
        dy = z
    else:
        assert (OH, OW) == (fh, fw), ((OH, OW), (fh, fw))
    wt = dgrad_weight(w)
    if Cout % 4:        # the convolution wants 16-byte channel vectors: zero channels on both operands
        cp = (Cout + 3) // 4 * 4
        dy = torch.nn.functional.pad(dy, (0, cp - Cout))
        wt = torch.nn.functional.pad(wt, (0, cp - Cout))
    pad_h, pad_w = dil * (KH - 1) - pad, dil * (KW - 1) - pad
    assert pad_h == pad_w and pad_h >= 0, "conv2d_dgrad: square kernels with pad <= dil*(K-1)"
    wx = weights.split_pairs_x3(wt) if (x3 and Cout % 32 == 0) else None
    if out is not None:
        return conv2d(dy, wt, stride=1, pad=pad_h, dil=dil, w_x3=wx, out=out, out_coff=out_coff, res1=out,
                      res1_coff=out_coff, _no_tape=True)
    return conv2d(dy, wt, stride=1, pad=pad_h, dil=dil, w_x3=wx, _no_tape=True)


def conv_epilogue_bwd(dy, y, scale=None, shift=None, act=0, res1=None, res2=None, want_dres=False, dscale=None, dshift=None,
                      accumulate=False, C=None, dy_coff=0, y_coff=0, res1_coff=0, res2_coff=0, dres1=None, dres1_coff=0,
                      dres2=None, dres2_coff=0, dres_accumulate=True, m_dev=None, pre=None, conv_raw=None):
    """Backward of conv2d's fused epilogue (tt_conv_epilogue_bwd): dy / y / res* [..., Cs] f32 channel-last views of the
    same M rows (row stride = last dim) -> (dconv [M, C] dense f32, dres, dscale [C], dshift [C]).  `dres1` / `dres2`:
    gradient buffers of the residual inputs, g is added to (or, dres_accumulate=False, written over) their channel
    windows; with want_dres a dense [M, C] copy of g is returned instead."""
    require_cuda(dy, y)
    C = C or (y.shape[-1] - y_coff)
    M = y.numel() // y.shape[-1]
    assert dy.numel() // dy.shape[-1] == M and dy.dtype == torch.float32 and y.dtype == torch.float32
    dev = y.device
    dconv = torch.empty(M, C, dtype=torch.float32, device=dev)
    dres = None
    if want_dres:
        assert dres1 is None
        dres = dres1 = torch.empty(M, C, dtype=torch.float32, device=dev)
        dres1_coff, dres_accumulate = 0, False
    if dscale is None:
        dscale = torch.empty(C, dtype=torch.float32, device=dev)
        dshift = torch.empty(C, dtype=torch.float32, device=dev)
        accumulate = False
    L = lib()
    L.tt_conv_epilogue_bwd_workspace_bytes.restype = ctypes.c_longlong
    nb = int(L.tt_conv_epilogue_bwd_workspace_bytes(_c(C)))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def cs(t):
        return 0 if t is None else t.shape[-1]
    check(L.tt_conv_epilogue_bwd(ptr(dy), _c(dy.shape[-1]), _c(dy_coff), ptr(y), _c(y.shape[-1]), _c(y_coff),
                                 ptr(res1), _c(cs(res1)), _c(res1_coff), ptr(res2), _c(cs(res2)), _c(res2_coff),
                                 ptr(scale), ptr(shift), _ll(M), _c(C), _c(act), ptr(dconv), _c(C), _c(0),
                                 ptr(dres1), _c(cs(dres1)), _c(dres1_coff), ptr(dres2), _c(cs(dres2)), _c(dres2_coff),
                                 _c(1 if dres_accumulate else 0), ptr(dscale if scale is not None else None), ptr(dshift),
                                 _c(1 if accumulate else 0), ptr(m_dev), ptr(pre), ptr(conv_raw), ptr(ws), _ll(nb), _st(y)),
          "tt_conv_epilogue_bwd")
    return dconv, dres, (dscale if scale is not None else None), dshift


def maxpool3x3s2_bwd(x, dy, dx):
    """dx += backward of maxpool3x3s2 (x: the forward input, dy: gradient of the pooled map); all f32 channel-last."""
    require_cuda(x, dy, dx)
    N, H, W, C = x.shape
    assert x.is_contiguous() and dy.is_contiguous() and dx.is_contiguous() and dx.shape == x.shape
    check(lib().tt_maxpool3x3s2_bwd(ptr(x), ptr(dy), ptr(dx), _c(N), _c(H), _c(W), _c(C), _st(x)), "tt_maxpool3x3s2_bwd")
    return dx


def bilinear_up2_bwd(dy, dx):
    """dx += backward of bilinear_up2 (align_corners x2)."""
    require_cuda(dy, dx)
    N, H, W, C = dx.shape
    assert dy.is_contiguous() and dx.is_contiguous() and tuple(dy.shape) == (N, 2 * H, 2 * W, C)
    check(lib().tt_bilinear_up2_bwd(ptr(dy), ptr(dx), _c(N), _c(H), _c(W), _c(C), _st(dy)), "tt_bilinear_up2_bwd")
    return dx


def channel_gate_bwd(x, gate, dy, dx, dgate, out_relu=None, dres=None):
    """dx += g * sigmoid(gate), dgate += sigmoid'(gate) * sum_hw g * x, g = dy (or dy * [out_relu > 0], then dres += g);
    all f32, x / dy / dx [N,H,W,C] contiguous."""
    require_cuda(x, gate, dy, dx, dgate)
    N, H, W, C = x.shape
    assert x.is_contiguous() and dy.is_contiguous() and dx.is_contiguous() and gate.is_contiguous() and dgate.is_contiguous()
    check(lib().tt_channel_gate_bwd(ptr(x), ptr(gate), ptr(dy), ptr(dx), ptr(dgate), _c(N), _c(H * W), _c(C),
                                    ptr(out_relu), ptr(dres), _st(x)), "tt_channel_gate_bwd")


def spatial_meanmax_bwd(x, dpool, dx):
    """dx += backward of spatial_pool(x, 1) (0.5 mean + 0.5 amax); x / dx [N,H,W,C] contiguous f32, dpool [N,C]."""
    require_cuda(x, dpool, dx)
    N, H, W, C = x.shape
    assert x.is_contiguous() and dx.is_contiguous() and dpool.is_contiguous()
    check(lib().tt_spatial_meanmax_bwd(ptr(x), ptr(dpool), ptr(dx), _c(N), _c(H * W), _c(C), _st(x)),
          "tt_spatial_meanmax_bwd")


def spatial_mean_bwd(dpool, dx, C, coff=0):
    """dx[n, :, :, coff:coff+C] += dpool[n] / (H*W)."""
    require_cuda(dpool, dx)
    N, H, W, Cs = dx.shape
    assert dpool.is_contiguous() and dx.is_contiguous() and tuple(dpool.shape) == (N, C)
    check(lib().tt_spatial_mean_bwd(ptr(dpool), ptr(dx), _c(N), _c(H * W), _c(C), _c(Cs), _c(coff), _st(dx)),
          "tt_spatial_mean_bwd")


def deform_im2col3x3_bwd(x, offsets, gcols, gx, goff, pad=1):
    """gx += , goff += backward of deform_im2col3x3 (f32; gx through atomics)."""
    require_cuda(x, offsets, gcols, gx, goff)
    N, H, W, C = x.shape
    assert all(t.is_contiguous() and t.dtype == torch.float32 for t in (x, offsets, gcols, gx, goff))
    check(lib().tt_deform_im2col3x3_bwd(ptr(x), ptr(offsets), ptr(gcols), ptr(gx), ptr(goff), _c(N), _c(H), _c(W), _c(C),
                                        _c(offsets.shape[-1]), _c(pad), _st(x)), "tt_deform_im2col3x3_bwd")


def broadcast_rows_bwd(dout, dv, out_coff):
    """dv (N, C) (row-strided) += per-image sums of dout (N,H,W,Ct)[..., out_coff:out_coff+C]."""
    N, C = dv.shape
    _, H, W, Ct = dout.shape
    check(lib().tt_broadcast_rows_bwd(ptr(dout), ptr(dv), _c(N), _c(H * W), _c(C), _c(Ct), _c(out_coff), _c(dv.stride(0)),
                                      _st(dout)), "tt_broadcast_rows_bwd")


def upsample_nearest_add_bwd(ddst, dsrc):
    """dsrc += sum of ddst over the pixels that sampled it (backward of upsample_nearest_add_ w.r.t. src)."""
    require_cuda(ddst, dsrc)
    N, H, W, C = ddst.shape
    _, h, w, _ = dsrc.shape
    assert ddst.is_contiguous() and dsrc.is_contiguous()
    check(lib().tt_upsample_nearest_add_bwd(ptr(ddst), ptr(dsrc), _c(N), _c(H), _c(W), _c(C), _c(h), _c(w), _st(ddst)),
          "tt_upsample_nearest_add_bwd")
    return dsrc


# ----------------------------------------------------------------------------- composite decoder kernels
def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def look_query_ln(qos, ref, wp, ctrl, raw_ctrl, temporal, static, meas, flat, maps, gamma, beta, row_stride=1552,
                  eps=1e-5):
    """look_gather_query + LayerNorm(1543) in one launch -> (B*4*120, row_stride) rows, zero padded."""
    B = wp.shape[0]
    assert flat.is_contiguous() and meas.is_contiguous() and wp.is_contiguous() and ctrl.is_contiguous()
    out = torch.empty(B * 4 * 120, row_stride, dtype=torch.float32, device=wp.device)
    arr, hw = _level_args(maps)
    check(lib().tt_look_query_ln(_c(B), ptr(qos), ptr(ref), ptr(wp), ptr(ctrl), _c(1 if raw_ctrl else 0), ptr(temporal),
                                 ptr(static), ptr(meas), ptr(flat), arr, hw, _c(dtype_code(maps[0])), ptr(gamma),
                                 ptr(beta), _f(eps), ptr(out), _c(row_stride), _st(wp)), "tt_look_query_ln")
    return out


def msda_sample_ln(value, offsets, logits, ref, level_hw, B, coff, gamma, beta, eps=1e-5, max_len=None):
    """msda_sample + LayerNorm(256): -> (raw rows, normalised rows), both (B*4*120, 256) f32.  `max_len` (device int from
    look_project_pack): the rows of slots >= max_len are skipped and left unwritten (sca_reduce_ln never reads them)."""
    R = B * 4 * 120
    out = torch.empty(R, 256, dtype=torch.float32, device=value.device)
    out_ln = torch.empty(R, 256, dtype=torch.float32, device=value.device)
    hw = (ctypes.c_int * 8)(*[v for pair in level_hw for v in pair])
    check(lib().tt_msda_sample_ln(_c(B), ptr(value), _c(dtype_code(value)), _c(value.shape[-1]), _c(coff), ptr(offsets),
                                  ptr(logits), ptr(ref), hw, ptr(gamma), ptr(beta), _f(eps), ptr(out), ptr(out_ln),
                                  ptr(max_len), _st(value)), "tt_msda_sample_ln")
    return out, out_ln


def msda_sample_proj_ln(maps, offsets, logits, ref, B, wvT, bias, vshift, gamma, beta, eps=1e-5, max_len=None):
    """msda_sample_ln without a projected value tensor (tt_msda_sample_proj_ln): `maps` = the four fpn_linear maps (B*4, H_l, W_l,
    256) f32, wvT (256 in, 256 out) = value_proj.weight^T, bias (256), vshift (4 levels, 4 cameras, 256) -> (raw rows, normalised
    rows), both (B*4*120, 256) f32."""
    R = B * 4 * 120
    dev = offsets.device
    assert all(m.dtype == torch.float32 and m.is_contiguous() and m.shape[-1] == 256 for m in maps)
    assert tuple(wvT.shape) == (256, 256) and wvT.is_contiguous() and tuple(vshift.shape) == (4, 4, 256) and vshift.is_contiguous()
    out = torch.empty(R, 256, dtype=torch.float32, device=dev)
    out_ln = torch.empty(R, 256, dtype=torch.float32, device=dev)
    arr, hw = _level_args(maps)
    check(lib().tt_msda_sample_proj_ln(_c(B), arr, hw, ptr(offsets), ptr(logits), ptr(ref), ptr(wvT), ptr(bias), ptr(vshift),
                                       ptr(gamma), ptr(beta), _f(eps), ptr(out), ptr(out_ln), ptr(max_len), _st(offsets)),
          "tt_msda_sample_proj_ln")
    return out, out_ln


def sca_reduce_ln(x, max_len, B, gamma, beta, eps=1e-5):
    out = torch.empty(B, 1024, dtype=torch.float32, device=x.device)
    check(lib().tt_sca_reduce_ln(_c(B), ptr(x), ptr(max_len), ptr(gamma), ptr(beta), _f(eps), ptr(out), _st(x)),
          "tt_sca_reduce_ln")
    return out


def dec_merge_in(fflat, look, temporal, meas, gamma, beta, eps=1e-5):
    B = look.shape[0]
    assert fflat.is_contiguous() and look.is_contiguous() and meas.is_contiguous()
    out = torch.empty(B * 4, 1024, dtype=torch.float32, device=look.device)
    check(lib().tt_dec_merge_in(_c(B), ptr(fflat), ptr(look), ptr(temporal), ptr(meas), ptr(gamma), ptr(beta), _f(eps),
                                ptr(out), _st(look)), "tt_dec_merge_in")
    return out


def dec_gru(wts, inp6, state, fut):
    """wts: dict from decoder_fused.prep_gru; inp6 (B,4,6), state (B,441,32), fut (B,4,441,32) all f32 contiguous."""
    B = state.shape[0]
    assert inp6.is_contiguous() and state.is_contiguous() and fut.is_contiguous()
    L = lib()
    L.tt_dec_gru_scratch_floats.restype = ctypes.c_longlong
    scratch = torch.empty(L.tt_dec_gru_scratch_floats(_c(B)), dtype=torch.float32, device=state.device)
    check(L.tt_dec_gru(_c(B), ptr(inp6), ptr(state), ptr(fut), ptr(scratch), wts["w0"], wts["wx"], wts["b0"],
                       wts["w2"], wts["b2"], ptr(wts["wd0"]), ptr(wts["bd0"]), ptr(wts["wd2"]), ptr(wts["bd2"]),
                       _st(state)), "tt_dec_gru")
    return fut


def dec_flatten(wts, maps, want_mids=False):
    """maps (N,441,32) f32 contiguous -> (N,256) [, mids (N, 100*64 + 16*128 + 4*256)]."""
    N = maps.shape[0]
    assert maps.is_contiguous() and maps.dtype == torch.float32
    out = torch.empty(N, 256, dtype=torch.float32, device=maps.device)
    mids = torch.empty(N, 100 * 64 + 16 * 128 + 4 * 256, dtype=torch.float32, device=maps.device) if want_mids else None
    L = lib()
    L.tt_dec_flatten_scratch_floats.restype = ctypes.c_longlong
    scratch = torch.empty(L.tt_dec_flatten_scratch_floats(_c(N)), dtype=torch.float32, device=maps.device)
    check(L.tt_dec_flatten(_c(N), ptr(maps), ptr(out), ptr(mids), ptr(scratch), wts["w"], wts["b"], ptr(wts["bn_scale"]),
                               ptr(wts["bn_shift"]), _st(maps)), "tt_dec_flatten")
    return (out, mids) if want_mids else out


def dec_bev_update(wts, bev, G, out):
    """bev (B,441,32), G (B,1152) -> out (B,441,32) (all f32 contiguous)."""
    B = bev.shape[0]
    assert bev.is_contiguous() and G.is_contiguous() and out.is_contiguous()
    L = lib()
    L.tt_dec_bev_update_scratch_floats.restype = ctypes.c_longlong
    scratch = torch.empty(L.tt_dec_bev_update_scratch_floats(_c(B)), dtype=torch.float32, device=bev.device)
    check(L.tt_dec_bev_update(_c(B), ptr(bev), ptr(G), ptr(out), _ll(441 * 32), None, _ll(0), ptr(scratch), ptr(wts["w0"]),
                              ptr(wts["b0"]), wts["w2"], ptr(wts["b2"]), _st(bev)), "tt_dec_bev_update")
    return out
