"""Checkpoint-tensor -> kernel-layout conversion (one-off, at load time; plumbing).

The reference keeps PyTorch layouts in its state_dict ([Cout,Cin,KH,KW] convs, [out,in] linears,
[Cin,Cout,2,2] transposed convs).  The MFMA kernels want K-contiguous rows
[Cout][KH][KW][Cin_padded]; BatchNorm (eval) folds into a per-channel f32 scale/shift that the
conv epilogue applies.
"""
import torch


X3 = "f32x3"      # precision mode: f32 storage, bf16x3 MFMA arithmetic (pre-split weights beside the f32 ones)


# precision mode of the MODEL: bf16x3 everywhere except the PAFPN's 3 x 3 convolutions, which read IEEE-half copies of their inputs and
# run the two-MFMA "h2" product (the one stage the storage-level emulation clears).  Every module but LSS treats it as X3.
X3H = "f32x3h"
# precision mode of ONE layer: half activation storage x f16 (hi, lo) weight pair (csrc/conv_h2.hip)
H2 = "h2"


def is_x3(dtype):
    return isinstance(dtype, str) and dtype in (X3, X3H)


def storage_dtype(dtype):
    """torch dtype activations / plain weights are stored in for a precision mode."""
    if is_x3(dtype):
        return torch.float32
    return torch.float16 if (isinstance(dtype, str) and dtype == H2) else dtype


def split_pairs_h2(w):
    """f32 weights [Cout, ..., K-contiguous] with K % 16 == 0 -> f16 tensor with a doubled last dimension holding, per 16 K
    elements, 64 B = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] with hi = f16(w), lo = f16(w - hi) (round to nearest even both):
    the B operand of conv_h2_kernel (tt_conv_desc.weight_h2)."""
    assert w.dtype == torch.float32 and w.is_contiguous() and w.shape[-1] % 16 == 0
    co = w.shape[0]
    k = w.numel() // co
    flat = w.reshape(co, k // 16, 16)
    hi = flat.to(torch.float16)
    lo = (flat - hi.float()).to(torch.float16)
    pair = torch.stack([hi, lo], 2).contiguous()            # (co, k/16, 2, 16) f16 = 64 B per group
    return pair.reshape(*w.shape[:-1], 2 * w.shape[-1]).contiguous()


def vec_of(dtype):
    return 4 if storage_dtype(dtype) == torch.float32 else 8


def split_pairs_x3(w):
    """f32 weights [Cout, ..., K-contiguous] with K % 16 == 0 -> the same shape (f32-typed bit container) holding, per
    16 K elements, 64 B = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] in bf16 with hi = bf16(w), lo = bf16(w - hi)
    (round to nearest even both): the B operand of conv_igemm_glds_kernel<..., X3>."""
    assert w.dtype == torch.float32 and w.is_contiguous()
    co = w.shape[0]
    k = w.numel() // co
    if k % 16 != 0:
        return None
    flat = w.reshape(co, k // 16, 16)
    hi = flat.to(torch.bfloat16)
    lo = (flat - hi.float()).to(torch.bfloat16)
    pair = torch.stack([hi, lo], 2).contiguous()            # (co, k/16, 2, 16) bf16 = 64 B per group
    return pair.view(torch.float32).reshape(w.shape).contiguous()


def pad_to(n, m):
    return (n + m - 1) // m * m


def split_pairs_frag(w):
    """f32 matrix [N, K] (N % 32 == 0, K % 16 == 0) -> FRAGMENT-MAJOR pair format for kernels that load the MFMA B
    operand straight from memory (csrc/dec_chain.hip, csrc/dec_spatial.hip): per (32-row block nb, 16-wide K step ks)
    2 KiB = [hi plane: lane 0..63 x 16 B | lo plane: lane x 16 B], lane = (k half h) * 32 + (row r), the 16 B being the
    8 bf16 of W[nb*32 + r][ks*16 + h*8 : +8].  A wave's fragment load is then ONE contiguous KiB (8 full cache lines)
    instead of 64 rows x 16 B scattered over 64 lines.  Returned as an f32-typed bit container of N*K elements."""
    assert w.dtype == torch.float32 and w.dim() == 2 and w.shape[0] % 32 == 0 and w.shape[1] % 16 == 0
    N, K = w.shape
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)

    def frag(t):                                   # [N, K] -> [NB, nsteps, h, r, e]
        return t.reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4)
    out = torch.stack([frag(hi), frag(lo)], 2).contiguous()          # [NB, nsteps, plane, h, r, e]
    return out.view(torch.float32).reshape(-1).contiguous()


def prep_conv_weight(w, dtype, cin_pad=None):
    """[Cout,Cin,KH,KW] -> [Cout,KH,KW,Cin_p] contiguous in `dtype` (zero-padded input channels)."""
    Cout, Cin, KH, KW = w.shape
    dtype = storage_dtype(dtype)
    cp = cin_pad or pad_to(Cin, vec_of(dtype))
    out = torch.zeros(Cout, KH, KW, cp, dtype=dtype, device=w.device)
    out[..., :Cin] = w.permute(0, 2, 3, 1).to(dtype)
    return out.contiguous()


def prep_linear_weight(w, dtype, in_pad=None):
    """[out,in] -> [out,1,1,in_p]."""
    return prep_conv_weight(w[:, :, None, None], dtype, in_pad)


def prep_deconv2x2_weight(w, dtype):
    """ConvTranspose2d(k=2,s=2) weight [Cin,Cout,2,2] -> [(dh*2+dw)*Cout+co, 1, 1, Cin_p]."""
    Cin, Cout, KH, KW = w.shape
    assert KH == 2 and KW == 2
    m = w.permute(2, 3, 1, 0).reshape(4 * Cout, Cin)
    return prep_linear_weight(m, dtype)


def fold_bn(bn_weight, bn_bias, running_mean, running_var, eps, conv_bias=None):
    """eval-mode BatchNorm after a conv -> (scale, shift) f32 with y = conv_nobias * scale + shift."""
    scale = bn_weight.float() / torch.sqrt(running_var.float() + eps)
    shift = bn_bias.float() - running_mean.float() * scale
    if conv_bias is not None:
        shift = shift + conv_bias.float() * scale
    return scale.contiguous(), shift.contiguous()


def to_channel_last(x, dtype=None, c_pad=None):
    """[N,C,H,W] -> [N,H,W,Cp] contiguous (zero-padded channels)."""
    N, C, H, W = x.shape
    dtype = storage_dtype(dtype or x.dtype)
    cp = c_pad or pad_to(C, vec_of(dtype))
    out = torch.zeros(N, H, W, cp, dtype=dtype, device=x.device)
    out[..., :C] = x.permute(0, 2, 3, 1).to(dtype)
    return out
