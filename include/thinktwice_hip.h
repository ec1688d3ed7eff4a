/*
 * thinktwice_hip.h -- C ABI of libthinktwice_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the per-frame forward path of OpenDriveLab/ThinkTwice
 * (reference: open_loop_training/).  Every entry point takes raw DEVICE
 * pointers, plain sizes and a hipStream_t passed as void*; buffers are
 * caller-owned; calls are asynchronous on the given stream; return value is
 * 0 on success and <0 on error (tt_last_error() holds the text).  Nothing
 * here ever calls exit() (the reference launcher does: ops/voxel_pooling/src/
 * voxel_pooling_forward_cuda.cu:51-55).
 *
 * dtype codes: TT_F32 = 0 (exact f32 MFMA path, parity mode), TT_BF16 = 1
 * (bf16 storage, f32 accumulate).
 */
#ifndef THINKTWICE_HIP_H
#define THINKTWICE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TT_F32 0
#define TT_BF16 1

/* activation codes for fused epilogues */
#define TT_ACT_NONE 0
#define TT_ACT_RELU 1
#define TT_ACT_SIGMOID 2
#define TT_ACT_GELU 3
#define TT_ACT_SOFTPLUS 4

const char* tt_last_error(void);
int tt_version(void);

/* ------------------------------------------------------------------------
 * B1: voxel pooling (Lift-Splat "splat").
 * Replaces voxel_pooling_forward_wrapper
 *   (ops/voxel_pooling/src/voxel_pooling_forward.cpp:24-37) and its kernel
 *   (ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:9-36), argument for
 *   argument: geom_xyz int32 [B,Np,3]; input_features f32 [B,Np,C];
 *   output_features f32 [B,Y,X,C] caller-allocated and pre-zeroed; pos_memo
 *   int32 [B,Np,3] caller-allocated and pre-filled with -1 (may be NULL in
 *   inference -- the reference always writes it).
 * ---------------------------------------------------------------------- */
int tt_voxel_pool_fwd(int batch_size, int num_points, int num_channels,
                      int num_voxel_x, int num_voxel_y, int num_voxel_z,
                      const int32_t* geom_xyz, const float* input_features,
                      float* output_features, int32_t* pos_memo, void* stream);

/* A9: VoxelPooling.backward (ops/voxel_pooling/voxel_pooling.py:57-69):
 * grad_in[b,p,:] = grad_out[b,:,y,x] for kept points, 0 elsewhere.
 * grad_out is the [B,Y,X,C] (channel-last) view of the reference's [B,C,Y,X]. */
int tt_voxel_pool_bwd(int batch_size, int num_points, int num_channels,
                      int num_voxel_x, int num_voxel_y,
                      const int32_t* pos_memo, const float* grad_out_bhwc,
                      float* grad_in, void* stream);

/* A7 + LSS.voxel_pooling_method index math (backbones/lss.py:474-512,629-631):
 * frustum point -> ego xyz -> int voxel index (C truncation toward zero).
 * mats: per (b,cam) two 4x4 row-major f32 matrices: inv(ida) and
 * sensor2ego @ inv(intrin)   [B*ncam][2][16].
 * frustum f32 [D,fH,fW,4]; geom_xyz out int32 [B, ncam*D*fH*fW, 3].
 * voxel_lo[3] = voxel_coord - voxel_size/2, voxel_size[3]. */
int tt_frustum_voxel_index(int batch_size, int num_cams, int D, int fH, int fW,
                           const float* frustum, const float* mats,
                           const float* voxel_lo, const float* voxel_size,
                           int32_t* geom_xyz, float* geom_f32_or_null,
                           void* stream);

/* A6+A8 fused: depth-softmax (x) context -> BEV without materialising the
 * [B,N,D,H,W,C] outer product (backbones/lss.py:583-615 + voxel pooling).
 * depth_logits [B*ncam, fH, fW, D] (channel-last), context [B*ncam, fH, fW, C]
 * (dtype = TT_F32 or TT_BF16), geom_xyz int32 as above,
 * out f32 [B, Y, X, out_cstride] written at channel offset out_coff, with the
 * rot90(flip) of encoder_decoder_framework.py:241 applied when rot_flip != 0.
 * out must be pre-zeroed. */
int tt_lift_splat_fwd(int batch_size, int num_cams, int D, int fH, int fW, int C,
                      int num_voxel_x, int num_voxel_y, int num_voxel_z,
                      const void* depth_logits, const void* context, int dtype,
                      const int32_t* geom_xyz, float* out, int out_cstride,
                      int out_coff, int rot_flip, void* stream);

/* ------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear on MFMA, channel-last activations.
 *   out[n,oh,ow,co] = act( scale[co]*sum(...) + shift[co] + shift_n[n%mod,co]
 *                          + res1 + res2 )
 * Covers nn.Conv2d / nn.Linear / ConvTranspose2d(k2,s2) call sites of
 * backbones/lss.py, encoder_decoder_framework.py, dense_heads/ (all files).
 * ---------------------------------------------------------------------- */
typedef struct tt_conv_desc {
    /* input  [N, H, W, in_cstride] read at channel offset in_coff, Cin used */
    const void* in;   int N, H, W, Cin, in_cstride, in_coff;
    long long in_nstride;           /* elements between images (0 => H*W*in_cstride) */
    /* weights [Cout_total][KH][KW][Cin] (K contiguous), dtype = dtype */
    const void* weight; int Cout, KH, KW, stride, pad, dil;
    /* output [N, OH, OW, out_cstride] at channel offset out_coff */
    void* out; int OH, OW, out_cstride, out_coff;
    long long out_nstride;          /* elements between images (0 => OH*OW*out_cstride) */
    int pixel_shuffle2;             /* 1: ConvTranspose2d k2 s2: Cout = 4*Cout_real,
                                       column j=(dh*2+dw)*Cout_real+co goes to pixel (2h+dh,2w+dw) */
    /* epilogue (all f32, nullable) */
    const float* scale; const float* shift;
    const float* shift_n; int shift_n_mod;
    const void* res1; int res1_cstride, res1_coff;
    const void* res2; int res2_cstride, res2_coff;
    int act;
    int dtype;       /* TT_F32 / TT_BF16: input, weight, residual storage type */
    int out_dtype;   /* storage type of out */
} tt_conv_desc;

int tt_conv2d_fwd(const tt_conv_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* THINKTWICE_HIP_H */
