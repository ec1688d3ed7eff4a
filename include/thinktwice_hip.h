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
#define TT_F16 2   /* IEEE half storage, f32 accumulate (v_mfma_f32_32x32x16_f16): same rate as bf16, 8x finer rounding */

/* activation codes for fused epilogues */
#define TT_ACT_NONE 0
#define TT_ACT_RELU 1
#define TT_ACT_SIGMOID 2
#define TT_ACT_GELU 3
#define TT_ACT_SOFTPLUS 4
#define TT_ACT_SOFTPLUS_CLAMP 5  /* clamp(softplus(x), min=1e-3): thinktwice_decoder.py:484 */

const char* tt_last_error(void);
/* measurement aid: name (template arguments spelled like rocprofv3 prints them) of the kernel that the last
 * tt_conv2d_fwd call of this thread launched; "" before the first call */
const char* tt_conv_last_kernel(void);
/* measurement aid: while set, every workgroup of the LDS-DMA conv kernel writes 4 wall-clock stamps (10 ns ticks: entry, first K tile
 * landed, K loop done, epilogue done) at stamps[blockIdx.x * 4]; null (default) = off */
int tt_conv_set_trace(void* stamps_or_null);
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

/* Same operator with a caller-provided workspace: atomics-free two-phase reduce (LDS-privatised
 * per-chunk cell sums, then an ordered per-cell gather).  Falls back to tt_voxel_pool_fwd when the
 * workspace is missing/small or the shape is unsupported (bytes query returns 0). */
long long tt_voxel_pool_workspace_bytes(int batch_size, int num_points, int num_channels,
                                        int num_voxel_x, int num_voxel_y);
int tt_voxel_pool_fwd_ws(int batch_size, int num_points, int num_channels,
                         int num_voxel_x, int num_voxel_y, int num_voxel_z,
                         const int32_t* geom_xyz, const float* input_features,
                         float* output_features, int32_t* pos_memo,
                         void* workspace, long long workspace_bytes, void* stream);

/* Planned forward for STATIC geometry (fixed camera rig: the same geom_xyz every frame; closed-loop ticks, the bench).
 * tt_voxel_pool_plan_build sorts the in-range points by (sample, cell) once; tt_voxel_pool_fwd_planned then streams
 * the point rows cell by cell with no index work, no atomics, deterministic.  Same result as tt_voxel_pool_fwd
 * (output_features is accumulated into, pre-zero it like the reference does). */
long long tt_voxel_pool_plan_bytes(int batch_size, int num_points, int num_voxel_x, int num_voxel_y);
long long tt_voxel_pool_plan_workspace_bytes(int batch_size, int num_points);
int tt_voxel_pool_plan_build(int batch_size, int num_points, int num_voxel_x, int num_voxel_y, int num_voxel_z,
                             const int32_t* geom_xyz, void* workspace, long long workspace_bytes, void* plan,
                             long long plan_bytes, void* stream);
long long tt_voxel_pool_planned_workspace_bytes(int batch_size, int num_points, int num_channels, int num_voxel_x,
                                                int num_voxel_y);
int tt_voxel_pool_fwd_planned(int batch_size, int num_points, int num_channels, int num_voxel_x, int num_voxel_y,
                              const void* plan, const float* input_features, float* output_features, void* workspace,
                              long long workspace_bytes, void* stream);

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
 * voxel_lo[3] = voxel_coord - voxel_size/2, voxel_size[3]: HOST pointers (module constants). */
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

/* Same with a caller-provided workspace (tt_lift_splat_workspace_bytes; its contents do not matter): every image strip
 * stores one partial row per BEV cell it touches plus its cell -> slot table, and a second kernel adds each cell's
 * partial rows to `out` in strip order -- no floating-point atomics, so equal inputs give bit-identical output.
 * ws == NULL, or a shape the strip kernel does not take (workspace_bytes == 0), is tt_lift_splat_fwd. */
long long tt_lift_splat_workspace_bytes(int batch_size, int num_cams, int D, int fH, int fW, int C,
                                        int num_voxel_x, int num_voxel_y);
int tt_lift_splat_fwd_ws(int batch_size, int num_cams, int D, int fH, int fW, int C,
                         int num_voxel_x, int num_voxel_y, int num_voxel_z,
                         const void* depth_logits, const void* context, int dtype,
                         const int32_t* geom_xyz, float* out, int out_cstride,
                         int out_coff, int rot_flip, void* ws, long long ws_bytes, void* stream);

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
    int dtype;       /* TT_F32 / TT_BF16 / TT_F16: input, weight, residual storage type */
    int out_dtype;   /* storage type of out: TT_F32, the operand dtype, or (f32 operands, no residuals) TT_F16 / TT_BF16 */
    /* sparse (spconv) mode: in = feature rows [*, in_cstride]; N = upper bound of output rows,
     * H=W=OH=OW=KH=1, KW = taps; gather_idx int32 [N][KW] = input row per (output row, tap) or -1
     * (the rulebook of tt_sp_rulebook); m_dev (nullable) = device int with the live row count. */
    const int* gather_idx;
    const int* m_dev;
    /* optional split-K workspace: f32 [N*OH*OW][Cout], zero-filled by the caller.  When given and the
     * launch would occupy < 128 workgroups with a long K loop, K is split across gridDim.y and a tiny
     * finalize kernel applies the epilogue (latency-bound decoder layers with M <= a few thousand). */
    float* splitk_ws;
    /* optional (dtype == TT_F32 only): the same weights pre-split into bf16 (hi, lo) pairs, per 16 K elements
     * 64 B = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] (thinktwice_amd/weights.py::split_pairs_x3).  When given
     * and the layer fits the LDS-DMA kernel it runs in "bf16x3" arithmetic -- a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on
     * the bf16 MFMA, ~1e-5 relative error, 16/3 of the f32-MFMA rate; otherwise the exact f32 path on `weight`. */
    const void* weight_x3;
    /* optional (gather mode): the tile plan of tt_sp_tile_plan for gather_idx -- output tile i covers rows
     * row_perm[256 i .. 256 i + 255] and multiplies only the taps set in the union of their masks */
    const int* row_perm;
    const unsigned* row_mask;
    /* split-K without atomics: `splitk_ws` holds this many [N*OH*OW][Cout] f32 slices (>= tt_conv2d_splitk_slices(d), no
     * zero fill needed); every K split stores its partial tile into its own slice and the finalize kernel adds the slices
     * in index order -- bit-reproducible.  0: the legacy form (one zero-filled slice, f32 atomics) */
    int splitk_slices;
    /* bf16x3 layers (weight_x3 given) whose activation tensor has no reader but bf16x3 convolutions can move the operand split
     * from the consumer's K loop (once per USE: nine times per element per column tile in a 3 x 3 layer) to the producer (once per
     * element).  "Pair format": per 16 channels 64 B = [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15] in bf16, hi = rne(v),
     * lo = rne(v - hi) -- the weights' format along the channel axis, f32-sized.  The sums are bit-identical to the f32 form.
     *   out_pair: write `out` (out_dtype TT_F32, no residuals, Cout / out_cstride / out_coff multiples of 16) in pair format;
     *   in_pair : `in` is in pair format (in_cstride / in_coff multiples of 16, Cin % 32 == 0, N*OH*OW > 4096, Cout <= 32 or
     *             >= 64): the launch is refused if the layer is outside the LDS-DMA kernel's contract. */
    int in_pair, out_pair;
    /* optional (dtype == TT_F16 only): "h2" arithmetic -- IEEE-half activations as stored x an f16 (hi, lo) weight pair, two
     * v_mfma_f32_32x32x16_f16 per product with f32 accumulation (exact to ~2^-22 with respect to the stored operands).
     * Layout: rows of 2*K halves, per 16 K elements 64 B = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15], hi = f16(w),
     * lo = f16(w - hi) (thinktwice_amd/weights.py::split_pairs_h2).  Needs Cin % 64 == 0; `weight` is then only a shape
     * carrier.  The PAFPN of the mixed mode runs it. */
    const void* weight_h2;
    /* optional second copy of the output, always f32, rows [N*OH*OW] of out2_cstride floats at channel offset out2_coff
     * (row-linear outputs only): a layer whose result is read both by a half-storage consumer and by an f32 one
     * (PAFPN fpn_convs.0: downsample_convs.0 reads the f16 copy, the UNet concat buffer takes the f32 one) */
    float* out2; int out2_cstride, out2_coff;
    /* optional: res1 is a coarser map [N][res1_up_h][res1_up_w][res1_cstride] read through NEAREST upsampling to (OH, OW)
     * (pixel (oh, ow) adds pixel (oh * res1_up_h / OH, ow * res1_up_w / OW)): `lat[i-1] += F.interpolate(lat[i], size=...)` of the
     * PAFPN top-down path (backbones/lss.py:301-305) inside the lateral conv.  0 = res1 has the output's own geometry.  Needs the
     * vector epilogue and more than 4096 output rows. */
    int res1_up_h, res1_up_w;
    /* 1: res1 is an f32 tensor although the operands are 16-bit (dtype TT_F16 / TT_BF16): the f32 sum chains of the mixed mode's
     * PAFPN beside its half conv inputs.  Vector epilogue only. */
    int res1_f32;
} tt_conv_desc;

int tt_conv2d_fwd(const tt_conv_desc* d, void* stream);
/* K splits tt_conv2d_fwd would use for this descriptor if it is given a split-K workspace (0: the layer does not split) */
int tt_conv2d_splitk_slices(const tt_conv_desc* d);

/* ------------------------------------------------------------------------
 * A chain of nn.Linear layers over R rows in ONE launch (the decoder's row-batched MLPs:
 * thinktwice_decoder.py:26-260 query_linear / ffn / output_proj / mlp / traj_offset_module / ctrl_offset_module /
 * flattened_BEV_feat_update_module and the coarse heads :419-445; MSDA sampling_offsets / attention_weights,
 * multi_scale_deformable_attn_function.py:431-447).  A workgroup owns 32 rows and walks the whole chain with the
 * intermediates in LDS; bf16x3 arithmetic (see tt_conv_desc.weight_x3).
 *   stage s:  y = act( W_s . in + bias + side_w . side[m, :side_k] + res[m, res_coff + n] )
 *   in = the chain input x (in_sel = -1) or the output of an earlier stage (in_sel = its index)
 * w: pair-format weights [N rounded up to 32][Kp] (weights.py::split_pairs_x3 of the zero-padded f32 matrix),
 * Kp % 16 == 0.  `out` (nullable): f32 rows in global memory.  Outputs that a later stage reads stay in LDS
 * (<= 160 KiB in total per 32 rows, checked).
 * ---------------------------------------------------------------------- */
typedef struct tt_chain_stage {
    const void* w; const float* bias;
    int K, Kp, N, act, in_sel;
    const float* res; int res_stride, res_coff;
    const float* side; const float* side_w; int side_stride, side_k;
    float* out; int out_stride, out_coff;
} tt_chain_stage;
/* n_split > 1 (single stage only): the N columns are dealt over n_split workgroups per 32 rows (few rows, wide N:
 * the BEV update's broadcast-channel term, 2048 -> 9 x 128) */
int tt_mlp_chain(const float* x, long long R, int x_stride, int nstages, const tt_chain_stage* stages, int n_split,
                 void* stream);
/* The same chain for FEW rows (the batch-1 tick: 1 - 32 rows, where a chain's time is the latency of streaming its weights
 * through one workgroup): the 32-column blocks of every stage are dealt over `n_groups` co-resident workgroups per 32 rows, the
 * waves of a workgroup split K; stage outputs a later stage reads go through `workspace` (f32, global) and the workgroups of
 * a row block meet at a ticket barrier between dependent stages.  Same arithmetic per product as tt_mlp_chain (bf16x3), the K
 * sum is taken in eight interleaved slices added in a fixed order.
 * Co-residency: the barrier needs every workgroup of the launch running.  The library bounds a launch to
 * tt_mlp_chain_wide_max_workgroups() = (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs) / 2 workgroups (two launches at a
 * time: the decoder's main and branch streams) and returns -1 beyond it (callers take tt_mlp_chain instead).  A barrier wait is
 * bounded; a wait that gives up is LOUD: the workgroup writes NaN from there on, the device's fault word (pinned host memory,
 * written by the kernel) is set, and from then on tt_mlp_chain_wide, tt_plan_run, tt_encoder_fwd and tt_decoder_fwd return -3
 * until tt_clear_device_faults().  tt_device_faults(): non-blocking read of that word for the current device (check it after any
 * synchronisation with the forward's streams -- thinktwice_amd does after the D2H copy of tt_action_post's result and at the
 * start of every forward); tt_mlp_chain_wide_faults(): the same after a hipDeviceSynchronize().  Ticket counters are per
 * device; launches recorded into a HIP graph claim theirs permanently (a replay reuses the recorded counter); when those
 * are used up the call returns -4 (nothing launched): record tt_mlp_chain instead. */
long long tt_mlp_chain_wide_workspace_bytes(long long R, int nstages, const tt_chain_stage* stages);
int tt_mlp_chain_wide(const float* x, long long R, int x_stride, int nstages, const tt_chain_stage* stages, int n_groups,
                      void* workspace, long long workspace_bytes, void* stream);
int tt_mlp_chain_wide_max_workgroups(void);
int tt_mlp_chain_wide_faults(void);
int tt_device_faults(void);
int tt_clear_device_faults(void);
/* test hook: polls before a barrier wait gives up (default 1 << 21, about a second; 0 forces a time-out; < 0 restores) */
int tt_mlp_chain_wide_set_max_spin(int polls);
/* debug: while set, every tt_mlp_chain_wide workgroup writes up to 64 wall-clock stamps (10 ns ticks; kernel entry, then per stage:
 * entry, barrier passed, first block's K loop + reduction done, stage done) at stamps[(row_group * n_groups + group) * 64] */
int tt_mlp_chain_wide_set_trace(void* stamps_or_null);

/* ------------------------------------------------------------------------
 * Spatial half of a refinement layer as persistent per-sample kernels (csrc/dec_spatial.hip), bf16x3 arithmetic.
 * All weight tensors are pair format (weights.py::split_pairs_x3), K order tap-major / channel-minor.
 * ---------------------------------------------------------------------- */
/* SpatialGRU (dense_heads/utils.py:53-106): inp6 [B][4][6] (waypoint xy, softplus ctrl), state [B][441][32] f32
 * channel-last -> fut [B][4][441][32].  w0/wx/b0/w2/b2: arrays of 3 (conv_update, conv_reset, conv_state_tilde):
 * w0 = state part of the .0 conv [32][9*32], wx = its constant-input part f32 [9][6][32], w2 = the .2 conv.
 * scratch: tt_dec_gru_scratch_floats(B) floats ([B][3][448][32] f32 -- the 441 pixels padded to 14 row blocks of 32 -- + two flag
 * words per sample: two workgroups per sample hand the state and the update gate over through it; a hand-over that times out
 * sets the device fault word, tt_device_faults). */
long long tt_dec_gru_scratch_floats(int B);
int tt_dec_gru(int B, const float* inp6, const float* state, float* fut, float* scratch, const void* const* w0,
               const float* const* wx, const float* const* b0, const void* const* w2, const float* const* b2,
               const void* wd0, const float* bd0, const void* wd2, const float* bd2, void* stream);
/* grid2feat (encoder_decoder_framework.py:228-234, thinktwice_decoder.py:405-415): maps x [441][32] -> [256];
 * 17 weight sets in the order documented in dec_spatial.hip; mids (nullable): the three SE-block outputs;
 * scratch: tt_dec_flatten_scratch_floats(maps) floats (the tensors between the per-layer launches of the 4x4 level on). */
long long tt_dec_flatten_scratch_floats(int maps);
int tt_dec_flatten(int maps, const float* in, float* out, float* mids_or_null, float* scratch, const void* const* w,
                   const float* const* b, const float* bn_scale, const float* bn_shift, void* stream);
/* debug: while set, workgroup 0 of tt_dec_gru / tt_dec_flatten writes a wall-clock stamp (10 ns ticks) after every phase (<= 64) */
int tt_dec_set_trace(void* stamps_or_null);
/* BEV_feat_update_module + residual (thinktwice_decoder.py:221-225,257): bev [B][441][32], G [B][9][128] = the
 * broadcast-channel term per tap (tt_mlp_chain over h), w0 [128][9*32] (bev part), w2[4] [32][9*32] per hidden chunk;
 * scratch: tt_dec_bev_update_scratch_floats(B) floats (the four hidden-channel chunks' partial maps + a ticket per sample). */
long long tt_dec_bev_update_scratch_floats(int B);
int tt_dec_bev_update(int B, const float* bev, const float* G, float* out, long long out_bstride, float* out2,
                      long long out2_bstride, float* scratch, const void* w0, const float* b0, const void* const* w2,
                      const float* b2, void* stream);

/* ------------------------------------------------------------------------
 * HBM-bound glue of the forward (channel-last; `dtype` = storage type of the activation).
 * Each replaces a torch call of the reference forward (call sites cited).
 * ---------------------------------------------------------------------- */
/* img (B*T*N,3,H,W) f32 NCHW -> channel-last with zero-padded channels (lss.py:517-519) */
int tt_nchw_to_nhwc_pad(const float* in, void* out, int N, int C, int H, int W, int Cp,
                        int out_dtype, void* stream);
/* Same conversion into the interior of a spatially padded buffer out[N][Hp][Wp][Cp] at pixel offset (top, left);
 * the border is NOT written (allocate it zeroed once).  Feeds the row-run form of the 7x7/2 ResNet stem, whose 8-pixel
 * input runs overhang the image (reference: mmdet ResNet.conv1 via backbones/lss.py:517-519 get_cam_feats). */
int tt_nchw_to_nhwc_border(const float* in, void* out, int N, int C, int H, int W, int Cp, int Hp, int Wp,
                           int top, int left, int out_dtype, void* stream);
/* channel-last -> NCHW f32 (outputs returned in the reference layout) */
int tt_nhwc_to_nchw(const void* in, float* out, int N, int C, int H, int W, int cstride, int coff,
                    int in_dtype, void* stream);
/* F.max_pool2d(x,3,2,1): mmdet ResNet stem (lss.py:401 -> [3P]) */
int tt_maxpool3x3s2(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream);
/* dst += F.interpolate(src, size=dst.shape, mode='nearest'): PAFPN top-down (lss.py:301-305) */
int tt_upsample_nearest_add(void* dst, const void* src, int N, int H, int W, int C, int h, int w,
                            int dtype, void* stream);
/* nn.Upsample(scale_factor=2, bilinear, align_corners=True): UNet (lss.py:267) */
int tt_bilinear_up2(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream);
/* the same, f32 in -> bf16x3 pair format out (tt_conv_desc.in_pair of the consuming convolution; C % 16 == 0) */
int tt_bilinear_up2_pair(const float* in, float* out, int N, int H, int W, int C, void* stream);
/* mode 0: AdaptiveAvgPool2d(1) (lss.py:80); mode 1: 0.5*mean+0.5*max (code/utils.py:91-92); out f32 [N,C] */
int tt_spatial_pool(const void* in, float* out, int N, int HW, int C, int cstride, int coff, int mode,
                    int dtype, void* stream);
/* out = out_act(x * gate_act(gate[n,c]) + res): SELayer (lss.py:158), SEModule+residual (utils.py:96,117-119) */
int tt_channel_gate(const void* x, const float* gate, const void* res, void* out, int N, int HW, int C,
                    int gate_act, int out_act, int dtype, void* stream);
/* rows: out = act(x*scale[c] + shift[c]): eval BatchNorm1d (lss.py:232, encoder_decoder_framework.py:134) */
int tt_affine_rows(const void* x, const float* scale, const float* shift, void* out, long long R, int C,
                   int x_stride, int out_stride, int act, int dtype, void* stream);
/* nn.LayerNorm over the last dim (MSDA:252,201,262; thinktwice_decoder.py:197) */
int tt_layernorm_rows(const void* x, const float* gamma, const float* beta, void* out, long long R,
                      int D, int x_stride, int out_stride, float eps, int dtype, void* stream);
/* strided channel-offset copy (torch.cat assembly); rot_flip!=0: torch.rot90(torch.flip(x,[2]),1,[2,3])
 * (encoder_decoder_framework.py:241,246) */
int tt_copy_nhwc(const void* in, void* out, int N, int H, int W, int C, int in_cstride, int in_coff,
                 int out_cstride, int out_coff, int rot_flip, int in_dtype, int out_dtype, void* stream);
/* out[n,p,coff+c] = v[n,c]: `.unsqueeze(-1).unsqueeze(-1).repeat` (thinktwice_decoder.py:42,257) */
int tt_broadcast_rows(const void* v, void* out, int N, int HW, int C, int v_stride, int out_cstride,
                      int out_coff, int dtype, void* stream);
/* op 0: a+b; 1: (1-b)*a; 2: (1-g)*a+g*b; 3: act(a)   (SpatialGRU.gru_cell, dense_heads/utils.py:93-106) */
int tt_ew(const void* a, const void* b, const void* g, void* out, long long R, int C, int a_stride,
          int a_coff, int b_stride, int b_coff, int g_stride, int g_coff, int o_stride, int o_coff,
          int op, int act, int dtype, void* stream);
/* torch.cat([...], -1) of up to 8 row-batched f32 pieces in one launch (decoder MLP inputs, thinktwice_decoder.py:
 * 236-260): piece s writes out[r, coffs[s] : coffs[s]+widths[s]] = srcs[s][((r / divs[s]) % mods[s]) * strides[s] + c]
 * (mods[s] = 0: no modulo; srcs[s] = NULL: zeros).  Pieces are contiguous in output-column order.  All arrays HOST. */
int tt_concat_rows(float* out, long long R, int out_stride, int nseg, const float* const* srcs, const int* strides,
                   const int* widths, const int* coffs, const int* divs, const int* mods, void* stream);

/* mmcv DeformConv2dPack (DCN v1, 3x3, stride 1, deform_groups 1) column builder (lss.py:189-197):
 * cols [N*H*W, 9, C] = bilinear samples of x [N,H,W,C] at the offset taps; offsets f32
 * [N,H,W,off_cstride] with channels [dy_0,dx_0,...,dy_8,dx_8].  The grouped GEMM runs on tt_conv2d_fwd. */
int tt_deform_im2col3x3(const void* x, const float* offsets, void* cols, int N, int H, int W, int C,
                        int off_cstride, int pad, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Look module (dense_heads/thinktwice_decoder.py:88-187 + multi_scale_deformable_attn_function.py).
 * All per-(sample,camera) query packing stays on the device (the reference does B*4 nonzero()
 * host syncs per decoder layer, thinktwice_decoder.py:131-137).  120 slots are always computed;
 * `max_len` (device int) bounds the final reduction exactly like the reference's padded length.
 * ---------------------------------------------------------------------- */
/* obtain_cam_ref_points_query (DEC:89-113,129-149): wp (B,4,2); lidar2img/ida_mat (B,4,4,4) f32.
 * -> ref_packed (B,4,120,2), query_of_slot (B,4,120) (-1 = padded), count (B,4), max_len (1). */
int tt_look_project_pack(int B, const float* wp, const float* lidar2img, const float* ida_mat,
                         float img_h, float img_w, float* ref_packed, int* query_of_slot,
                         int* count, int* max_len, void* stream);
/* query rows (B*4*120, row_stride>=1543) f32 = [ctrl 4 | xyz 3 | emb 128 | meas 128 | flat 256 |
 * F.grid_sample of the 4 fpn_linear maps (c*4+lvl) 1024] (DEC:119-127,148,164-171); padded slots = 0.
 * level_maps: 4 HOST-array device pointers to [B*4,H_l,W_l,256] maps; level_hw: 8 host ints. */
int tt_look_gather_query(int B, const int* query_of_slot, const float* ref_packed, const float* wp,
                         const float* ctrl_softplus, const float* temporal_embedding,
                         const float* static_embedding, const float* measurement_feat,
                         const float* flattened_feat, const void* const* level_maps,
                         const int* level_hw, int maps_dtype, float* out, int row_stride, void* stream);
/* MSDeformableAttention3D core (MSDA:478-526 + mmcv multi_scale_deformable_attn_pytorch [3P]):
 * value [B*4, sum(HW), 256] (8 heads x 32), offsets f32 [R,512], logits f32 [R,256], R = B*4*120. */
int tt_msda_sample(int B, const void* value, int value_dtype, const float* offsets, const float* logits,
                   const float* ref_packed, const int* level_hw, float* out, void* stream);
/* Same core over a value tensor whose rows hold `value_cstride` >= 256 channels, sampling the 256-channel window at
 * `value_coff`: lets ONE value-projection GEMM (Cout = layers x 256) serve all five refinement layers, each layer's
 * attention reading its own window (the reference projects per layer, thinktwice_decoder.py:392-398 inside the loop
 * of 428-447; the projections only depend on the FPN maps). */
int tt_msda_sample_strided(int B, const void* value, int value_dtype, int value_cstride, int value_coff,
                           const float* offsets, const float* logits, const float* ref_packed, const int* level_hw,
                           float* out, void* stream);
/* SpatialCrossAttention "mask & average" incl. its batch-coupling bug (MSDA:338-342):
 * out (B, 4*256) = sum_{s=B}^{max_len-1} x[b,cam,s,:] / B. */
int tt_sca_reduce(int B, const float* x, const int* max_len, float* out, void* stream);

/* Composite-decoder variants of the four row producers above with the nn.LayerNorm that follows them folded in
 * (query_linear.0 thinktwice_decoder.py:131, ffn.norm MSDA:262, output_proj.0 DEC:141, mlp.0 DEC:197): each output
 * feeds tt_mlp_chain directly.  tt_look_query_ln: `ctrl` raw (raw_ctrl=1: softplus applied here) or softplus'ed. */
int tt_look_query_ln(int B, const int* query_of_slot, const float* ref_packed, const float* wp, const float* ctrl,
                     int raw_ctrl, const float* temporal_embedding, const float* static_embedding,
                     const float* measurement_feat, const float* flattened_feat, const void* const* level_maps,
                     const int* level_hw, int maps_dtype, const float* gamma, const float* beta, float eps, float* out,
                     int row_stride, void* stream);
int tt_msda_sample_ln(int B, const void* value, int value_dtype, int value_cstride, int value_coff,
                      const float* offsets, const float* logits, const float* ref_packed, const int* level_hw,
                      const float* gamma, const float* beta, float eps, float* out, float* out_ln,
                      const int* max_len_or_null, void* stream);
/* The same rows WITHOUT a projected value tensor ("sample first, project after"): level_maps = the four fpn_linear maps
 * [B*4][H_l][W_l][256] f32, wvT = value_proj.weight transposed [in 256][out 256], bias [256], vshift [level 4][camera 4][256] =
 * W (cams_embeds + level_embeds) (thinktwice_decoder.py:392-393).  value_proj is applied to the attention-weighted sum of the raw
 * rows; the bias / embedding terms enter through the in-bounds weight of each level (zero padding).  Replaces
 * multi_scale_deformable_attn_function.py:474 (value_proj over every position) + :497-525 in the composite decoder. */
int tt_msda_sample_proj_ln(int B, const void* const* level_maps, const int* level_hw, const float* offsets,
                           const float* logits, const float* ref_packed, const float* wvT, const float* bias,
                           const float* vshift, const float* gamma, const float* beta, float eps, float* out,
                           float* out_ln, const int* max_len_or_null, void* stream);   /* max_len (device, from tt_look_project_pack): rows of
                      slots >= *max_len are skipped and left unwritten -- tt_sca_reduce_ln never reads them */
int tt_sca_reduce_ln(int B, const float* x, const int* max_len, const float* gamma, const float* beta, float eps,
                     float* out, void* stream);
/* row (b, t) of the refinement layer's mlp input: LayerNorm(cat([fflat(b,t) | look(b) | 0 | temporal(t) | meas(b)])) */
int tt_dec_merge_in(int B, const float* fflat, const float* look, const float* temporal_embedding,
                    const float* measurement_feat, const float* gamma, const float* beta, float eps, float* out,
                    void* stream);

/* ------------------------------------------------------------------------
 * LiDAR branch (backbones/lidarnet.py:87-96; bodies are third-party: mmcv Voxelization, mmdet3d
 * HardSimpleVFE / SparseEncoder, spconv).  Active-row counts live in DEVICE ints; every kernel is
 * launched over an upper bound (`max_*`) and exits early, so there is no host synchronisation
 * (the reference syncs at lidarnet.py:90 `coors[-1,0]+1`).  coords are int32 (b,z,y,x).
 * ---------------------------------------------------------------------- */
long long tt_lidar_voxelize_workspace_bytes(long long num_points_total);
/* hard voxelisation + mean VFE: points f32 [B,Np,nfeat]; pc_range_lo / voxel_size: 3 HOST floats;
 * grid_xyz: 3 HOST ints; voxels with z index >= z_limit are dropped (sparse_shape z, CFG:170).
 * -> voxel_feats f32 [<=B*Np, nfeat] (mean of the first <=max_points points in point order),
 *    coords int32 [<=B*Np,4], num_voxels (device int). */
int tt_lidar_voxelize(const float* points, int B, int Np, int nfeat, const float* pc_range_lo,
                      const float* voxel_size, const int* grid_xyz, int z_limit, int max_points,
                      void* workspace, long long workspace_bytes, float* voxel_feats, int* coords,
                      int* num_voxels, void* stream);
/* the same with mmcv's max_voxels cap: voxels are created in order of first appearance (point order) and at most
 * `max_voxels` per sample exist; points of later voxels are dropped (configs/thinktwice.py:161-165: 120000 train / 160000
 * eval; call site backbones/lidarnet.py:88).  Needed only when a sample holds more points than the cap (else
 * tt_lidar_voxelize gives the same rows); rows come out in cell order like there */
long long tt_lidar_voxelize_capped_workspace_bytes(long long num_points_total);
int tt_lidar_voxelize_capped(const float* points, int B, int Np, int nfeat, const float* pc_range_lo, const float* voxel_size,
                             const int* grid_xyz, int z_limit, int max_points, int max_voxels, void* workspace,
                             long long workspace_bytes, float* voxel_feats, int* coords, int* num_voxels, void* stream);
/* dense index volume of one resolution level: vol int32 [batch, D, H, W] = feature row or -1 */
int tt_sp_volume_build(const int* coords, const int* num_rows, long long max_rows, int batch,
                       const int* dims_zyx, int* vol, void* stream);
/* spconv SparseConv3d output-site generation (mark + scan + compact, no atomics; rows come out in cell
 * order); kernel_stride_pad = {kz,ky,kx,sz,sy,sx,pz,py,px} (host).  Also fills the OUTPUT level's volume. */
long long tt_sp_strided_outputs_workspace_bytes(long long out_cells);
int tt_sp_strided_outputs(const int* in_coords, const int* in_rows, long long max_in, int batch,
                          const int* kernel_stride_pad, const int* out_dims_zyx, void* workspace,
                          long long workspace_bytes, int* out_vol, int* out_coords, int* out_rows,
                          long long max_out, void* stream);
/* nbr[o][k] = input row feeding output o through tap k, or -1 (SubM: out coords == in coords) */
int tt_sp_rulebook(const int* out_coords, const int* out_rows, long long max_out,
                   const int* kernel_stride_pad, const int* in_dims_zyx, const int* in_vol, int* nbr,
                   void* stream);
/* the sparse convolution itself = tt_conv2d_fwd with gather_idx = this rulebook (MFMA gathered GEMM) */
/* Tile plan of a rulebook for the mask-sorted gathered GEMM: row_perm = the output rows sorted by their tap-occupancy
 * mask (bit t = tap t has an input), row_mask_sorted = the masks in that order (rows beyond *num_rows: 0xFFFFFFFF, last).
 * A 256-row tile of the sorted order then visits only the UNION of its rows' taps instead of all KV (spconv's
 * "implicit GEMM with mask sort"); pairs (nullable, pre-zeroed) += the number of existing (row, tap) pairs. */
long long tt_sp_tile_plan_workspace_bytes(long long max_rows);
int tt_sp_tile_plan(const int* nbr, const int* num_rows, long long max_rows, int KV, void* workspace,
                    long long workspace_bytes, int* row_perm, unsigned* row_mask_sorted,
                    unsigned long long* pairs_or_null, void* stream);
/* SparseConvTensor.dense() + view(N, C*D, H, W) (lidarnet.py:53-56), channel-last: dense
 * [B, H, W, C*D] with channel c*D+z; `dense` must be pre-zeroed. */
int tt_sp_to_dense(const void* feats, const int* coords, const int* num_rows, long long max_rows, int C,
                   const int* dims_zyx, void* dense, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * fused camera preprocessing (undistort grid_sample + bilinear resize + crop + /255 +
 * ImageNet normalise; datasets/pipelines/transform.py:283-286,346-356,144,163) in one gather kernel.
 * raw uint8 [NI,H,W,3]; mapx/mapy f32 [H,W] = cv2.initUndistortRectifyMap output (pixel units);
 * mean3/std3: HOST floats.  Writes channel-last out_nhwc [NI,out_h,out_w,Cp] (dtype) and/or NCHW f32.
 * ---------------------------------------------------------------------- */
int tt_preprocess_images(const uint8_t* raw_hwc, int num_images, int H, int W, const float* mapx,
                         const float* mapy, int resized_h, int resized_w, int crop_y, int crop_x,
                         int out_h, int out_w, const float* mean3, const float* std3, void* out_nhwc,
                         int out_channels_padded, int out_dtype, float* out_nchw_or_null, void* stream);

/* ------------------------------------------------------------------------
 * SURVEY 8f-4 (training step), optimizer half: the reference's `optimizer_config = dict(grad_clip=dict(max_norm=100,
 * norm_type=2))` and `optimizer = dict(type='AdamW', lr=1e-4, weight_decay=1e-7)` (configs/thinktwice.py:282-287:
 * torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW through mmcv's OptimizerHook) over ONE flat f32 parameter /
 * gradient buffer.  No host synchronisation: the clip coefficient stays on the device.
 * ---------------------------------------------------------------------- */
/* out_norm_scale[0] = ||grad||_2, [1] = min(1, max_norm / (norm + 1e-6)); workspace_1024: 1024 floats */
int tt_grad_norm_clip(const float* grad, long long n, float max_norm, float* workspace_1024,
                      float* out_norm_scale, void* stream);
/* one AdamW update of `n` parameters in place (p, exp_avg, exp_avg_sq); `step` >= 1 is the 1-based step count of
 * the bias correction; the gradient is multiplied by *grad_scale_or_null (device scalar, e.g. out_norm_scale + 1) */
int tt_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step,
                  const float* grad_scale_or_null, void* stream);
/* The same update with the step count ON THE DEVICE: the bias corrections use *good_steps_dev + 1; tt_adamw_advance, issued once
 * after the launches of one optimizer step, increments the count unless the clip factor is NaN (non-finite gradient norm: the
 * update was skipped, and the count stays with the moments whatever the host does meanwhile). */
int tt_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                      float beta1, float beta2, float eps, float weight_decay, const int* good_steps_dev,
                      const float* grad_scale_or_null, void* stream);
int tt_adamw_advance(int* good_steps_dev, const float* grad_scale_or_null, void* stream);

/* ----------------------------------------------------------------------
 * SURVEY 8f-4 / row A24, loss half of the training forward: device reductions for every term of
 * ThinkTwiceDecoder.loss (thinktwice_decoder.py:536-619), the focal segmentation loss (utils.py:31-47 at
 * encoder_decoder_framework.py:172-176) and the depth BCE (encoder_decoder_framework.py:179-190, 441-481).
 * One launch per term, f64 accumulation, workgroup partials combined in index order by the last workgroup
 * (bit-reproducible).  `workspace`: tt_loss_workspace_bytes() bytes, ZEROED once by the caller, reusable by
 * consecutive calls on one stream.  Outputs are device floats.
 * ---------------------------------------------------------------------- */
long long tt_loss_workspace_bytes(void);
/* pred [n_outer][repeat][inner] against target [n_outer][inner] (broadcast over `repeat`; NULL = zeros),
 * F.smooth_l1_loss(beta=1).  reduce != 0: out[0] = scale * mean(min(l, clamp_max)) (clamp_max <= 0: no clamp,
 * torch.clamp(..., -5, 5) of DEC:51-52 is clamp_max = 5); reduce == 0: out[i] = scale * l_i (reduction='none'). */
int tt_loss_smooth_l1(const float* pred, const float* target, long long n_outer, int repeat, long long inner,
                      float clamp_max, float scale, int reduce, float* out, void* workspace, void* stream);
/* gradients of the two mean-reduced decoder terms w.r.t. the prediction, dense in the prediction's layout (written, not
 * accumulated): dpred = d[scale * mean(...)]/dpred.  tt_loss_smooth_l1_bwd: zero where the clamp is active. */
int tt_loss_smooth_l1_bwd(const float* pred, const float* target, long long n_outer, int repeat, long long inner,
                          float clamp_max, float scale, float* dpred, void* stream);
int tt_loss_beta_kl_bwd(const float* target_alpha, const float* target_beta, const float* pred_alpha,
                        const float* pred_beta, long long n_outer, int repeat, long long inner, float scale,
                        float* dpred_alpha, float* dpred_beta, void* stream);
/* out[0] = scale * mean KL(Beta(target_alpha, target_beta) || Beta(pred_alpha, pred_beta)), target [n_outer][inner]
 * broadcast over pred [n_outer][repeat][inner] (torch.distributions.kl_divergence, DEC:553-556, 571-573) */
int tt_loss_beta_kl(const float* target_alpha, const float* target_beta, const float* pred_alpha,
                    const float* pred_beta, long long n_outer, int repeat, long long inner, float scale, float* out,
                    void* workspace, void* stream);
/* out[c] = mean_r |pred[r][c] - target[r][c]|, cols <= 8, pred rows pred_row_stride floats apart (DEC:544-551).
 * With pred_beta / target_beta non-NULL the operands are Beta parameters and their modes mapped to [-1, 1]
 * (_get_action_beta, DEC:622-637) are compared. */
int tt_loss_l1_cols(const float* pred, const float* pred_beta, long long pred_row_stride, const float* target,
                    const float* target_beta, long long rows, int cols, float* out, void* workspace, void* stream);
/* logits_cl: channel-last [BN][H/factor][W/factor][row_stride >= num_classes]; labels [BN][H][W] float class ids
 * (255 = ignore) sampled at (y*factor, x*factor); out[0] = 10 * focal(alpha .5, gamma 2) of the mean cross entropy;
 * out[1], out[2] = that mean cross entropy and the number of contributing pixels (`aux` of the backward): out holds 3 floats.
 * tt_loss_seg_focal_bwd: dlogits_cl = upstream (device scalar, NULL = 1) * d out[0] / d logits, padding channels 0. */
int tt_loss_seg_focal(const float* logits_cl, int row_stride, int num_classes, const float* labels, int BN, int H,
                      int W, int factor, float* out, void* workspace, void* stream);
int tt_loss_seg_focal_bwd(const float* logits_cl, int row_stride, int num_classes, const float* labels, int BN, int H,
                          int W, int factor, const float* aux, const float* upstream_or_null, float* dlogits_cl,
                          void* stream);
/* logits_cl: channel-last [BN][H/factor][W/factor][row_stride >= D]; gt_depth [BN][H][W] metres (0 = no return);
 * bins of d_step from d_lo; out[0] = sum of BCE-with-logits over the foreground cells / max(1, #foreground) */
int tt_loss_depth_bce(const float* logits_cl, int row_stride, int D, const float* gt_depth, int BN, int H, int W,
                      int factor, float d_lo, float d_step, float* out, void* workspace, void* stream);
/* out of tt_loss_depth_bce holds 2 floats (loss, divisor = `aux` here); dlogits_cl = upstream * d loss / d logits */
int tt_loss_depth_bce_bwd(const float* logits_cl, int row_stride, int D, const float* gt_depth, int BN, int H, int W,
                          int factor, float d_lo, float d_step, const float* aux, const float* upstream_or_null,
                          float* dlogits_cl, void* stream);

/* ----------------------------------------------------------------------
 * SURVEY 8f-4, backward of the convolution (what autograd + cuDNN do under loss.backward() in the reference,
 * apis/mmdet_train.py:72-79).  Weight gradient of tt_conv2d_fwd's channel-last convolution:
 *   dw[co][kh][kw][ci] (+)= sum_{n,oh,ow} dy[n][oh][ow][dy_coff + co] * x[n][oh*s - p + kh*d][ow*s - p + kw*d][x_coff + ci]
 * exact f32 products on the f32 MFMA, deterministic (split partial sums added in a fixed order).  dw has the weight
 * layout [Cout][KH][KW][cin_pad] (padding channels are written as 0).  The input gradient needs no entry of its own:
 * it is tt_conv2d_fwd of dy with the 180-degree-rotated, Cin/Cout-transposed weights (thinktwice_amd/ops.py::conv2d_dgrad).
 * ---------------------------------------------------------------------- */
long long tt_conv2d_wgrad_workspace_bytes(int N, int OH, int Cout, int Cin, int cin_pad, int KH, int KW);
int tt_conv2d_wgrad(const float* x, int N, int H, int W, int Cin, int x_cstride, int x_coff, const float* dy, int OH,
                    int OW, int Cout, int dy_cstride, int dy_coff, int KH, int KW, int stride, int pad, int dil,
                    int cin_pad, int accumulate, float* dw, void* workspace, long long workspace_bytes, void* stream);
/* Same contract in the forward's bf16x3 arithmetic (dy and x split into bf16 hi + lo, three MFMAs per product, f32
 * accumulation: ~2^-17 relative per product): layers with >= 128 channels on both sides run on an LDS-staged
 * v_mfma_f32_32x32x16_bf16 kernel (the contraction index -- the pixel -- is the slow index of both operands: tiles of 32
 * pixels are staged as they lie in memory and read column-wise); everything else falls through to tt_conv2d_wgrad's kernels.
 * Same workspace query, same determinism. */
int tt_conv2d_wgrad_x3(const float* x, int N, int H, int W, int Cin, int x_cstride, int x_coff, const float* dy, int OH,
                    int OW, int Cout, int dy_cstride, int dy_coff, int KH, int KW, int stride, int pad, int dil,
                    int cin_pad, int accumulate, float* dw, void* workspace, long long workspace_bytes, void* stream);
/* Backward of tt_conv2d_fwd's fused epilogue  y = act(scale[c]*conv + shift[c] + res1 + res2)  from dy and the saved
 * output y (all [M][*] channel-last f32 with channel stride / offset):  g = dy * act'(.)  (TT_ACT_NONE / RELU / SIGMOID);
 * dconv = g * scale[c] (feeds tt_conv2d_wgrad and the dgrad convolution); dres / dres2 (optional): g added to
 * (dres_accumulate) or written over the gradient windows of the two residual inputs -- overwrite is for a layer whose
 * output was written in place over its residual; dshift[c] (+)= sum_m g, dscale[c] (+)= sum_m g * conv (skipped when
 * scale is NULL; needs res1 / res2 to still hold their forward values).  Folded BatchNorm: dgamma = (dscale -
 * mean*dshift)/sigma, dbeta = dshift.  Deterministic. */
long long tt_conv_epilogue_bwd_workspace_bytes(int C);
int tt_conv_epilogue_bwd(const float* dy, int dy_cstride, int dy_coff, const float* y, int y_cstride, int y_coff,
                         const float* res1, int res1_cstride, int res1_coff, const float* res2, int res2_cstride,
                         int res2_coff, const float* scale, const float* shift, long long M, int C, int act, float* dconv,
                         int dconv_cstride, int dconv_coff, float* dres, int dres_cstride, int dres_coff, float* dres2,
                         int dres2_cstride, int dres2_coff, int dres_accumulate, float* dscale, float* dshift,
                         int accumulate, const int* m_dev_or_null, const float* pre_or_null, const float* conv_raw_or_null,
                         void* workspace, long long workspace_bytes, void* stream);
/* (pre_or_null: dense [M][C] pre-activation scale*conv + shift + res1 + res2, recomputed by running the layer once more
 *  without its activation -- required for GELU / softplus, whose derivative cannot be read off the output.
 *  conv_raw_or_null: dense [M][C] raw convolution output, recomputed likewise without scale / shift: dscale = sum g * conv
 *  is then exact; without it conv is reconstructed as (pre - shift - res) / scale, which loses everything when a BatchNorm
 *  gamma is (near) zero -- zero_init_residual, pruned channels -- or a sigmoid saturates; a zero scale contributes 0) */
/* Sparse (rulebook) convolution backward.  Weight gradient dw[co][0][t][ci] (+)= sum_m dy[m][co] * x[nbr[m][t]][ci] over the
 * live rows m < *m_dev (f32 MFMA, deterministic).  The input gradient is the forward gathered GEMM itself on a transposed
 * rulebook: for a submanifold layer the rulebook is its own transpose under tap reversal; for a strided layer
 * tt_sp_inverse_rulebook builds inv[j][t] = m  <=>  nbr[m][t] = j  (inv pre-filled with -1).  tt_sp_from_dense is the
 * backward of tt_sp_to_dense (gathers the dense gradient back to the rows, accumulating). */
long long tt_gather_conv_wgrad_workspace_bytes(long long M, int Cout, int Cin, int cin_pad, int taps);
int tt_gather_conv_wgrad(const float* x, int x_cstride, int Cin, const int* nbr, const int* m_dev, long long M, int taps,
                         const float* dy, int dy_cstride, int Cout, int cin_pad, int accumulate, float* dw, void* workspace,
                         long long workspace_bytes, void* stream);
int tt_sp_inverse_rulebook(const int* nbr, const int* m_dev, long long M, int taps, int* inv_prefilled_minus1, void* stream);
int tt_sp_from_dense(const float* gdense, const int* coords, const int* num_rows, long long max_rows, int C, int D, int H,
                     int W, float* grows, void* stream);
/* backward of tt_maxpool3x3s2 (F.max_pool2d(x,3,2,1)): dx[n][ih][iw][c] += sum of dy over the <= 4 windows whose FIRST
 * maximum (scan order kh, kw, as torch) is this pixel; gather form, no atomics */
int tt_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* backward of tt_upsample_nearest_add w.r.t. src: dsrc[n][sy][sx][c] += sum of ddst over the dst pixels that read (sy, sx) */
int tt_upsample_nearest_add_bwd(const float* ddst, float* dsrc, int N, int H, int W, int C, int h, int w, void* stream);
/* backward of tt_bilinear_up2 (x2, align_corners=True): dx [N][H][W][C] += weights^T dy [N][2H][2W][C] */
int tt_bilinear_up2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* backward of tt_channel_gate (sigmoid gate): dx += g * s, dgate[n][c] += s(1-s) sum g*x, with g = dy, or -- when the
 * forward was out = relu(x*s + res), pass the saved out -- g = dy * [out > 0] and dres += g (optional) */
int tt_channel_gate_bwd(const float* x, const float* gate, const float* dy, float* dx, float* dgate, int N, int HW, int C,
                        const float* out_relu_or_null, float* dres_or_null, void* stream);
/* backward of tt_spatial_pool mode 1 (0.5 mean + 0.5 amax, ties share the amax gradient evenly); x dense [N][HW][C] */
int tt_spatial_meanmax_bwd(const float* x, const float* dpool, float* dx, int N, int HW, int C, void* stream);
/* backward of tt_spatial_pool mode 0 (mean): dx[n][p][coff + c] += dpool[n][c] / HW */
int tt_spatial_mean_bwd(const float* dpool, float* dx, int N, int HW, int C, int cstride, int coff, void* stream);
/* backward of tt_ew (same ops and row-strided channel windows): gv = dout * act'(saved out); da / db / dg (each optional)
 * are accumulated.  Activations: none, ReLU, sigmoid, softplus (from the saved output). */
int tt_ew_bwd(int op, int act, long long R, int C, const float* a, int a_stride, int a_coff, const float* b, int b_stride,
              int b_coff, const float* g, int g_stride, int g_coff, const float* out, int o_stride, int o_coff,
              const float* dout, int d_stride, int d_coff, float* da, int da_stride, int da_coff, float* db, int db_stride,
              int db_coff, float* dg, int dg_stride, int dg_coff, void* stream);
/* Backward of the look module's kernels (f32; scatters with atomics).  The waypoint / control inputs of a refinement layer
 * are detached in the reference (thinktwice_decoder.py:429-430), so there is no gradient for them or the reference points.
 * tt_look_gather_query_bwd: dout = gradient of the query rows; adds into the embeddings, the per-sample vectors and the four
 * FPN-side maps.  tt_msda_sample_bwd: adds into the value tensor's channel window, the offsets [R][512] and the attention
 * logits [R][256] (softmax included).  tt_sca_reduce_bwd: dx[bc][k] += dout[bc] / B for the slots the forward summed. */
int tt_look_gather_query_bwd(int B, const int* query_of_slot, const float* ref_packed, const float* dout, int row_stride,
                             float* dtemporal, float* dstatic, float* dmeas, float* dflat, float* const* dlevel_maps,
                             const int* level_hw, void* stream);
int tt_msda_sample_bwd(int B, const float* value, int value_cstride, int value_coff, const float* offsets,
                       const float* logits, const float* ref_packed, const int* level_hw, const float* dout, float* dvalue,
                       float* doffsets, float* dlogits, void* stream);
int tt_sca_reduce_bwd(int B, const float* dout, const int* max_len, float* dx, void* stream);
/* backward of tt_layernorm_rows: dx += , dgamma += , dbeta +=  (deterministic; workspace tt_layernorm_rows_bwd_workspace_bytes) */
long long tt_layernorm_rows_bwd_workspace_bytes(long long R, int D);
int tt_layernorm_rows_bwd(const float* x, const float* gamma, const float* dout, float* dx, float* dgamma, float* dbeta,
                          long long R, int D, int x_stride, int dout_stride, int dx_stride, float eps, void* workspace,
                          long long workspace_bytes, void* stream);
/* backward of one piece of tt_concat_rows: dsrc[(r / div) % mod or r / div][c] += dout[r][coff + c] summed over r */
int tt_concat_piece_bwd(const float* dout, int out_stride, int coff, long long R, int C, int div, int mod, float* dsrc,
                        int src_stride, int src_rows, void* stream);
/* backward of tt_broadcast_rows: dv[n][c] += sum_p dout[n][p][coff + c] */
int tt_broadcast_rows_bwd(const float* dout, float* dv, int N, int HW, int C, int cstride, int coff, int v_stride,
                          void* stream);
/* backward of tt_lift_splat_fwd (f32, rot_flip = 0): grad_out is the [B][Y][X][out_cstride] BEV gradient (channel window at
 * out_coff); grad_context [B*ncam][fH][fW][C] and grad_depth_logits [B*ncam][fH][fW][D] are accumulated (softmax included) */
int tt_lift_splat_bwd(int batch_size, int num_cams, int D, int fH, int fW, int C, int num_voxel_x, int num_voxel_y,
                      int num_voxel_z, const float* depth_logits, const float* context, const int32_t* geom_xyz,
                      const float* grad_out, int out_cstride, int out_coff, float* grad_depth_logits, float* grad_context,
                      void* stream);
/* backward of tt_deform_im2col3x3: gx += (f32 atomics), goffsets[pix][2 tap (+1)] += the sampling-position gradients */
int tt_deform_im2col3x3_bwd(const float* x, const float* offsets, const float* gcols, float* gx, float* goffsets, int N,
                            int H, int W, int C, int off_cstride, int pad, void* stream);

/* SURVEY 8f-1, LiDAR side: merge of the two 180-degree half sweeps of the closed-loop tick
 * (leaderboard/team_code/thinktwice_agent.py:340-352).  prev / now: (n, 4) f32 (x, y, z, intensity) device rows;
 * rel_transform_3x4: HOST pointer to the first three rows of inv(T_now) @ T_prev (row-major); out: (n_prev + n_now, 4)
 * = [transformed prev | now] with z += z_shift (2.5). */
int tt_lidar_merge_half_sweeps(const float* prev_xyzi, int n_prev, const float* now_xyzi, int n_now,
                               const float* rel_transform_3x4, float z_shift, float* out_xyzi, void* stream);

/* ------------------------------------------------------------------------
 * train-mode BatchNorm (batch statistics) over channel-last rows -- what every nn.BatchNorm2d / BatchNorm1d
 * (torch.nn.SyncBatchNorm under configs/thinktwice.py:39 SyncBN=True, apis/mmdet_train.py:86-87) computes under
 * model.train() (norm_eval=False, configs/thinktwice.py:146).  The convolution in front writes its raw output z (bias in);
 * `groups` equal row groups are normalised with their own statistics (the camera trunk's T sweeps run as one batch here,
 * one pass per sweep in the reference, lss.py:690-717); m_dev != NULL: sparse rows, live count on the device, groups == 1.
 *   stats  : double [groups][2C + 2] = per-channel sum | sum of squares | row count | 0 -- the block a SyncBN all-reduce
 *            (SUM) runs over, between tt_bn_stats and tt_bn_finalize
 *   scale (= gamma * invstd) / shift (= beta) / mean / invstd : float [groups][C]; running_mean / running_var (nullable) are updated like torch does
 *            (momentum, unbiased variance), group 0 (the key sweep) first, like lss.py:689-714
 *   tt_bn_apply      out[m][out_coff + c] = act((z - mean) * scale + shift + res1 + res2), act in {TT_ACT_NONE, TT_ACT_RELU}
 *            (centred like torch: folding the mean into the shift loses ulp(mean * scale) where the variance is ~0)
 *   tt_bn_bwd_reduce dy := dy * act'(y) in place, dres += it, sums = double [groups][2C] = sum g | sum g * xhat (LOCAL;
 *            dgamma / dbeta are these; under SyncBN they are all-reduced before tt_bn_bwd_apply)
 *   tt_bn_bwd_apply  dz = scale * (g - sum_g / n - xhat * sum_gxhat / n), n = stats[g][2C]
 * ---------------------------------------------------------------------- */
long long tt_bn_workspace_bytes(int C, int groups);
int tt_bn_stats(const float* z, long long M, int C, int z_cstride, int z_coff, const int* m_dev, int groups, double* stats,
                void* workspace, long long workspace_bytes, void* stream);
int tt_bn_finalize(const double* stats, int C, int groups, const float* gamma, const float* beta, float eps, float momentum,
                   float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* invstd,
                   void* stream);
int tt_bn_apply(const float* z, long long M, int C, int z_cstride, int z_coff, const int* m_dev, int groups,
                const float* scale, const float* shift, const float* mean, const float* res1, int r1_cstride, int r1_coff, const float* res2,
                int r2_cstride, int r2_coff, int act, float* out, int out_cstride, int out_coff, void* stream);
int tt_bn_bwd_reduce(float* dy, int dy_cstride, int dy_coff, const float* y, int y_cstride, int y_coff, const float* z,
                     int z_cstride, int z_coff, long long M, int C, const int* m_dev, int groups, const float* mean,
                     const float* invstd, int act, float* dres1, int d1_cstride, int d1_coff, float* dres2, int d2_cstride,
                     int d2_coff, double* sums, void* workspace, long long workspace_bytes, void* stream);
int tt_bn_bwd_apply(const float* g, int g_cstride, int g_coff, const float* z, int z_cstride, int z_coff, long long M, int C,
                    const int* m_dev, int groups, const double* sums, const double* stats, const float* scale,
                    const float* mean, const float* invstd, float* dz, int dz_cstride, int dz_coff, void* stream);
/* nn.Dropout(p) of the DepthNet ASPP (lss.py:91,110) in train mode: keep-mask from a counter-based generator (seed, element
 * index -> splitmix64), out = x * mask / (1 - p); `mask` (uint8 [n], nullable) receives the keep bits for the backward, or --
 * mask_in != NULL -- supplies them (parity tests replay the reference's torch mask). */
int tt_dropout_fwd(const float* x, float* out, long long n, float p, unsigned long long seed, const uint8_t* mask_in,
                   uint8_t* mask_out, void* stream);
int tt_dropout_bwd(const float* dout, const uint8_t* mask, float* dx, long long n, float p, void* stream);

/* ------------------------------------------------------------------------
 * action post-processing of a closed-loop tick, on the device (one thread) or on the host.
 * Replaces, in ONE call:
 *   EncoderDecoder.process_action / _get_action_beta  (code/encoder_decoder_framework.py:268-304): Beta mode of the last
 *       refinement stage's control branch -> (steer, throttle, brake)_ctrl;
 *   EncoderDecoder.control_pid + PIDController.step   (encoder_decoder_framework.py:309-390, code/utils.py:7-29): waypoint
 *       PID with its two n-entry error windows -> (steer, throttle, brake)_traj;
 *   the agent's arbitration + stuck detector          (leaderboard/team_code/thinktwice_agent.py:463-509)
 *       -> the final (steer, throttle, brake).
 * The reference copies mu / sigma / waypoints to the host and does this in numpy; here the model outputs are read where
 * they are and a tick needs exactly one small device->host copy (`out`).
 *   cfg    : the reference's `cfg` scalars (configs/thinktwice.py:44-57) + stuck_threshold (thinktwice_agent.py: 800).
 *   state  : caller-owned; zero-initialised == the reference's fresh deques ([0]*n) and stuck_detector = 0.
 *   mu_last / sigma_last : 2 floats each = pred['mu_branches'][0, -1, :], pred['sigma_branches'][0, -1, :]
 *   wp_last: 8 floats = pred['pred_wp'][0, -1, :, :] (4 waypoints x (x, y));  target_xy: 2 floats.
 *   stuck_desired_speed <= 0: unused (the reference's default -1).
 *   out    : TT_ACTION_OUT doubles, see the index names below.
 * tt_action_post: mu/sigma/wp/state/out are DEVICE pointers, target/speed by value; asynchronous on `stream`.
 * tt_action_post_host: every pointer is a HOST pointer; no GPU involved (same arithmetic, same source).
 * ---------------------------------------------------------------------- */
#define TT_PID_WINDOW_MAX 64
typedef struct tt_action_cfg {
    double turn_KP, turn_KI, turn_KD, speed_KP, speed_KI, speed_KD;
    double brake_speed, brake_ratio, clip_delta, aim_dist, angle_thresh, dist_thresh;
    int turn_n, speed_n, stuck_threshold, reserved;
} tt_action_cfg;
typedef struct tt_action_state {
    double turn_window[TT_PID_WINDOW_MAX], speed_window[TT_PID_WINDOW_MAX];
    int turn_head, speed_head;      /* ring position of the OLDEST entry (= where the next error is written) */
    int stuck_detector, reserved;
} tt_action_state;
enum {
    TT_ACT_STEER = 0, TT_ACT_THROTTLE, TT_ACT_BRAKE,                 /* the final control of the tick            */
    TT_ACT_STEER_CTRL, TT_ACT_THROTTLE_CTRL, TT_ACT_BRAKE_CTRL,      /* process_action                           */
    TT_ACT_STEER_TRAJ, TT_ACT_THROTTLE_TRAJ, TT_ACT_BRAKE_TRAJ,      /* control_pid (brake: 0 / 1)               */
    TT_ACT_DESIRED_SPEED, TT_ACT_ANGLE, TT_ACT_ANGLE_LAST, TT_ACT_ANGLE_TARGET, TT_ACT_ANGLE_FINAL, TT_ACT_DELTA,
    TT_ACT_AIM_X, TT_ACT_AIM_Y,                                      /* control_pid metadata                     */
    TT_ACT_IS_TURN, TT_ACT_IS_STUCK, TT_ACT_STUCK_DETECTOR,          /* arbitration                              */
    TT_ACTION_OUT = 24
};
int tt_action_post(const float* mu_last, const float* sigma_last, const float* wp_last, float speed, float target_x,
                   float target_y, float stuck_desired_speed, const tt_action_cfg* cfg_host, tt_action_state* state,
                   double* out, void* stream);
int tt_action_post_host(const float* mu_last, const float* sigma_last, const float* wp_last, float speed, float target_x,
                        float target_y, float stuck_desired_speed, const tt_action_cfg* cfg, tt_action_state* state,
                        double* out);
/* the three stages one by one on the host, for callers that keep the reference's call structure (process_action, then
 * control_pid, then the agent's own arbitration): each fills its slots of `out` (the rest is zeroed) */
int tt_action_ctrl_host(const float* mu_last, const float* sigma_last, double* out);
int tt_action_pid_host(const float* wp_last, float speed, float target_x, float target_y, float stuck_desired_speed,
                       const tt_action_cfg* cfg, tt_action_state* state, double* out);
/* the arbitration stage alone (thinktwice_agent.py:463-509), for a host that already holds the two heads' controls:
 * fills TT_ACT_STEER / THROTTLE / BRAKE / IS_TURN / IS_STUCK / STUCK_DETECTOR of `out`, updates state->stuck_detector */
int tt_action_arbitrate_host(double steer_ctrl, double throttle_ctrl, double brake_ctrl, double throttle_traj,
                             double brake_traj, float speed, const tt_action_cfg* cfg, tt_action_state* state, double* out);

/* ------------------------------------------------------------------------
 * launch plans -- the forward sequenced from C (csrc/plan.cpp).
 * The reference sequences the forward in Python (encoder_decoder_framework.py:194-250: extract_sensor_feat ->
 * get_fusion_feat -> ThinkTwiceDecoder.forward, thinktwice_decoder.py:419-489).  A plan is that sequence as data: the ordered
 * C-ABI calls of one forward, every pointer as (buffer id, byte offset) -- buffer 0 = packed weights, 1 = activation arena,
 * 2.. = inputs -- the stream slot of each call and the cross-stream dependencies.  thinktwice_amd/plan.py compiles a plan by
 * recording one forward of the Python mirror; this runtime binds a host's buffers and issues the calls on its streams.  Plans
 * (and the weights blob) are files: a non-Python host runs the path with tt_plan_load + tt_plan_bind + tt_encoder_fwd /
 * tt_decoder_fwd (tools/plan_host.cpp).  Argument kinds of tt_plan_add_call: 0 = integer (ivals), 1 = float / double (fvals),
 * 2 = device pointer (buffers[i], ivals[i] = byte offset), 3 = host blob (ivals[i] = offset returned by tt_plan_add_blob;
 * device pointers stored inside a blob are declared with tt_plan_add_reloc), 4 = NULL.
 * ---------------------------------------------------------------------- */
/* buffer plumbing as C-ABI entries, so that a plan holds NO torch kernel: zeros / constant fills (32-bit pattern) and
 * contiguous device-to-device copies */
int tt_fill_u32(void* dst, long long n_words, unsigned pattern, void* stream);
int tt_copy_bytes(void* dst, const void* src, long long nbytes, void* stream);
typedef struct tt_plan tt_plan;
tt_plan* tt_plan_create(void);
void tt_plan_destroy(tt_plan* plan);
int tt_plan_set_buffer(tt_plan* plan, int id, long long bytes, const char* name);
long long tt_plan_add_blob(tt_plan* plan, const void* bytes, long long n);
int tt_plan_add_reloc(tt_plan* plan, long long blob_offset, int buffer, long long offset);
int tt_plan_add_call(tt_plan* plan, const char* entry, int nargs, const int* kinds, const int* buffers, const long long* ivals,
                     const double* fvals, int stream_slot);
int tt_plan_add_sync(tt_plan* plan, int waiter_stream_slot, int signal_stream_slot);
/* Builder only.  Declare one allocation of the activation arena (address order, no overlap), then let the plan re-place the
 * allocations by liveness: tt_plan_compact_arena rewrites every pointer into buffer `arena_buffer` (call arguments, pointers
 * inside argument blobs, outputs) and returns the arena's new size in bytes (< 0 on error, plan unchanged).  Memory is handed
 * on only between allocations whose every use is on ONE stream (stream order makes the reuse safe); outputs keep their own.
 * (replaces the caching allocator torch gives the reference's eager forward, encoder_decoder_framework.py:194-235) */
int tt_plan_add_arena_alloc(tt_plan* plan, long long offset, long long nbytes);
long long tt_plan_compact_arena(tt_plan* plan, int arena_buffer, long long align);
/* outputs: where a host finds a result tensor -- byte offset of element 0 inside `buffer`, shape and ELEMENT strides (f32) */
int tt_plan_add_output(tt_plan* plan, const char* name, int buffer, long long offset, int ndim, const long long* shape,
                       const long long* stride);
int tt_plan_num_ops(const tt_plan* plan);
int tt_plan_num_calls(const tt_plan* plan);
int tt_plan_num_streams(const tt_plan* plan);
int tt_plan_num_buffers(const tt_plan* plan);
long long tt_plan_buffer_bytes(const tt_plan* plan, int id);
const char* tt_plan_buffer_name(const tt_plan* plan, int id);
int tt_plan_num_outputs(const tt_plan* plan);
int tt_plan_output(const tt_plan* plan, int i, const char** name, int* buffer, long long* offset, int* ndim, long long* shape8,
                   long long* stride8);
int tt_plan_save(const tt_plan* plan, const char* path);
tt_plan* tt_plan_load(const char* path);
/* bases[id]: device (or, for host-read arguments, host) address of buffer id, tt_plan_buffer_bytes(id) bytes each */
int tt_plan_bind(tt_plan* plan, void* const* bases, int nbases);
/* issue the whole plan / one half on the caller's streams (hipStream_t as void*), asynchronously */
int tt_plan_run(tt_plan* plan, void* const* streams, int nstreams);
int tt_plan_run_range(tt_plan* plan, void* const* streams, int nstreams, int first_op, int num_ops);
/* the encoder half (camera + LiDAR encoders, BEV fusion + flatten network: encoder_decoder_framework.py:194-235) and the
 * decoder half (ThinkTwiceDecoder.forward, thinktwice_decoder.py:419-489) of a bound forward plan */
int tt_encoder_fwd(tt_plan* plan, void* const* streams, int nstreams);
int tt_decoder_fwd(tt_plan* plan, void* const* streams, int nstreams);

#ifdef __cplusplus
}
#endif
#endif /* THINKTWICE_HIP_H */
