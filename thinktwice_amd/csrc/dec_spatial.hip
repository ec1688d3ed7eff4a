// Spatial half of the look-and-predict decoder as persistent per-sample kernels (one launch per stage instead of ~120):
//   tt_dec_gru         SpatialGRU, 4 steps x 8 convs on the 21x21x32 BEV state (dense_heads/utils.py:53-106,
//                      thinktwice_decoder.py:26-47 PredictionModule)
//   tt_dec_flatten     conv21_10 -> MLP10 -> conv10_4 -> MLP4 -> conv4_2 -> MLP2 -> output_fc, the grid2feat network
//                      (encoder_decoder_framework.py:228-234, thinktwice_decoder.py:405-415; SEBasicBlock code/utils.py:84-121)
//   tt_dec_bev_update  BEV_feat_update_module on cat([bev, h broadcast]) + residual (thinktwice_decoder.py:221-225,257)
//
// One workgroup owns one sample (one map): the maps live in LDS for the whole kernel, the convolutions are implicit
// GEMMs on the bf16 MFMA in "bf16x3" arithmetic (see dec_chain.hip), separated by workgroup barriers only.
//   * LDS maps are stored in PAIR FORMAT: per pixel, per 16 channels, 64 B = [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15]
//     (bf16), i.e. a wave's A fragment for (pixel = lane & 31, channel half = lane >> 5) is two ds_read_b128 with no
//     conversion; a value is split once, when the producing epilogue stores it.  Pixel stride = Cp*4 + 16 B, an odd
//     number of 16 B slots, so the 32 consecutive pixels of a row block hit distinct LDS slots.
//   * Weights stream from L2 straight into the B operand registers (a ring of PF K-steps in flight), stored
//     FRAGMENT-MAJOR on the host so that each load instruction covers one contiguous KiB (lane-scattered 16 B pieces
//     of 64 different rows ran the address coalescer at ~0.4 lines/clock: 20 us per 21x21 conv instead of ~3).
//   * Spatially CONSTANT input channels (the 6 GRU input channels, the 2048 broadcast `h` channels of the BEV update)
//     never enter the GEMM: their contribution is sum over the VALID taps of G[tap][n] = W[n, tap, const] . x, which
//     depends on the pixel only through its border class (3 row classes x 3 column classes); the 9 class sums are
//     tabulated once per conv and added in the epilogue.  The BEV update's K drops from 18,720 to 288 that way.
#include "conv_common.h"

namespace tt {

// the three products of a K step go to TWO accumulators (summed at the end): a back-to-back MFMA pair on the same
// accumulator waits for the first one's last pass, and these kernels have a single output block per wave
__device__ __forceinline__ void mfma3s(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x16& c,
                                       f32x16& c2) {
    Mfma<uint16_t>::run(al, bh, c2);
    Mfma<uint16_t>::run(ah, bh, c);
    Mfma<uint16_t>::run(ah, bl, c2);
}

__device__ __forceinline__ void pair_store(unsigned char* map, int PS, int pix, int c, float v) {
    const uint16_t hi = f32_to_bf16(v);
    const uint16_t lo = f32_to_bf16(v - bf16_to_f32(hi));
    unsigned char* p = map + (size_t)pix * PS + (c >> 4) * 64 + ((c >> 3) & 1) * 16 + (c & 7) * 2;
    *reinterpret_cast<uint16_t*>(p) = hi;
    *reinterpret_cast<uint16_t*>(p + 32) = lo;
}

__device__ __forceinline__ float pair_load(const unsigned char* map, int PS, int pix, int c) {
    const unsigned char* p = map + (size_t)pix * PS + (c >> 4) * 64 + ((c >> 3) & 1) * 16 + (c & 7) * 2;
    return bf16_to_f32(*reinterpret_cast<const uint16_t*>(p)) + bf16_to_f32(*reinterpret_cast<const uint16_t*>(p + 32));
}

// four consecutive channels c .. c + 3 (c % 4 == 0) of one pixel: two 8-byte LDS stores
__device__ __forceinline__ void pair_store4(unsigned char* map, int PS, int pix, int c, const float4& v) {
    unsigned char* p = map + (size_t)pix * PS + (c >> 4) * 64 + ((c >> 3) & 1) * 16 + (c & 7) * 2;
    const uint16_t h0 = f32_to_bf16(v.x), h1 = f32_to_bf16(v.y), h2 = f32_to_bf16(v.z), h3 = f32_to_bf16(v.w);
    const uint16_t l0 = f32_to_bf16(v.x - bf16_to_f32(h0)), l1 = f32_to_bf16(v.y - bf16_to_f32(h1));
    const uint16_t l2 = f32_to_bf16(v.z - bf16_to_f32(h2)), l3 = f32_to_bf16(v.w - bf16_to_f32(h3));
    *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
    *reinterpret_cast<uint2*>(p + 32) = make_uint2((unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16));
}

// a 441 x 32 f32 map (16 B aligned) from global memory into a pair-format LDS map: every load of a thread issued before its
// first store (one element per step left each step waiting for its own L2 round trip)
template <int NT>
__device__ __forceinline__ void load_map32(unsigned char* map, const float* src, int tid) {
    constexpr int kN4 = 441 * 32 / 4, kPer = (kN4 + NT - 1) / NT, kBatch = 4;
    const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll 1
    for (int u0 = 0; u0 < kPer; u0 += kBatch) {
        float4 v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int e = tid + (u0 + u) * NT;
            v[u] = s4[e < kN4 ? e : 0];
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int e = tid + (u0 + u) * NT;
            if (e < kN4) pair_store4(map, 144, e >> 3, (e & 7) * 4, v[u]);
        }
    }
}

// Implicit-GEMM convolution of an LDS-resident pair-format map by the whole workgroup.
//   in_map: H x W pixels, Cp channels (multiple of 16), pixel stride PS; `zero`: >= 64 zeroed bytes (padding taps)
//   w: fragment-major pair-format weights (weights.py::split_pairs_frag) of the [N rounded up to 32][KH*KW*Cp]
//      matrix (K order tap-major, channel-minor): a wave's B operand is two contiguous, fully coalesced 1 KiB loads
//   work items = (32-row block, group of NBG 32-column blocks), dealt to the waves round-robin (item % nwaves)
//   epi(m, n, v, i): called for every accumulator element of a valid (m < OH*OW, n < N) output; i = register index
template <int NBG, int PF, typename Epi>
__device__ __forceinline__ void conv_lds(const unsigned char* in_map, int H, int W, int Cp, int PS,
                                         const unsigned char* zero, int stride, int pad, int KH, int KW,
                                         const unsigned char* w, int N, int OH, int OW, int wave, int nwaves, int lane,
                                         Epi epi) {
    // Launder the lane id: everything below (row / pixel / address arithmetic of 16 accumulator rows per conv) is
    // invariant across the calls of a kernel, and the compiler would otherwise hoist all of it out of the step loop
    // and keep ~100 registers of addresses alive (measured: 570 B/lane of spills at the 128-register budget).
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(lane));
#endif
    const int M = OH * OW, MB = (M + 31) >> 5, NB = (N + 31) >> 5, NG = (NB + NBG - 1) / NBG;
    const int gpt = Cp >> 4;                       // K steps (16 channels) per tap
    const int nsteps = KH * KW * gpt;
    const size_t blk_bytes = (size_t)nsteps * 2048;   // fragment-major weights: bytes per 32-row block
    const int r = lane & 31, h = lane >> 5;
    for (int item = wave; item < MB * NG; item += nwaves) {
        const int mb = item / NG, ng = item - mb * NG;
        const int m = mb * 32 + r;
        const bool m_ok = m < M;
        const int oh = m_ok ? m / OW : 0, ow = m_ok ? m - (m / OW) * OW : 0;
        const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
        f32x16 acc[NBG], acc2[NBG];
        const unsigned char* bp[NBG];
        bool live[NBG];
#pragma unroll
        for (int q = 0; q < NBG; ++q) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] = acc2[q][i] = 0.f;
            const int nb = ng * NBG + q;
            live[q] = nb < NB;
            bp[q] = w + (size_t)(live[q] ? nb : 0) * blk_bytes + (h * 32 + r) * 16;
        }
        // Weight ring: PF K-steps in flight.  Every load is UNCONDITIONAL (the step index is clamped to the last
        // step, dead blocks re-read block 0): with loads under branches the compiler cannot count what is outstanding
        // and drains the ring (vmcnt(0)) at every loop head.
        uint4 bh[PF][NBG], bl[PF][NBG];
        const int last = nsteps - 1;
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int kp = p < last ? p : last;
#pragma unroll
            for (int q = 0; q < NBG; ++q) {
                bh[p][q] = *reinterpret_cast<const uint4*>(bp[q] + (size_t)kp * 2048);
                bl[p][q] = *reinterpret_cast<const uint4*>(bp[q] + (size_t)kp * 2048 + 1024);
            }
        }
#pragma unroll 1
        for (int ks0 = 0; ks0 < nsteps; ks0 += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int ks = ks0 + p;
                if (ks < nsteps) {
                    const int tap = ks / gpt, g = ks - tap * gpt;
                    const int kh = tap / KW, kw = tap - kh * KW;
                    const int ih = ih0 + kh, iw = iw0 + kw;
                    const bool ok = m_ok && ih >= 0 && ih < H && iw >= 0 && iw < W;
                    const unsigned char* ap = ok ? in_map + (size_t)(ih * W + iw) * PS + g * 64 + h * 16 : zero + h * 16;
                    const uint4 ah = *reinterpret_cast<const uint4*>(ap);
                    const uint4 al = *reinterpret_cast<const uint4*>(ok ? ap + 32 : ap);
#pragma unroll
                    for (int q = 0; q < NBG; ++q)
                        if (live[q]) mfma3s(ah, al, bh[p][q], bl[p][q], acc[q], acc2[q]);
                }
                {
                    const int kn = ks + PF < last ? ks + PF : last;
#pragma unroll
                    for (int q = 0; q < NBG; ++q) {
                        bh[p][q] = *reinterpret_cast<const uint4*>(bp[q] + (size_t)kn * 2048);
                        bl[p][q] = *reinterpret_cast<const uint4*>(bp[q] + (size_t)kn * 2048 + 1024);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NBG; ++q) {
            if (!live[q]) continue;
            const int n = (ng * NBG + q) * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int mm = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (mm < M && n < N) epi(mm, n, acc[q][i] + acc2[q][i], i);
            }
        }
    }
}

// The 3 x 3 / pad 1 / stride 1 convolution of a 21 x 21 x 32 LDS map to 32 channels (every GRU conv, the BEV update) with all
// shape arithmetic at compile time.  In the generic conv_lds* loops the tap decomposition and the window tests cost ~40 VALU
// instructions per (row block, K step) beside 3 MFMAs -- at 4 cycles per wave64 VALU instruction that is ~1.7x the MFMA time, so
// those loops are VALU-bound (9 us per conv against 2.9 us of MFMA work).  Here: the nine window tests of a lane are lane masks
// computed once per conv, the tap offset is an immediate of the ds_read, and a K step costs one select per row block.
// Same products in the same order per accumulator as conv_lds (bit-identical results).
template <int MBG, typename Epi>
__device__ __forceinline__ void conv3x3_map32(const unsigned char* in_map, const unsigned char* zero, const unsigned char* w,
                                              int wave, int nwaves, int lane, Epi epi) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(lane));
#endif
    constexpr int HW = 21, M = 441, MB = 14, PS = 144, PF = 6;         // PF = the six K steps (3 taps x 2 channel groups) of a kernel row
    static_assert(M <= MB * 32, "row blocks");
    constexpr int MG = (MB + MBG - 1) / MBG;
    const int r = lane & 31, h = lane >> 5;
    const unsigned char* zl = zero + h * 16;
    for (int mg = wave; mg < MG; mg += nwaves) {
        const unsigned char* base[MBG];          // pixel (oh - 1, ow - 1) of the lane's output pixel (+ its 16 B half)
        bool rowok[MBG][3], colok[MBG][3];
        f32x16 acc[MBG], acc2[MBG];
#pragma unroll
        for (int q = 0; q < MBG; ++q) {
            const int m = (mg * MBG + q) * 32 + r;
            const bool m_ok = m < M;
            const int oh = m_ok ? m / HW : 0, ow = m_ok ? m - (m / HW) * HW : 0;
            base[q] = in_map + ((oh - 1) * HW + (ow - 1)) * PS + h * 16;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                rowok[q][t] = m_ok && oh + t - 1 >= 0 && oh + t - 1 < HW;
                colok[q][t] = ow + t - 1 >= 0 && ow + t - 1 < HW;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] = acc2[q][i] = 0.f;
        }
        // weight ring: the six K steps of a kernel row in flight; the refill for row kh + 1 is issued behind the MFMAs of row kh
        // (a rolled loop over kh: in a fully unrolled body the compiler sinks every load to just before its use)
        const unsigned char* bp = w + (h * 32 + r) * 16;
        uint4 bh[PF], bl[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            bh[p] = *reinterpret_cast<const uint4*>(bp + (size_t)p * 2048);
            bl[p] = *reinterpret_cast<const uint4*>(bp + (size_t)p * 2048 + 1024);
        }
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
            const unsigned char* rowbase[MBG];
            bool rk[MBG];
#pragma unroll
            for (int q = 0; q < MBG; ++q) {
                rowbase[q] = base[q] + kh * (HW * PS);
                rk[q] = kh == 0 ? rowok[q][0] : (kh == 1 ? rowok[q][1] : rowok[q][2]);
            }
            const int knext = kh < 2 ? kh + 1 : 2;                     // (the last row re-reads itself: loads stay unconditional)
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                constexpr int kGpt = 2;
                const int kw = p / kGpt, g = p % kGpt;
                const int toff = kw * PS + g * 64;                     // compile-time
                uint4 ah[MBG], al[MBG];
#pragma unroll
                for (int q = 0; q < MBG; ++q) {
                    const unsigned char* sel = (rk[q] && colok[q][kw]) ? rowbase[q] : zl - toff;
                    ah[q] = *reinterpret_cast<const uint4*>(sel + toff);
                    al[q] = *reinterpret_cast<const uint4*>(sel + toff + 32);
                }
#pragma unroll
                for (int q = 0; q < MBG; ++q) Mfma<uint16_t>::run(al[q], bh[p], acc2[q]);
#pragma unroll
                for (int q = 0; q < MBG; ++q) Mfma<uint16_t>::run(ah[q], bh[p], acc[q]);
#pragma unroll
                for (int q = 0; q < MBG; ++q) Mfma<uint16_t>::run(ah[q], bl[p], acc2[q]);
                bh[p] = *reinterpret_cast<const uint4*>(bp + (size_t)(knext * PF + p) * 2048);
                bl[p] = *reinterpret_cast<const uint4*>(bp + (size_t)(knext * PF + p) * 2048 + 1024);
            }
        }
        // epi(mg, q, i, v): accumulator element i of row block mg * MBG + q -- output pixel (mg * MBG + q) * 32 + (i & 3) +
        // 8 * (i >> 2) + 4 * (lane >> 5), channel lane & 31.  Called for EVERY element, also the rows beyond pixel 440 of the last
        // block: the caller decides what to guard (its LDS maps are padded to 448 pixels so that it need not)
#pragma unroll
        for (int q = 0; q < MBG; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) epi(mg, q, i, acc[q][i] + acc2[q][i]);
    }
}

// border class of pixel (y, x) of an H x W map for a 3x3 / pad 1 conv: 3 * row class + column class
__device__ __forceinline__ int border_class(int y, int x, int H, int W) {
    const int rc = y == 0 ? 0 : (y == H - 1 ? 2 : 1);
    const int cc = x == 0 ? 0 : (x == W - 1 ? 2 : 1);
    return rc * 3 + cc;
}
// is tap (kh, kw) of a 3x3 / pad 1 conv inside the map for a pixel of border class cls?
__device__ __forceinline__ bool tap_valid(int cls, int kh, int kw) {
    const int rc = cls / 3, cc = cls - rc * 3;
    return !((rc == 0 && kh == 0) || (rc == 2 && kh == 2) || (cc == 0 && kw == 0) || (cc == 2 && kw == 2));
}

// sigmoid on the hardware exp2 / reciprocal (2 + 1 instructions instead of the ~35 of expf and an IEEE division -- the two gate
// epilogues of the GRU were 1100-1400 VALU instructions per wave): relative error ~1e-6, below the bf16x3 products feeding it
__device__ __forceinline__ float sigmoid_fast(float v) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v));
}

// ------------------------------------------------------------------------------------------------ conv-GRU
constexpr int kGruWaves = 7;             // two 32-pixel row blocks of the 441-pixel map per wave (conv_lds_mb<2>)
constexpr int kBevWaves = 7;             // BEV update: two row blocks per wave too (its second conv accumulates in the wave's registers)
constexpr int kMapHW = 21, kMapPix = 441, kMapC = 32;
constexpr int kMapPixPad = 448;        // 14 row blocks of 32
constexpr int kPS32 = kMapC * 4 + 16;    // 144 B per pixel

struct GruArgs {
    const float* inp6;        // [B][4][6]   (waypoint xy, softplus(ctrl) x4) per future step
    const float* state;       // [B][441][32] f32 channel-last BEV state
    float* fut;               // [B][4][441][32] f32
    float* scratch;           // [B][3][448][32] f32: update gate (role 1 -> role 0), previous state of the running step (written
                              // and re-read by the SAME lane), new state (role 0 -> role 1); then [B][2] flag words
    const unsigned char* w0[3];   // conv_update.0 / conv_reset.0 / conv_state_tilde.0: state part, pair [32][9*32]
    const float* wx[3];           // ... constant-input part, f32 [9][6][32]
    const float* b0[3];
    const unsigned char* w2[3];   // .2 convs, pair [32][9*32]
    const float* b2[3];
    const unsigned char* wd0; const float* bd0;   // conv_decoder.0 / .2
    const unsigned char* wd2; const float* bd2;
    unsigned* flags;          // [B][2], zero at launch: [0] states published by role 0, [1] update gates published by role 1
    int* fault;               // the device's host-mapped fault word (dec_chain.hip): set if a wait gave up
    int max_spin;
    long long* trace;         // debug (tt_dec_set_trace): wall-clock stamps of workgroup 0 after every phase, or null
};

// Flag hand-over between the two workgroups of a sample (agent scope: they may sit on different XCDs, whose L2s are not
// coherent for plain stores).  post: every thread's stores of the phase, then thread 0 releases and bumps the flag.  wait:
// thread 0 polls (bounded: a partner that is not running within max_spin polls makes the wait give up LOUDLY -- the device's
// fault word is set and the caller poisons everything it writes from then on, as in tt_mlp_chain_wide), then acquires.
__device__ __forceinline__ void flag_post(unsigned* f, unsigned value) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(f, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool flag_wait(const unsigned* f, unsigned target, int* fault, int max_spin, int* gave_up) {
    if (threadIdx.x == 0) {
        int spin = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spin > max_spin) {
                __hip_atomic_store(fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                *gave_up = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return *gave_up != 0;
}

// TWO workgroups per sample.  A step of the GRU is eight 441 x 32 x 288 convolutions, MFMA-bound on the one CU that ran them all
// (~50 us a step on the batch-1 tick's critical path), but only four of them are on the recurrence:
//   role 0 (blockIdx.x = 0): reset gate (conv_reset.0 / .2), candidate (conv_state_tilde.0 / .2 on (1 - r) * S), the blend
//                            S' = (1 - u) * S + u * cand with the update gate it gets from role 1; publishes S';
//   role 1:                  update gate (conv_update.0 / .2) of the state role 0 published, then -- off the recurrence -- the
//                            decoder convs (conv_decoder.0 / .2) of that same state = the previous step's output map.
// Every convolution is the one the single workgroup ran, on the same LDS image (pair(S) rebuilt from the published f32 state =
// the pair the blend stored): results are bit-identical to it.
__global__ __launch_bounds__(kGruWaves * 64) void dec_gru_kernel(const GruArgs a) {
    // LDS: two maps padded to 448 pixels (the epilogues store the 7 rows beyond pixel 440 of the last row block unguarded), the
    // zero page, the class sums [3 convs][16 class slots][32] (slot 15: rows beyond the map).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Smap = smem;                                   // state (or (1-r)*state) map
    unsigned char* Hmap = smem + kMapPixPad * kPS32;              // hidden map of the current two-conv block
    unsigned char* zero = Hmap + kMapPixPad * kPS32;              // 64 zero bytes
    float* Gc = reinterpret_cast<float*>(zero + 64);
    __shared__ int gave_up;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = blockIdx.x, b = blockIdx.y;
    int tri = 0;
    auto stamp = [&]() {
        if (a.trace && b == 0 && role == 0 && tid == 0 && tri < 64) a.trace[tri++] = (long long)wall_clock64();
    };
    stamp();
    if (a.trace && b == 0 && role == 0 && tid == 0) a.trace[62] = (long long)clock64();      // shader-clock ticks (vs the 100 MHz stamps)
    if (tid == 0) gave_up = 0;
    if (tid < 16) reinterpret_cast<uint32_t*>(zero)[tid] = 0u;
    load_map32<kGruWaves * 64>(Smap, a.state + (size_t)b * kMapPix * kMapC, tid);
    for (int e = tid; e < 3 * 16 * 32; e += kGruWaves * 64)
        if (((e >> 5) & 15) >= 9) Gc[e] = 0.f;      // slot 15 (rows beyond the map) and the unused slots: finite; 0-8 are written per step
    // The epilogues were the larger half of a conv (30-44 VALU instructions per accumulator element at 4 cycles each: pixel ->
    // (y, x) division, border class, guards, addresses).  A lane's 32 output pixels are the same in all 32 convs, so their border
    // classes are packed once (4 bits each, 15 = beyond the map) and every address below is `lane base + immediate`.
    const int r = lane & 31, h = lane >> 5;
    const int m_lane = wave * 64 + 4 * h;                         // + q * 32 + (i & 3) + 8 * (i >> 2)
    unsigned long long clsp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        clsp[q] = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m_lane + q * 32 + (i & 3) + 8 * (i >> 2);
            const int y = m / kMapHW, x = m - y * kMapHW;
            const unsigned long long c = m < kMapPix ? (unsigned long long)border_class(y, x, kMapHW, kMapHW) : 15ull;
            clsp[q] |= c << (4 * i);
        }
    }
    auto cls_of = [&](int q, int i) { return (int)((clsp[q] >> (4 * i)) & 15ull); };
    // `ml`: m_lane laundered before every conv -- the per-element addresses are invariant across the convs and the time steps, and
    // the compiler would otherwise hoist all of them (32 elements x 5 arrays) out of the loops and spill
    int ml = m_lane;
    auto fresh_rows = [&]() {
        ml = m_lane;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(ml));
#endif
    };
    auto row_of = [&](int q, int i) { return ml + q * 32 + (i & 3) + 8 * (i >> 2); };
    stamp();
    float* ug = a.scratch + (size_t)b * 3 * kMapPixPad * kMapC;   // padded like the maps: written / read unguarded
    float* sg = ug + kMapPixPad * kMapC;
    float* sn = sg + kMapPixPad * kMapC;
    unsigned* f_state = a.flags + 2 * b;
    unsigned* f_gate = f_state + 1;
    bool poisoned = false;
    const float kNaN = __builtin_nanf("");
    // class sums of the constant-input contribution of the first convs cv0 .. cv1 (+ their bias) for step t
    auto class_sums = [&](int t, int cv0, int cv1) {
        const float* x = a.inp6 + ((size_t)b * 4 + t) * 6;
        for (int ge = cv0 * 288 + tid; ge < (cv1 + 1) * 288; ge += kGruWaves * 64) {        // entry = (conv, class, n)
            const int cv = ge / 288, e = ge - cv * 288, cls = e >> 5, n = e & 31;
            const float* wxc = cv == 0 ? a.wx[0] : (cv == 1 ? a.wx[1] : a.wx[2]);
            const float* b0c = cv == 0 ? a.b0[0] : (cv == 1 ? a.b0[1] : a.b0[2]);
            float xv[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) xv[j] = x[j];
            float s = b0c[n];
#pragma unroll 3
            for (int tap = 0; tap < 9; ++tap) {
                float g = 0.f;
#pragma unroll
                for (int j = 0; j < 6; ++j) g += wxc[(tap * 6 + j) * 32 + n] * xv[j];      // 18 independent loads in flight
                if (tap_valid(cls, tap / 3, tap % 3)) s += g;
            }
            Gc[(cv * 16 + cls) * 32 + n] = s;
        }
    };
    auto first_conv = [&](int cv) {    // H = relu(conv_cv.0([x, S]))
        fresh_rows();
        conv3x3_map32<2>(Smap, zero, a.w0[cv], wave, kGruWaves, lane, [&](int, int q, int i, float v) {
            v += Gc[(cv * 16 + cls_of(q, i)) * 32 + r];
            pair_store(Hmap, kPS32, row_of(q, i), r, v > 0.f ? v : 0.f);
        });
    };
    if (role == 0) {
        // ---------------------------------------------------------------- the recurrence
        const float b2r = a.b2[1][r], b2c = a.b2[2][r];
        for (int t = 0; t < 4; ++t) {
            class_sums(t, 1, 2);
            __syncthreads();
            stamp();
            first_conv(1);
            __syncthreads();
            stamp();
            fresh_rows();
            conv3x3_map32<2>(Hmap, zero, a.w2[1], wave, kGruWaves, lane, [&](int, int q, int i, float v) {
                const int m = row_of(q, i);
                const float rg = sigmoid_fast(v + b2r);
                const float s = pair_load(Smap, kPS32, m, r);
                sg[m * kMapC + r] = s;
                pair_store(Smap, kPS32, m, r, (1.f - rg) * s);        // own element only: no other reader now
            });
            __syncthreads();
            stamp();
            first_conv(2);
            __syncthreads();
            stamp();
            poisoned = flag_wait(f_gate, (unsigned)(t + 1), a.fault, a.max_spin, &gave_up) || poisoned;     // update gate of step t
            fresh_rows();
            conv3x3_map32<2>(Hmap, zero, a.w2[2], wave, kGruWaves, lane, [&](int, int q, int i, float v) {
                const int m = row_of(q, i);
                const float cand = v + b2c;
                const float u = ug[m * kMapC + r];
                float ns = (1.f - u) * sg[m * kMapC + r] + u * cand;
                if (poisoned) ns = kNaN;
                pair_store(Smap, kPS32, m, r, ns);
                sn[m * kMapC + r] = ns;
            });
            flag_post(f_state, (unsigned)(t + 1));
            stamp();
        }
    } else {
        // ---------------------------------------------------------------- update gate + decoder
        const float b2u = a.b2[0][r], bd0 = a.bd0[r], bd2 = a.bd2[r];
        for (int t = 0; t <= 4; ++t) {
            if (t < 4) class_sums(t, 0, 0);
            if (t > 0) {       // the state after step t - 1
                poisoned = flag_wait(f_state, (unsigned)t, a.fault, a.max_spin, &gave_up) || poisoned;
                load_map32<kGruWaves * 64>(Smap, sn, tid);
            }
            __syncthreads();
            if (t < 4) {
                first_conv(0);
                __syncthreads();
                fresh_rows();
                conv3x3_map32<2>(Hmap, zero, a.w2[0], wave, kGruWaves, lane, [&](int, int q, int i, float v) {
                    ug[row_of(q, i) * kMapC + r] = poisoned ? kNaN : sigmoid_fast(v + b2u);
                });
                flag_post(f_gate, (unsigned)(t + 1));
            }
            if (t > 0) {
                __syncthreads();
                fresh_rows();
                conv3x3_map32<2>(Smap, zero, a.wd0, wave, kGruWaves, lane, [&](int, int q, int i, float v) {
                    v += bd0;
                    pair_store(Hmap, kPS32, row_of(q, i), r, v > 0.f ? v : 0.f);
                });
                __syncthreads();
                float* fo = a.fut + (((size_t)b * 4 + (t - 1)) * kMapPix) * kMapC;
                fresh_rows();
                conv3x3_map32<2>(Hmap, zero, a.wd2, wave, kGruWaves, lane, [&](int, int q, int i, float v) {
                    if (cls_of(q, i) != 15) fo[(size_t)row_of(q, i) * kMapC + r] = poisoned ? kNaN : v + bd2;      // exactly 441 rows
                });
                __syncthreads();
            }
        }
    }
    if (a.trace && b == 0 && role == 0 && tid == 0) a.trace[63] = (long long)clock64();
}

// ------------------------------------------------------------------------------------------------ grid2feat
// tt_dec_flatten = dec_flatten_kernel (one workgroup per 21x21x32 map: conv21_10, MLP10, conv10_4) + eleven launches of
// dec_tail_conv_kernel (everything from the 4x4 level on, rows of all maps packed, columns dealt over workgroups).  Weight sets (pair format, eval BatchNorm scale folded into the rows, shift = bias):
//   0 conv21_10 | 1 2 MLP10.conv1/2 | 3 4 MLP10.se.fc1/2 | 5 conv10_4 | 6 7 MLP4.conv1/2 | 8 9 MLP4.se.fc1/2 |
//   10 conv4_2 | 11 12 MLP2.conv1/2 | 13 14 MLP2.se.fc1/2 | 15 output_fc.0 (as a 2x2 conv over the 2x2x256 map) |
//   16 output_fc.3
constexpr int kFlatWaves = 8;
constexpr int kFlatSets = 17;

struct FlatArgs {
    const float* in;          // [maps][441][32] f32 channel-last
    float* out;               // [maps][256] f32
    float* mids;              // optional [maps][(100*64 + 16*128 + 4*256)] f32: the 10x10, 4x4, 2x2 block outputs
    float* x4;                // [maps][16][128] f32: conv10_4's output, where the kernel hands over to the column-split tail
    const unsigned char* w[kFlatSets];
    const float* b[kFlatSets];
    const float* bn_scale;    // output_fc.2 (eval BatchNorm1d) scale / shift [512]
    const float* bn_shift;
    long long* trace;         // debug (tt_dec_set_trace), as in GruArgs
};

__device__ __forceinline__ int ps_of(int C) { return C * 4 + 16; }

// SEBasicBlock (code/utils.py:99-121) on an LDS map, IN PLACE: X (C channels; input, residual and output) ->
// Y1 = relu(bn1(conv1 X)) (2C channels) -> Y2 = relu(bn2(conv2 Y1)) (C) -> gate = sigmoid(fc2 relu(fc1 pool(Y2))) ->
// X = relu(Y2 * gate + X).  `vec`: scratch for two 1-pixel maps of C channels + C floats of gate.
template <typename Stamp>
__device__ __forceinline__ void se_block(const FlatArgs& a, int set, int HW, int C, unsigned char* X, unsigned char* Y1,
                                         unsigned char* Y2, unsigned char* vec, const unsigned char* zero, int wave,
                                         int lane, int tid, Stamp stamp) {
    const int PS = ps_of(C), PS1 = ps_of(2 * C), M = HW * HW;
    conv_lds<1, 4>(X, HW, HW, C, PS, zero, 1, 1, 3, 3, a.w[set], 2 * C, HW, HW, wave, kFlatWaves, lane,
                   [&](int m, int n, float v, int) { v += a.b[set][n]; pair_store(Y1, PS1, m, n, v > 0.f ? v : 0.f); });
    __syncthreads();
    stamp();
    conv_lds<1, 4>(Y1, HW, HW, 2 * C, PS1, zero, 1, 1, 3, 3, a.w[set + 1], C, HW, HW, wave, kFlatWaves, lane,
                   [&](int m, int n, float v, int) { v += a.b[set + 1][n]; pair_store(Y2, PS, m, n, v > 0.f ? v : 0.f); });
    __syncthreads();
    stamp();
    unsigned char* s0 = vec;                       // pooled vector as a 1-pixel map
    unsigned char* s1 = vec + ps_of(C);            // fc1 output
    float* gate = reinterpret_cast<float*>(vec + 2 * ps_of(C));
    float* red = gate + C;                         // [threads / C][C][2] partial (sum, max)
    {
        // every thread takes a slice of the pixels of one channel (one thread per channel walked all M pixels alone: ~10 us)
        const int nparts = kFlatWaves * 64 / C, c = tid % C, part = tid / C;
        float sum = 0.f, mx = -INFINITY;
        for (int m = part; m < M; m += nparts) {
            const float v = pair_load(Y2, PS, m, c);
            sum += v;
            mx = fmaxf(mx, v);
        }
        red[(part * C + c) * 2] = sum;
        red[(part * C + c) * 2 + 1] = mx;
        __syncthreads();
        if (tid < C) {
            for (int q = 1; q < nparts; ++q) {
                sum += red[(q * C + tid) * 2];
                mx = fmaxf(mx, red[(q * C + tid) * 2 + 1]);
            }
            pair_store(s0, PS, 0, tid, 0.5f * (sum / (float)M) + 0.5f * mx);
        }
    }
    __syncthreads();
    stamp();
    conv_lds<1, 4>(s0, 1, 1, C, PS, zero, 1, 0, 1, 1, a.w[set + 2], C, 1, 1, wave, kFlatWaves, lane,
                   [&](int, int n, float v, int) { v += a.b[set + 2][n]; pair_store(s1, PS, 0, n, v > 0.f ? v : 0.f); });
    __syncthreads();
    conv_lds<1, 4>(s1, 1, 1, C, PS, zero, 1, 0, 1, 1, a.w[set + 3], C, 1, 1, wave, kFlatWaves, lane,
                   [&](int, int n, float v, int) { gate[n] = 1.f / (1.f + expf(-(v + a.b[set + 3][n]))); });
    __syncthreads();
    stamp();
    for (int e = tid; e < M * C; e += kFlatWaves * 64) {
        const int m = e / C, c = e - m * C;
        const float v = pair_load(Y2, PS, m, c) * gate[c] + pair_load(X, PS, m, c);
        pair_store(X, PS, m, c, v > 0.f ? v : 0.f);
    }
    __syncthreads();
}

__global__ __launch_bounds__(kFlatWaves * 64) void dec_flatten_kernel(const FlatArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int map = blockIdx.x;
    int tri = 0;
    auto stamp = [&]() {
        if (a.trace && map == 0 && tid == 0 && tri < 64) a.trace[tri++] = (long long)wall_clock64();
    };
    stamp();
    // LDS plan (bytes).  Region A [0, 63504): the input map; once conv21_10 has consumed it, MLP10's 128-channel hidden
    // map.  Region B: the 10x10x64 map X10 and MLP10's Y2.  Then zero page, vectors.
    constexpr int kIn = kMapPix * kPS32;                               // 63504
    constexpr int kOffB = (kIn + 15) / 16 * 16;
    constexpr int kSz10 = 100 * (64 * 4 + 16);                         // 27200
    constexpr int kZero = kOffB + 2 * kSz10;
    constexpr int kVec = kZero + 64;                                   // 2 x (64*4+16) + 64*4 + 8*64*2*4
    unsigned char* Rin = smem;
    unsigned char* X10 = smem + kOffB;
    unsigned char* Y10b = X10 + kSz10;
    unsigned char* Y10a = smem;                                        // 100 x (128*4+16) = 52800 B, in region A
    unsigned char* zero = smem + kZero;
    unsigned char* vec = smem + kVec;

    if (tid < 16) reinterpret_cast<uint32_t*>(zero)[tid] = 0u;
    const float* src = a.in + (size_t)map * kMapPix * kMapC;
    load_map32<kFlatWaves * 64>(Rin, src, tid);
    __syncthreads();
    stamp();
    // conv21_10: 3x3 stride 2, no padding, 32 -> 64, ReLU
    conv_lds<1, 4>(Rin, kMapHW, kMapHW, 32, kPS32, zero, 2, 0, 3, 3, a.w[0], 64, 10, 10, wave, kFlatWaves, lane,
                   [&](int m, int n, float v, int) { v += a.b[0][n]; pair_store(X10, ps_of(64), m, n, v > 0.f ? v : 0.f); });
    __syncthreads();
    stamp();
    se_block(a, 1, 10, 64, X10, Y10a, Y10b, vec, zero, wave, lane, tid, stamp);     // X10 updated in place
    stamp();
    float* mid = a.mids ? a.mids + (size_t)map * (100 * 64 + 16 * 128 + 4 * 256) : nullptr;
    if (mid)
        for (int e = tid; e < 100 * 64; e += kFlatWaves * 64) mid[e] = pair_load(X10, ps_of(64), e >> 6, e & 63);
    // conv10_4: 3x3 stride 2, no padding, 64 -> 128, ReLU -- straight to global memory: the 4x4 and 2x2 levels and the two
    // linears are weight streams (15.6 of the network's 16.8 MB) and run as the column-split stages below
    float* x4 = a.x4 + (size_t)map * 16 * 128;
    conv_lds<1, 4>(X10, 10, 10, 64, ps_of(64), zero, 2, 0, 3, 3, a.w[5], 128, 4, 4, wave, kFlatWaves, lane,
                   [&](int m, int n, float v, int) { v += a.b[5][n]; x4[m * 128 + n] = v > 0.f ? v : 0.f; });
    stamp();
}

// ------------------------------------------------------------------------------------------------ grid2feat, 4x4 level on
// One workgroup streamed all 16.8 MB of grid2feat's weights for its map at the ~64 GB/s a single CU gets out of the L2 (267 us
// at 4 maps, 340 us at 32: 15.6 MB of it for the 16-, 4- and 1-pixel levels).  From conv10_4's output on, the ROWS of all maps
// are packed into one GEMM per layer (rows = maps x output pixels: a 32-row block holds 2 / 8 / 32 maps) and the COLUMNS are
// dealt over workgroups, so a layer's weights are streamed once per 32 / 64 rows by N / 32 CUs at a time.  One launch per
// layer (the stream order is the barrier); a workgroup = one 32-column block x one group of maps:
//   * the group's input maps are staged in LDS in pair format (as in the kernel above).  Three staging modes: the stored
//     tensor; the SE block's pooled vector 0.5 mean + 0.5 max over the pixels (input of se.fc1); the SE block's output
//     relu(y * gate + x) (input of the layer after the block -- column block 0 also stores it, for `mids`);
//   * the eight waves split the K steps (wave w: steps w, w + 8, ...: a ring of 8 x 2 KiB of weights in flight per wave), each
//     against all row blocks of the group; the partial tiles are summed through LDS in wave order (deterministic).
constexpr int kTailWaves = 8, kTailPF = 8;

struct TailArgs {
    const float* in;          // mode 0: [maps][HW * HW][C]; modes 1, 2: the block's y [maps][P_in][C]
    const float* gate;        // mode 2: [maps][C]
    const float* res;         // mode 2: the block's input x [maps][P_in][C]
    float* stage_out;         // mode 2, optional: relu(y * gate + x), map stride stage_out_ms floats
    long long stage_out_ms;
    int mode, maps, MG;       // MG maps per workgroup (gridDim.y = ceil(maps / MG)); MG * OHW * OHW <= 32 * MBG
    int HW, C, KH, KW, pad, OHW, P_in;
    const unsigned char* w;   // fragment-major pair format, as in conv_lds
    const float* bias;
    const float* bn_scale;    // optional affine after the activation (output_fc.2)
    const float* bn_shift;
    int N, act;               // act: 0 ReLU, 1 sigmoid
    float* out;               // [maps * OHW * OHW][N]
};

template <int MBG>
__global__ __launch_bounds__(kTailWaves * 64) void dec_tail_conv_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = blockIdx.x, map0 = blockIdx.y * a.MG;
    const int nm = min(a.MG, a.maps - map0);
    const int C = a.C, PS = C * 4 + 16, HW2 = a.HW * a.HW;
    unsigned char* Xs = smem;
    unsigned char* zero = smem + (size_t)a.MG * HW2 * PS;
    float(*part)[MBG][32][32] = reinterpret_cast<float(*)[MBG][32][32]>(zero + 64);
    if (tid < 16) reinterpret_cast<uint32_t*>(zero)[tid] = 0u;
    // ---- stage the group's maps: four channels per thread and step, kStageUn independent loads in flight per thread (one
    // element per step left every step waiting for its own L2 round trip: 22 us for the 16 K elements of MLP4.conv2's input)
    constexpr int kStageUn = 4, kStep = kTailWaves * 64;
    const int C4 = C >> 2, per_map4 = HW2 * C4, ngr = nm * per_map4;
    auto put4 = [&](int ml, int q4, float4 v) {
        const int pix = q4 / C4, c = (q4 - pix * C4) * 4;
        unsigned char* p = Xs + (size_t)(ml * HW2 + pix) * PS + (c >> 4) * 64 + ((c >> 3) & 1) * 16 + (c & 7) * 2;
        const uint16_t h0 = f32_to_bf16(v.x), h1 = f32_to_bf16(v.y), h2 = f32_to_bf16(v.z), h3 = f32_to_bf16(v.w);
        const uint16_t l0 = f32_to_bf16(v.x - bf16_to_f32(h0)), l1 = f32_to_bf16(v.y - bf16_to_f32(h1));
        const uint16_t l2 = f32_to_bf16(v.z - bf16_to_f32(h2)), l3 = f32_to_bf16(v.w - bf16_to_f32(h3));
        *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
        *reinterpret_cast<uint2*>(p + 32) = make_uint2((unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16));
    };
    const float4* in4 = reinterpret_cast<const float4*>(a.in);
    for (int e0 = tid; e0 < ngr; e0 += kStep * kStageUn) {
        int ml[kStageUn], q4[kStageUn];
        float4 v[kStageUn];
#pragma unroll
        for (int u = 0; u < kStageUn; ++u) {
            const int e = e0 + u * kStep < ngr ? e0 + u * kStep : e0;
            ml[u] = e / per_map4;
            q4[u] = e - ml[u] * per_map4;
        }
        if (a.mode == 0) {
#pragma unroll
            for (int u = 0; u < kStageUn; ++u) v[u] = in4[(size_t)(map0 + ml[u]) * per_map4 + q4[u]];
        } else if (a.mode == 1) {                                   // HW = 1: q4 = channel group; pool the P_in pixels
#pragma unroll
            for (int u = 0; u < kStageUn; ++u) {
                const float4* y = in4 + (size_t)(map0 + ml[u]) * a.P_in * C4 + q4[u];
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll 4
                for (int p = 0; p < a.P_in; ++p) {
                    const float4 t = y[(size_t)p * C4];
                    sum.x += t.x; sum.y += t.y; sum.z += t.z; sum.w += t.w;
                    mx.x = fmaxf(mx.x, t.x); mx.y = fmaxf(mx.y, t.y); mx.z = fmaxf(mx.z, t.z); mx.w = fmaxf(mx.w, t.w);
                }
                const float inv = 1.f / (float)a.P_in;
                v[u] = make_float4(0.5f * (sum.x * inv) + 0.5f * mx.x, 0.5f * (sum.y * inv) + 0.5f * mx.y,
                                   0.5f * (sum.z * inv) + 0.5f * mx.z, 0.5f * (sum.w * inv) + 0.5f * mx.w);
            }
        } else {
            float4 g[kStageUn], x[kStageUn];
#pragma unroll
            for (int u = 0; u < kStageUn; ++u) {
                const size_t map = (size_t)(map0 + ml[u]);
                v[u] = in4[map * per_map4 + q4[u]];
                g[u] = reinterpret_cast<const float4*>(a.gate)[map * C4 + q4[u] % C4];
                x[u] = reinterpret_cast<const float4*>(a.res)[map * per_map4 + q4[u]];
            }
#pragma unroll
            for (int u = 0; u < kStageUn; ++u) {
                v[u] = make_float4(fmaxf(v[u].x * g[u].x + x[u].x, 0.f), fmaxf(v[u].y * g[u].y + x[u].y, 0.f),
                                   fmaxf(v[u].z * g[u].z + x[u].z, 0.f), fmaxf(v[u].w * g[u].w + x[u].w, 0.f));
                if (a.stage_out && nb == 0 && e0 + u * kStep < ngr)
                    *reinterpret_cast<float4*>(a.stage_out + (size_t)(map0 + ml[u]) * a.stage_out_ms + (size_t)q4[u] * 4) = v[u];
            }
        }
#pragma unroll
        for (int u = 0; u < kStageUn; ++u)
            if (e0 + u * kStep < ngr) put4(ml[u], q4[u], v[u]);
    }
    __syncthreads();
    // ---- K loop
    const int r = lane & 31, h = lane >> 5;
    const int P = a.OHW * a.OHW, rows = nm * P;
    const int gpt = C >> 4, gsh = __builtin_ctz(gpt);
    const int nsteps = a.KH * a.KW * gpt, last = nsteps - 1;
    const int mine = (nsteps - wave + kTailWaves - 1) / kTailWaves;
    int base[MBG];
    unsigned mask[MBG];
#pragma unroll
    for (int q = 0; q < MBG; ++q) {
        const int row = q * 32 + r;
        const bool ok = row < rows;
        const int ml = ok ? row / P : 0, pix = ok ? row - (row / P) * P : 0;
        const int oy = pix / a.OHW, ox = pix - oy * a.OHW;
        base[q] = (ml * HW2 + (oy - a.pad) * a.HW + (ox - a.pad)) * PS + h * 16;
        unsigned mk = 0;
        if (ok)
            for (int kh = 0; kh < a.KH; ++kh)
                for (int kw = 0; kw < a.KW; ++kw) {
                    const int iy = oy - a.pad + kh, ix = ox - a.pad + kw;
                    if (iy >= 0 && iy < a.HW && ix >= 0 && ix < a.HW) mk |= 1u << (kh * a.KW + kw);
                }
        mask[q] = mk;
    }
    f32x16 acc[MBG], acc2[MBG];
#pragma unroll
    for (int q = 0; q < MBG; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = acc2[q][i] = 0.f;
    const unsigned char* bp = a.w + (size_t)nb * nsteps * 2048 + (h * 32 + r) * 16;
    uint4 bh[kTailPF], bl[kTailPF];
    auto load_b = [&](int step, int p) {
        const int kk = step < last ? step : last;                  // unconditional loads (conv_lds): clamped to the last step
        bh[p] = *reinterpret_cast<const uint4*>(bp + (size_t)kk * 2048);
        bl[p] = *reinterpret_cast<const uint4*>(bp + (size_t)kk * 2048 + 1024);
    };
#pragma unroll
    for (int p = 0; p < kTailPF; ++p) load_b(wave + kTailWaves * p, p);
#pragma unroll 1
    for (int i = 0; i < mine; i += kTailPF) {
#pragma unroll
        for (int p = 0; p < kTailPF; ++p) {
            const int step = wave + kTailWaves * (i + p);
            if (i + p < mine) {
                const int tap = step >> gsh, g = step & (gpt - 1);
                const int kh = tap / a.KW, kw = tap - kh * a.KW;
                const int toff = (kh * a.HW + kw) * PS + g * 64;
#pragma unroll
                for (int q = 0; q < MBG; ++q) {
                    const bool ok = (mask[q] >> tap) & 1u;
                    const unsigned char* ap = ok ? Xs + base[q] + toff : zero + h * 16;
                    const uint4 ah = *reinterpret_cast<const uint4*>(ap);
                    const uint4 al = *reinterpret_cast<const uint4*>(ok ? ap + 32 : ap);
                    mfma3s(ah, al, bh[p], bl[p], acc[q], acc2[q]);
                }
            }
            if (i + kTailPF < mine) load_b(step + kTailWaves * kTailPF, p);    // wave-uniform: the last round reloads nothing
        }
    }
    // ---- partial tiles -> LDS, summed in wave order.  C/D map: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int q = 0; q < MBG; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) part[wave][q][(i & 3) + 8 * (i >> 2) + 4 * h][r] = acc[q][i] + acc2[q][i];
    __syncthreads();
    const int col = tid & 31, n = nb * 32 + col;
    if (n >= a.N) return;
    const float bias = a.bias[n];
    const float sc = a.bn_scale ? a.bn_scale[n] : 1.f, sh = a.bn_scale ? a.bn_shift[n] : 0.f;
#pragma unroll
    for (int j = 0; j < 2 * MBG; ++j) {
        const int rg = (tid >> 5) + 16 * j;
        if (rg >= rows) break;
        float v = part[0][rg >> 5][rg & 31][col];
#pragma unroll
        for (int q = 1; q < kTailWaves; ++q) v += part[q][rg >> 5][rg & 31][col];
        v += bias;
        v = a.act == 1 ? 1.f / (1.f + expf(-v)) : (v > 0.f ? v : 0.f);
        a.out[((size_t)map0 * P + rg) * a.N + n] = v * sc + sh;
    }
}

// ------------------------------------------------------------------------------------------------ BEV update
// new_bev = conv2(relu(conv0(cat([bev, h broadcast])))) + bev  (thinktwice_decoder.py:221-225,257).  The 2048 broadcast
// channels enter as G[b][tap][n] = W0[n, 32:, tap] . h[b] (one wide linear over the B rows, tt_mlp_chain) and are added
// per border class; the 128 hidden channels are produced and consumed 32 at a time so that both maps fit in LDS.
struct BevArgs {
    const float* bev;         // [B][441][32] f32
    const float* G;           // [B][9][128] f32
    float* out;               // new bev, f32, row (b, pixel) at out + b*out_bstride + pixel*32
    float* out2;              // optional second copy (the stacked per-layer outputs), same addressing with out2_bstride
    long long out_bstride, out2_bstride;
    const unsigned char* w0;  // pair [128][9*32]: the bev part of BEV_feat_update_module.0
    const float* b0;          // [128]
    const unsigned char* w2[4];   // pair [32][9*32] each: BEV_feat_update_module.2 restricted to hidden channels 32c..32c+31
    const float* b2;          // [32]
    float* part;              // scratch [B][4][448][32] f32: the four hidden-channel chunks' partial outputs
    unsigned* tickets;        // [B], zero at launch
};

// One workgroup per (sample, chunk of 32 hidden channels): out = sum_c conv2_c(relu(conv0_c([bev, h]))) + b2 + bev.  A sample
// is eight 441 x 32 x 288 convolutions, MFMA-bound on the ONE CU that ran them (58 us on the batch-1 tick's critical path);
// dealt over four CUs, the last workgroup of a sample to finish adds the four partial maps IN CHUNK ORDER -- the order the
// single workgroup accumulated them in its registers, so the result is bit-identical to it.
__global__ __launch_bounds__(kBevWaves * 64) void dec_bev_update_kernel(const BevArgs a) {
    // (maps padded to 448 pixels, 16 class slots, packed border classes: see dec_gru_kernel)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Bmap = smem;
    unsigned char* Hmap = smem + kMapPixPad * kPS32;
    unsigned char* zero = Hmap + kMapPixPad * kPS32;
    float* Gc = reinterpret_cast<float*>(zero + 64);              // [16 class slots][32], bias included
    __shared__ unsigned s_old;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x, b = blockIdx.y;
    if (tid < 16) reinterpret_cast<uint32_t*>(zero)[tid] = 0u;
    const float* src = a.bev + (size_t)b * kMapPix * kMapC;
    load_map32<kBevWaves * 64>(Bmap, src, tid);
    const float* g = a.G + (size_t)b * 9 * 128 + c * 32;
    for (int e = tid; e < 16 * 32; e += kBevWaves * 64) {
        const int cls = e >> 5, n = e & 31;
        float s = 0.f;
        if (cls < 9) {
            s = a.b0[c * 32 + n];
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw)
                    if (tap_valid(cls, kh, kw)) s += g[(kh * 3 + kw) * 128 + n];
        }
        Gc[e] = s;
    }
    const int r = lane & 31, h = lane >> 5;
    const int m_lane = wave * 64 + 4 * h;                         // + q * 32 + (i & 3) + 8 * (i >> 2)
    unsigned long long clsp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        clsp[q] = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m_lane + q * 32 + (i & 3) + 8 * (i >> 2);
            const int y = m / kMapHW, x = m - y * kMapHW;
            clsp[q] |= (m < kMapPix ? (unsigned long long)border_class(y, x, kMapHW, kMapHW) : 15ull) << (4 * i);
        }
    }
    __syncthreads();
    int ml = m_lane;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(ml));
#endif
    conv3x3_map32<2>(Bmap, zero, a.w0 + (size_t)c * 32 * (9 * 32 * 4), wave, kBevWaves, lane,
                     [&](int, int q, int i, float v) {
                         v += Gc[(int)((clsp[q] >> (4 * i)) & 15ull) * 32 + r];
                         pair_store(Hmap, kPS32, ml + q * 32 + (i & 3) + 8 * (i >> 2), r, v > 0.f ? v : 0.f);
                     });
    __syncthreads();
    // the chunk's partial output: all 448 rows of the padded map (the buffer is padded likewise), read back by one workgroup
    float* mine = a.part + ((size_t)b * 4 + c) * kMapPixPad * kMapC;
    conv3x3_map32<2>(Hmap, zero, a.w2[c], wave, kBevWaves, lane, [&](int, int q, int i, float v) {
        mine[(size_t)(ml + q * 32 + (i & 3) + 8 * (i >> 2)) * kMapC + r] = v;
    });
    // last workgroup of the sample: agent-scope release of the partial map, ticket, acquire (the XCDs' L2s are not coherent
    // with each other for plain stores)
    __threadfence();
    __syncthreads();
    if (tid == 0) s_old = __hip_atomic_fetch_add(a.tickets + b, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_old != 3u) return;
    __threadfence();
    const float* p = a.part + (size_t)b * 4 * kMapPixPad * kMapC;
    constexpr int kChunk = kMapPixPad * kMapC;
    for (int e = tid; e < kMapPix * kMapC; e += kBevWaves * 64) {
        float acc = __builtin_nontemporal_load(p + e);
        acc += __builtin_nontemporal_load(p + kChunk + e);
        acc += __builtin_nontemporal_load(p + 2 * kChunk + e);
        acc += __builtin_nontemporal_load(p + 3 * kChunk + e);
        const float v = acc + a.b2[e & 31] + src[e];
        a.out[(size_t)b * a.out_bstride + e] = v;
        if (a.out2) a.out2[(size_t)b * a.out2_bstride + e] = v;
    }
}

}  // namespace tt

using namespace tt;

static long long* g_dec_trace = nullptr;
extern "C" int tt_dec_set_trace(void* stamps_or_null) {
    g_dec_trace = static_cast<long long*>(stamps_or_null);
    return 0;
}

extern "C" long long tt_dec_gru_scratch_floats(int B) { return (long long)B * (3 * kMapPixPad * kMapC + 2); }

extern "C" int tt_dec_gru(int B, const float* inp6, const float* state, float* fut, float* scratch, const void* const* w0,
                          const float* const* wx, const float* const* b0, const void* const* w2, const float* const* b2,
                          const void* wd0, const float* bd0, const void* wd2, const float* bd2, void* stream) {
    TT_REQUIRE(B > 0 && inp6 && state && fut && scratch && w0 && wx && b0 && w2 && b2 && wd0 && bd0 && wd2 && bd2, "tt_dec_gru: null");
    TT_REQUIRE((reinterpret_cast<uintptr_t>(state) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0,
               "tt_dec_gru: state / scratch must be 16 B aligned");
    if (int rc = refuse_after_fault("tt_dec_gru")) return rc;
    GruArgs a;
    a.inp6 = inp6; a.state = state; a.fut = fut; a.scratch = scratch;
    for (int c = 0; c < 3; ++c) {
        TT_REQUIRE(w0[c] && wx[c] && b0[c] && w2[c] && b2[c], "tt_dec_gru: null weight %d", c);
        a.w0[c] = (const unsigned char*)w0[c]; a.wx[c] = wx[c]; a.b0[c] = b0[c];
        a.w2[c] = (const unsigned char*)w2[c]; a.b2[c] = b2[c];
    }
    a.wd0 = (const unsigned char*)wd0; a.bd0 = bd0; a.wd2 = (const unsigned char*)wd2; a.bd2 = bd2;
    a.trace = g_dec_trace;
    // the two workgroups of a sample hand the state / the update gate over through flags behind the scratch maps (zeroed here,
    // on the stream: recordable into a HIP graph); a wait that gives up sets the device's fault word (dec_chain.hip)
    a.flags = reinterpret_cast<unsigned*>(scratch + (size_t)B * 3 * kMapPixPad * kMapC);
    a.fault = device_fault_word();
    TT_REQUIRE(a.fault, "tt_dec_gru: cannot allocate the host-mapped fault word");
    a.max_spin = pair_wait_max_spin();
    TT_REQUIRE(hipMemsetAsync(a.flags, 0, (size_t)B * 2 * sizeof(unsigned), (hipStream_t)stream) == hipSuccess,
               "tt_dec_gru: hipMemsetAsync failed");
    const size_t smem = (size_t)2 * kMapPixPad * kPS32 + 64 + 3 * 16 * 32 * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_gru_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
        attr = true;
    }
    hipLaunchKernelGGL(dec_gru_kernel, dim3(2u, (unsigned)B), dim3(kGruWaves * 64), smem, (hipStream_t)stream, a);
    return check_launch("tt_dec_gru");
}

// floats of scratch per map: x4, MLP4's y1 / y2, its pooled fc1 output and gate, the block output, the 2x2 level likewise, fc0
constexpr int kFlatScratch = 2048 + 4096 + 2048 + 128 + 128 + 2048 + 1024 + 2048 + 1024 + 256 + 256 + 1024 + 512;

extern "C" long long tt_dec_flatten_scratch_floats(int maps) { return (long long)maps * kFlatScratch; }

extern "C" int tt_dec_flatten(int maps, const float* in, float* out, float* mids_or_null, float* scratch,
                              const void* const* w, const float* const* b, const float* bn_scale, const float* bn_shift,
                              void* stream) {
    TT_REQUIRE(maps > 0 && in && out && scratch && w && b && bn_scale && bn_shift, "tt_dec_flatten: null");
    TT_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(mids_or_null) & 15) == 0, "tt_dec_flatten: in / scratch / mids must be 16 B aligned");
    FlatArgs a;
    a.in = in; a.out = out; a.mids = mids_or_null; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
    a.trace = g_dec_trace;
    for (int i = 0; i < kFlatSets; ++i) {
        TT_REQUIRE(w[i] && b[i], "tt_dec_flatten: null weight set %d", i);
        a.w[i] = (const unsigned char*)w[i];
        a.b[i] = b[i];
    }
    const size_t M = (size_t)maps;
    float* x4 = scratch;                 float* y4a = x4 + M * 2048;      float* y4b = y4a + M * 4096;
    float* s4 = y4b + M * 2048;          float* g4 = s4 + M * 128;        float* x4o = g4 + M * 128;
    float* x2 = x4o + M * 2048;          float* y2a = x2 + M * 1024;      float* y2b = y2a + M * 2048;
    float* s2 = y2b + M * 1024;          float* g2 = s2 + M * 256;        float* x2o = g2 + M * 256;
    float* h512 = x2o + M * 1024;
    a.x4 = x4;
    const size_t smem = (size_t)((kMapPix * kPS32 + 15) / 16 * 16) + 2 * 100 * (64 * 4 + 16) + 64 +
                        2 * (64 * 4 + 16) + 64 * 4 + kFlatWaves * 64 * 2 * 4 + 64;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_flatten_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_tail_conv_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_tail_conv_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(dec_flatten_kernel, dim3((unsigned)maps), dim3(kFlatWaves * 64), smem, (hipStream_t)stream, a);
    // one column-split stage: `set` = weight set; input map HW x HW x C, KH x KW / pad -> OHW x OHW x N
    auto stage = [&](int set, int mode, const float* src, const float* gate, const float* res, float* stage_out,
                     long long stage_out_ms, int P_in, int HW, int C, int K, int pad, int OHW, int N, int act, bool bn,
                     float* dst, int MG, int MBG) {
        TailArgs t;
        t.in = src; t.gate = gate; t.res = res; t.stage_out = stage_out; t.stage_out_ms = stage_out_ms;
        t.mode = mode; t.maps = maps; t.MG = MG; t.HW = HW; t.C = C; t.KH = K; t.KW = K; t.pad = pad; t.OHW = OHW; t.P_in = P_in;
        t.w = a.w[set]; t.bias = a.b[set]; t.bn_scale = bn ? bn_scale : nullptr; t.bn_shift = bn ? bn_shift : nullptr;
        t.N = N; t.act = act; t.out = dst;
        const size_t lds = (size_t)MG * HW * HW * (C * 4 + 16) + 64 + (size_t)kTailWaves * MBG * 32 * 32 * 4;
        const dim3 grid((unsigned)((N + 31) / 32), (unsigned)((maps + MG - 1) / MG));
        if (MBG == 2) hipLaunchKernelGGL(dec_tail_conv_kernel<2>, grid, dim3(kTailWaves * 64), lds, (hipStream_t)stream, t);
        else hipLaunchKernelGGL(dec_tail_conv_kernel<1>, grid, dim3(kTailWaves * 64), lds, (hipStream_t)stream, t);
    };
    const long long ms = 100 * 64 + 16 * 128 + 4 * 256;
    float* mid4 = mids_or_null ? mids_or_null + 100 * 64 : nullptr;
    float* mid2 = mids_or_null ? mids_or_null + 100 * 64 + 16 * 128 : nullptr;
    // MLP4 (SEBasicBlock on the 4x4x128 map): conv1, conv2, se.fc1 on the pooled vector, se.fc2 -> gate
    stage(6, 0, x4, nullptr, nullptr, nullptr, 0, 0, 4, 128, 3, 1, 4, 256, 0, false, y4a, 4, 2);
    stage(7, 0, y4a, nullptr, nullptr, nullptr, 0, 0, 4, 256, 3, 1, 4, 128, 0, false, y4b, 4, 2);
    stage(8, 1, y4b, nullptr, nullptr, nullptr, 0, 16, 1, 128, 1, 0, 1, 128, 0, false, s4, 32, 1);
    stage(9, 0, s4, nullptr, nullptr, nullptr, 0, 0, 1, 128, 1, 0, 1, 128, 1, false, g4, 32, 1);
    // conv4_2 on the block's output relu(y * gate + x): 3x3, no padding, 128 -> 256
    stage(10, 2, y4b, g4, x4, mid4 ? mid4 : x4o, mid4 ? ms : 2048, 16, 4, 128, 3, 0, 2, 256, 0, false, x2, 8, 1);
    // MLP2 on the 2x2x256 map
    stage(11, 0, x2, nullptr, nullptr, nullptr, 0, 0, 2, 256, 3, 1, 2, 512, 0, false, y2a, 8, 1);
    stage(12, 0, y2a, nullptr, nullptr, nullptr, 0, 0, 2, 512, 3, 1, 2, 256, 0, false, y2b, 8, 1);
    stage(13, 1, y2b, nullptr, nullptr, nullptr, 0, 4, 1, 256, 1, 0, 1, 256, 0, false, s2, 32, 1);
    stage(14, 0, s2, nullptr, nullptr, nullptr, 0, 0, 1, 256, 1, 0, 1, 256, 1, false, g2, 32, 1);
    // output_fc.0 over the pixel-major flattening of the block's 2x2x256 output = a 2x2 "valid" conv; ReLU; BatchNorm1d (eval)
    stage(15, 2, y2b, g2, x2, mid2 ? mid2 : x2o, mid2 ? ms : 1024, 4, 2, 256, 2, 0, 1, 512, 0, true, h512, 16, 1);
    stage(16, 0, h512, nullptr, nullptr, nullptr, 0, 0, 1, 512, 1, 0, 1, 256, 0, false, out, 32, 1);
    return check_launch("tt_dec_flatten");
}

extern "C" long long tt_dec_bev_update_scratch_floats(int B) { return (long long)B * (4 * kMapPixPad * kMapC + 1); }

extern "C" int tt_dec_bev_update(int B, const float* bev, const float* G, float* out, long long out_bstride, float* out2,
                                 long long out2_bstride, float* scratch, const void* w0, const float* b0,
                                 const void* const* w2, const float* b2, void* stream) {
    TT_REQUIRE(B > 0 && bev && G && out && scratch && w0 && b0 && w2 && b2, "tt_dec_bev_update: null");
    TT_REQUIRE((reinterpret_cast<uintptr_t>(bev) & 15) == 0, "tt_dec_bev_update: bev must be 16 B aligned");
    BevArgs a;
    a.bev = bev; a.G = G; a.out = out; a.out2 = out2; a.out_bstride = out_bstride; a.out2_bstride = out2_bstride;
    a.w0 = (const unsigned char*)w0; a.b0 = b0; a.b2 = b2;
    for (int c = 0; c < 4; ++c) {
        TT_REQUIRE(w2[c], "tt_dec_bev_update: null w2[%d]", c);
        a.w2[c] = (const unsigned char*)w2[c];
    }
    a.part = scratch;
    a.tickets = reinterpret_cast<unsigned*>(scratch + (size_t)B * 4 * kMapPixPad * kMapC);
    TT_REQUIRE(hipMemsetAsync(a.tickets, 0, (size_t)B * sizeof(unsigned), (hipStream_t)stream) == hipSuccess,
               "tt_dec_bev_update: hipMemsetAsync failed");
    const size_t smem = (size_t)2 * kMapPixPad * kPS32 + 64 + 16 * 32 * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_bev_update_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr = true;
    }
    hipLaunchKernelGGL(dec_bev_update_kernel, dim3(4u, (unsigned)B), dim3(kBevWaves * 64), smem, (hipStream_t)stream, a);
    return check_launch("tt_dec_bev_update");
}
