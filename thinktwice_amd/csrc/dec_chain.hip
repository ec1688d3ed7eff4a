// Row-batched MLP chains of the look-and-predict decoder in ONE launch (tt_mlp_chain).
//
// The reference decoder (thinktwice_decoder.py:26-260, multi_scale_deformable_attn_function.py:197-344) is a long
// sequence of nn.Linear layers over a few hundred to a few thousand rows: query_linear -> sampling_offsets /
// attention_weights, ffn.w_1 -> w_2 (+ residual), output_proj, mlp -> traj / ctrl offset heads, the coarse heads.
// Launched one kernel per layer they are ~5 us each of pure latency.  Here a workgroup owns 32 rows and walks the
// whole chain: intermediates never leave LDS, only the outputs a later kernel needs are written to HBM.
//
// Arithmetic: "bf16x3" on the bf16 MFMA (the exact-f32 MFMA is 16x slower): every f32 operand is a bf16 (hi, lo) pair,
// a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi accumulated in f32 (relative error ~1e-5 per dot product).  Weights arrive
// pre-split from the host in FRAGMENT-MAJOR order (weights.py::split_pairs_frag: per 32-column block and 16-wide K
// step one contiguous KiB of hi halves + one of lo halves, indexed by lane), so a wave's B operand is two fully
// coalesced 1 KiB loads; intermediates are split ONCE when a stage writes them to LDS ("pair format": per row, per 16
// K elements 64 B = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15]), so a wave's A fragment is two ds_read_b128 with no
// conversion; the chain input is split when it is loaded from HBM.
//
// Layout of a stage: out[32 rows][N] = A[32][K] * W^T.  v_mfma_f32_32x32x16_bf16: lane l supplies row / column
// (l & 31) and the 8-element K chunk (l >> 5) of a 16-element K step.  The 8 waves of the workgroup take the 32-column
// blocks of N round-robin, up to 4 accumulator blocks per wave at a time, so one A fragment feeds up to 12 MFMAs.
// Weights stream from L2 straight into the B operand registers (each element is used once per workgroup: staging
// them through LDS would only add a round trip), double-buffered one K step ahead.
#include <atomic>
#include <mutex>
#include <string.h>

#include "conv_common.h"

namespace tt {

constexpr int kChainMaxStages = 12;
constexpr int kChainWaves = 8;
constexpr int kChainNBW = 2;      // accumulator blocks per wave per pass
constexpr int kChainPF = 4;       // weight K-steps in flight per wave

struct ChainStage {
    const void* w;        // fragment-major pair-format weights of the [N padded to 32][Kp] matrix
    const float* bias;    // [N] or null
    const float* res;     // optional residual rows (global f32): v += res[m * res_stride + res_coff + n]
    const float* side;    // optional extra input columns (global f32 [R][side_stride]): v += sum_j side[m][j] * side_w[n][j]
    const float* side_w;  // [N][side_k] f32
    float* out;           // optional global f32 output: out[m * out_stride + out_coff + n]
    int K, Kp, N, act;
    int in_sel;           // -1: the chain input x; s >= 0: the LDS output of stage s
    int res_stride, res_coff, side_stride, side_k;
    int out_stride, out_coff;
    int lds_off;          // byte offset of this stage's LDS output (pair format); -1: not kept
    int lds_stride;       // bytes per row of that buffer (Kp_next * 4 + 16: an odd number of 16 B slots, conflict-free)
};

struct ChainArgs {
    const float* x;
    long long R;
    int x_stride, nstages;
    int nbw;              // 32-column blocks per wave per pass (1..kChainNBW); 1 when N is split over gridDim.y
    ChainStage st[kChainMaxStages];
};

__device__ __forceinline__ void split8(const float (&x)[8], uint4& hi, uint4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);
        const float r0 = x[2 * e] - __uint_as_float(h[e] << 16);
        const float r1 = x[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u);
        l[e] = pack_bf16x2(r0, r1);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// the cross terms and the main term go to two accumulators (summed in the epilogue): a back-to-back MFMA pair on the
// same accumulator waits for the first one's last pass
__device__ __forceinline__ void mfma3(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x16& c,
                                      f32x16& c2) {
    Mfma<uint16_t>::run(al, bh, c2);
    Mfma<uint16_t>::run(ah, bh, c);
    Mfma<uint16_t>::run(ah, bl, c2);
}

// store one f32 value as a (hi, lo) bf16 pair at element (row, col) of a pair-format LDS buffer
__device__ __forceinline__ void lds_store_pair(unsigned char* buf, int stride, int row, int col, float v) {
    const uint16_t hi = f32_to_bf16(v);
    const uint16_t lo = f32_to_bf16(v - bf16_to_f32(hi));
    unsigned char* p = buf + (size_t)row * stride + (col >> 4) * 64 + ((col >> 3) & 1) * 16 + (col & 7) * 2;
    *reinterpret_cast<uint16_t*>(p) = hi;
    *reinterpret_cast<uint16_t*>(p + 32) = lo;
}

__global__ __launch_bounds__(kChainWaves * 64) void mlp_chain_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r_ = lane & 31, h_ = lane >> 5;
    const long long m0 = (long long)blockIdx.x * 32;
    const long long m = m0 + r_;
    const bool row_ok = m < a.R;

    for (int s = 0; s < a.nstages; ++s) {
        const ChainStage& S = a.st[s];
        const int NB = (S.N + 31) >> 5;
        const int nsteps = S.Kp >> 4;
        const bool from_x = S.in_sel < 0;
        const unsigned char* abuf = from_x ? nullptr : smem + a.st[S.in_sel].lds_off;
        const int astride = from_x ? 0 : a.st[S.in_sel].lds_stride;
        const float* xrow = a.x + (row_ok ? m : 0) * a.x_stride;

        const int per_pass = kChainWaves * a.nbw;
        for (int base = per_pass * blockIdx.y; base < NB; base += per_pass * gridDim.y) {
            // launder the lane-derived indices: the row / column address arithmetic of the epilogue is invariant across
            // passes and stages and would otherwise be hoisted and kept live through the K loops (spills)
            int r = r_, h = h_;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(r), "+v"(h));
#endif
            // this pass: blocks nb0, nb0 + 8, ... (nbw of them, those < NB); gridDim.y > 1 splits a single wide stage
            const int nb0 = base + wave;
            const unsigned char* bptr[kChainNBW];
            bool live[kChainNBW];
#pragma unroll
            for (int q = 0; q < kChainNBW; ++q) {
                const int nb = nb0 + q * kChainWaves;
                live[q] = q < a.nbw && nb < NB;
                // fragment-major weights: block nb, step ks at ((nb * nsteps + ks) * 2 + plane) KiB, lane-contiguous
                bptr[q] = reinterpret_cast<const unsigned char*>(S.w) + (size_t)(live[q] ? nb : 0) * nsteps * 2048 +
                          (h * 32 + r) * 16;
            }
            f32x16 acc[kChainNBW], acc2[kChainNBW];
#pragma unroll
            for (int q = 0; q < kChainNBW; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[q][i] = acc2[q][i] = 0.f;

            // Weight ring of kChainPF K-steps; every load is UNCONDITIONAL (step clamped to the last one, dead blocks
            // re-read block 0, rows beyond R re-read row 0) so the compiler counts the outstanding loads statically
            // instead of draining them at every loop head.
            const int last = nsteps - 1;
            auto load_b = [&](int ks, uint4 (&bh)[kChainNBW], uint4 (&bl)[kChainNBW]) {
                const int kk = ks < last ? ks : last;
#pragma unroll
                for (int q = 0; q < kChainNBW; ++q) {
                    const unsigned char* p = bptr[q] + (size_t)kk * 2048;
                    bh[q] = *reinterpret_cast<const uint4*>(p);
                    bl[q] = *reinterpret_cast<const uint4*>(p + 1024);
                }
            };
            uint4 bh[kChainPF][kChainNBW], bl[kChainPF][kChainNBW];
#pragma unroll
            for (int p = 0; p < kChainPF; ++p) load_b(p, bh[p], bl[p]);
            if (from_x) {
                // A from HBM/L2 (f32 rows, x_stride >= Kp checked by the host): a ring of raw fragments, split on use
                const float* xa = xrow + h * 8;
                float4 ar0[kChainPF], ar1[kChainPF];
#pragma unroll
                for (int p = 0; p < kChainPF; ++p) {
                    const int kk = p < last ? p : last;
                    ar0[p] = *reinterpret_cast<const float4*>(xa + kk * 16);
                    ar1[p] = *reinterpret_cast<const float4*>(xa + kk * 16 + 4);
                }
#pragma unroll 1
                for (int ks = 0; ks < nsteps; ks += kChainPF) {
#pragma unroll
                    for (int p = 0; p < kChainPF; ++p) {
                        const int k = ks + p;
                        const float v[8] = {ar0[p].x, ar0[p].y, ar0[p].z, ar0[p].w, ar1[p].x, ar1[p].y, ar1[p].z, ar1[p].w};
                        uint4 ah, al;
                        split8(v, ah, al);
                        if (k < nsteps) {
#pragma unroll
                            for (int q = 0; q < kChainNBW; ++q)
                                if (live[q]) mfma3(ah, al, bh[p][q], bl[p][q], acc[q], acc2[q]);
                        }
                        const int kn = k + kChainPF < last ? k + kChainPF : last;
                        ar0[p] = *reinterpret_cast<const float4*>(xa + kn * 16);
                        ar1[p] = *reinterpret_cast<const float4*>(xa + kn * 16 + 4);
                        load_b(k + kChainPF, bh[p], bl[p]);
                    }
                }
            } else {
                const unsigned char* pa = abuf + (size_t)r * astride + h * 16;
#pragma unroll 1
                for (int ks = 0; ks < nsteps; ks += kChainPF) {
#pragma unroll
                    for (int p = 0; p < kChainPF; ++p) {
                        const int k = ks + p;
                        const int kk = k < last ? k : last;
                        const uint4 ah = *reinterpret_cast<const uint4*>(pa + kk * 64);
                        const uint4 al = *reinterpret_cast<const uint4*>(pa + kk * 64 + 32);
                        if (k < nsteps) {
#pragma unroll
                            for (int q = 0; q < kChainNBW; ++q)
                                if (live[q]) mfma3(ah, al, bh[p][q], bl[p][q], acc[q], acc2[q]);
                        }
                        load_b(k + kChainPF, bh[p], bl[p]);
                    }
                }
            }

            // epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
#pragma unroll
            for (int q = 0; q < kChainNBW; ++q) {
                if (!live[q]) continue;
                const int n = (nb0 + q * kChainWaves) * 32 + r;
                const bool n_ok = n < S.N;
                const float bias = (S.bias && n_ok) ? S.bias[n] : 0.f;
                float sw[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) sw[j] = (S.side && n_ok && j < S.side_k) ? S.side_w[n * S.side_k + j] : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
                    const long long mr = m0 + rr;
                    const bool ok = n_ok && mr < a.R;
                    float v = (acc[q][i] + acc2[q][i]) + bias;
                    if (S.side && ok) {
                        const float* sp = S.side + mr * S.side_stride;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (j < S.side_k) v += sp[j] * sw[j];
                    }
                    if (S.res && ok) v += S.res[mr * S.res_stride + S.res_coff + n];
                    v = apply_act(v, S.act);
                    if (!ok) v = 0.f;
                    if (S.out && ok) S.out[mr * S.out_stride + S.out_coff + n] = v;
                    if (S.lds_off >= 0) lds_store_pair(smem + S.lds_off, S.lds_stride, rr, n, v);
                }
            }
        }
        __syncthreads();   // this stage's LDS output is complete (and its input buffer free) before the next stage
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The WIDE form (tt_mlp_chain_wide): few rows, so the time of a chain is the latency of streaming its weights -- one workgroup
// keeps ~128 KiB of loads in flight and gets ~45 GB/s out of the L2 (a 4.4 MB chain: 90-120 us; the batch-1 tick runs 38 of
// them back to back).  Here the COLUMNS of every stage are dealt over `G` co-resident workgroups per 32 rows (gridDim.y) and
// the eight waves of a workgroup split the K steps of a column block between them (partial tiles summed through LDS in wave
// order: deterministic).  Stage outputs a later stage reads go through an f32 scratch in global memory; between dependent
// stages the G workgroups of a row block meet at a ticket barrier (agent-scope release / acquire: the XCDs' L2s are not
// coherent with each other for ordinary stores).  The ticket counters live in a library-owned pool, are claimed round-robin per
// launch and reset themselves with the launch's last arrival.
constexpr int kWidePF = 8;               // K steps in flight per wave (8 x 4 KiB: weights + activation fragments)
constexpr int kWideMaxSpin = 1 << 21;    // default bail-out of a ticket wait (~1 s): a wrong count must not hang the device

struct WideStage {
    ChainStage s;
    const float* in;      // stage input: the chain input rows, or an earlier stage's scratch (fragment-major)
    int in_stride;        // row stride in floats; fragment-major: 16-column steps per row block
    int in_frag;
    float* keep;          // scratch of this stage's output (null: no later stage reads it), fragment-major
    int keep_stride;      // its steps per row block = N rounded up to 32, / 16
    int sync_before;      // the row block's workgroups meet before this stage
};

struct WideArgs {
    long long R;
    int nstages, nsync;
    int rb;               // 32-row blocks per workgroup (1, 2, 4 or 8); its eight waves = rb row blocks x 8 / rb K slices
    unsigned* tickets;    // one counter (64 B apart) per row group
    int* fault;           // HOST-mapped (pinned) word of this device, set with a system-scope store if a ticket wait gave up:
                          // the host reads it without a copy (tt_device_faults) and every later entry point refuses to run
    int max_spin;         // polls before a wait gives up (kWideMaxSpin; tests force a time-out with 0)
    long long* trace;     // debug (tt_mlp_chain_wide_set_trace): 64 wall-clock stamps (10 ns ticks) per workgroup, or null
    WideStage st[kChainMaxStages];
};

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global store (~1 us per
// stage here), which only the ticket barrier needs
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// Returns true if the wait GAVE UP (the launch's workgroups were not all running within max_spin polls: the chip was
// oversubscribed beyond what the host-side occupancy check allows for, or a count is wrong).  A time-out is made LOUD, never
// silent: the device's host-mapped fault word is set (system scope), and the caller poisons every value the workgroup writes
// from here on with NaN, so the launch's outputs cannot be mistaken for results.  The workgroup still arrives at every later
// barrier, so the counter reaches its total and resets itself for the slot's next launch.
__device__ __forceinline__ bool ticket_barrier(unsigned* ctr, unsigned target, int* fault, int max_spin, int* gave_up) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spin = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spin > max_spin) {
                __hip_atomic_store(fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                *gave_up = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // one cache invalidate after the wait, not one per poll
    }
    __syncthreads();
    return *gave_up != 0;
}

// One stage of the wide chain with a ring of PF K steps per wave (PF = 2, 4, 8 by the stage's steps per wave: every ring slot
// is loaded unconditionally, and a workgroup's loads go through one L1 at 64 B / clock -- a ring deeper than the stage has
// steps would spend more time on duplicate loads than the stage's own weights take).
template <int PF>
__device__ __forceinline__ void wide_stage(const WideArgs& a, const WideStage& W, float (*part)[32][32], unsigned* ctr,
                                           unsigned& arrived, long long* tr, int& tri, int* gave_up, bool& poisoned) {
    const ChainStage& S = W.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.y, g = blockIdx.y;
    const int RB = a.rb, KS = kChainWaves / RB;
    const int rb = wave & (RB - 1), ks = wave / RB;
    const long long m0g = (long long)blockIdx.x * RB * 32;       // first row of the workgroup
    const long long m0 = m0g + rb * 32;                          // first row of this wave's block
    auto stamp = [&]() {
        if (tr && tid == 0 && tri < 64) tr[tri++] = (long long)wall_clock64();
    };
    const int NB = (S.N + 31) >> 5;
    const int nsteps = S.Kp >> 4;
    const int last = nsteps - 1;
    const int mine = (nsteps - ks + KS - 1) / KS;                // K steps ks, ks + KS, ... (<= 0: none)
    int r = lane & 31, h = lane >> 5;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(r), "+v"(h));
#endif
    // The WEIGHT half of the first block's ring is issued before the barrier: it does not depend on the other workgroups.
    uint4 bh[PF], bl[PF];
    float4 a0[PF], a1[PF];
    const unsigned char* bp = reinterpret_cast<const unsigned char*>(S.w) + (size_t)(g < NB ? g : 0) * nsteps * 2048 +
                              (h * 32 + r) * 16;
    auto load_b = [&](int step, int p) {
        const int kk = step < last ? step : last;
        const unsigned char* q = bp + (size_t)kk * 2048;
        bh[p] = *reinterpret_cast<const uint4*>(q);
        bl[p] = *reinterpret_cast<const uint4*>(q + 1024);
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) load_b(ks + KS * p, p);
    if (W.sync_before) {
        arrived += (unsigned)G;
        poisoned = ticket_barrier(ctr, arrived, a.fault, a.max_spin, gave_up) || poisoned;
    }
    stamp();
    // activation fragments: lanes of rows beyond R load nothing (a 1-row chain issues 2 of 64 lanes).  Row-major input (the
    // chain input): 32 B of row m0 + r per step; an earlier stage's scratch is FRAGMENT-MAJOR (per row block and step two
    // contiguous KiB indexed by lane), so those loads are fully coalesced.
    const bool row_live = m0 + r < a.R;
    const float* ap;
    int a_step, a_off1;
    if (W.in_frag) {
        ap = W.in + ((size_t)(m0 >> 5) * W.in_stride * 128 + (size_t)(h * 32 + r)) * 4;
        a_step = 512; a_off1 = 256;
    } else {
        ap = W.in + (row_live ? m0 + r : 0) * W.in_stride + h * 8;
        a_step = 16; a_off1 = 4;
    }
    auto load_a = [&](int step, int p) {
        const int kk = step < last ? step : last;
        if (row_live) {
            a0[p] = *reinterpret_cast<const float4*>(ap + (size_t)kk * a_step);
            a1[p] = *reinterpret_cast<const float4*>(ap + (size_t)kk * a_step + a_off1);
        } else {
            a0[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            a1[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    for (int nb = g; nb < NB; nb += G) {
        if (nb != g) {
            bp += (size_t)G * nsteps * 2048;
#pragma unroll
            for (int p = 0; p < PF; ++p) load_b(ks + KS * p, p);
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) load_a(ks + KS * p, p);
        // epilogue operands, loaded now so that their latency is not exposed after the K loop (first two row slots of the
        // thread: all of them when the workgroup has one row block)
        const int col = tid & 31, n = nb * 32 + col;
        const bool n_ok = n < S.N;
        const float bias = (S.bias && n_ok) ? S.bias[n] : 0.f;
        // (straight-line code: clamped addresses + selects under wave-uniform branches, not one branch per element)
        float sw[8], pre[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) sw[j] = 0.f;
        const int n_c = n_ok ? n : 0;
        if (S.side) {
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = S.side_w[n_c * S.side_k + (j < S.side_k ? j : 0)];
#if defined(__HIP_DEVICE_COMPILE__)
            // all eight loads issued before the first select (the compiler otherwise sinks each into its own branch: 8 serial
            // round trips)
            asm volatile("" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]), "+v"(wv[4]), "+v"(wv[5]), "+v"(wv[6]), "+v"(wv[7]));
#endif
#pragma unroll
            for (int j = 0; j < 8; ++j) sw[j] = (n_ok && j < S.side_k) ? wv[j] : 0.f;
        }
        constexpr int kPre = PF < 8 ? 2 : 0;                       // (the deep ring has no registers to spare, and its stages
                                                                  // are bound by their weight stream, not by this latency)
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            const long long mr = m0g + (tid >> 5) + 16 * j;
            const long long mr_c = mr < a.R ? mr : 0;
            float v = 0.f;
            if (S.res) v = S.res[mr_c * S.res_stride + S.res_coff + n_c];
            if (S.side) {
                const float* sp = S.side + mr_c * S.side_stride;
                float sv[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) sv[jj] = sp[jj < S.side_k ? jj : 0];
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]), "+v"(sv[5]), "+v"(sv[6]), "+v"(sv[7]));
#endif
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v += sv[jj] * sw[jj];
            }
            pre[j] = v;                                           // used only where (mr < R && n < N)
        }

        f32x16 acc, acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = acc2[i] = 0.f;
#pragma unroll 1
        for (int i = 0; i < mine; i += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const float v[8] = {a0[p].x, a0[p].y, a0[p].z, a0[p].w, a1[p].x, a1[p].y, a1[p].z, a1[p].w};
                uint4 ah, al;
                split8(v, ah, al);
                if (i + p < mine) mfma3(ah, al, bh[p], bl[p], acc, acc2);
                if (i + PF < mine) {                 // wave-uniform: the last ring round reloads nothing
                    const int nxt = ks + KS * (i + p + PF);
                    load_b(nxt, p);
                    load_a(nxt, p);
                }
            }
        }
        // this wave's partial tile.  C/D map: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
#pragma unroll
        for (int i = 0; i < 16; ++i) part[wave][(i & 3) + 8 * (i >> 2) + 4 * h][r] = acc[i] + acc2[i];
        lds_barrier();
        if (nb == g) stamp();
        // epilogue: thread -> column `col` of rows (tid >> 5) + 16 j of the workgroup's RB * 32 rows; the K slices of a row
        // block are summed in slice order
        for (int j = 0; j < 2 * RB; ++j) {
            const int rg = (tid >> 5) + 16 * j;                   // row within the workgroup
            const int b = rg >> 5, rr = rg & 31;
            const long long mr = m0g + rg;
            const bool ok = n_ok && mr < a.R;
            float v = part[b][rr][col];
            for (int q = 1; q < KS; ++q) v += part[q * RB + b][rr][col];
            v += bias;
            if (j < kPre) {
                v += j == 0 ? pre[0] : pre[1];
            } else {
                const long long mr_c = ok ? mr : 0;
                float t = 0.f;
                if (S.res) t = S.res[mr_c * S.res_stride + S.res_coff + n_c];
                if (S.side) {
                    const float* sp = S.side + mr_c * S.side_stride;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) t += sp[jj < S.side_k ? jj : 0] * sw[jj];
                }
                v += t;
            }
            v = apply_act(v, S.act);
            if (poisoned) v = __builtin_nanf("");                 // a barrier of this workgroup timed out: never a plausible number
            if (!ok) v = 0.f;
            if (S.out && ok) S.out[mr * S.out_stride + S.out_coff + n] = v;
            if (W.keep && mr < a.R)                              // fragment-major scratch (see load_a); padding columns = 0
                W.keep[((((size_t)(mr >> 5) * W.keep_stride + (n >> 4)) * 2 + ((n >> 2) & 1)) * 64 + ((n >> 3) & 1) * 32 + rr) * 4 +
                       (n & 3)] = v;
        }
        lds_barrier();      // `part` is free again; the global stores are NOT waited for here (the next ticket barrier does)
    }
    stamp();
}

__global__ __launch_bounds__(kChainWaves * 64) void mlp_chain_wide_kernel(const WideArgs a) {
    __shared__ float part[kChainWaves][32][32];
    __shared__ int gave_up;
    const int tid = threadIdx.x;
    const int G = gridDim.y, g = blockIdx.y;
    unsigned* ctr = a.tickets + (size_t)blockIdx.x * 16;
    unsigned arrived = 0;
    bool poisoned = false;
    if (tid == 0) gave_up = 0;                                   // (published by the first ticket barrier's __syncthreads)
    long long* tr = a.trace ? a.trace + ((size_t)blockIdx.x * G + g) * 64 : nullptr;
    int tri = 0;
    if (tr && tid == 0) tr[tri++] = (long long)wall_clock64();
    const int KS = kChainWaves / a.rb;
    for (int s = 0; s < a.nstages; ++s) {
        const WideStage& W = a.st[s];
        if (tr && tid == 0 && tri < 64) tr[tri++] = (long long)wall_clock64();
        const int per_wave = ((W.s.Kp >> 4) + KS - 1) / KS;      // K steps of the busiest wave (uniform over the workgroup)
        if (per_wave <= 2) wide_stage<2>(a, W, part, ctr, arrived, tr, tri, &gave_up, poisoned);
        else if (per_wave <= 4) wide_stage<4>(a, W, part, ctr, arrived, tr, tri, &gave_up, poisoned);
        else wide_stage<8>(a, W, part, ctr, arrived, tr, tri, &gave_up, poisoned);
    }
    // the launch's last arrival puts the ticket back to zero for the next launch that claims this slot
    __syncthreads();
    if (tid == 0) {
        const unsigned total = (unsigned)(a.nsync + 1) * (unsigned)G;
        const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == total) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace tt

using namespace tt;

extern "C" int tt_mlp_chain(const float* x, long long R, int x_stride, int nstages, const tt_chain_stage* st,
                            int n_split, void* stream) {
    TT_REQUIRE(x && st && R > 0 && nstages >= 1 && nstages <= kChainMaxStages, "tt_mlp_chain: bad arguments");
    TT_REQUIRE(n_split >= 1 && (n_split == 1 || nstages == 1), "tt_mlp_chain: n_split > 1 needs a single stage");
    TT_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && x_stride % 4 == 0, "tt_mlp_chain: x must be 16 B aligned rows");
    ChainArgs a;
    a.x = x; a.R = R; a.x_stride = x_stride; a.nstages = nstages;
    a.nbw = n_split > 1 ? 1 : kChainNBW;
    // LDS plan: an output is kept while a LATER stage reads it; buffers are placed first-fit over the live ranges
    int last_use[kChainMaxStages];
    for (int s = 0; s < nstages; ++s) last_use[s] = -1;
    for (int s = 0; s < nstages; ++s) {
        TT_REQUIRE(st[s].in_sel < s, "tt_mlp_chain: stage %d reads stage %d", s, st[s].in_sel);
        if (st[s].in_sel >= 0) last_use[st[s].in_sel] = s;
    }
    size_t off[kChainMaxStages], len[kChainMaxStages];
    size_t total = 0;
    for (int s = 0; s < nstages; ++s) {
        const tt_chain_stage& d = st[s];
        TT_REQUIRE(d.w && d.K > 0 && d.N > 0 && d.Kp % 16 == 0 && d.Kp >= d.K && d.K % 4 == 0,
                   "tt_mlp_chain: stage %d: K=%d Kp=%d N=%d", s, d.K, d.Kp, d.N);
        TT_REQUIRE((reinterpret_cast<uintptr_t>(d.w) & 15) == 0, "tt_mlp_chain: stage %d weights unaligned", s);
        TT_REQUIRE(d.side_k >= 0 && d.side_k <= 8 && (!d.side || d.side_w), "tt_mlp_chain: stage %d side input", s);
        TT_REQUIRE(d.in_sel >= 0 || x_stride >= d.Kp,
                   "tt_mlp_chain: stage %d: x rows (stride %d) must cover the padded K = %d (finite padding)", s, x_stride,
                   d.Kp);
        TT_REQUIRE(d.in_sel < 0 || st[d.in_sel].N == d.K, "tt_mlp_chain: stage %d: K=%d but stage %d has N=%d", s, d.K,
                   d.in_sel, d.in_sel >= 0 ? st[d.in_sel].N : 0);
        ChainStage& S = a.st[s];
        S.w = d.w; S.bias = d.bias; S.res = d.res; S.side = d.side; S.side_w = d.side_w; S.out = d.out;
        S.K = d.K; S.Kp = d.Kp; S.N = d.N; S.act = d.act; S.in_sel = d.in_sel;
        S.res_stride = d.res_stride; S.res_coff = d.res_coff; S.side_stride = d.side_stride; S.side_k = d.side_k;
        S.out_stride = d.out_stride; S.out_coff = d.out_coff;
        S.lds_off = -1; S.lds_stride = 0;
        off[s] = 0; len[s] = 0;
        if (last_use[s] >= 0) {
            const int np = (d.N + 15) / 16 * 16;          // the consumer's Kp
            S.lds_stride = np * 4 + 16;
            len[s] = (size_t)32 * S.lds_stride;
            // Two-ended placement against the buffers still live at stage s (those read at or after s): a buffer goes
            // to the END of the 160 KiB window when its input sits in the lower half, else to the start -- a chain then
            // ping-pongs between the two ends and a long-lived buffer in the middle never splits the free space.
            constexpr size_t kBudget = 160 * 1024;
            auto clashes = [&](size_t pos) -> int {
                for (int t = 0; t < s; ++t)
                    if (len[t] != 0 && last_use[t] >= s && pos < off[t] + len[t] && off[t] < pos + len[s]) return t;
                return -1;
            };
            const bool in_low = d.in_sel < 0 || off[d.in_sel] + len[d.in_sel] / 2 < kBudget / 2;
            bool placed = false;
            size_t pos = 0;
            for (int attempt = 0; attempt < 2 && !placed; ++attempt) {
                const bool top = (attempt == 0) ? (d.in_sel >= 0 && in_low) : !(d.in_sel >= 0 && in_low);
                if (top) {
                    if (len[s] > kBudget) break;
                    long long p = (long long)((kBudget - len[s]) / 16 * 16);
                    int t;
                    while (p >= 0 && (t = clashes((size_t)p)) >= 0) p = ((long long)off[t] - (long long)len[s]) / 16 * 16;
                    if (p >= 0 && clashes((size_t)p) < 0) { pos = (size_t)p; placed = true; }
                } else {
                    size_t p = 0;
                    int t;
                    while ((t = clashes(p)) >= 0) p = (off[t] + len[t] + 15) / 16 * 16;
                    if (p + len[s] <= kBudget) { pos = p; placed = true; }
                }
            }
            TT_REQUIRE(placed, "tt_mlp_chain: intermediates of stage %d do not fit in 160 KiB of LDS", s);
            off[s] = pos;
            S.lds_off = (int)pos;
            if (pos + len[s] > total) total = pos + len[s];
        }
    }
    for (int s = 0; s < nstages; ++s)
        if (a.st[s].in_sel >= 0)
            TT_REQUIRE(a.st[a.st[s].in_sel].lds_stride == a.st[s].Kp * 4 + 16, "tt_mlp_chain: stage %d Kp mismatch", s);
    TT_REQUIRE(total <= 160 * 1024, "tt_mlp_chain: intermediates need %zu B of LDS (> 160 KiB)", total);
    if (total < 16) total = 16;
    static size_t attr = 0;
    if (total > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr = 160 * 1024;
    }
    const unsigned blocks = (unsigned)((R + 31) / 32);
    hipLaunchKernelGGL(mlp_chain_kernel, dim3(blocks, (unsigned)n_split), dim3(kChainWaves * 64), total,
                       (hipStream_t)stream, a);
    return check_launch("tt_mlp_chain");
}

// ---- tt_mlp_chain_wide
// Host state of the wide form, PER DEVICE and behind one mutex (the library may be driven from several host threads / devices):
//  * the ticket pool: kEagerSlots counters claimed round-robin by plain launches (a collision needs > kEagerSlots row groups in
//    flight at once), and kCapturedSlots claimed PERMANENTLY by launches recorded into a HIP graph (a captured launch replays
//    with the slot baked into its arguments, so no later launch may be handed the same counter);
//  * the fault word: one int in pinned, host-mapped memory per device.  A barrier that gives up sets it from the device with a
//    system-scope store; the host reads it without a copy or a synchronisation of its own (tt_device_faults), and once it is
//    set tt_mlp_chain_wide / tt_encoder_fwd / tt_decoder_fwd / tt_plan_run refuse to run until tt_clear_device_faults();
//  * the co-residency capacity: hipOccupancyMaxActiveBlocksPerMultiprocessor x the device's CU count.  A launch may use at most
//    capacity / kWideConcurrent workgroups (the decoder runs two chains at a time, on its main and its branch stream), so the
//    spinning workgroups of concurrent launches can never fill the chip and keep their own peers out.  hipLaunchCooperativeKernel
//    would give the same guarantee per launch, but it goes through the device's single cooperative queue (the two streams'
//    chains would serialise) and is not recordable into a HIP graph; the occupancy bound + bounded spin + loud fault is the
//    form that keeps both.
namespace {
constexpr int kEagerSlots = 4096;        // 64 B each; a launch claims one per row group
constexpr int kCapturedSlots = 12288;
constexpr int kMaxDevices = 64;
constexpr int kWideConcurrent = 2;
struct WidePool {
    unsigned* tickets = nullptr;         // kEagerSlots + kCapturedSlots counters, 64 B apart, device memory
    unsigned next_eager = 0, next_captured = 0;
    int capacity = 0;                    // co-resident workgroups of mlp_chain_wide_kernel on this device
};
WidePool g_pools[kMaxDevices];
int* g_fault_words = nullptr;            // pinned host-mapped: one int (64 B apart) per device
std::mutex g_wide_mutex;
std::atomic<int> g_wide_max_spin{kWideMaxSpin};
long long* g_trace = nullptr;            // tt_mlp_chain_wide_set_trace

int* fault_word(int dev) {               // (g_wide_mutex held, or after the first successful call)
    if (!g_fault_words) {
        void* p = nullptr;
        if (hipHostMalloc(&p, (size_t)kMaxDevices * 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        memset(p, 0, (size_t)kMaxDevices * 64);
        g_fault_words = static_cast<int*>(p);
    }
    return g_fault_words + (size_t)dev * 16;
}

// (g_wide_mutex held)
WidePool* wide_pool(int dev) {
    WidePool& P = g_pools[dev];
    if (P.tickets) return &P;
    if (!fault_word(dev)) {
        tt::set_error("tt_mlp_chain_wide: cannot allocate the host-mapped fault word");
        return nullptr;
    }
    int occ = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, tt::mlp_chain_wide_kernel, kChainWaves * 64, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || occ < 1 || cus < 1) {
        (void)hipGetLastError();
        tt::set_error("tt_mlp_chain_wide: cannot query the kernel's occupancy on device %d", dev);
        return nullptr;
    }
    void* p = nullptr;
    const size_t bytes = (size_t)(kEagerSlots + kCapturedSlots) * 64;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) {
        (void)hipGetLastError();
        tt::set_error("tt_mlp_chain_wide: cannot allocate the ticket pool");
        return nullptr;
    }
    P.capacity = occ * cus;
    P.tickets = static_cast<unsigned*>(p);
    return &P;
}
}  // namespace

// Non-blocking: the fault word of the CURRENT device (0 = no barrier has timed out since the last clear).  Meaningful for the
// work the caller has already synchronised with; any entry point that synchronises anyway (the D2H copy of tt_action_post's
// result, the end of a plan run in tools/plan_host.cpp) checks it there.
extern "C" int tt_device_faults(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -2;
    if (!g_fault_words) return 0;
    return __atomic_load_n(g_fault_words + (size_t)dev * 16, __ATOMIC_ACQUIRE);
}

extern "C" int tt_clear_device_faults(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -2;
    if (g_fault_words) __atomic_store_n(g_fault_words + (size_t)dev * 16, 0, __ATOMIC_RELEASE);
    return 0;
}

int* tt::device_fault_word() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_wide_mutex);
    return fault_word(dev);
}

int tt::pair_wait_max_spin() { return g_wide_max_spin.load(); }

// The check every forward entry point makes first: a barrier time-out is sticky (like a device fault), because whatever ran
// after it consumed poisoned data.
int tt::refuse_after_fault(const char* what) {
    const int f = tt_device_faults();
    if (f > 0) {
        tt::set_error("%s: refused -- an earlier tt_mlp_chain_wide launch on this device gave up waiting at its barrier (its "
                      "outputs are NaN); results since then are invalid.  tt_clear_device_faults() re-arms the device", what);
        return -3;
    }
    return 0;
}

extern "C" int tt_mlp_chain_wide_set_max_spin(int polls) {
    g_wide_max_spin.store(polls < 0 ? kWideMaxSpin : polls);
    return 0;
}

// Workgroups one launch may use on the current device (co-resident capacity / kWideConcurrent), or < 0 on error.
extern "C" int tt_mlp_chain_wide_max_workgroups(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -2;
    std::lock_guard<std::mutex> lock(g_wide_mutex);
    WidePool* P = wide_pool(dev);
    return P ? P->capacity / kWideConcurrent : -2;
}

static inline long long wide_keep_stride(int N) { return (N + 31) / 32 * 32; }
// 32-row blocks per workgroup: with several row blocks the waves of a workgroup share the weight stream (same K slice, different
// rows) instead of every row block streaming all the weights through its own workgroup
static inline int wide_rb(long long row_blocks) { return row_blocks <= 1 ? 1 : (row_blocks == 2 ? 2 : 4); }
static inline long long wide_rows(long long R) {
    const long long row_blocks = (R + 31) / 32;
    const int rb = wide_rb(row_blocks);
    return (row_blocks + rb - 1) / rb * rb * 32;
}

extern "C" long long tt_mlp_chain_wide_workspace_bytes(long long R, int nstages, const tt_chain_stage* st) {
    if (!st || R <= 0 || nstages < 1 || nstages > kChainMaxStages) return -1;
    const long long rows = wide_rows(R);
    long long total = 0;
    for (int s = 0; s < nstages; ++s) {
        bool kept = false;
        for (int t = s + 1; t < nstages; ++t) kept = kept || st[t].in_sel == s;
        if (kept) total += (rows * wide_keep_stride(st[s].N) * 4 + 255) / 256 * 256;
    }
    return total < 256 ? 256 : total;
}

extern "C" int tt_mlp_chain_wide_set_trace(void* stamps_or_null) {
    g_trace = static_cast<long long*>(stamps_or_null);
    return 0;
}

// Blocking form (tests): waits for the device, then reads the fault word.
extern "C" int tt_mlp_chain_wide_faults(void) {
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return tt_device_faults();
}

extern "C" int tt_mlp_chain_wide(const float* x, long long R, int x_stride, int nstages, const tt_chain_stage* st,
                                 int n_groups, void* workspace, long long workspace_bytes, void* stream) {
    TT_REQUIRE(x && st && R > 0 && nstages >= 1 && nstages <= kChainMaxStages, "tt_mlp_chain_wide: bad arguments");
    TT_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && x_stride % 4 == 0, "tt_mlp_chain_wide: x must be 16 B aligned rows");
    const long long row_blocks = (R + 31) / 32;
    const int rb = wide_rb(row_blocks);
    const long long row_groups = (row_blocks + rb - 1) / rb;
    // host-side argument checks first (no device is touched until they pass)
    TT_REQUIRE(n_groups >= 1 && n_groups <= 64, "tt_mlp_chain_wide: %d column groups: 1 - 64 workgroups per row group can be "
               "co-resident", n_groups);
    const long long need = tt_mlp_chain_wide_workspace_bytes(R, nstages, st);
    TT_REQUIRE(workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
               "tt_mlp_chain_wide: workspace of %lld B needed (%lld given)", need, workspace_bytes);
    if (int rc = tt::refuse_after_fault("tt_mlp_chain_wide")) return rc;
    int dev = 0;
    TT_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices, "tt_mlp_chain_wide: no current device");
    std::lock_guard<std::mutex> lock(g_wide_mutex);     // pool creation + slot claim (+ the launch: slots stay in claim order)
    WidePool* P = wide_pool(dev);
    if (!P) return -2;
    TT_REQUIRE(n_groups >= 1 && n_groups <= 64 && row_groups * n_groups * kWideConcurrent <= P->capacity,
               "tt_mlp_chain_wide: %lld row groups x %d column groups: at most %d workgroups per launch can be guaranteed "
               "co-resident on this device (%d resident, %d launches at a time)", row_groups, n_groups,
               P->capacity / kWideConcurrent, P->capacity, kWideConcurrent);
    WideArgs a;
    a.R = R; a.nstages = nstages; a.nsync = 0; a.rb = rb;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    int synced_upto = -1;                 // stages <= this index are complete on every workgroup of the row block
    for (int s = 0; s < nstages; ++s) {
        const tt_chain_stage& d = st[s];
        TT_REQUIRE(d.in_sel < s, "tt_mlp_chain_wide: stage %d reads stage %d", s, d.in_sel);
        TT_REQUIRE(d.w && d.K > 0 && d.N > 0 && d.Kp % 16 == 0 && d.Kp >= d.K && d.K % 4 == 0,
                   "tt_mlp_chain_wide: stage %d: K=%d Kp=%d N=%d", s, d.K, d.Kp, d.N);
        TT_REQUIRE((reinterpret_cast<uintptr_t>(d.w) & 15) == 0, "tt_mlp_chain_wide: stage %d weights unaligned", s);
        TT_REQUIRE(d.side_k >= 0 && d.side_k <= 8 && (!d.side || d.side_w), "tt_mlp_chain_wide: stage %d side input", s);
        TT_REQUIRE(d.in_sel >= 0 || x_stride >= d.Kp,
                   "tt_mlp_chain_wide: stage %d: x rows (stride %d) must cover the padded K = %d (finite padding)", s, x_stride,
                   d.Kp);
        TT_REQUIRE(d.in_sel < 0 || st[d.in_sel].N == d.K, "tt_mlp_chain_wide: stage %d: K=%d but stage %d has N=%d", s, d.K,
                   d.in_sel, d.in_sel >= 0 ? st[d.in_sel].N : 0);
        WideStage& W = a.st[s];
        ChainStage& S = W.s;
        S.w = d.w; S.bias = d.bias; S.res = d.res; S.side = d.side; S.side_w = d.side_w; S.out = d.out;
        S.K = d.K; S.Kp = d.Kp; S.N = d.N; S.act = d.act; S.in_sel = d.in_sel;
        S.res_stride = d.res_stride; S.res_coff = d.res_coff; S.side_stride = d.side_stride; S.side_k = d.side_k;
        S.out_stride = d.out_stride; S.out_coff = d.out_coff;
        S.lds_off = -1; S.lds_stride = 0;
        bool kept = false;
        for (int t = s + 1; t < nstages; ++t) kept = kept || st[t].in_sel == s;
        W.keep = nullptr; W.keep_stride = 0;
        if (kept) {
            W.keep = reinterpret_cast<float*>(ws);
            W.keep_stride = (int)wide_keep_stride(d.N) / 16;
            ws += (wide_rows(R) * wide_keep_stride(d.N) * 4 + 255) / 256 * 256;
        }
        if (d.in_sel < 0) {
            W.in = x; W.in_stride = x_stride; W.in_frag = 0; W.sync_before = 0;
        } else {
            W.in = a.st[d.in_sel].keep; W.in_stride = a.st[d.in_sel].keep_stride; W.in_frag = 1;
            // the consumer's padded K (a multiple of 16) lies inside the producer's 32-column blocks: finite (zero) padding
            W.sync_before = d.in_sel > synced_upto ? 1 : 0;
            if (W.sync_before) { synced_upto = s - 1; ++a.nsync; }
        }
    }
    if (n_groups == 1) {                   // one workgroup per row block: nobody to wait for
        for (int s = 0; s < nstages; ++s) a.st[s].sync_before = 0;
        a.nsync = 0;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    if (cap == hipStreamCaptureStatusActive) {          // replayed with these arguments: the slots are this launch's for good
        if (P->next_captured + row_groups > (unsigned)kCapturedSlots) {
            // a process that keeps re-capturing graphs: captured launches own their slots for good.  Distinct code: the caller
            // records the one-workgroup-per-row-block chain (tt_mlp_chain) instead (ops.mlp_chain does)
            tt::set_error("tt_mlp_chain_wide: the %d ticket slots reserved for graph-captured launches are used up", kCapturedSlots);
            return -4;
        }
        a.tickets = P->tickets + (size_t)(kEagerSlots + P->next_captured) * 16;
        P->next_captured += (unsigned)row_groups;
    } else {
        if (P->next_eager + row_groups > (unsigned)kEagerSlots) P->next_eager = 0;
        a.tickets = P->tickets + (size_t)P->next_eager * 16;
        P->next_eager += (unsigned)row_groups;
    }
    a.fault = fault_word(dev);
    a.max_spin = g_wide_max_spin.load();
    a.trace = g_trace;
    hipLaunchKernelGGL(mlp_chain_wide_kernel, dim3((unsigned)row_groups, (unsigned)n_groups), dim3(kChainWaves * 64), 0,
                       (hipStream_t)stream, a);
    return check_launch("tt_mlp_chain_wide");
}
