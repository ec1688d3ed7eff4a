// Shared host/device helpers for libthinktwice_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/thinktwice_hip.h"

namespace tt {

void set_error(const char* fmt, ...);
// Non-zero (and the error text set) if a tt_mlp_chain_wide barrier has timed out on the current device since the last
// tt_clear_device_faults(): forward entry points return it first (csrc/dec_chain.hip).
int refuse_after_fault(const char* what);
// The current device's host-mapped fault word (allocated on first use; null on failure) and the poll bound of a cross-workgroup
// wait (tt_mlp_chain_wide_set_max_spin): for kernels whose workgroups wait for each other (tt_mlp_chain_wide, tt_dec_gru).
int* device_fault_word();
int pair_wait_max_spin();

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

#define TT_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ::tt::set_error(__VA_ARGS__);     \
            return -1;                        \
        }                                     \
    } while (0)

constexpr int kWave = 64;
constexpr int kNumCU = 256;

static inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// bf16 <-> f32 bit helpers (round-to-nearest-even), usable on device.
__device__ __forceinline__ float bf16_to_f32(uint16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
// f32 -> bf16 (round to nearest even) on the gfx950 converter: one v_cvt_pk_bf16_f32 instead of the ~6-op integer
// sequence.  The software form made every bf16 epilogue VALU-bound (measured: 1.37 TB/s of output regardless of
// the write pattern, profiles/README.md).
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
typedef __bf16 tt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float tt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    tt_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, tt_bf16x2));
}

// IEEE half storage (TT_F16): a distinct 2-byte type so the kernels can be instantiated for it beside bf16
// (uint16_t).  Same MFMA rate as bf16, 3 more mantissa bits: the 16-bit mode whose outputs stay inside the 1e-3
// tolerance of the reference.
struct f16_t {
    uint16_t bits;
};
typedef _Float16 tt_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float f16_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }   // RNE
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    tt_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, tt_f16x2));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kVec = 4;  // elements per 16-byte vector
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<uint16_t> {  // bf16 storage
    static constexpr int kVec = 8;
    __device__ static __forceinline__ float ld(const uint16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(uint16_t* p, float v) { *p = f32_to_bf16(v); }
};

template <> struct Elem<f16_t> {  // IEEE half storage
    static constexpr int kVec = 8;
    __device__ static __forceinline__ float ld(const f16_t* p) { return f16_to_f32(p->bits); }
    __device__ static __forceinline__ void st(f16_t* p, float v) { p->bits = f32_to_f16(v); }
};

// Two 16-bit elements packed in one dword <-> two floats (low half = element 0).
template <typename T> struct Pair16;
template <> struct Pair16<uint16_t> {
    __device__ static __forceinline__ void unpack(uint32_t w, float& a, float& b) {
        a = __uint_as_float(w << 16);
        b = __uint_as_float(w & 0xffff0000u);
    }
    __device__ static __forceinline__ uint32_t pack(float a, float b) { return pack_bf16x2(a, b); }
};
template <> struct Pair16<f16_t> {
    __device__ static __forceinline__ void unpack(uint32_t w, float& a, float& b) {
        const tt_f16x2 h = __builtin_bit_cast(tt_f16x2, w);
        a = (float)h.x;
        b = (float)h.y;
    }
    __device__ static __forceinline__ uint32_t pack(float a, float b) { return pack_f16x2(a, b); }
};

// Store one value as the 16-bit type named by a runtime dtype code (TT_BF16 / TT_F16).
__device__ __forceinline__ void store16(void* base, long long o, float v, int dtype) {
    reinterpret_cast<uint16_t*>(base)[o] = (dtype == TT_F16) ? f32_to_f16(v) : f32_to_bf16(v);
}

// Inline everywhere.  The conv epilogues only reach the transcendental cases from ROLLED loops (an unrolled use
// per accumulator element once blew the kernels up to ~50k ISA lines; an out-of-line call instead pinned the hot
// path's registers to the call ABI and spilled them).
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case TT_ACT_NONE: return v;
        case TT_ACT_RELU: return v > 0.f ? v : 0.f;
        case TT_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case TT_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        case TT_ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));
        case TT_ACT_SOFTPLUS_CLAMP: return fmaxf(v > 20.f ? v : log1pf(expf(v)), 1e-3f);
        default: return v;
    }
}

}  // namespace tt
