// Shared device helpers for the gfx950 kernels of librgda_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rgda_hip.h"

#define RGDA_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return RGDA_ERR_LAUNCH; \
    } while (0)

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
static __device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// fp32 -> bf16, round-to-nearest-even: the hardware conversion of gfx950 (v_cvt_pk_bf16_f32)
static __device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    f32x2_t v = {lo, hi};
    bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned, b);
}
static __device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// v + v[lane ^ O] for O = 8 / 16 / 32 without the LDS crossbar: DPP row rotate (rows are 16 lanes) and the gfx950
// half / row-pair exchanges.  `__shfl_xor` is a ds_bpermute_b32: 32 of them per wave in a conv epilogue, from all eight
// waves of the workgroup at once, cost ~3000 cycles on the shared LDS pipe (scripts/dev/dev_conv_phases2.py).
template <int O>
static __device__ __forceinline__ float xor_add(float v) {
    static_assert(O == 8 || O == 16 || O == 32, "xor_add: 8, 16 or 32");
    const unsigned u = __float_as_uint(v);
    if constexpr (O == 8) {
        return v + __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    } else if constexpr (O == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // rows (0,1) and (2,3) exchanged
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
        auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // the two half-waves exchanged
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
}

static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

#define NREP RGDA_STAT_REPLICAS

// ---- order-independent per-channel accumulators (include/rgda_hip.h: rgda_stat_t, 64-bit fixed point).
// A workgroup's partial sum (reduced in a fixed order inside the workgroup) -> round(v * 2^frac) -> integer atomic.
static __device__ __forceinline__ long long stat_fix(float v, int frac) {
    // round(v * 2^frac) as a 64-bit integer without fp64 or the software float -> int64 routine (40+ instructions at the
    // tail of every convolution workgroup): the scaling by a power of two is exact in fp32; x = hi * 2^31 + lo with
    // hi = trunc(x / 2^31) and the remainder lo (same sign, |lo| < 2^31, exactly representable: it is the low part of a
    // 24-bit mantissa), each converted by a native 32-bit instruction.
    float x = ldexpf(v, frac);
    if (!(fabsf(x) < 0x1p61f)) x = (v == v && fabsf(v) != __builtin_inff()) ? copysignf(0x1p61f, v) : 0.f;   // clamp; NaN / Inf add nothing
    const float hi = truncf(x * 0x1p-31f);
    const float lo = x - hi * 0x1p31f;
    return ((long long)__float2int_rz(hi) << 31) + (long long)__float2int_rn(lo);
}
static __device__ __forceinline__ void stat_add(rgda_stat_t* p, float v, int frac) {
    atomicAdd((unsigned long long*)p, (unsigned long long)stat_fix(v, frac));
}
// total of statistic `which` (0: first sum, 1: second) of channel c over the replicas of one row group, as a double
static __device__ __forceinline__ double stat_total(const rgda_stat_t* __restrict__ st, int C, int c, int which, int frac) {
    long long t = 0;
#pragma unroll
    for (int r = 0; r < NREP; ++r) t += st[(size_t)(2 * r + which) * C + c];
    return (double)t * (1.0 / (double)(1ll << frac));
}

// Tuning hooks (tile overrides, per-workgroup timestamps, ablation switches) read environment variables.  They exist
// only in a tuning build (`make TUNING=1`, what scripts/dev/dev_*.py expect); the product library never looks at the
// environment.
#include <stdlib.h>
#ifdef RGDA_TUNING
#define TUNE_ENV(name) getenv(name)
#else
#define TUNE_ENV(name) ((const char*)nullptr)
#endif
static inline hipStream_t to_stream(rgda_stream_t s) { return (hipStream_t)s; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
