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
// A partial that is out of range (|v * 2^frac| >= 2^59) or not finite adds RGDA_STAT_POISON = 3 * 2^60 instead, and
// stat_total() reads a total of magnitude >= 2^59 as +inf: with in-range totals below 2^59, k poisons move the 64-bit sum
// by (3 k mod 16) * 2^60, i.e. out of range for every k that is not a multiple of 16 -- a diverging run shows inf / NaN in
// its BatchNorm outputs instead of silently wrapped (sign-flipped) statistics.  ONE comparison on the total: a check per
// replica cost 0.14 ms per step in the BatchNorm preambles.  In-range totals: |sum| < 2^59 / 2^frac = 8.6e9 forward,
// 5.2e5 backward.
#define RGDA_STAT_POISON (3ll << 60)
static __device__ __forceinline__ long long stat_fix(float v, int frac) {
    // round(v * 2^frac) as a 64-bit integer without fp64 or the software float -> int64 routine (40+ instructions at the
    // tail of every convolution workgroup): the scaling by a power of two is exact in fp32; x = hi * 2^31 + lo with
    // hi = trunc(x / 2^31) and the remainder lo (same sign, |lo| < 2^31, exactly representable: it is the low part of a
    // 24-bit mantissa), each converted by a native 32-bit instruction.
    const float x = ldexpf(v, frac);
    if (!(fabsf(x) < 0x1p59f)) return RGDA_STAT_POISON;       // out of range, Inf or NaN
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
    if (t >= (1ll << 59) || t <= -(1ll << 59)) return (double)__builtin_inff();      // poisoned or out of range
    return (double)t * (1.0 / (double)(1ll << frac));
}

// mean, biased variance and 1 / sqrt(var + eps) of channel c from one row group's accumulators: fp64 from the exact
// integer totals (every consumer goes through here, so forward, backward and the running update see the same values)
static __device__ __forceinline__ void stat_moments(const rgda_stat_t* __restrict__ st, int C, int c, double invM, float eps,
                                                    float& mean, float& var, float& istd) {
    const double m = stat_total(st, C, c, 0, RGDA_STAT_FRAC_FWD) * invM;
    double v = stat_total(st, C, c, 1, RGDA_STAT_FRAC_FWD) * invM - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    istd = 1.f / sqrtf((float)v + eps);       // (fp32: a double-precision sqrt + divide per channel shows in the apply passes)
}

// ---- BatchNorm (+ ReLU) applied on the CONSUMER's operand path (include/rgda_hip.h: rgda_bn_operand).
// The producing convolution leaves its raw output and the per-channel accumulators; the consumer (a convolution, the
// stem's max-pool) rebuilds (scale, shift) per channel in its prologue and applies  a = relu(fma(x, scale, shift))  to the
// operand on its way to the matrix pipe -- the activation is never written.  ONE formula everywhere (forward operand,
// the backward kernels' ReLU sign and the activation they hand to the weight gradient), so all of them see the same bits:
//     scale = gamma * invstd,  shift = fma(-mean, scale, beta),  a = max(fma(x, scale, shift), 0) rounded to bf16 (RNE)
struct BnOperand {
    const rgda_stat_t* stats;
    const float* gamma;
    const float* beta;
    float* mi;
    float* rm;
    float* rv;
    long long* nbt;
    float eps, mom;
    int groups, relu, C;
    int rows_per_group;        // rows of the operand map in one group
};
static __device__ __forceinline__ void bn_scale_shift(float mean, float istd, float gamma, float beta, float& sc, float& sh) {
    sc = gamma * istd;
    sh = __builtin_fmaf(-mean, sc, beta);
}
static __device__ __forceinline__ float bn_affine(float x, float sc, float sh) { return __builtin_fmaf(x, sc, sh); }
typedef float f32x2v __attribute__((ext_vector_type(2)));
// eight packed bf16 values -> relu(fma(x, sc, sh)) (relu optional), packed bf16 again
static __device__ __forceinline__ uint4 bn_operand8(uint4 v, const float (&sc)[8], const float (&sh)[8], bool relu) {
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f32x2v x = {__uint_as_float(w[e] << 16), __uint_as_float(w[e] & 0xffff0000u)};
        f32x2v s = {sc[2 * e], sc[2 * e + 1]}, h = {sh[2 * e], sh[2 * e + 1]};
        f32x2v f = __builtin_elementwise_fma(x, s, h);
        if (relu) { f.x = fmaxf(f.x, 0.f); f.y = fmaxf(f.y, 0.f); }
        w[e] = pack2bf(f.x, f.y);
    }
    return uint4{w[0], w[1], w[2], w[3]};
}
// Prologue of a consumer workgroup of NT threads: (scale, shift) of all C channels of row group `grp` -> tab[0..C),
// tab[C..2C) (LDS; the caller puts a barrier between this and the first use).  The `leader` workgroup of the launch also
// publishes (mean, invstd) of every group for the backward pass and updates the running statistics, group after group
// (the reference runs the source batch, then the target batch: tools/train_ssl_reg.py:210-212).
template <int NT>
static __device__ __forceinline__ void bn_operand_table(const BnOperand& b, int grp, bool leader, float* tab) {
    const int C = b.C;
    const double invM = 1.0 / (double)b.rows_per_group;
    for (int c = threadIdx.x; c < C; c += NT) {
        float mean, var, istd, sc, sh;
        stat_moments(b.stats + (size_t)grp * NREP * 2 * C, C, c, invM, b.eps, mean, var, istd);
        bn_scale_shift(mean, istd, b.gamma[c], b.beta[c], sc, sh);
        tab[c] = sc;
        tab[C + c] = sh;
        if (leader) {
            const float unb = (b.rows_per_group > 1) ? (float)b.rows_per_group / (float)(b.rows_per_group - 1) : 1.f;
            float rm = b.rm ? b.rm[c] : 0.f, rv = b.rm ? b.rv[c] : 0.f;
            for (int g = 0; g < b.groups; ++g) {
                float gm, gv, gi;
                stat_moments(b.stats + (size_t)g * NREP * 2 * C, C, c, invM, b.eps, gm, gv, gi);
                if (b.mi) { b.mi[(size_t)g * 2 * C + c] = gm; b.mi[(size_t)g * 2 * C + C + c] = gi; }
                rm = (1.f - b.mom) * rm + b.mom * gm;
                rv = (1.f - b.mom) * rv + b.mom * gv * unb;
            }
            if (b.rm) { b.rm[c] = rm; b.rv[c] = rv; }
            if (c == 0 && b.nbt) *b.nbt += b.groups;
        }
    }
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
// clear a workspace region from inside an entry point: the library's own fill kernel (rgda_fill_zero) where the pointer is
// 16-byte aligned -- hipMemsetAsync runs as one or two of the runtime's blit kernels per call
static inline int zero_bytes(void* p, size_t bytes, rgda_stream_t stream) {
    if (!((uintptr_t)p & 15)) return rgda_fill_zero(p, bytes, stream);
    return hipMemsetAsync(p, 0, bytes, (hipStream_t)stream) == hipSuccess ? RGDA_OK : RGDA_ERR_LAUNCH;
}
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
