// HBM-bound passes of the conv stack on PxC bf16 activations: BatchNorm (train) forward /
// backward, InstanceNorm, MaxPool, the spatial linear maps of the PPM head, the stem im2col
// and the 6-class classifier.  Every thread owns one 16-byte channel vector (8 bf16) and walks
// rows, so global accesses are 16 B per lane and row-contiguous; per-channel parameters live in
// registers; reductions are wave shuffle -> LDS -> one atomic per channel per workgroup.
#include <stdlib.h>
#include "common.h"

// thread layout shared by the per-channel passes: VPB channel-vectors per block (<= 256),
// RPB = 256 / VPB row lanes.
struct RowLayout {
    int vpr;     // vectors per row (C/8)
    int vpb;     // vectors handled per block (power of two <= 256)
    int rpb;     // row lanes per block
};
static RowLayout row_layout(int C) {
    RowLayout L;
    L.vpr = C / 8;
    int v = 1;
    while (v < L.vpr && v < 256) v <<= 1;
    L.vpb = v;
    L.rpb = 256 / v;
    return L;
}

// per-channel reductions: ~1024 workgroups, every thread sees at least 4 rows
static int reduce_rows_per_block(long long M, const RowLayout& L) {
    int ny = cdiv(L.vpr, L.vpb);
    long long want = 512 / ny;
    if (const char* e = TUNE_ENV("RGDA_RED_WANT")) want = atoi(e) / ny;     // tuning experiments only
    if (want < 1) want = 1;
    long long rows = (M + want - 1) / want;
    long long minrows = (long long)L.rpb * 4;
    if (rows < minrows) rows = minrows;
    rows = (rows + L.rpb - 1) / L.rpb * L.rpb;
    return (int)rows;
}

static __device__ __forceinline__ void load8(const bf16_t* p, float (&f)[8]) {
    u16x8 v = *(const u16x8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = bf2f(v[e]);
}
static __device__ __forceinline__ void cvt8(const u16x8& v, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = bf2f(v[e]);
}
// rows handled per thread per batch in the streaming kernels: all loads of a batch are issued before the first
// use, so every wave keeps ROW_BATCH (x up to 3 operands) 16-byte loads in flight.  Per kernel (whole-step A/B on one
// box, scripts/dev/multi_ab.sh): the forward apply is flat between 3 and 4 (8: +0.4 ms); the backward apply, which
// streams three operands, wants 2 (1: +0.25, 3: +0.1, 4: +0.25 ms -- the registers of a deeper batch cost occupancy)
#ifndef ROW_BATCH
#define ROW_BATCH 4
#endif
#ifndef RB_BWD
#define RB_BWD 2
#endif
#ifndef RB_APPLY
#define RB_APPLY ROW_BATCH
#endif
#ifndef RB_REDUCE
#define RB_REDUCE ROW_BATCH
#endif
#ifndef RB_BWD
#define RB_BWD ROW_BATCH
#endif
#ifndef RB_INORM
#define RB_INORM ROW_BATCH
#endif
#ifndef RB_MIX
#define RB_MIX ROW_BATCH
#endif
#ifndef RB_CLS
#define RB_CLS ROW_BATCH
#endif
// sign mask of 8 packed bf16 values: bit e = [value e > 0]
static __device__ __forceinline__ unsigned char relu_bits(const uint4& v) {
    auto pos = [](unsigned h) { return (unsigned)((h & 0x7fffu) != 0 && (h & 0x8000u) == 0); };
    return (unsigned char)(pos(v.x & 0xffffu) | pos(v.x >> 16) << 1 | pos(v.y & 0xffffu) << 2 | pos(v.y >> 16) << 3 |
                           pos(v.z & 0xffffu) << 4 | pos(v.z >> 16) << 5 | pos(v.w & 0xffffu) << 6 | pos(v.w >> 16) << 7);
}
static __device__ __forceinline__ void store8(bf16_t* p, const float (&f)[8]) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    *(uint4*)p = v;
}

// reduce two 8-vectors over the row lanes of a block (fixed order) and add them to out0[c], out1[c] as fixed-point
// integers (order-independent totals, common.h: stat_add)
static __device__ __forceinline__ void block_reduce_atomic(float (&s)[8], float (&q)[8], int cvl, int rl, int vpb,
                                                           int rpb, int cglobal, bool cok, rgda_stat_t* out0,
                                                           rgda_stat_t* out1, float* lds, int frac) {
    // lds: [rpb][vpb*8][2]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        lds[((rl * vpb + cvl) * 8 + e) * 2 + 0] = s[e];
        lds[((rl * vpb + cvl) * 8 + e) * 2 + 1] = q[e];
    }
    __syncthreads();
    if (rl == 0 && cok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.f, b = 0.f;
            for (int r = 0; r < rpb; ++r) {
                a += lds[((r * vpb + cvl) * 8 + e) * 2 + 0];
                b += lds[((r * vpb + cvl) * 8 + e) * 2 + 1];
            }
            stat_add(out0 + cglobal + e, a, frac);
            stat_add(out1 + cglobal + e, b, frac);
        }
    }
}

// ------------------------------------------------------------------ BN statistics (standalone)
__global__ void __launch_bounds__(256) bn_stats_kernel(const bf16_t* __restrict__ x, int ldx, rgda_stat_t* stats,
                                                       long long M, int C, int vpb, int rpb, int rows_per_block) {
    __shared__ float lds[256 * 16];
    const int cvl = threadIdx.x % vpb, rl = threadIdx.x / vpb;
    const int cg = (blockIdx.y * vpb + cvl) * 8;
    const bool cok = cg < C;
    float s[8] = {0}, q[8] = {0};
    long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    if (cok)
        for (long long r = r0 + rl; r < r1; r += rpb) {
            float f[8];
            load8(x + r * ldx + cg, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
        }
    rgda_stat_t* rep = stats + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) & (NREP - 1)) * 2 * C;
    block_reduce_atomic(s, q, cvl, rl, vpb, rpb, cg, cok, rep, rep + C, lds, RGDA_STAT_FRAC_FWD);
}

extern "C" int rgda_bn_stats(const void* x, int ldx, rgda_stat_t* stats, int64_t M, int C, rgda_stream_t stream) {
    if (!x || !stats || M <= 0 || C <= 0 || (C & 7) || (ldx & 7)) return RGDA_ERR_ARG;
    RowLayout L = row_layout(C);
    if (L.vpb > 16) { L.vpb = 16; L.rpb = 16; }          // fewer closing atomics per workgroup, see rgda_bn_bwd_reduce
    int rows_per_block = reduce_rows_per_block(M, L);
    dim3 grid(cdiv(M, rows_per_block), cdiv(L.vpr, L.vpb));
    bn_stats_kernel<<<grid, 256, 0, to_stream(stream)>>>((const bf16_t*)x, ldx, stats, M, C, L.vpb, L.rpb, rows_per_block);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// ------------------------------------------------------------------ BN finalize
__global__ void __launch_bounds__(256) bn_finalize_kernel(const rgda_stat_t* stats, float* mi, float* rm, float* rv,
                                                          long long* nbt, double M, int C, int groups, float eps,
                                                          float mom) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    for (int g = 0; g < groups; ++g) {          // group after group: the reference runs src then tgt
        const rgda_stat_t* st = stats ? stats + (size_t)g * NREP * 2 * C : nullptr;
        float* m = mi + (size_t)g * 2 * C;
        if (st) {
            float mean, var, istd;                  // biased variance (normalisation)
            stat_moments(st, C, c, 1.0 / M, eps, mean, var, istd);
            m[c] = mean;
            m[C + c] = istd;
            if (rm) {
                const float unb = (M > 1) ? (float)M / (float)(M - 1.0) : 1.f;   // unbiased (running update)
                rm[c] = (1.f - mom) * rm[c] + mom * mean;
                rv[c] = (1.f - mom) * rv[c] + mom * var * unb;
            }
            if (c == 0 && nbt) *nbt += 1;
        } else {
            m[c] = rm[c];
            m[C + c] = 1.f / sqrtf(rv[c] + eps);
        }
    }
}

extern "C" int rgda_bn_finalize(const rgda_stat_t* stats, float* mi, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, int64_t M, int C, int groups, float eps,
                                float momentum, rgda_stream_t stream) {
    if (!mi || C <= 0 || groups < 1 || (M % groups) || (!stats && (!running_mean || !running_var))) return RGDA_ERR_ARG;
    if (stats && M / groups < 2) return RGDA_ERR_ARG;   // "Expected more than 1 value per channel when training"
    bn_finalize_kernel<<<cdiv(C, 256), 256, 0, to_stream(stream)>>>(stats, mi, running_mean, running_var,
                                                                      (long long*)num_batches_tracked,
                                                                      (double)(M / groups), C, groups, eps, momentum);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// per-channel batch statistics of one row group from the replicated (sum, sumsq) accumulators
static __device__ __forceinline__ void group_stats(const rgda_stat_t* __restrict__ stats, int grp, int C, int cg, float invM,
                                                   float eps, float (&mean)[8], float (&istd)[8], float (&var)[8]) {
    const rgda_stat_t* st = stats + (size_t)grp * NREP * 2 * C;
#pragma unroll
    for (int e = 0; e < 8; ++e) stat_moments(st, C, cg + e, (double)invM, eps, mean[e], var[e], istd[e]);
}

// ------------------------------------------------------------------ BN apply (forward)
__global__ void __launch_bounds__(256) bn_apply_kernel(const bf16_t* __restrict__ x, int ldx,
                                                       const float* __restrict__ mi, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const bf16_t* __restrict__ res,
                                                       int ldres, const float* __restrict__ nscale, int rpi,
                                                       bf16_t* __restrict__ y, int ldy, long long M, int C, int relu,
                                                       int vpb, int rpb, int rows_per_block, int bpg,
                                                       const rgda_stat_t* __restrict__ stats, float* mi_out, float* rm,
                                                       float* rv, long long* nbt, int groups, float eps, float mom,
                                                       uint8_t* __restrict__ mask_out) {
    const int cvl = threadIdx.x % vpb, rl = threadIdx.x / vpb;
    const int cg = (blockIdx.y * vpb + cvl) * 8;
    const int grp = blockIdx.x / bpg, chunk = blockIdx.x % bpg;     // M = rows of ONE group
    // (vpb <= 16 from elementwise_grid: 128 channels per workgroup.  1 KB, not 16: a small footprint lets these
    // workgroups sit on a CU next to the other stream's 72-147 KB convolution workgroups)
    __shared__ float smean[128], sistd[128];
    if (stats) {
        // train mode with the finalize step folded in: the workgroup rebuilds mean / invstd of its vpb*8 channels
        // from the conv epilogue's replicated accumulators, one channel per thread, and shares them through LDS
        const double invM = 1.0 / (double)M;
        const rgda_stat_t* st = stats + (size_t)grp * NREP * 2 * C;
        for (int c = threadIdx.x; c < vpb * 8; c += 256) {
            int cc = blockIdx.y * vpb * 8 + c;
            if (cc < C) {
                float m, v0, is;
                stat_moments(st, C, cc, invM, eps, m, v0, is);
                smean[c] = m;
                sistd[c] = is;
                if (chunk == 0) {
                    // the first workgroup of a group's channel block also publishes (mean, invstd) for the backward pass;
                    // group 0's updates the running statistics, group after group (the reference runs src then tgt) --
                    // here, one channel per thread with all loads independent, not as a serial tail of a few threads
                    // (that tail, ~3 dependent memory round trips in one workgroup, was half the kernel on small maps)
                    mi_out[(size_t)grp * 2 * C + cc] = m;
                    mi_out[(size_t)grp * 2 * C + C + cc] = is;
                    if (grp == 0 && rm) {
                        const float unb = (M > 1) ? (float)M / (float)(M - 1) : 1.f;
                        float gm[8], gv[8];             // groups <= 8 (checked by the host)
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            if (g < groups) {
                                float gis;
                                stat_moments(stats + (size_t)g * NREP * 2 * C, C, cc, invM, eps, gm[g], gv[g], gis);
                            }
                        }
                        float a = rm[cc], b = rv[cc];
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            if (g < groups) {
                                a = (1.f - mom) * a + mom * gm[g];
                                b = (1.f - mom) * b + mom * gv[g] * unb;
                            }
                        }
                        rm[cc] = a;
                        rv[cc] = b;
                        if (cc == 0 && nbt) *nbt += groups;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (cg >= C) return;
    float mean[8], sc[8], sh[8];
    if (stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mean[e] = smean[cvl * 8 + e];
            sc[e] = sistd[cvl * 8 + e] * gamma[cg + e];
            sh[e] = beta[cg + e];
        }
    } else {
        mi += (size_t)grp * 2 * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mean[e] = mi[cg + e];
            sc[e] = mi[C + cg + e] * gamma[cg + e];
            sh[e] = beta[cg + e];
        }
    }
    long long r0 = (long long)grp * M + (long long)chunk * rows_per_block;
    long long r1 = min((long long)(grp + 1) * M, r0 + rows_per_block);
    for (long long rb = r0 + rl; rb < r1; rb += (long long)rpb * RB_APPLY) {
        u16x8 xv[RB_APPLY], rv[RB_APPLY];
#pragma unroll
        for (int u = 0; u < RB_APPLY; ++u) {
            const long long r = rb + (long long)u * rpb;
            if (r < r1) {
                xv[u] = *(const u16x8*)(x + r * ldx + cg);
                if (res) rv[u] = *(const u16x8*)(res + r * ldres + cg);
            }
        }
#pragma unroll
        for (int u = 0; u < RB_APPLY; ++u) {
            const long long r = rb + (long long)u * rpb;
            if (r >= r1) break;
            float f[8];
            cvt8(xv[u], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean[e]) * sc[e] + sh[e];
            if (res) {
                float g[8];
                cvt8(rv[u], g);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += g[e];
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
            }
            if (nscale) {
                const float* ns = nscale + (r / rpi) * C + cg;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] *= ns[e];
            }
            uint4 pk;
            pk.x = pack2bf(f[0], f[1]); pk.y = pack2bf(f[2], f[3]); pk.z = pack2bf(f[4], f[5]); pk.w = pack2bf(f[6], f[7]);
            *(uint4*)(y + r * ldy + cg) = pk;
            if (mask_out) mask_out[r * (C >> 3) + (cg >> 3)] = relu_bits(pk);
        }
    }
}

// M = rows of ONE group; grid.x = blocks-per-group * groups
static void elementwise_grid(long long M, int C, int groups, RowLayout& L, int& rows_per_block, int& bpg, dim3& grid,
                             int rows_mult = 8) {
    L = row_layout(C);
    // at most 128 channels (16 vectors) x 16 row lanes per workgroup: every workgroup first rebuilds (or loads) the
    // statistics of its channels, 16 floats each, which must stay small next to the rows it streams; measured on the
    // whole step (A/B on one box): cap 16 -> 21.23, 32 -> 21.30, 64 -> 21.3, none -> 21.70 ms
    if (L.vpb > 16) { L.vpb = 16; L.rpb = 16; }
    if (const char* e = TUNE_ENV("RGDA_BN_VPB")) {                  // tuning experiments only
        int v = atoi(e);
        RowLayout full = row_layout(C);                             // (above 16: the paths WITHOUT the statistics prologue only)
        if (v <= full.vpb) { L.vpb = v; L.rpb = 256 / v; }
    }
    if (const char* e = TUNE_ENV("RGDA_BN_ROWS")) rows_mult = atoi(e);   // tuning experiments only
    rows_per_block = L.rpb * rows_mult;
    while ((long long)cdiv(M, rows_per_block) * groups * cdiv(L.vpr, L.vpb) > 8192) rows_per_block *= 2;
    bpg = cdiv(M, rows_per_block);
    grid = dim3(bpg * groups, cdiv(L.vpr, L.vpb));
}

extern "C" int rgda_bn_apply(const void* x, int ldx, const float* mi, const float* gamma, const float* beta,
                             const void* res, int ldres, const float* nscale, int rows_per_image, void* y, int ldy,
                             int64_t M, int C, int relu, int groups, rgda_stream_t stream) {
    if (!x || !mi || !gamma || !beta || !y || M <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldy & 7)) return RGDA_ERR_ARG;
    if (res && (ldres & 7)) return RGDA_ERR_ARG;
    if (nscale && rows_per_image <= 0) return RGDA_ERR_ARG;
    if (groups < 1 || (M % groups)) return RGDA_ERR_ARG;
    RowLayout L; int rpbk, bpg; dim3 grid;
    elementwise_grid(M / groups, C, groups, L, rpbk, bpg, grid);
    bn_apply_kernel<<<grid, 256, 0, to_stream(stream)>>>((const bf16_t*)x, ldx, mi, gamma, beta, (const bf16_t*)res,
                                                          ldres, nscale, rows_per_image, (bf16_t*)y, ldy, M / groups, C,
                                                          relu, L.vpb, L.rpb, rpbk, bpg, nullptr, nullptr, nullptr,
                                                          nullptr, nullptr, groups, 0.f, 0.f, nullptr);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

extern "C" int rgda_bn_train_apply(const void* x, int ldx, const rgda_stat_t* stats, float* mi, float* running_mean,
                                   float* running_var, int64_t* num_batches_tracked, const float* gamma,
                                   const float* beta, const void* res, int ldres, const float* nscale,
                                   int rows_per_image, void* y, int ldy, uint8_t* relu_mask, int64_t M, int C,
                                   int relu, int groups, float eps, float momentum, rgda_stream_t stream) {
    if (!x || !stats || !mi || !gamma || !beta || !y || M <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldy & 7)) return RGDA_ERR_ARG;
    if (relu_mask && !relu) return RGDA_ERR_ARG;
    if (res && (ldres & 7)) return RGDA_ERR_ARG;
    if (nscale && rows_per_image <= 0) return RGDA_ERR_ARG;
    if (groups < 1 || groups > 8 || (M % groups) || M / groups < 2) return RGDA_ERR_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return RGDA_ERR_ARG;
    RowLayout L; int rpbk, bpg; dim3 grid;
    // fatter workgroups than the plain apply: each one first rebuilds its channels' statistics (128 loads/thread)
    elementwise_grid(M / groups, C, groups, L, rpbk, bpg, grid);
    bn_apply_kernel<<<grid, 256, 0, to_stream(stream)>>>((const bf16_t*)x, ldx, nullptr, gamma, beta, (const bf16_t*)res,
                                                          ldres, nscale, rows_per_image, (bf16_t*)y, ldy, M / groups, C,
                                                          relu, L.vpb, L.rpb, rpbk, bpg, stats, mi, running_mean,
                                                          running_var, (long long*)num_batches_tracked, groups, eps,
                                                          momentum, relu_mask);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// ------------------------------------------------------------------ BN backward
template <bool FROM_X>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const bf16_t* __restrict__ g, int ldg,
                                                            const bf16_t* __restrict__ y, int ldy,
                                                            const uint8_t* __restrict__ rmask,
                                                            const bf16_t* __restrict__ x, int ldx,
                                                            const float* __restrict__ mi, const float* __restrict__ nscale,
                                                            int rpi, rgda_stat_t* sums, long long M, int C, int relu, int vpb,
                                                            int rpb, int rows_per_block, int bpg,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta) {
    __shared__ float lds[256 * 16];
    const int cvl = threadIdx.x % vpb, rl = threadIdx.x / vpb;
    const int cg = (blockIdx.y * vpb + cvl) * 8;
    const bool cok = cg < C;
    const int grp = blockIdx.x / bpg, chunk = blockIdx.x % bpg;     // M = rows of ONE group
    mi += (size_t)grp * 2 * C;
    sums += (size_t)grp * NREP * 2 * C;
    float s[8] = {0}, q[8] = {0};
    if (cok) {
        float mean[8], istd[8], fsc[8], fsh[8];      // fsc / fsh: the forward operand path's (scale, shift), relu == 2 only
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mean[e] = mi[cg + e]; istd[e] = mi[C + cg + e];
            fsc[e] = 0.f; fsh[e] = 0.f;
            if constexpr (FROM_X) bn_scale_shift(mean[e], istd[e], gamma[cg + e], beta[cg + e], fsc[e], fsh[e]);
        }
        long long r0 = (long long)grp * M + (long long)chunk * rows_per_block;
        long long r1 = min((long long)(grp + 1) * M, r0 + rows_per_block);
        for (long long rb = r0 + rl; rb < r1; rb += (long long)rpb * RB_REDUCE) {
            u16x8 gv[RB_REDUCE], xv[RB_REDUCE], yv[RB_REDUCE];
            unsigned mb[RB_REDUCE];
#pragma unroll
            for (int u = 0; u < RB_REDUCE; ++u) {
                const long long r = rb + (long long)u * rpb;
                if (r < r1) {
                    gv[u] = *(const u16x8*)(g + r * ldg + cg);
                    xv[u] = *(const u16x8*)(x + r * ldx + cg);
                    if (!FROM_X && relu) {
                        if (rmask) mb[u] = rmask[r * (C >> 3) + (cg >> 3)];
                        else yv[u] = *(const u16x8*)(y + r * ldy + cg);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < RB_REDUCE; ++u) {
                const long long r = rb + (long long)u * rpb;
                if (r >= r1) break;
                float gf[8], xf[8];
                cvt8(gv[u], gf);
                cvt8(xv[u], xf);
                if constexpr (FROM_X) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) gf[e] = (bn_affine(xf[e], fsc[e], fsh[e]) > 0.f) ? gf[e] : 0.f;
                } else if (relu) {
                    if (rmask) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) gf[e] = ((mb[u] >> e) & 1u) ? gf[e] : 0.f;
                    } else {
                        float yf[8];
                        cvt8(yv[u], yf);
#pragma unroll
                        for (int e = 0; e < 8; ++e) gf[e] = (yf[e] > 0.f) ? gf[e] : 0.f;
                    }
                }
                if (nscale) {
                    const float* ns = nscale + (r / rpi) * C + cg;
#pragma unroll
                    for (int e = 0; e < 8; ++e) gf[e] *= ns[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += gf[e]; q[e] += gf[e] * ((xf[e] - mean[e]) * istd[e]); }
            }
        }
    }
    rgda_stat_t* rep = sums + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) & (NREP - 1)) * 2 * C;
    block_reduce_atomic(s, q, cvl, rl, vpb, rpb, cg, cok, rep, rep + C, lds, RGDA_STAT_FRAC_BWD);
}

extern "C" int rgda_bn_bwd_reduce(const void* g, int ldg, const void* y, int ldy, const uint8_t* relu_mask,
                                  const void* x, int ldx, const float* mi, const float* nscale, int rows_per_image,
                                  rgda_stat_t* sums, int64_t M, int C, int relu, const float* gamma, const float* beta,
                                  int groups, rgda_stream_t stream) {
    if (!g || !x || !mi || !sums || (relu == 1 && !y && !relu_mask) || M <= 0 || C <= 0 || (C & 7) || (ldg & 7) || (ldx & 7)) return RGDA_ERR_ARG;
    if (relu < 0 || relu > 2 || (relu == 2 && (!gamma || !beta))) return RGDA_ERR_ARG;
    if (groups < 1 || (M % groups)) return RGDA_ERR_ARG;
    hipStream_t st = to_stream(stream);
    RowLayout L = row_layout(C);
    // 128 channels x 16 row lanes per workgroup: the atomics that end a workgroup are (workgroups x channels each), so
    // a workgroup that spans a whole 2048-channel row issues 8 x as many as sixteen that split it.  Measured
    // (scripts/dev/dev_bnred.py, M = 16384, C = 2048): full rows 74 us, 64-vector cap 41, 16-vector cap 30 us
    if (L.vpb > 16) { L.vpb = 16; L.rpb = 16; }
    if (const char* e = TUNE_ENV("RGDA_RED_VPB")) { int v = atoi(e); if (v < L.vpb) { L.vpb = v; L.rpb = 256 / v; } }   // tuning only
    const long long Mg = M / groups;
    int rows_per_block = reduce_rows_per_block(Mg * groups, L);
    if (rows_per_block > Mg) rows_per_block = (int)((Mg + L.rpb - 1) / L.rpb * L.rpb);
    int bpg = cdiv(Mg, rows_per_block);
    dim3 grid(bpg * groups, cdiv(L.vpr, L.vpb));
    if (relu == 2)
        bn_bwd_reduce_kernel<true><<<grid, 256, 0, st>>>((const bf16_t*)g, ldg, (const bf16_t*)y, ldy, relu_mask, (const bf16_t*)x, ldx,
                                                         mi, nscale, rows_per_image, sums, Mg, C, relu, L.vpb, L.rpb,
                                                         rows_per_block, bpg, gamma, beta);
    else
        bn_bwd_reduce_kernel<false><<<grid, 256, 0, st>>>((const bf16_t*)g, ldg, (const bf16_t*)y, ldy, relu_mask, (const bf16_t*)x, ldx,
                                                          mi, nscale, rows_per_image, sums, Mg, C, relu, L.vpb, L.rpb,
                                                          rows_per_block, bpg, gamma, beta);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

template <bool FROM_X>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const bf16_t* __restrict__ g, int ldg,
                                                           const bf16_t* __restrict__ y, int ldy,
                                                           const uint8_t* __restrict__ rmask,
                                                           const bf16_t* __restrict__ x, int ldx,
                                                           const float* __restrict__ mi, const float* __restrict__ gamma,
                                                           const float* __restrict__ nscale, int rpi,
                                                           const rgda_stat_t* __restrict__ sums, bf16_t* __restrict__ dx,
                                                           int lddx, bf16_t* __restrict__ gmask, int ldgm, float* dgamma,
                                                           float* dbeta, long long M, int C, int relu, int vpb, int rpb,
                                                           int rows_per_block, int bpg, const float* __restrict__ beta,
                                                           bf16_t* __restrict__ act_out, int ldact) {
    const int cvl = threadIdx.x % vpb, rl = threadIdx.x / vpb;
    const int cg = (blockIdx.y * vpb + cvl) * 8;
    const int grp = blockIdx.x / bpg, chunk = blockIdx.x % bpg;     // M = rows of ONE group
    mi += (size_t)grp * 2 * C;
    sums += (size_t)grp * NREP * 2 * C;
    // per-channel constants of the workgroup's vpb * 8 channels: one channel per thread (coalesced across threads; the
    // 2 * NREP replicated partial sums are 16 loads per CHANNEL, not per thread: 16 row lanes used to fetch the same 38
    // vectors each, more load instructions than the rows a workgroup streams), shared through LDS
    extern __shared__ __attribute__((aligned(16))) float sk_dyn[];      // [6][vpb * 8]
    const int nch = vpb * 8;
    float* const sk0 = sk_dyn;          // mean, istd, k0 = gamma * istd, k1 = sum(g') / M, k2 = sum(g' xhat) / M,
                                        // (relu == 2) the forward operand path's shift (its scale is k0)
    const float invM = 1.f / (float)M;
    for (int c = threadIdx.x; c < vpb * 8; c += 256) {
        const int cc = blockIdx.y * vpb * 8 + c;
        if (cc < C) {
            const float t1 = (float)stat_total(sums, C, cc, 0, RGDA_STAT_FRAC_BWD);
            const float t2 = (float)stat_total(sums, C, cc, 1, RGDA_STAT_FRAC_BWD);
            const float is = mi[C + cc];
            sk0[0 * nch + c] = mi[cc];
            sk0[1 * nch + c] = is;
            sk0[2 * nch + c] = gamma[cc] * is;
            sk0[3 * nch + c] = t1 * invM;
            sk0[4 * nch + c] = t2 * invM;
            if constexpr (FROM_X) {
                float fsc, fsh;
                bn_scale_shift(mi[cc], is, gamma[cc], beta[cc], fsc, fsh);      // fsc == sk0[2][c]: the same product
                sk0[5 * nch + c] = fsh;
            }
            if (chunk == 0 && grp == 0 && dgamma) {
                // ONE workgroup per channel block folds the sums of every group into the parameter gradients, group
                // after group (a fixed order: nothing depends on which group's workgroup retires first).  The add itself
                // is the fire-and-forget atomic: a plain read-modify-write puts two dependent memory round trips in front
                // of this workgroup's barrier (+0.7 ms per step when the weight gradients keep the memory system busy)
                const int groups = gridDim.x / bpg;
                float dg = t2, db = t1;
                for (int g2 = 1; g2 < groups; ++g2) {
                    dg += (float)stat_total(sums + (size_t)g2 * NREP * 2 * C, C, cc, 1, RGDA_STAT_FRAC_BWD);
                    db += (float)stat_total(sums + (size_t)g2 * NREP * 2 * C, C, cc, 0, RGDA_STAT_FRAC_BWD);
                }
                atomicAdd(dgamma + cc, dg);
                atomicAdd(dbeta + cc, db);
            }
        }
    }
    __syncthreads();
    if (cg >= C) return;
    float mean[8], istd[8], k0[8], k1[8], k2[8], fsh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mean[e] = sk0[0 * nch + cvl * 8 + e];
        istd[e] = sk0[1 * nch + cvl * 8 + e];
        k0[e] = sk0[2 * nch + cvl * 8 + e];
        k1[e] = sk0[3 * nch + cvl * 8 + e];
        k2[e] = sk0[4 * nch + cvl * 8 + e];
        if constexpr (FROM_X) fsh[e] = sk0[5 * nch + cvl * 8 + e]; else fsh[e] = 0.f;
    }
    long long r0 = (long long)grp * M + (long long)chunk * rows_per_block;
    long long r1 = min((long long)(grp + 1) * M, r0 + rows_per_block);
    for (long long rb = r0 + rl; rb < r1; rb += (long long)rpb * RB_BWD) {
        u16x8 gv[RB_BWD], xv[RB_BWD], yv[RB_BWD];
        unsigned mb[RB_BWD];
#pragma unroll
        for (int u = 0; u < RB_BWD; ++u) {
            const long long r = rb + (long long)u * rpb;
            if (r < r1) {
                gv[u] = *(const u16x8*)(g + r * ldg + cg);
                xv[u] = *(const u16x8*)(x + r * ldx + cg);
                if (!FROM_X && relu) {
                    if (rmask) mb[u] = rmask[r * (C >> 3) + (cg >> 3)];
                    else yv[u] = *(const u16x8*)(y + r * ldy + cg);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RB_BWD; ++u) {
            const long long r = rb + (long long)u * rpb;
            if (r >= r1) break;
            float gf[8], xf[8];
            cvt8(gv[u], gf);
            cvt8(xv[u], xf);
            if constexpr (FROM_X) {
                // the unit's activation was never written (it ran on its consumer's operand path): its ReLU sign, and
                // the activation itself for the consumer's weight gradient, are recomputed with the forward's own formula
                float af[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    af[e] = bn_affine(xf[e], k0[e], fsh[e]);
                    gf[e] = (af[e] > 0.f) ? gf[e] : 0.f;
                    af[e] = fmaxf(af[e], 0.f);
                }
                if (act_out) store8(act_out + r * ldact + cg, af);
            } else if (relu) {
                if (rmask) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) gf[e] = ((mb[u] >> e) & 1u) ? gf[e] : 0.f;
                } else {
                    float yf[8];
                    cvt8(yv[u], yf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) gf[e] = (yf[e] > 0.f) ? gf[e] : 0.f;
                }
            }
            if (nscale) {
                const float* ns = nscale + (r / rpi) * C + cg;
#pragma unroll
                for (int e = 0; e < 8; ++e) gf[e] *= ns[e];
            }
            if (gmask) store8(gmask + r * ldgm + cg, gf);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = k0[e] * (gf[e] - k1[e] - (xf[e] - mean[e]) * istd[e] * k2[e]);
            store8(dx + r * lddx + cg, o);
        }
    }
}

extern "C" int rgda_bn_bwd_apply(const void* g, int ldg, const void* y, int ldy, const uint8_t* relu_mask,
                                 const void* x, int ldx,
                                 const float* mi, const float* gamma, const float* nscale, int rows_per_image,
                                 const rgda_stat_t* sums, void* dx, int lddx, void* gmask, int ldgm, float* dgamma,
                                 float* dbeta, int64_t M, int C, int relu, const float* beta, void* act_out, int ldact,
                                 int groups, rgda_stream_t stream) {
    if (!g || !x || !mi || !gamma || !sums || !dx || (relu == 1 && !y && !relu_mask) || M <= 0 || C <= 0 || (C & 7)) return RGDA_ERR_ARG;
    if (relu < 0 || relu > 2 || (relu == 2 && !beta) || (act_out && (relu != 2 || (ldact & 7) || ldact < C))) return RGDA_ERR_ARG;
    if ((ldg & 7) || (ldx & 7) || (lddx & 7) || (gmask && (ldgm & 7)) || ((dgamma == nullptr) != (dbeta == nullptr)))
        return RGDA_ERR_ARG;
    if (groups < 1 || (M % groups)) return RGDA_ERR_ARG;
    RowLayout L; int rpbk, bpg; dim3 grid;
    elementwise_grid(M / groups, C, groups, L, rpbk, bpg, grid);
    if (relu == 2)
        bn_bwd_apply_kernel<true><<<grid, 256, (size_t)6 * L.vpb * 8 * sizeof(float), to_stream(stream)>>>(
            (const bf16_t*)g, ldg, (const bf16_t*)y, ldy, relu_mask, (const bf16_t*)x, ldx, mi, gamma, nscale, rows_per_image, sums,
            (bf16_t*)dx, lddx, (bf16_t*)gmask, ldgm, dgamma, dbeta, M / groups, C, relu, L.vpb, L.rpb, rpbk, bpg, beta,
            (bf16_t*)act_out, ldact);
    else
        bn_bwd_apply_kernel<false><<<grid, 256, (size_t)6 * L.vpb * 8 * sizeof(float), to_stream(stream)>>>(
            (const bf16_t*)g, ldg, (const bf16_t*)y, ldy, relu_mask, (const bf16_t*)x, ldx, mi, gamma, nscale, rows_per_image, sums,
            (bf16_t*)dx, lddx, (bf16_t*)gmask, ldgm, dgamma, dbeta, M / groups, C, relu, L.vpb, L.rpb, rpbk, bpg, beta,
            (bf16_t*)act_out, ldact);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// ------------------------------------------------------------------ BatchNorm of SMALL maps, several layers per launch
// The PPM branches (regda/models/Encoder.py:30-51): conv -> BatchNorm -> ReLU on s x s maps, s = 1, 2, 3, 6 -- 8 to 288 rows
// per statistics group and 512 channels.  As the general kernels these are 4 forward launches and 8 backward launches per
// head of 6 - 8 us each, every one a statistics hand-off through memory for a tensor of a few hundred rows.  Here ONE
// workgroup owns 64 channels of one layer and ALL of its rows: it forms the statistics itself (fixed summation order: run
// to run identical), applies them, and up to eight layers share a launch.  Same formulas as the general kernels
// (var = E[x^2] - mean^2 in double, invstd in fp32, y = (x - mean) * (invstd * gamma) + beta; dx = gamma * invstd *
// (g' - mean(g') - xhat * mean(g' xhat))); the statistics are sums of the STORED bf16 values, as the convolution epilogue's are.
constexpr int BN_SMALL_MAX = 8;
struct BnSmallFwd {
    const bf16_t* x; bf16_t* y; uint8_t* mask_out;
    const float* gamma; const float* beta; float* mi; float* rm; float* rv; long long* nbt;
    int ldx, ldy, M /* rows of one group */, C, groups, relu;
    float eps, mom;
};
struct BnSmallBwd {
    const bf16_t* g; const bf16_t* y; const uint8_t* rmask; const bf16_t* x; bf16_t* dx;
    const float* mi; const float* gamma; float* dgamma; float* dbeta;
    int ldg, ldy, ldx, lddx, M /* rows of one group */, C, groups, relu;
};
template <class A> struct BnSmallGroup { A a[BN_SMALL_MAX]; int start[BN_SMALL_MAX + 1]; int n; };

// A workgroup is 512 threads = 8 channel vectors (64 channels) x 64 row lanes and keeps ITS rows in registers (packed bf16,
// BN_SMALL_RPT rows per thread and group, TWO groups at a time, every load in flight at once): memory is read once, the
// statistics and the apply both run from registers, and the per-channel parameters are fetched next to the rows -- one
// memory round trip in front of the reduction, none behind it.  (First form: 256 threads walking the rows 16 at a time,
// twice, group after group: 45 / 59 us per launch, a chain of ~100 dependent load round trips for the 6 x 6 maps; rows in
// registers but group after group with the parameters loaded behind the reduction: 28 us.)
constexpr int BN_SMALL_RPT = 5;                         // rows per thread: 64 x 5 = 320 rows of one group
constexpr int BN_SMALL_LANES = 64;                      // row lanes
constexpr int BN_SMALL_CV = 8;                          // channel vectors (of 8 channels) per workgroup
constexpr int BN_SMALL_NT = BN_SMALL_CV * BN_SMALL_LANES, BN_SMALL_WAVES = BN_SMALL_NT / 64;
constexpr int BN_SMALL_ROWS = BN_SMALL_LANES * BN_SMALL_RPT;
constexpr int BN_SMALL_GP = 2;                          // groups resident at a time

// totals of this thread's eight channels over the row lanes: the eight row lanes of a wave by three fixed shuffles, then
// the waves through LDS, every thread summing in wave order.  NP planes at once (one barrier pair for all of them).
template <int NP>
static __device__ __forceinline__ void small_totals(float (*red)[BN_SMALL_WAVES][BN_SMALL_CV * 8], float (&v)[NP][8], int cvl,
                                                    int wave, int lane) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = v[p][e];
            x += __shfl_xor(x, 8, 64);
            x += __shfl_xor(x, 16, 64);
            x += __shfl_xor(x, 32, 64);
            v[p][e] = x;
        }
    __syncthreads();                    // the previous pass's readers are done
    if (lane < BN_SMALL_CV) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[p][wave][cvl * 8 + e] = v[p][e];
    }
    __syncthreads();
}
// (the totals stay in LDS and are summed where they are used: 2 x groups x 8 doubles per thread held in registers next to
// the resident rows made both kernels spill)
static __device__ __forceinline__ double small_total(float (*red)[BN_SMALL_WAVES][BN_SMALL_CV * 8], int plane, int c) {
    double x = 0.0;
#pragma unroll
    for (int l = 0; l < BN_SMALL_WAVES; ++l) x += (double)red[plane][l][c];
    return x;
}

__global__ void __launch_bounds__(BN_SMALL_NT) bn_small_fwd_kernel(BnSmallGroup<BnSmallFwd> G) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < BN_SMALL_MAX; ++i)
        if (i < G.n && (int)blockIdx.x >= G.start[i]) p = i;
    const BnSmallFwd& a = G.a[p];
    const int cb = (int)blockIdx.x - G.start[p];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cvl = threadIdx.x & (BN_SMALL_CV - 1), rl = threadIdx.x / BN_SMALL_CV;
    const int cg = (cb * BN_SMALL_CV + cvl) * 8;
    const bool cok = cg < a.C;
    __shared__ float red[2 * BN_SMALL_GP][BN_SMALL_WAVES][BN_SMALL_CV * 8];
    const int C = a.C, M = a.M;
    float run_m[8], run_v[8], gam[8], bet[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        run_m[e] = (cok && a.rm) ? a.rm[cg + e] : 0.f; run_v[e] = (cok && a.rm) ? a.rv[cg + e] : 0.f;
        gam[e] = cok ? a.gamma[cg + e] : 0.f; bet[e] = cok ? a.beta[cg + e] : 0.f;
    }
    const float unb = (M > 1) ? (float)M / (float)(M - 1) : 1.f;
    const double invMd = 1.0 / (double)M;       // (as the general kernels: sums times 1 / M)
    for (int g0 = 0; g0 < a.groups; g0 += BN_SMALL_GP) {
        u16x8 xv[BN_SMALL_GP][BN_SMALL_RPT];
#pragma unroll
        for (int gi = 0; gi < BN_SMALL_GP; ++gi)
#pragma unroll
            for (int u = 0; u < BN_SMALL_RPT; ++u) {
                const int r = rl + BN_SMALL_LANES * u;
                if (cok && g0 + gi < a.groups && r < M) xv[gi][u] = *(const u16x8*)(a.x + (unsigned)(((g0 + gi) * M + r) * a.ldx + cg));
            }
        float sq[2 * BN_SMALL_GP][8];
#pragma unroll
        for (int gi = 0; gi < BN_SMALL_GP; ++gi) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sq[2 * gi][e] = 0.f; sq[2 * gi + 1][e] = 0.f; }
#pragma unroll
            for (int u = 0; u < BN_SMALL_RPT; ++u) {
                if (cok && g0 + gi < a.groups && rl + BN_SMALL_LANES * u < M) {
                    float f[8];
                    cvt8(xv[gi][u], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { sq[2 * gi][e] += f[e]; sq[2 * gi + 1][e] += f[e] * f[e]; }
                }
            }
        }
        small_totals<2 * BN_SMALL_GP>(red, sq, cvl, wave, lane);
        if (!cok) continue;
#pragma unroll
        for (int gi = 0; gi < BN_SMALL_GP; ++gi)
#pragma unroll
            for (int u = 0; u < BN_SMALL_RPT; ++u) asm volatile("" : "+v"(xv[gi][u]));      // (re-converted below, not kept as floats)
#pragma unroll
        for (int gi = 0; gi < BN_SMALL_GP; ++gi) {
            const int grp = g0 + gi;
            if (grp >= a.groups) break;
            float mean[8], sc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double m = small_total(red, 2 * gi, cvl * 8 + e) * invMd;
                double v = small_total(red, 2 * gi + 1, cvl * 8 + e) * invMd - m * m;
                if (v < 0.0) v = 0.0;
                const float is = 1.f / sqrtf((float)v + a.eps);
                mean[e] = (float)m;
                sc[e] = is * gam[e];
                run_m[e] = (1.f - a.mom) * run_m[e] + a.mom * (float)m;
                run_v[e] = (1.f - a.mom) * run_v[e] + a.mom * (float)v * unb;
                if (rl == 0) { a.mi[(size_t)grp * 2 * C + cg + e] = (float)m; a.mi[(size_t)grp * 2 * C + C + cg + e] = is; }
            }
#pragma unroll
            for (int u = 0; u < BN_SMALL_RPT; ++u) {
                const int r = rl + BN_SMALL_LANES * u;
                if (r >= M) break;
                float f[8];
                cvt8(xv[gi][u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f[e] = (f[e] - mean[e]) * sc[e] + bet[e];
                    if (a.relu) f[e] = fmaxf(f[e], 0.f);
                }
                uint4 pk;
                pk.x = pack2bf(f[0], f[1]); pk.y = pack2bf(f[2], f[3]); pk.z = pack2bf(f[4], f[5]); pk.w = pack2bf(f[6], f[7]);
                const int rr = grp * M + r;
                *(uint4*)(a.y + (unsigned)(rr * a.ldy + cg)) = pk;
                if (a.mask_out) a.mask_out[(unsigned)(rr * (C >> 3) + (cg >> 3))] = relu_bits(pk);
            }
        }
    }
    if (cok && rl == 0 && a.rm) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { a.rm[cg + e] = run_m[e]; a.rv[cg + e] = run_v[e]; }
    }
    if (cb == 0 && threadIdx.x == 0 && a.nbt) *a.nbt += a.groups;
}

// USE_Y: the ReLU gate is read from y (one more resident row set) instead of from the sign mask
template <bool USE_Y>
__global__ void __launch_bounds__(BN_SMALL_NT) bn_small_bwd_kernel(BnSmallGroup<BnSmallBwd> G) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < BN_SMALL_MAX; ++i)
        if (i < G.n && (int)blockIdx.x >= G.start[i]) p = i;
    const BnSmallBwd& a = G.a[p];
    const int cb = (int)blockIdx.x - G.start[p];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cvl = threadIdx.x & (BN_SMALL_CV - 1), rl = threadIdx.x / BN_SMALL_CV;
    const int cg = (cb * BN_SMALL_CV + cvl) * 8;
    const bool cok = cg < a.C;
    __shared__ float red[2 * BN_SMALL_GP][BN_SMALL_WAVES][BN_SMALL_CV * 8];
    const int C = a.C, M = a.M;
    float dg[8] = {0}, db[8] = {0}, gam[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gam[e] = cok ? a.gamma[cg + e] : 0.f;
    for (int g0 = 0; g0 < a.groups; g0 += BN_SMALL_GP) {
        float mean[BN_SMALL_GP][8], istd[BN_SMALL_GP][8];
        u16x8 gv[BN_SMALL_GP][BN_SMALL_RPT], xv[BN_SMALL_GP][BN_SMALL_RPT], yv[USE_Y ? BN_SMALL_GP : 1][USE_Y ? BN_SMALL_RPT : 1];
        unsigned mb[BN_SMALL_GP][BN_SMALL_RPT];
#pragma unroll
        for (int gi = 0; gi < BN_SMALL_GP; ++gi) {
            const bool gok = cok && g0 + gi < a.groups;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                mean[gi][e] = gok ? a.mi[(size_t)(g0 + gi) * 2 * C + cg + e] : 0.f;
                istd[gi][e] = gok ? a.mi[(size_t)(g0 + gi) * 2 * C + C + cg + e] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < BN_SMALL_RPT; ++u) {
                const int r = (g0 + gi) * M + rl + BN_SMALL_LANES * u;      // (32-bit offsets: a small map)
                mb[gi][u] = 0xffu;
                if (gok && rl + BN_SMALL_LANES * u < M) {
                    gv[gi][u] = *(const u16x8*)(a.g + (unsigned)(r * a.ldg + cg));
                    xv[gi][u] = *(const u16x8*)(a.x + (unsigned)(r * a.ldx + cg));
                    if (a.relu) {
                        if constexpr (USE_Y) yv[gi][u] = *(const u16x8*)(a.y + (unsigned)(r * a.ldy + cg));
                        else mb[gi][u] = a.rmask[(unsigned)(r * (C >> 3) + (cg >> 3))];
                    }
                }
            }
        }
        auto row = [&](int gi, int u, float (&gf)[8], float (&xh)[8]) {       // g' and xhat of a register row
            float xf[8];
            cvt8(gv[gi][u], gf);
            cvt8(xv[gi][u], xf);
            if constexpr (USE_Y) {
                if (a.relu) {
                    float yf[8];
                    cvt8(yv[gi][u], yf);
#pragma unroll
                    for (int e = 0; e < 8; ++e) gf[e] = (yf[e] > 0.f) ? gf[e] : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) gf[e] = ((mb[gi][u] >> e) & 1u) ? gf[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) xh[e] = (xf[e] - mean[gi][e]) * istd[gi][e];
        };
        float sq[2 * BN_SMALL_GP][8];
#pragma unroll
        for (int gi = 0; gi < BN_SMALL_GP; ++gi) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sq[2 * gi][e] = 0.f; sq[2 * gi + 1][e] = 0.f; }
#pragma unroll
            for (int u = 0; u < BN_SMALL_RPT; ++u) {
                if (cok && g0 + gi < a.groups && rl + BN_SMALL_LANES * u < M) {
                    float gf[8], xh[8];
                    row(gi, u, gf, xh);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { sq[2 * gi][e] += gf[e]; sq[2 * gi + 1][e] += gf[e] * xh[e]; }
                }
            }
        }
        small_totals<2 * BN_SMALL_GP>(red, sq, cvl, wave, lane);
        if (!cok) continue;
        // the apply RE-converts the packed rows: left to common-subexpression elimination, the 2 x 5 x 16 floats of the
        // first pass stay live across the reduction and the kernel spills
#pragma unroll
        for (int gi = 0; gi < BN_SMALL_GP; ++gi)
#pragma unroll
            for (int u = 0; u < BN_SMALL_RPT; ++u) asm volatile("" : "+v"(gv[gi][u]), "+v"(xv[gi][u]));
        const float invM = 1.f / (float)M;
#pragma unroll
        for (int gi = 0; gi < BN_SMALL_GP; ++gi) {
            if (g0 + gi >= a.groups) break;
            float k0[8], k1[8], k2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t1 = (float)small_total(red, 2 * gi, cvl * 8 + e), t2 = (float)small_total(red, 2 * gi + 1, cvl * 8 + e);
                k0[e] = gam[e] * istd[gi][e];
                k1[e] = t1 * invM;
                k2[e] = t2 * invM;
                db[e] += t1;
                dg[e] += t2;
            }
#pragma unroll
            for (int u = 0; u < BN_SMALL_RPT; ++u) {
                if (rl + BN_SMALL_LANES * u >= M) break;
                float gf[8], xh[8], o[8];
                row(gi, u, gf, xh);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = k0[e] * (gf[e] - k1[e] - xh[e] * k2[e]);
                store8(a.dx + (unsigned)(((g0 + gi) * M + rl + BN_SMALL_LANES * u) * a.lddx + cg), o);
            }
        }
    }
    if (cok && rl == 0 && a.dgamma) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { atomicAdd(a.dgamma + cg + e, dg[e]); atomicAdd(a.dbeta + cg + e, db[e]); }
    }
}

// rows of one group the small kernels accept (BN_SMALL_LANES row lanes x BN_SMALL_RPT rows in registers)
#define RGDA_BN_SMALL_MAX_ROWS BN_SMALL_ROWS

extern "C" int rgda_bn_train_small(const rgda_bn_small_fwd_desc* descs, int n, rgda_stream_t stream) {
    if (!descs || n < 0) return RGDA_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        const rgda_bn_small_fwd_desc& d = descs[i];
        if (!d.x || !d.y || !d.gamma || !d.beta || !d.mi || d.M <= 0 || d.C <= 0 || (d.C & 7) || (d.ldx & 7) || (d.ldy & 7) ||
            d.ldx < d.C || d.ldy < d.C || d.groups < 1 || d.groups > 8 || (d.M % d.groups) || d.M / d.groups < 2 ||
            d.M / d.groups > RGDA_BN_SMALL_MAX_ROWS || (d.relu_mask && !d.relu) || d.M * (int64_t)(d.ldx > d.ldy ? d.ldx : d.ldy) >= (1ll << 31) ||
            ((d.running_mean == nullptr) != (d.running_var == nullptr)))
            return RGDA_ERR_ARG;
    }
    for (int i0 = 0; i0 < n; i0 += BN_SMALL_MAX) {
        BnSmallGroup<BnSmallFwd> G;
        G.n = (n - i0 < BN_SMALL_MAX) ? n - i0 : BN_SMALL_MAX;
        G.start[0] = 0;
        for (int i = 0; i < BN_SMALL_MAX; ++i) {
            if (i < G.n) {
                const rgda_bn_small_fwd_desc& d = descs[i0 + i];
                BnSmallFwd& a = G.a[i];
                a.x = (const bf16_t*)d.x; a.y = (bf16_t*)d.y; a.mask_out = d.relu_mask; a.gamma = d.gamma; a.beta = d.beta;
                a.mi = d.mi; a.rm = d.running_mean; a.rv = d.running_var; a.nbt = (long long*)d.num_batches_tracked;
                a.ldx = d.ldx; a.ldy = d.ldy; a.M = (int)(d.M / d.groups); a.C = d.C; a.groups = d.groups; a.relu = d.relu;
                a.eps = d.eps; a.mom = d.momentum;
                G.start[i + 1] = G.start[i] + cdiv(d.C, BN_SMALL_CV * 8);
            } else {
                G.start[i + 1] = G.start[i];
            }
        }
        bn_small_fwd_kernel<<<G.start[G.n], BN_SMALL_NT, 0, to_stream(stream)>>>(G);
        RGDA_CHECK_LAUNCH();
    }
    return RGDA_OK;
}

extern "C" int rgda_bn_bwd_small(const rgda_bn_small_bwd_desc* descs, int n, rgda_stream_t stream) {
    if (!descs || n < 0) return RGDA_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        const rgda_bn_small_bwd_desc& d = descs[i];
        if (!d.g || !d.x || !d.dx || !d.mi || !d.gamma || (d.relu && !d.y && !d.relu_mask) || d.relu < 0 || d.relu > 1 || d.M <= 0 ||
            d.C <= 0 || (d.C & 7) || (d.ldg & 7) || (d.ldx & 7) || (d.lddx & 7) || (d.y && (d.ldy & 7)) || d.groups < 1 ||
            d.groups > 8 || (d.M % d.groups) || d.M / d.groups > RGDA_BN_SMALL_MAX_ROWS ||
            d.M * (int64_t)(d.ldg > d.ldx ? (d.ldg > d.lddx ? d.ldg : d.lddx) : (d.ldx > d.lddx ? d.ldx : d.lddx)) >= (1ll << 31) ||
            ((d.dgamma == nullptr) != (d.dbeta == nullptr)))
            return RGDA_ERR_ARG;
    }
    // the gate's source is a compile-time variant: descriptors that read y and descriptors that read the mask go to separate launches
    for (int use_y = 0; use_y < 2; ++use_y) {
        BnSmallGroup<BnSmallBwd> G;
        G.n = 0; G.start[0] = 0;
        auto flush = [&]() -> int {
            if (!G.n) return RGDA_OK;
            for (int i = G.n; i < BN_SMALL_MAX; ++i) G.start[i + 1] = G.start[G.n];
            if (use_y) bn_small_bwd_kernel<true><<<G.start[G.n], BN_SMALL_NT, 0, to_stream(stream)>>>(G);
            else bn_small_bwd_kernel<false><<<G.start[G.n], BN_SMALL_NT, 0, to_stream(stream)>>>(G);
            G.n = 0;
            RGDA_CHECK_LAUNCH();
            return RGDA_OK;
        };
        for (int i = 0; i < n; ++i) {
            const rgda_bn_small_bwd_desc& d = descs[i];
            if ((d.relu && !d.relu_mask) != (use_y != 0)) continue;
            BnSmallBwd& a = G.a[G.n];
            a.g = (const bf16_t*)d.g; a.y = (const bf16_t*)d.y; a.rmask = d.relu_mask; a.x = (const bf16_t*)d.x;
            a.dx = (bf16_t*)d.dx; a.mi = d.mi; a.gamma = d.gamma; a.dgamma = d.dgamma; a.dbeta = d.dbeta;
            a.ldg = d.ldg; a.ldy = d.ldy; a.ldx = d.ldx; a.lddx = d.lddx; a.M = (int)(d.M / d.groups); a.C = d.C;
            a.groups = d.groups; a.relu = d.relu;
            G.start[G.n + 1] = G.start[G.n] + cdiv(d.C, BN_SMALL_CV * 8);
            if (++G.n == BN_SMALL_MAX) { if (int rc = flush()) return rc; }
        }
        if (int rc = flush()) return rc;
    }
    return RGDA_OK;
}

// ------------------------------------------------------------------ MaxPool 3x3 / 2 / pad 1
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho,
                                                          int Wo) {
    const int vpr = C / 8;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    long long total = (long long)N * Ho * Wo * vpr;
    if (i >= total) return;
    int cv = (int)(i % vpr);
    long long p = i / vpr;
    int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), n = (int)(p / ((long long)Wo * Ho));
    float best[8];
    unsigned char bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    bool first = true;
    for (int kh = 0; kh < 3; ++kh) {
        int hi = ho * 2 - 1 + kh;
        if (hi < 0 || hi >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            int wi = wo * 2 - 1 + kw;
            if (wi < 0 || wi >= W) continue;
            float f[8];
            load8(x + ((size_t)(n * H + hi) * W + wi) * C + cv * 8, f);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (first || f[e] > best[e]) { best[e] = f[e]; bi[e] = (unsigned char)(kh * 3 + kw); }
            first = false;
        }
    }
    store8(y + (size_t)p * C + cv * 8, best);
    uint2 pk;
    pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
    pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
    *(uint2*)(idx + (size_t)p * C + cv * 8) = pk;
}

__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const bf16_t* __restrict__ gy, const uint8_t* __restrict__ idx,
                                                          bf16_t* __restrict__ gx, int N, int H, int W, int C, int Ho,
                                                          int Wo) {
    const int vpr = C / 8;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    long long total = (long long)N * H * W * vpr;
    if (i >= total) return;
    int cv = (int)(i % vpr);
    long long p = i / vpr;
    int wi = (int)(p % W), hi = (int)((p / W) % H), n = (int)(p / ((long long)W * H));
    float acc[8] = {0};
    for (int ho = (hi) / 2; ho <= (hi + 1) / 2; ++ho) {
        if (ho < 0 || ho >= Ho) continue;
        int kh = hi - (ho * 2 - 1);
        if (kh < 0 || kh > 2) continue;
        for (int wo = (wi) / 2; wo <= (wi + 1) / 2; ++wo) {
            if (wo < 0 || wo >= Wo) continue;
            int kw = wi - (wo * 2 - 1);
            if (kw < 0 || kw > 2) continue;
            size_t o = ((size_t)(n * Ho + ho) * Wo + wo) * C + cv * 8;
            uint2 pk = *(const uint2*)(idx + o);
            float g[8];
            load8(gy + o, g);
            unsigned tapid = (unsigned)(kh * 3 + kw);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned b = ((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xffu;
                if (b == tapid) acc[e] += g[e];
            }
        }
    }
    store8(gx + (size_t)p * C + cv * 8, acc);
}

// The stem's MaxPool with conv1's BatchNorm + ReLU applied on ITS operand path (rgda_bn_operand): x is the RAW output of
// the stem convolution, the pooled value is max over the window of relu(fma(x, scale, shift)) -- the 134 MB activation of
// 16 images of 512 x 512 is never written or read back.  Workgroups own whole rows of ONE image (one statistics group).
__global__ void __launch_bounds__(256) maxpool_fwd_bnin_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                               uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho,
                                                               int Wo, BnOperand b, int blocks_per_image) {
    extern __shared__ __attribute__((aligned(16))) float mp_tab[];         // [2][C]
    const int vpr = C / 8;
    const int n = blockIdx.x / blocks_per_image;
    const long long per_image = (long long)Ho * Wo * vpr;
    const long long i = (long long)(blockIdx.x % blocks_per_image) * 256 + threadIdx.x;
    bn_operand_table<256>(b, n / (N / b.groups), blockIdx.x == 0, mp_tab);
    __syncthreads();
    if (i >= per_image) return;
    const int cv = (int)(i % vpr);
    const long long p = i / vpr;
    const int wo = (int)(p % Wo), ho = (int)(p / Wo);
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = mp_tab[cv * 8 + e]; sh[e] = mp_tab[C + cv * 8 + e]; }
    float best[8];
    unsigned char bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    bool first = true;
    for (int kh = 0; kh < 3; ++kh) {
        int hi = ho * 2 - 1 + kh;
        if (hi < 0 || hi >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            int wi = wo * 2 - 1 + kw;
            if (wi < 0 || wi >= W) continue;
            float f[8];
            load8(x + ((size_t)(n * H + hi) * W + wi) * C + cv * 8, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // the value the materialised activation would hold: rounded to bf16 before the comparison
                float a = bn_affine(f[e], sc[e], sh[e]);
                if (b.relu) a = fmaxf(a, 0.f);
                a = bf2f(f2bf(a));
                if (first || a > best[e]) { best[e] = a; bi[e] = (unsigned char)(kh * 3 + kw); }
            }
            first = false;
        }
    }
    const size_t o = ((size_t)n * Ho * Wo + p) * C + cv * 8;
    store8(y + o, best);
    uint2 pk;
    pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
    pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
    *(uint2*)(idx + o) = pk;
}

extern "C" int rgda_maxpool_fwd_bnin(const rgda_bn_operand* bn_in, const void* x, void* y, uint8_t* idx, int N, int H, int W,
                                     int C, int Ho, int Wo, rgda_stream_t stream) {
    if (!bn_in || !x || !y || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || C > 512) return RGDA_ERR_ARG;
    if (Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return RGDA_ERR_ARG;
    if (!bn_in->stats || !bn_in->gamma || !bn_in->beta || bn_in->groups < 1 || (N % bn_in->groups) ||
        (bn_in->running_mean == nullptr) != (bn_in->running_var == nullptr))
        return RGDA_ERR_ARG;
    if ((long long)N * H * W / bn_in->groups < 2) return RGDA_ERR_ARG;
    BnOperand b;
    b.stats = bn_in->stats; b.gamma = bn_in->gamma; b.beta = bn_in->beta; b.mi = bn_in->mi;
    b.rm = bn_in->running_mean; b.rv = bn_in->running_var; b.nbt = (long long*)bn_in->num_batches_tracked;
    b.eps = bn_in->eps; b.mom = bn_in->momentum; b.groups = bn_in->groups; b.relu = bn_in->relu; b.C = C;
    b.rows_per_group = (int)((long long)N * H * W / bn_in->groups);
    const long long per_image = (long long)Ho * Wo * (C / 8);
    const int bpi = cdiv(per_image, 256);
    if ((long long)bpi * N > 0x7fffffffLL) return RGDA_ERR_ARG;
    maxpool_fwd_bnin_kernel<<<bpi * N, 256, (size_t)2 * C * sizeof(float), to_stream(stream)>>>(
        (const bf16_t*)x, (bf16_t*)y, idx, N, H, W, C, Ho, Wo, b, bpi);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

extern "C" int rgda_maxpool_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int Ho, int Wo,
                                rgda_stream_t stream) {
    if (!x || !y || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return RGDA_ERR_ARG;
    if (Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return RGDA_ERR_ARG;
    long long total = (long long)N * Ho * Wo * (C / 8);
    maxpool_fwd_kernel<<<cdiv(total, 256), 256, 0, to_stream(stream)>>>((const bf16_t*)x, (bf16_t*)y, idx, N, H, W, C, Ho, Wo);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}
extern "C" int rgda_maxpool_bwd(const void* gy, const uint8_t* idx, void* gx, int N, int H, int W, int C, int Ho,
                                int Wo, rgda_stream_t stream) {
    if (!gy || !gx || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return RGDA_ERR_ARG;
    long long total = (long long)N * H * W * (C / 8);
    maxpool_bwd_kernel<<<cdiv(total, 256), 256, 0, to_stream(stream)>>>((const bf16_t*)gy, idx, (bf16_t*)gx, N, H, W, C, Ho, Wo);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// ------------------------------------------------------------------ InstanceNorm2d (no affine)
// block = (image n, 64 channels): 8 channel vectors x 32 pixel lanes.
__global__ void __launch_bounds__(256) instnorm_fwd_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* y0, bf16_t* y1,
                                                           int ldy, float* feat, float* mi, int HW, int C, float eps) {
    __shared__ float lds[32 * 64 * 2];
    __shared__ float tile[64][33];
    const int n = blockIdx.y, c0 = blockIdx.x * 64;
    const int cv = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int cg = c0 + cv * 8;
    const bool cok = cg < C;
    const bf16_t* xb = x + (size_t)n * HW * ldx;
    float s[8] = {0}, q[8] = {0};
    if (cok)
        for (int pb = rl; pb < HW; pb += 32 * RB_INORM) {           // RB_INORM loads in flight per thread
            u16x8 v[RB_INORM];
#pragma unroll
            for (int u = 0; u < RB_INORM; ++u)
                if (pb + u * 32 < HW) v[u] = *(const u16x8*)(xb + (size_t)(pb + u * 32) * ldx + cg);
#pragma unroll
            for (int u = 0; u < RB_INORM; ++u) {
                if (pb + u * 32 >= HW) break;
                float f[8];
                cvt8(v[u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
            }
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        lds[(rl * 64 + cv * 8 + e) * 2] = s[e];
        lds[(rl * 64 + cv * 8 + e) * 2 + 1] = q[e];
    }
    __syncthreads();
    float mean[8], istd[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < 32; ++r) { a += lds[(r * 64 + cv * 8 + e) * 2]; b += lds[(r * 64 + cv * 8 + e) * 2 + 1]; }
        float m = a / (float)HW;
        float var = fmaxf(b / (float)HW - m * m, 0.f);
        mean[e] = m;
        istd[e] = 1.f / sqrtf(var + eps);
    }
    if (rl == 0 && cok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mi[((size_t)n * 2 + 0) * C + cg + e] = mean[e];
            mi[((size_t)n * 2 + 1) * C + cg + e] = istd[e];
        }
    }
    for (int pb = 0; pb < HW; pb += 32) {
        int p = pb + rl;
        float f[8] = {0};
        if (cok && p < HW) {
            load8(xb + (size_t)p * ldx + cg, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean[e]) * istd[e];
            size_t o = ((size_t)n * HW + p) * ldy + cg;
            if (y0) store8(y0 + o, f);
            if (y1) store8(y1 + o, f);
        }
        if (feat) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[cv * 8 + e][rl] = f[e];
            __syncthreads();
            int ch = threadIdx.x >> 2, part = threadIdx.x & 3;
            if (c0 + ch < C) {
                float* dst = feat + ((size_t)n * C + c0 + ch) * HW + pb + part * 8;
                if (pb + part * 8 + 8 <= HW && !(HW & 3)) {          // two 16-byte stores
                    *(float4*)dst = make_float4(tile[ch][part * 8 + 0], tile[ch][part * 8 + 1], tile[ch][part * 8 + 2], tile[ch][part * 8 + 3]);
                    *(float4*)(dst + 4) = make_float4(tile[ch][part * 8 + 4], tile[ch][part * 8 + 5], tile[ch][part * 8 + 6], tile[ch][part * 8 + 7]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (pb + part * 8 + e < HW) dst[e] = tile[ch][part * 8 + e];
                }
            }
        }
    }
}

extern "C" int rgda_instnorm_fwd(const void* x, int ldx, void* y0, void* y1, int ldy, float* feat_nchw, float* mi,
                                 int N, int HW, int C, float eps, rgda_stream_t stream) {
    if (!x || !mi || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || (ldx & 7) || ((y0 || y1) && (ldy & 7))) return RGDA_ERR_ARG;
    dim3 grid(cdiv(C, 64), N);
    instnorm_fwd_kernel<<<grid, 256, 0, to_stream(stream)>>>((const bf16_t*)x, ldx, (bf16_t*)y0, (bf16_t*)y1, ldy,
                                                              feat_nchw, mi, HW, C, eps);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

__global__ void __launch_bounds__(256) instnorm_bwd_kernel(const bf16_t* __restrict__ ga, const bf16_t* __restrict__ gb,
                                                           int ldg, const bf16_t* __restrict__ gc, int ldgc,
                                                           const bf16_t* __restrict__ x, int ldx,
                                                           const float* __restrict__ mi, bf16_t* __restrict__ dx, int lddx,
                                                           int HW, int C) {
    __shared__ float lds[32 * 64 * 2];
    const int n = blockIdx.y, c0 = blockIdx.x * 64;
    const int cv = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int cg = c0 + cv * 8;
    const bool cok = cg < C;
    float mean[8] = {0}, istd[8] = {0};
    if (cok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { mean[e] = mi[((size_t)n * 2) * C + cg + e]; istd[e] = mi[((size_t)n * 2 + 1) * C + cg + e]; }
    }
    // both passes: all loads of a batch of rows (up to four tensors each) are issued before the first use
    constexpr int IB = 2;
    auto fetch = [&](int pb, u16x8 (&va)[IB], u16x8 (&vb)[IB], u16x8 (&vc)[IB], u16x8 (&vx)[IB]) {
#pragma unroll
        for (int u = 0; u < IB; ++u) {
            const int p = pb + u * 32;
            if (p >= HW) break;
            const size_t row = (size_t)n * HW + p;
            if (ga) va[u] = *(const u16x8*)(ga + row * ldg + cg);
            if (gb) vb[u] = *(const u16x8*)(gb + row * ldg + cg);
            if (gc) vc[u] = *(const u16x8*)(gc + row * ldgc + cg);
            vx[u] = *(const u16x8*)(x + row * ldx + cg);
        }
    };
    auto gsum = [&](const u16x8& a8, const u16x8& b8, const u16x8& c8, float (&g)[8]) {
        float t8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
        if (ga) { cvt8(a8, t8);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] += t8[e]; }
        if (gb) { cvt8(b8, t8);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] += t8[e]; }
        if (gc) { cvt8(c8, t8);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] += t8[e]; }
    };
    float s[8] = {0}, q[8] = {0};
    if (cok)
        for (int pb = rl; pb < HW; pb += 32 * IB) {
            u16x8 va[IB], vb[IB], vc[IB], vx[IB];
            fetch(pb, va, vb, vc, vx);
#pragma unroll
            for (int u = 0; u < IB; ++u) {
                if (pb + u * 32 >= HW) break;
                float g[8], f[8];
                gsum(va[u], vb[u], vc[u], g);
                cvt8(vx[u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += g[e]; q[e] += g[e] * ((f[e] - mean[e]) * istd[e]); }
            }
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        lds[(rl * 64 + cv * 8 + e) * 2] = s[e];
        lds[(rl * 64 + cv * 8 + e) * 2 + 1] = q[e];
    }
    __syncthreads();
    float k1[8], k2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < 32; ++r) { a += lds[(r * 64 + cv * 8 + e) * 2]; b += lds[(r * 64 + cv * 8 + e) * 2 + 1]; }
        k1[e] = a / (float)HW;
        k2[e] = b / (float)HW;
    }
    if (cok)
        for (int pb = rl; pb < HW; pb += 32 * IB) {
            u16x8 va[IB], vb[IB], vc[IB], vx[IB];
            fetch(pb, va, vb, vc, vx);
#pragma unroll
            for (int u = 0; u < IB; ++u) {
                const int p = pb + u * 32;
                if (p >= HW) break;
                float g[8], f[8], o[8];
                gsum(va[u], vb[u], vc[u], g);
                cvt8(vx[u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = istd[e] * (g[e] - k1[e] - (f[e] - mean[e]) * istd[e] * k2[e]);
                store8(dx + ((size_t)n * HW + p) * lddx + cg, o);
            }
        }
}

extern "C" int rgda_instnorm_bwd(const void* ga, const void* gb, int ldg, const void* gc, int ldgc, const void* x,
                                 int ldx, const float* mi, void* dx, int lddx, int N, int HW, int C,
                                 rgda_stream_t stream) {
    if (!x || !mi || !dx || (!ga && !gb && !gc) || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (lddx & 7) ||
        ((ga || gb) && (ldg & 7)) || (gc && (ldgc & 7)))
        return RGDA_ERR_ARG;
    dim3 grid(cdiv(C, 64), N);
    instnorm_bwd_kernel<<<grid, 256, 0, to_stream(stream)>>>((const bf16_t*)ga, (const bf16_t*)gb, ldg,
                                                              (const bf16_t*)gc, ldgc,
                                                              (const bf16_t*)x, ldx, mi, (bf16_t*)dx, lddx, HW, C);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// ------------------------------------------------------------------ spatial linear map
// out[n][i][c] (+)= sum_j Mx[i][j] * in[n][j][c]; block = (output row i, image n, CVB channel vectors);
// 256 / CVB thread slices split the J range.  Long rows (adaptive-pool matrices: J = H*W, a bin touches a
// few dozen of them) use 8 vectors x 32 slices and walk only the row's nonzero span; short rows (bilinear
// upsampling from an s x s grid) use 32 vectors x 8 slices.
template <int CVB>
__global__ void __launch_bounds__(256) spatial_mix_kernel(const bf16_t* __restrict__ in, int ldin,
                                                          const float* __restrict__ Mx, void* out, int ldout, int I,
                                                          int J, int C, int accumulate, int out_f32) {
    constexpr int SL = 256 / CVB;
    __shared__ float red[SL][CVB][8];
    __shared__ int span[2];
    const int i = blockIdx.x, n = blockIdx.y;
    const int cvl = threadIdx.x % CVB, sl = threadIdx.x / CVB;
    const int cv = blockIdx.z * CVB + cvl;
    const bool cok = cv < C / 8;
    const float* mrow = Mx + (size_t)i * J;
    int lo = 0, hi = J - 1;
    if (J >= 256) {
        if (threadIdx.x == 0) { span[0] = J; span[1] = -1; }
        __syncthreads();
        int l = J, h = -1;
        for (int j = threadIdx.x; j < J; j += 256)
            if (mrow[j] != 0.f) { l = min(l, j); h = max(h, j); }
        if (h >= 0) { atomicMin(&span[0], l); atomicMax(&span[1], h); }
        __syncthreads();
        lo = span[0]; hi = span[1];
    }
    float acc[8] = {0};
    if (cok)
        for (int jb = lo + sl; jb <= hi; jb += SL * RB_MIX) {
            float m[RB_MIX];
            u16x8 v[RB_MIX];
#pragma unroll
            for (int u = 0; u < RB_MIX; ++u) {
                const int j = jb + u * SL;
                m[u] = (j <= hi) ? mrow[j] : 0.f;
                if (m[u] != 0.f) v[u] = *(const u16x8*)(in + ((size_t)n * J + j) * ldin + cv * 8);
            }
#pragma unroll
            for (int u = 0; u < RB_MIX; ++u) {
                if (m[u] == 0.f) continue;
                float f[8];
                cvt8(v[u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += m[u] * f[e];
            }
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[sl][cvl][e] = acc[e];
    __syncthreads();
    if (sl != 0 || !cok) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float a = 0.f;
        for (int q = 0; q < SL; ++q) a += red[q][cvl][e];
        acc[e] = a;
    }
    size_t o = ((size_t)n * I + i) * ldout + cv * 8;
    if (out_f32) {
        float* op = (float*)out + o;
#pragma unroll
        for (int e = 0; e < 8; ++e) op[e] = accumulate ? op[e] + acc[e] : acc[e];
    } else {
        bf16_t* op = (bf16_t*)out + o;
        if (accumulate) {
            float f[8];
            load8(op, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
        store8(op, acc);
    }
}

extern "C" int rgda_spatial_mix(const void* in, int ldin, const float* Mx, void* out, int ldout, int N, int I, int J,
                                int C, int accumulate, int out_f32, rgda_stream_t stream) {
    if (!in || !Mx || !out || N <= 0 || I <= 0 || J <= 0 || C <= 0 || (C & 7) || (ldin & 7) || (ldout & 7)) return RGDA_ERR_ARG;
    if (J >= 256) {
        dim3 grid(I, N, cdiv(C / 8, 8));
        spatial_mix_kernel<8><<<grid, 256, 0, to_stream(stream)>>>((const bf16_t*)in, ldin, Mx, out, ldout, I, J, C, accumulate, out_f32);
    } else {
        dim3 grid(I, N, cdiv(C / 8, 32));
        spatial_mix_kernel<32><<<grid, 256, 0, to_stream(stream)>>>((const bf16_t*)in, ldin, Mx, out, ldout, I, J, C, accumulate, out_f32);
    }
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// out[n][i][c] = sum_q sum_j Mq[i][j] * inq[n][j][c] for up to 4 (matrix, tensor) pairs with small J (the
// transposed adaptive-pool maps of the four PPM scales): one pass, one bf16 store per output vector.
struct MixSrc { const bf16_t* in; const float* Mx; int ldin; int J; };
struct MixSrc4 { MixSrc s[4]; int n; };
// Workgroup = (output row i, image n).  The matrix rows of all sources (<= 1024 entries together) are staged in LDS
// once; the 256 threads are CVB channel vectors x 256/CVB slices of that entry list, zeros are skipped from LDS
// (the tap-shifted bilinear rows of the PPM heads have <= 117 nonzeros out of 450), slices are summed in a fixed
// order (deterministic).
template <int CVB>
__global__ void __launch_bounds__(256) spatial_mix_multi_kernel(MixSrc4 src, bf16_t* __restrict__ out, int ldout, int I,
                                                                int C) {
    constexpr int SL = 256 / CVB;
    __shared__ float s_w[1024];
    __shared__ float red[SL][CVB][8];
    const int i = blockIdx.x, n = blockIdx.y;
    int total = 0;
    for (int q = 0; q < src.n; ++q) {
        const MixSrc& m = src.s[q];
        for (int j = threadIdx.x; j < m.J; j += 256) s_w[total + j] = m.Mx[(size_t)i * m.J + j];
        total += m.J;
    }
    __syncthreads();
    const int cvl = threadIdx.x % CVB, sl = threadIdx.x / CVB;
    for (int cv0 = 0; cv0 < C / 8; cv0 += CVB) {
        const int cv = cv0 + cvl;
        float acc[8] = {0};
        if (cv < C / 8) {
            int base = 0;
            for (int q = 0; q < src.n; ++q) {
                const MixSrc& m = src.s[q];
                const bf16_t* in = m.in + (size_t)n * m.J * m.ldin + cv * 8;
                for (int j = sl; j < m.J; j += SL) {
                    const float w = s_w[base + j];
                    if (w == 0.f) continue;
                    float f[8];
                    load8(in + (size_t)j * m.ldin, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += w * f[e];
                }
                base += m.J;
            }
        }
        if (SL > 1) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) red[sl][cvl][e] = acc[e];
            __syncthreads();
            if (sl == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float a = 0.f;
                    for (int r = 0; r < SL; ++r) a += red[r][cvl][e];
                    acc[e] = a;
                }
            }
        }
        if (sl == 0 && cv < C / 8) store8(out + ((size_t)n * I + i) * ldout + cv * 8, acc);
    }
}

extern "C" int rgda_spatial_mix_multi(int nsrc, const void* const* ins, const int* ldins, const float* const* mats,
                                      const int* Js, void* out, int ldout, int N, int I, int C, rgda_stream_t stream) {
    if (nsrc < 1 || nsrc > 4 || !ins || !ldins || !mats || !Js || !out || N <= 0 || I <= 0 || C <= 0 || (C & 7) ||
        (ldout & 7))
        return RGDA_ERR_ARG;
    MixSrc4 src;
    src.n = nsrc;
    for (int q = 0; q < nsrc; ++q) {
        if (!ins[q] || !mats[q] || Js[q] <= 0 || (ldins[q] & 7)) return RGDA_ERR_ARG;
        src.s[q].in = (const bf16_t*)ins[q]; src.s[q].Mx = mats[q]; src.s[q].ldin = ldins[q]; src.s[q].J = Js[q];
    }
    int jt = 0;
    for (int q = 0; q < nsrc; ++q) jt += Js[q];
    if (jt > 1024) return RGDA_ERR_UNSUPPORTED;
    dim3 grid(I, N);
    if (C / 8 >= 256) spatial_mix_multi_kernel<256><<<grid, 256, 0, to_stream(stream)>>>(src, (bf16_t*)out, ldout, I, C);
    else if (C / 8 >= 128) spatial_mix_multi_kernel<128><<<grid, 256, 0, to_stream(stream)>>>(src, (bf16_t*)out, ldout, I, C);
    else if (C / 8 >= 64) spatial_mix_multi_kernel<64><<<grid, 256, 0, to_stream(stream)>>>(src, (bf16_t*)out, ldout, I, C);
    else spatial_mix_multi_kernel<32><<<grid, 256, 0, to_stream(stream)>>>(src, (bf16_t*)out, ldout, I, C);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// ------------------------------------------------------------------ stem im2col (7x7 / 2 / pad 3, 3 channels)
// col[(n,ho,wo)][k] with k = (kh*7 + kw)*3 + c (zero for k >= 147 and for padding).  Workgroup = 64 output pixels of
// one output row: the 7 input rows x 133 columns x 3 channels they touch are staged in LDS as bf16 once (coalesced
// fp32 row reads), then every thread assembles 16-byte column vectors from LDS.  (One thread per vector reading the
// image directly spent its time in 8 scattered 4-byte loads and their index arithmetic: 1.8 TB/s of writes.)
__global__ void __launch_bounds__(256) stem_im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ col, int N,
                                                          int H, int W, int Ho, int Wo, int Kp) {
    constexpr int PX = 64, SW = 2 * PX + 5;           // staged columns: wi = 2*wo0 - 3 .. 2*(wo0 + 63) + 3
    __shared__ bf16_t patch[3][7][SW + 1];
    const int wtiles = (Wo + PX - 1) / PX;
    const int wo0 = (blockIdx.x % wtiles) * PX;
    const int ho = (blockIdx.x / wtiles) % Ho, n = blockIdx.x / (wtiles * Ho);
    for (int i = threadIdx.x; i < 21 * SW; i += 256) {
        const int row = i / SW, x = i % SW;           // row = c*7 + kh
        const int c = row / 7, kh = row % 7;
        const int hi = ho * 2 - 3 + kh, wi = wo0 * 2 - 3 + x;
        float v = 0.f;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = img[((size_t)(n * 3 + c) * H + hi) * W + wi];
        patch[c][kh][x] = f2bf(v);
    }
    __syncthreads();
    // thread = (column vector v, pixel lane): its eight patch offsets are fixed, only the pixel moves
    const int vpr = Kp / 8;
    const int lanes = 256 / vpr;                       // pixel lanes (10 for Kp = 192; the remaining threads idle)
    const int v = threadIdx.x % vpr, pl = threadIdx.x / vpr;
    if (pl >= lanes) return;
    int off[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = v * 8 + e;
        const int tap = k / 3, c = k % 3, kh = tap / 7, kw = tap % 7;
        off[e] = (k < 147) ? (c * 7 + kh) * (SW + 1) + kw : -1;
    }
    const bf16_t* flat = &patch[0][0][0];
    const int npx = min(PX, Wo - wo0);
    bf16_t* dst = col + ((size_t)(n * Ho + ho) * Wo + wo0) * Kp + v * 8;
    for (int px = pl; px < npx; px += lanes) {
        u16x8 out;
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = (off[e] < 0) ? (bf16_t)0 : flat[off[e] + 2 * px];
        *(u16x8*)(dst + (size_t)px * Kp) = out;
    }
}

extern "C" int rgda_stem_im2col(const float* img, void* col, int N, int H, int W, int Ho, int Wo, int Kp,
                                rgda_stream_t stream) {
    if (!img || !col || N <= 0 || H <= 0 || W <= 0 || Kp < 152 || Kp > 2048 || (Kp & 7)) return RGDA_ERR_ARG;
    if (Ho != (H + 6 - 7) / 2 + 1 || Wo != (W + 6 - 7) / 2 + 1) return RGDA_ERR_ARG;
    const long long blocks = (long long)N * Ho * cdiv(Wo, 64);
    if (blocks > 0x7fffffffLL) return RGDA_ERR_ARG;
    stem_im2col_kernel<<<(int)blocks, 256, 0, to_stream(stream)>>>(img, (bf16_t*)col, N, H, W, Ho, Wo, Kp);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// ------------------------------------------------------------------ classifier (1x1, ncls outputs, bias)
// one wavefront per pixel: lanes split the C hidden channels, ncls dot products, shuffle reduce
template <int NC>
__global__ void __launch_bounds__(256) classifier_fwd_kernel(const bf16_t* __restrict__ hid, int ldh,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ logits, int N, int HW, int C) {
    const int lane = threadIdx.x & 63;
    long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (long long)N * HW) return;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    for (int k = lane * 8; k < C; k += 512) {
        float f[8];
        load8(hid + m * ldh + k, f);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[c] += f[e] * w[c * C + k + e];
    }
    int n = (int)(m / HW), p = (int)(m % HW);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float s = wave_sum(acc[c]);
        if (lane == 0) logits[((size_t)n * NC + c) * HW + p] = s + bias[c];
    }
}

// dhidden[m][k] = sum_c g[n][c][p] * w[c][k];   dW[c][k] += sum_m g * hidden[m][k];  db[c] += sum_m g
template <int NC>
__global__ void __launch_bounds__(256) classifier_bwd_kernel(const bf16_t* __restrict__ hid, int ldh,
                                                             const float* __restrict__ w, const float* __restrict__ gl,
                                                             bf16_t* __restrict__ dhid, int lddh, float* dw, float* db,
                                                             int N, int HW, int C, int rows_per_block, float* part) {
    // block: channel vectors x row lanes, like the BN passes
    extern __shared__ float lds[];       // [rpb][vpb*8][NC]
    const int vpr = C / 8;
    int vpb = 1;
    while (vpb < vpr && vpb < 256) vpb <<= 1;
    const int rpb = 256 / vpb;
    const int cvl = threadIdx.x % vpb, rl = threadIdx.x / vpb;
    const int cg = (blockIdx.y * vpb + cvl) * 8;
    const bool cok = cg < C;
    float wreg[NC][8], dwacc[NC][8], dbacc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        dbacc[c] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { wreg[c][e] = cok ? w[c * C + cg + e] : 0.f; dwacc[c][e] = 0.f; }
    }
    long long M = (long long)N * HW;
    long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    if (cok)
        for (long long mb = r0 + rl; mb < r1; mb += (long long)rpb * RB_CLS) {
            float gq[RB_CLS][NC];
            u16x8 hv[RB_CLS];
#pragma unroll
            for (int u = 0; u < RB_CLS; ++u) {          // all loads of the batch first
                const long long m = mb + (long long)u * rpb;
                if (m < r1) {
                    int n = (int)(m / HW), p = (int)(m % HW);
#pragma unroll
                    for (int c = 0; c < NC; ++c) gq[u][c] = gl[((size_t)n * NC + c) * HW + p];
                    hv[u] = *(const u16x8*)(hid + m * ldh + cg);
                }
            }
#pragma unroll
            for (int u = 0; u < RB_CLS; ++u) {
                const long long m = mb + (long long)u * rpb;
                if (m >= r1) break;
                float h[8], o[8];
                cvt8(hv[u], h);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    dbacc[c] += gq[u][c];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { o[e] += gq[u][c] * wreg[c][e]; dwacc[c][e] += gq[u][c] * h[e]; }
                }
                store8(dhid + m * lddh + cg, o);
            }
        }
    // reduce dW over the row lanes
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) lds[((rl * vpb + cvl) * 8 + e) * NC + c] = dwacc[c][e];
    __syncthreads();
    if (rl == 0 && cok) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = 0.f;
                for (int r = 0; r < rpb; ++r) a += lds[((r * vpb + cvl) * 8 + e) * NC + c];
                if (part) part[(size_t)blockIdx.x * (NC * C + 64) + c * C + cg + e] = a;   // per-workgroup partial, no atomics
                else atomicAdd(dw + c * C + cg + e, a);
            }
    }
    if (blockIdx.y == 0 && cvl == 0) {
        // every row lane of channel-vector 0 saw a disjoint set of rows
        if (!part) {
#pragma unroll
            for (int c = 0; c < NC; ++c) atomicAdd(db + c, dbacc[c]);
        }
    }
    if (part && blockIdx.y == 0) {
        __shared__ float dbl2[64][NC];
        if (cvl == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) dbl2[rl][c] = dbacc[c];
        }
        __syncthreads();
        if (threadIdx.x < NC) {
            float a = 0.f;
            for (int r = 0; r < rpb; ++r) a += dbl2[r][threadIdx.x];
            part[(size_t)blockIdx.x * (NC * C + 64) + NC * C + threadIdx.x] = a;
        }
    }
}

// dW[c][k] += sum over the workgroups' partials (rows of NC*C + 64 floats: dW then db).  Workgroup = 32 outputs x
// 8 slices of the partial rows (eight loads in flight per thread), slices summed through LDS in a fixed order.
template <int NC>
__global__ void __launch_bounds__(256) classifier_bwd_reduce_kernel(const float* __restrict__ part, int blocks, float* dw,
                                                                    float* db, int C) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    const int stride = NC * C + 64;
    const bool ok = i < NC * C + NC;
    float a = 0.f;
    if (ok) {
        const int per = (blocks + 7) / 8;
        const int b0 = sl * per, b1 = min(blocks, b0 + per);
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + u) * stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; b < b1; ++b) a += part[(size_t)b * stride + i];
    }
    red[sl][o] = a;
    __syncthreads();
    if (sl == 0 && ok) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][o];
        if (i < NC * C) dw[i] += t; else db[i - NC * C] += t;
    }
}

extern "C" int rgda_classifier_fwd(const void* hidden, int ldh, const float* w, const float* bias, float* logits,
                                   int N, int HW, int C, int ncls, rgda_stream_t stream) {
    if (!hidden || !w || !bias || !logits || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || (ldh & 7)) return RGDA_ERR_ARG;
    if (ncls != 6) return RGDA_ERR_UNSUPPORTED;
    long long M = (long long)N * HW;
    classifier_fwd_kernel<6><<<cdiv(M, 4), 256, 0, to_stream(stream)>>>((const bf16_t*)hidden, ldh, w, bias, logits, N, HW, C);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

extern "C" size_t rgda_classifier_bwd_workspace(int64_t M, int C, int ncls) {
    return (size_t)cdiv(M, 64) * ((size_t)ncls * C + 64) * 4;
}

extern "C" int rgda_classifier_bwd(const void* hidden, int ldh, const float* w, const float* glogits, void* dhidden,
                                   int lddh, float* dw, float* db, int N, int HW, int C, int ncls, void* ws,
                                   size_t ws_bytes, rgda_stream_t stream) {
    if (!hidden || !w || !glogits || !dhidden || !dw || !db || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || (ldh & 7) ||
        (lddh & 7))
        return RGDA_ERR_ARG;
    if (ncls != 6) return RGDA_ERR_UNSUPPORTED;
    long long M = (long long)N * HW;
    RowLayout L = row_layout(C);
    hipStream_t st = to_stream(stream);
    if (ws) {
        // 64 rows per workgroup (the whole chip), per-workgroup partial dW / db in the workspace, then one small
        // deterministic reduction: ncls*C atomics per workgroup onto the SAME addresses forced few, long workgroups
        // (64 of them at M = 16384: a quarter of the CUs, 70 us)
        if (ws_bytes < rgda_classifier_bwd_workspace(M, C, ncls)) return RGDA_ERR_WORKSPACE;
        const int blocks = cdiv(M, 64);
        dim3 grid(blocks, cdiv(L.vpr, L.vpb));
        classifier_bwd_kernel<6><<<grid, 256, (size_t)256 * 8 * 6 * 4, st>>>(
            (const bf16_t*)hidden, ldh, w, glogits, (bf16_t*)dhidden, lddh, dw, db, N, HW, C, 64, (float*)ws);
        RGDA_CHECK_LAUNCH();
        classifier_bwd_reduce_kernel<6><<<cdiv(6 * C + 6, 32), 256, 0, st>>>((const float*)ws, blocks, dw, db, C);
        RGDA_CHECK_LAUNCH();
        return RGDA_OK;
    }
    // without a workspace: atomics, few long workgroups
    int rows_per_block = 256;
    if (const char* e = TUNE_ENV("RGDA_CLS_ROWS")) rows_per_block = atoi(e);     // tuning experiments only
    while ((long long)cdiv(M, rows_per_block) * cdiv(L.vpr, L.vpb) > 1024) rows_per_block *= 2;
    dim3 grid(cdiv(M, rows_per_block), cdiv(L.vpr, L.vpb));
    classifier_bwd_kernel<6><<<grid, 256, (size_t)256 * 8 * 6 * 4, st>>>(
        (const bf16_t*)hidden, ldh, w, glogits, (bf16_t*)dhidden, lddh, dw, db, N, HW, C, rows_per_block, nullptr);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}
