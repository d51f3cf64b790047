// Implicit-GEMM convolution for gfx950 on pixel-major bf16 activations ("PxC": [N*H*W][C]).
//
//   forward / data-gradient : D[co][pixel] = sum_{tap,ci} W[co][tap][ci] * X[src(pixel,tap)][ci]
//   weight-gradient         : dW[co][tap][ci] += sum_pixel dY[pixel][co] * X[src(pixel,tap)][ci]
//
// v_mfma_f32_32x32x16_bf16, fp32 accumulate.  Workgroups of WC x WP wavefronts (4 or 8), every wavefront owns a
// (BC/WC)x(BP/WP) block of 32x32 accumulators.  Operand tiles (64 channels of one filter tap) go HBM/L2 -> LDS by
// LDS-DMA (`buffer_load_dwordx4 ... lds`) into a 2- or 3-stage ring, ordered by counted `s_waitcnt vmcnt(N)` and one
// raw `s_barrier` per K tile; padding / out-of-range rows carry an out-of-range buffer offset and the hardware
// deposits zeros, so the gathered, zero-padded pixel rows of the implicit GEMM need no second pass.
//   * forward / data gradient: both operands are K(channel)-contiguous -> 128-byte LDS rows; the 16-byte-slot XOR
//     swizzle (slot ^= (row>>1)&7) is applied on the SOURCE address (the DMA image is lane-linear) and again on the
//     read: conflict-free for the 16-lane groups of ds_read_b128.
//   * weight gradient: both operands are K(pixel)-STRIDED -> rows of pixels as they lie in memory, 64-byte granules
//     XOR-swizzled with the row index, ds_read_b64_tr_b16 transposing reads (4 k-rows x 16 columns per 16-lane group).
// Epilogue (forward): accumulators -> bf16 -> LDS -> 16-byte coalesced row stores, with the
// optional residual add and the per-channel sum / sum-of-squares of BatchNorm folded in.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include <type_traits>

struct ConvArgs {
    const bf16_t* x;
    const bf16_t* w;
    bf16_t* y;
    const bf16_t* res;
    rgda_stat_t* stats;
    int ldx, ldy, ldres;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, dil, mode;
    int M;
    int tiles_c, tiles_p;
    int rows_per_group;   // BatchNorm statistics are kept per group of rows (src / tgt batch)
    int howo_shift, wo_shift;   // log2(Ho * Wo), log2(Wo) when those are powers of two (every map of a 512 x 512 step), else -1
    unsigned long long* dbg;   // optional per-workgroup phase timestamps (tuning builds only)
    int tpw;                   // conv3x3_c64_kernel: image rows per workgroup
    int skip;                  // tuning only: 1 = no DMA inside the K loop, 2 = no LDS reads / MFMA
    // optional fused BatchNorm-backward reduction of the CONSUMER of this (data-gradient) output: with g = the stored
    // result, g' = g * [bn_y > 0] * nscale, xhat = (bn_x - mean) * invstd, `stats` receives sum(g'), sum(g' * xhat)
    const unsigned char* res_mask;  // [M][Cout/8] sign bits: the residual is added only where its bit is set
    const bf16_t* bn_y;
    const unsigned char* bn_mask;   // [M][Cout/8] ReLU sign bits of bn_y (read instead of bn_y when given)
    const bf16_t* bn_x;
    const float* bn_mi;
    const float* bn_nscale;
    int bn_ldy, bn_ldx, bn_rpi, bn_relu;
    // optional fused inference-mode BatchNorm (+ residual + ReLU): y = act((conv - rm) / sqrt(rv + eps) * gamma + beta + res)
    const float* ev_rm;
    const float* ev_rv;
    const float* ev_gamma;
    const float* ev_beta;
    float ev_eps;
    int ev_relu;
    // bn_relu == 2: the ReLU sign of the consumer's BatchNorm output is recomputed from bn_x (that unit's activation
    // was never written: it ran on its consumer's operand path), with the forward's own formula (common.h: bn_affine)
    const float* bn_gamma;
    const float* bn_beta;
    // optional BatchNorm (+ ReLU) of the PRODUCER of x applied on this convolution's operand path (xf.stats != nullptr)
    BnOperand xf;
};

// Per-workgroup timestamps and K-loop ablation switches exist in tuning builds only (`make TUNING=1`): in the product
// library the conditions below are compile-time constants and the instrumented branches are not generated.
#ifdef RGDA_TUNING
#define TDBG(a) ((a).dbg)
#define TSKIP(a) ((a).skip)
#else
#define TDBG(a) ((unsigned long long*)nullptr)
#define TSKIP(a) 0
#endif

static __device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // block b runs on XCD b%8: give every XCD a contiguous range of logical tiles (bijective form)
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

static __device__ __forceinline__ void src_coord(int mode, int base, int t, int dil, int stride, int lim, int& out,
                                                 bool& ok) {
    if (mode == 0) {
        out = base + t * dil;
        ok = (out >= 0) && (out < lim);
    } else {
        int th = base - t * dil;
        ok = th >= 0;
        if (stride == 1) {
            out = th;
        } else {
            ok = ok && (th % stride == 0);
            out = th / stride;
        }
        ok = ok && (out < lim);
    }
}

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// LDS-DMA issued from inline assembly (weight-gradient kernels; the halo pieces of conv3x3_halo_kernel<.., XF>, whose
// in-place transform the compiler would otherwise order behind EVERY DMA in flight: `s_waitcnt vmcnt(0)` per LDS store).  Through the builtin the compiler knows the
// instruction writes LDS and -- it cannot prove which bytes -- puts `s_waitcnt vmcnt(0)` in front of the
// transposing LDS reads of the K loop, which drains the tiles prefetched for the NEXT iterations as well and
// serialises memory latency with the MFMAs.  The kernels order DMA against LDS reads themselves (explicit
// vmcnt + one barrier per K tile), so the DMA is kept opaque.  M0 carries the wave-uniform LDS destination.
typedef __attribute__((ext_vector_type(4))) int i32x4;
static __device__ __forceinline__ i32x4 dma_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r = {(int)(unsigned)b, (int)((unsigned)(b >> 32) & 0xffffu), (int)bytes, 0x00020000};
    return r;
}
static __device__ __forceinline__ void dma16_to_lds(i32x4 rsrc, const unsigned char* lds_dst, int voffset, int soffset) {
    const unsigned m0 = (unsigned)(__UINTPTR_TYPE__)LDS_PTR(lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(m0), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}


// ---- the row pass of the epilogue: bf16 C tile [pixel][cout] in LDS -> 16-byte row pieces in memory, one variant per
// fused epilogue.  Every thread owns one 8-channel vector of BP / RPP rows; the LDS reads and the global loads of up to
// four rows are issued together (the pass is latency- and instruction-bound: ~40 % of a small-K 1x1 workgroup's life
// went here when it was one generic loop with a load -> wait -> store chain per row), the arithmetic is two-wide
// (v_pk_*_f32) on bf16 pairs unpacked with a shift / a mask.
enum { EPI_PLAIN = 0, EPI_STATS, EPI_RES, EPI_RES_STATS, EPI_EV, EPI_BNX, EPI_BNX2 };   // BNX2: BNX with the ReLU sign recomputed from bn_x
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned u32x4r;
#define RGDA_BNIN_MAX_C 512      /* most input channels a BatchNorm-on-the-operand-path launch serves (LDS table) */
static __device__ __forceinline__ f32x2 bf2f_pair(unsigned u) {
    f32x2 r;
    r.x = __uint_as_float(u << 16);
    r.y = __uint_as_float(u & 0xffff0000u);
    return r;
}
static __device__ __forceinline__ unsigned pack2bf(f32x2 v) { return pack2bf(v.x, v.y); }

// the per-channel constants of a fused epilogue for this thread's 8-channel vector: EV: k0 = scale, k1 = shift; BNX: k0 =
// invstd, k1 = mean; BNX2 also k2, k3 = scale, shift of the forward operand path
struct EpiParams { f32x2 k0[4], k1[4], k2[4], k3[4]; };
template <int KIND>
static __device__ __forceinline__ void epilogue_params(const ConvArgs& a, int m0, int co, bool cok, EpiParams& P) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        P.k0[e] = f32x2{0.f, 0.f}; P.k1[e] = f32x2{0.f, 0.f};
        P.k2[e] = f32x2{0.f, 0.f}; P.k3[e] = f32x2{0.f, 0.f};
    }
    constexpr bool BNX = KIND == EPI_BNX || KIND == EPI_BNX2;
    if (BNX && cok) {
        const float* mi = a.bn_mi + (size_t)(m0 / a.rows_per_group) * 2 * a.Cout + co;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            P.k1[e] = f32x2{mi[2 * e], mi[2 * e + 1]};
            P.k0[e] = f32x2{mi[a.Cout + 2 * e], mi[a.Cout + 2 * e + 1]};
        }
        if constexpr (KIND == EPI_BNX2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float sc, sh;
                bn_scale_shift(P.k1[e >> 1][e & 1], P.k0[e >> 1][e & 1], a.bn_gamma[co + e], a.bn_beta[co + e], sc, sh);
                P.k2[e >> 1][e & 1] = sc;
                P.k3[e >> 1][e & 1] = sh;
            }
        }
    }
    if (KIND == EPI_EV && cok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = a.ev_gamma[co + e] / sqrtf(a.ev_rv[co + e] + a.ev_eps);
            P.k0[e >> 1][e & 1] = sc;
            P.k1[e >> 1][e & 1] = a.ev_beta[co + e] - a.ev_rm[co + e] * sc;
        }
    }
}

// FULL: every row and channel of the tile exists (the caller's host side guarantees it): no per-lane conditions, exactly
// NIT row stores per thread.  pre: the constants, prepared once by a caller that runs many tiles of one channel tile and
// one statistics group (conv1x1_stream_kernel); nullptr: prepared here.
// WMAP: the tile is a band of 32-pixel image rows inside a WIDER map (conv3x3_halo_wide_kernel): tile row r lies at memory
// row m0 + (r / 32) W + r % 32 instead of m0 + r
template <int WMAP>
static __device__ __forceinline__ int tile_row(const ConvArgs& a, int row) {
    return WMAP ? ((row >> 5) * a.W + (row & 31)) : row;
}
template <int BC, int BP, int NT, int KIND, bool FULL = false, int WMAP = 0>
static __device__ __forceinline__ void epilogue_rows(const ConvArgs& a, const unsigned char* smem, int m0, int c0,
                                                     float (&s)[8], float (&q)[8], const EpiParams* pre = nullptr) {
    constexpr int CSTR = BC * 2 + 16;
    constexpr int VPR = BC / 8, RPP = NT / VPR, NIT = BP / RPP;
    constexpr int CH = NIT < 4 ? NIT : 4;                  // rows in flight per thread
    static_assert(NIT % CH == 0, "row passes");
    const int t = threadIdx.x;
    const int cv = t % VPR, rr = t / VPR;
    const int co = c0 + cv * 8;
    const bool cok = FULL || co < a.Cout;
    EpiParams P;
    if (pre) P = *pre; else epilogue_params<KIND>(a, m0, co, cok, P);
    f32x2 (&k0)[4] = P.k0, (&k1)[4] = P.k1, (&k2)[4] = P.k2, (&k3)[4] = P.k3;
    f32x2 s2[4], q2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { s2[e] = f32x2{s[2 * e], s[2 * e + 1]}; q2[e] = f32x2{q[2 * e], q[2 * e + 1]}; }
    constexpr bool BNX = KIND == EPI_BNX || KIND == EPI_BNX2;
    const bool has_res = (KIND == EPI_RES || KIND == EPI_RES_STATS) || ((KIND == EPI_EV || BNX) && a.res);
    const bool res_gate = (KIND != EPI_EV) && has_res && a.res_mask;
    constexpr bool STATS = KIND == EPI_STATS || KIND == EPI_RES_STATS;
#pragma unroll
    for (int p0 = 0; p0 < NIT; p0 += CH) {
        uint4 val[CH], rv[CH], xv[CH], yv[CH];
        unsigned rmb[CH], bmb[CH];
        bool ok[CH];
        // ---- everything this chunk reads, issued back to back
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int row = rr + (p0 + i) * RPP;
            const int m = m0 + tile_row<WMAP>(a, row);
            ok[i] = FULL || (cok && m < a.M);
            val[i] = *(const uint4*)(smem + row * CSTR + cv * 16);
            rv[i] = uint4{0u, 0u, 0u, 0u}; xv[i] = rv[i]; yv[i] = rv[i]; rmb[i] = 0xffu; bmb[i] = 0xffu;
            if (ok[i]) {
                if (has_res) {
                    rv[i] = *(const uint4*)(a.res + (size_t)m * a.ldres + co);
                    if (res_gate) rmb[i] = a.res_mask[(size_t)m * (a.Cout >> 3) + (co >> 3)];
                }
                if (BNX) {
                    xv[i] = *(const uint4*)(a.bn_x + (size_t)m * a.bn_ldx + co);
                    if (KIND == EPI_BNX && a.bn_relu) {
                        if (a.bn_mask) bmb[i] = a.bn_mask[(size_t)m * (a.Cout >> 3) + (co >> 3)];
                        else yv[i] = *(const uint4*)(a.bn_y + (size_t)m * a.bn_ldy + co);
                    }
                }
            }
        }
        // ---- arithmetic + the row store
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int m = m0 + tile_row<WMAP>(a, rr + (p0 + i) * RPP);
            const unsigned vw[4] = {val[i].x, val[i].y, val[i].z, val[i].w};
            const unsigned rw[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
            unsigned ow[4];
            if (KIND == EPI_PLAIN || KIND == EPI_STATS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ow[e] = vw[e];
            } else if (KIND == EPI_EV) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x2 f = bf2f_pair(vw[e]) * k0[e] + k1[e];
                    if (has_res) f += bf2f_pair(rw[e]);
                    if (a.ev_relu) { f.x = fmaxf(f.x, 0.f); f.y = fmaxf(f.y, 0.f); }
                    ow[e] = pack2bf(f);
                }
            } else {                                        // residual (gated by the ReLU mask of the layer it crossed)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned r = rw[e];
                    if (res_gate) {
                        const unsigned b0 = (rmb[i] >> (2 * e)) & 1u, b1 = (rmb[i] >> (2 * e + 1)) & 1u;
                        r &= (b0 ? 0x0000ffffu : 0u) | (b1 ? 0xffff0000u : 0u);
                    }
                    ow[e] = has_res ? pack2bf(bf2f_pair(vw[e]) + bf2f_pair(r)) : vw[e];
                }
            }
            if (ok[i]) {
                if (STATS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x2 f = bf2f_pair(ow[e]);
                        s2[e] += f;
                        q2[e] += f * f;
                    }
                }
                if (BNX) {
                    const unsigned xw[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
                    const unsigned yw[4] = {yv[i].x, yv[i].y, yv[i].z, yv[i].w};
                    f32x2 ns[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) ns[e] = f32x2{1.f, 1.f};
                    if (a.bn_nscale) {
                        const float* np = a.bn_nscale + (size_t)(m / a.bn_rpi) * a.Cout + co;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ns[e] = f32x2{np[2 * e], np[2 * e + 1]};
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f32x2 g = bf2f_pair(ow[e]);
                        if constexpr (KIND == EPI_BNX2) {
                            const f32x2 xx = bf2f_pair(xw[e]);
                            g.x = (bn_affine(xx.x, k2[e].x, k3[e].x) > 0.f) ? g.x : 0.f;
                            g.y = (bn_affine(xx.y, k2[e].y, k3[e].y) > 0.f) ? g.y : 0.f;
                        } else if (a.bn_relu) {
                            if (a.bn_mask) {
                                g.x = ((bmb[i] >> (2 * e)) & 1u) ? g.x : 0.f;
                                g.y = ((bmb[i] >> (2 * e + 1)) & 1u) ? g.y : 0.f;
                            } else {
                                const f32x2 yy = bf2f_pair(yw[e]);
                                g.x = (yy.x > 0.f) ? g.x : 0.f;
                                g.y = (yy.y > 0.f) ? g.y : 0.f;
                            }
                        }
                        if (a.bn_nscale) g *= ns[e];
                        s2[e] += g;
                        q2[e] += g * ((bf2f_pair(xw[e]) - k1[e]) * k0[e]);
                    }
                }
                if (!(TSKIP(a) & 128)) *(uint4*)(a.y + (size_t)m * a.ldy + co) = uint4{ow[0], ow[1], ow[2], ow[3]};
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[2 * e] = s2[e].x; s[2 * e + 1] = s2[e].y; q[2 * e] = q2[e].x; q[2 * e + 1] = q2[e].y; }
}

// ---- epilogue shared by the convolution kernels: accumulators -> bf16 C tile [pixel][cout] in LDS (`smem`, BP * CSTR
// bytes + the reduction scratch behind it) -> 16-byte rows to memory, with the fused residual / inference BatchNorm /
// BatchNorm statistics / BatchNorm-backward sums.  s, q: the per-channel sums of this thread's channel vector,
// carried by the caller; flush: reduce them over the workgroup and add them to statistics replica `replica`.
// KIND >= 0: the caller was compiled for one fused variant (conv1x1_stream_kernel: with the seven variants inside its tile
// loop the compiler merges their pending-load states at the back edge and opens every tile with `s_waitcnt vmcnt(0)`, and
// the kernel carries 193 registers instead of 114); KIND < 0: chosen here from the arguments.
template <int BC, int BP, int WC, int WP, bool RAW = false, int KIND = -1, int WMAP = 0>
static __device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[BC / WC / 32][BP / WP / 32],
                                                     unsigned char* smem, int m0, int c0, float (&s)[8], float (&q)[8],
                                                     bool flush, int replica, const EpiParams* pre = nullptr) {
    constexpr int NW = WC * WP, NT = 64 * NW;
    constexpr int FI = BC / WC / 32, FJ = BP / WP / 32;
    constexpr int CSTR = BC * 2 + 16;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wc = wave % WC, wp = wave / WC;
    const int lrow = lane & 31, lk = lane >> 5;
    // ---- epilogue: accumulators -> bf16 C tile [pixel][cout] in LDS
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            int px = wp * (BP / WP) + j * 32 + lrow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int co = wc * (BC / WC) + i * 32 + 8 * g + 4 * lk;
                uint2 pk;
                pk.x = pack2bf(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1]);
                pk.y = pack2bf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                *(uint2*)(smem + px * CSTR + co * 2) = pk;
            }
        }
    if constexpr (RAW) {
        // LDS only: `__syncthreads()` also waits for every memory operation in flight (conv1x1_stream_kernel: the tiles
        // prefetched for the next iterations and the previous tile's row stores)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
        __syncthreads();
    }
    if (TDBG(a) && t == 0) TDBG(a)[(16384 + blockIdx.x) * 4 + 0] = __builtin_readcyclecounter();
    constexpr int VPR = BC / 8;              // 16-byte vectors per C row
    const int cv = t % VPR;
    // the row pass, specialised per fused variant (uniform branches; each variant is one straight block)
    if constexpr (KIND >= 0) epilogue_rows<BC, BP, NT, KIND, RAW, WMAP>(a, smem, m0, c0, s, q, pre);
    else if (a.ev_rm) epilogue_rows<BC, BP, NT, EPI_EV, false, WMAP>(a, smem, m0, c0, s, q);
    else if (a.bn_x) {
        if (a.bn_relu == 2) epilogue_rows<BC, BP, NT, EPI_BNX2, false, WMAP>(a, smem, m0, c0, s, q);
        else epilogue_rows<BC, BP, NT, EPI_BNX, false, WMAP>(a, smem, m0, c0, s, q);
    }
    else if (a.res) {
        if (a.stats) epilogue_rows<BC, BP, NT, EPI_RES_STATS, false, WMAP>(a, smem, m0, c0, s, q);
        else epilogue_rows<BC, BP, NT, EPI_RES, false, WMAP>(a, smem, m0, c0, s, q);
    } else if (a.stats) epilogue_rows<BC, BP, NT, EPI_STATS, false, WMAP>(a, smem, m0, c0, s, q);
    else epilogue_rows<BC, BP, NT, EPI_PLAIN, false, WMAP>(a, smem, m0, c0, s, q);
    if (TDBG(a) && t == 0) TDBG(a)[(16384 + blockIdx.x) * 4 + 1] = __builtin_readcyclecounter();
    if (a.stats && flush) {
        // lanes with equal cv inside a wave: strides VPR, 2*VPR, ... < 64
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (VPR <= 8) { s[e] = xor_add<8>(s[e]); q[e] = xor_add<8>(q[e]); }
            if constexpr (VPR <= 16) { s[e] = xor_add<16>(s[e]); q[e] = xor_add<16>(q[e]); }
            if constexpr (VPR <= 32) { s[e] = xor_add<32>(s[e]); q[e] = xor_add<32>(q[e]); }
        }
        float* red = (float*)(smem + BP * CSTR);     // [NW waves][2][BC]
        if (lane < VPR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 2 + 0) * BC + cv * 8 + e] = s[e];
                red[(wave * 2 + 1) * BC + cv * 8 + e] = q[e];
            }
        }
        // LDS only: `__syncthreads()` would also wait for this thread's ROW STORES above to be acknowledged by memory
        // (vmcnt counts stores on gfx9), ~1000 cycles the statistics do not depend on
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (TDBG(a) && t == 0) TDBG(a)[(16384 + blockIdx.x) * 4 + 2] = __builtin_readcyclecounter();
        if (t < 2 * BC) {
            int which = t / BC, c = t % BC;
            float tot = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) tot += red[(wv * 2 + which) * BC + c];
            if (c0 + c < a.Cout) {
                rgda_stat_t* dst = &a.stats[(((size_t)(m0 / a.rows_per_group) * NREP + replica) * 2 + which) * a.Cout + c0 + c];
#ifdef RGDA_TUNING      // timing experiments (wrong results): 16 = no statistics atomics, 32 = fp32 atomics on the same words
                if (TSKIP(a) & 16) {}
                else if (TSKIP(a) & 32) atomicAdd((float*)dst, tot);
                else
#endif
                stat_add(dst, tot, a.bn_x ? RGDA_STAT_FRAC_BWD : RGDA_STAT_FRAC_FWD);
            }
        }
    }
}

// XF: the pixel operand is  relu(BatchNorm(x))  of the producing convolution's RAW output x, applied on the way into LDS
// (ConvArgs::xf, common.h: BnOperand): the rows travel global -> registers -> fma / max / bf16 -> ds_write_b128 instead of by
// LDS-DMA (padding rows are written as zeros: the activation of a padding pixel is 0, not relu(shift)); the weight tiles
// keep the DMA ring.  Written for the 2-stage ring (two workgroups per CU hide each other's load latency).
// (the body: `bid` = the workgroup's index inside ITS problem -- blockIdx.x for a single launch, blockIdx.x minus the problem's
// first workgroup for conv_igemm_grouped_kernel)
template <int BC, int BP, int STAGES, int WC, int WP, bool PIPE, bool XF>
static __device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, const int bid) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource builtins exist in the device pass only
    constexpr int NW = WC * WP;                         // waves per workgroup
    constexpr int FI = BC / WC / 32, FJ = BP / WP / 32;  // 32x32 accumulators per wave
    constexpr int WL = BC / (NW * 8), XL = BP / (NW * 8);   // LDS-DMA instructions per wave per tile (1 KiB each)
    constexpr int LD = WL + XL;
    constexpr int TILE = (BC + BP) * 128;               // bytes of one K tile (64 channels)
    constexpr int CSTR = BC * 2 + 16;                   // epilogue row stride (bytes)
    constexpr int EPI = BP * CSTR + NW * BC * 2 * 4;
    constexpr int SMEM = (STAGES * TILE > EPI) ? STAGES * TILE : EPI;
    constexpr int XFTAB = XF ? RGDA_BNIN_MAX_C * 2 * 4 : 0;      // (scale, shift) of the operand's channels
    static_assert(!XF || (STAGES == 2 && !PIPE), "the operand-transform loop is written for the 2-stage ring");
    __shared__ __attribute__((aligned(256))) unsigned char smem[SMEM + XFTAB];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wc = wave % WC, wp = wave / WC;
    const int logical = xcd_remap(bid, a.tiles_c * a.tiles_p);
    const int c0 = (logical % a.tiles_c) * BC;
    const int m0 = (logical / a.tiles_c) * BP;
    const int taps = a.KH * a.KW;
    // LDS-DMA geometry: instruction q of this wave fills rows (q*NW + wave)*8 .. +8 of the tile; lane -> (row, slot).
    // Loads are buffer_load_dwordx4 ... lds: per-lane 32-bit byte offset (fixed for W rows, recomputed once
    // per filter tap for the gathered pixel rows) + ONE scalar offset per K tile; rows that are padding /
    // out of range carry an out-of-range offset and the hardware deposits zeros (tests/test_hw_semantics_gpu.py).
    const int lrow8 = lane >> 3, lslot = lane & 7;
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, (int)((size_t)a.Cout * taps * a.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((((size_t)a.N * a.H * a.W - 1) * a.ldx + a.Cin) * 2), 0x00020000);

    // ---- per-thread pixel rows (fixed for the whole K loop)
    int xn[XL], xh[XL], xw[XL], xsw[XL];
    bool xm[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        int r = (i * NW + wave) * 8 + lrow8;
        int m = m0 + r;
        xm[i] = m < a.M;
        int mm = xm[i] ? m : 0;
        int n, rem, ho, wo;       // 32-bit divisions cost ~50 instructions each: shifts where the map sizes allow
        if (a.howo_shift >= 0) { n = mm >> a.howo_shift; rem = mm & ((1 << a.howo_shift) - 1); }
        else { n = mm / (a.Ho * a.Wo); rem = mm % (a.Ho * a.Wo); }
        if (a.wo_shift >= 0) { ho = rem >> a.wo_shift; wo = rem & ((1 << a.wo_shift) - 1); }
        else { ho = rem / a.Wo; wo = rem % a.Wo; }
        xn[i] = n * a.H * a.W;
        if (a.mode == 0) { xh[i] = ho * a.stride - a.pad; xw[i] = wo * a.stride - a.pad; }
        else             { xh[i] = ho + a.pad;            xw[i] = wo + a.pad; }
        // DMA: the swizzle lives on the SOURCE side (the LDS image is lane-linear); XF: the lane fetches its own logical
        // slot (so that its channels, hence its scale / shift registers, are the same for every row) and swizzles the STORE
        xsw[i] = XF ? lslot * 16 : (lslot ^ ((r >> 1) & 7)) * 16;
    }
    int wvo[WL];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        int r = (i * NW + wave) * 8 + lrow8;
        int co = c0 + r;
        wvo[i] = (co < a.Cout) ? (co * taps * a.Cin * 2 + (lslot ^ ((r >> 1) & 7)) * 16) : OOB;
    }

    // ---- loader state
    int tap = 0, kh = 0, kw = 0, ci0 = 0;
    int xvo[XL];
    auto retap = [&]() {
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            int hi, wi;
            bool okh, okw;
            src_coord(a.mode, xh[i], kh, a.dil, a.stride, a.H, hi, okh);
            src_coord(a.mode, xw[i], kw, a.dil, a.stride, a.W, wi, okw);
            xvo[i] = (xm[i] && okh && okw) ? ((xn[i] + hi * a.W + wi) * a.ldx * 2 + xsw[i]) : OOB;
        }
    };
    retap();
    auto issue = [&](int stage) {
        unsigned char* wb = smem + stage * TILE + wave * 1024;
        const int so_w = (tap * a.Cin + ci0) * 2, so_x = ci0 * 2;
#pragma unroll
        for (int i = 0; i < WL; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(wb + i * NW * 1024), 16, wvo[i], so_w, 0, 0);
        unsigned char* xb = wb + BC * 128;
#pragma unroll
        for (int i = 0; i < XL; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, LDS_PTR(xb + i * NW * 1024), 16, xvo[i], so_x, 0, 0);
    };
    auto advance = [&]() {
        ci0 += 64;
        if (ci0 >= a.Cin) {
            ci0 = 0;
            ++tap;
            if (++kw == a.KW) { kw = 0; ++kh; }
            retap();
        }
    };

    f32x16 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT = taps * (a.Cin >> 6);
    unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, twait = 0, tbar = 0;
    if (TDBG(a)) tq0 = __builtin_readcyclecounter();
    const int lrow = lane & 31, lk = lane >> 5;
    int stage = 0;
    if constexpr (XF) {
        float* xtab = (float*)(smem + SMEM);
        auto issue_w = [&](int st) {
            unsigned char* wb = smem + st * TILE + wave * 1024;
            const int so_w = (tap * a.Cin + ci0) * 2;
#pragma unroll
            for (int i = 0; i < WL; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(wb + i * NW * 1024), 16, wvo[i], so_w, 0, 0);
        };
        u32x4r xr[XL];
        bool xok[XL];
        auto load_x = [&]() {
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                xok[i] = xvo[i] != OOB;
                xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xvo[i], ci0 * 2, 0);
            }
        };
        issue_w(0);
        load_x();
        if (!(TSKIP(a) & 256)) bn_operand_table<64 * NW>(a.xf, m0 / a.rows_per_group, logical == 0, xtab);
        __syncthreads();
        bf16x8 af[4][FI], bfr[4][FJ];
        int ci_cur = 0;                                     // first channel of the tile held in xr
        for (int kt = 0; kt < KT; ++kt) {
            // ---- tile kt: registers -> BatchNorm + ReLU -> its LDS stage (the stage's last readers finished before the
            // previous barrier); the weight tile of the same stage was DMA'd one iteration ago
            {
                float sc[8], sh[8];
                const f32x4* ts = (const f32x4*)(xtab + ci_cur + lslot * 8);
                const f32x4* th = (const f32x4*)(xtab + a.Cin + ci_cur + lslot * 8);
                const f32x4 s0 = ts[0], s1 = ts[1], h0 = th[0], h1 = th[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; sh[e] = h0[e]; sh[4 + e] = h1[e]; }
                unsigned char* xb = smem + stage * TILE + BC * 128;
#pragma unroll
                for (int i = 0; i < XL; ++i) {
                    const int r = (i * NW + wave) * 8 + lrow8;
                    uint4 v = uint4{xr[i][0], xr[i][1], xr[i][2], xr[i][3]};
                    if (!(TSKIP(a) & 512)) v = xok[i] ? bn_operand8(v, sc, sh, a.xf.relu != 0) : uint4{0u, 0u, 0u, 0u};
                    *(uint4*)(xb + r * 128 + ((lslot ^ ((r >> 1) & 7)) << 4)) = v;
                }
            }
            WAIT_VMCNT(0);                                   // this wave's pieces of weight tile kt have landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < KT) {
                advance();
                issue_w(stage ^ 1);
                load_x();
                ci_cur = ci0;
            }
            const unsigned char* wb = smem + stage * TILE;
            const unsigned char* xb = wb + BC * 128;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int i = 0; i < FI; ++i) {
                    int r = wc * (BC / WC) + i * 32 + lrow;
                    af[kk][i] = *(const bf16x8*)(wb + r * 128 + (((kk * 2 + lk) ^ ((r >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    int r = wp * (BP / WP) + j * 32 + lrow;
                    bfr[kk][j] = *(const bf16x8*)(xb + r * 128 + (((kk * 2 + lk) ^ ((r >> 1) & 7)) << 4));
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < FJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][i], bfr[kk][j], acc[i][j], 0, 0, 0);
            stage ^= 1;
        }
    } else if constexpr (PIPE) {
        // ---- software-pipelined K loop: the barrier of K tile kt+1 sits BETWEEN the two halves of tile kt's MFMAs.
        // Fragments of k-steps 0-1 of the next tile are read behind the barrier and land while k-steps 2-3 of this
        // tile run; fragments of k-steps 2-3 are read at the top and land while k-steps 0-1 run.  The MFMA stream of
        // a wave is then continuous -- the plain loop pays the LDS latency (~250 cycles) once per K tile with the
        // matrix pipe idle.  All three stages are in flight: tile kt+3 refills the stage of tile kt as soon as every
        // wave is past the barrier (its k-step 2-3 reads are complete by then: explicit lgkmcnt(0)).
        static_assert(!PIPE || STAGES == 3, "the pipelined loop is written for a 3-stage ring");
        issue(0);
        if (1 < KT) { advance(); issue(1); }
        if (2 < KT) { advance(); issue(2); }
        { const int younger = min(KT - 1, 2); if (younger == 2) WAIT_VMCNT(2 * LD); else if (younger == 1) WAIT_VMCNT(LD); else WAIT_VMCNT(0); }
        __builtin_amdgcn_s_barrier();
        if (TDBG(a)) tq1 = __builtin_readcyclecounter();
        bf16x8 fa[4][FI], fb[4][FJ];
        auto read_half = [&](int st, int h) {
            const unsigned char* wb = smem + st * TILE;
            const unsigned char* xb = wb + BC * 128;
#pragma unroll
            for (int kk = 2 * h; kk < 2 * h + 2; ++kk) {
#pragma unroll
                for (int i = 0; i < FI; ++i) {
                    int r = wc * (BC / WC) + i * 32 + lrow;
                    fa[kk][i] = *(const bf16x8*)(wb + r * 128 + (((kk * 2 + lk) ^ ((r >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    int r = wp * (BP / WP) + j * 32 + lrow;
                    fb[kk][j] = *(const bf16x8*)(xb + r * 128 + (((kk * 2 + lk) ^ ((r >> 1) & 7)) << 4));
                }
            }
        };
        auto mfma_half = [&](int h) {
#pragma unroll
            for (int kk = 2 * h; kk < 2 * h + 2; ++kk)
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < FJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
        };
        // (spreading each half's fragment reads between the other half's MFMAs with sched_group_barrier, what pays in
        // conv3x3_halo_kernel<.., NS = 4>, measured no different here: round 5, scripts/dev/iso_set.py)
        read_half(0, 0);
        // (the last tile is peeled: a branch around the barrier block would merge two different LDS-counter states
        // and make the compiler wait for the NEXT tile's fragments before this tile's second half)
        for (int kt = 0; kt + 1 < KT; ++kt) {
            const int nxt = (stage == 2) ? 0 : stage + 1;
            read_half(stage, 1);
            mfma_half(0);
#ifdef RGDA_TUNING      // timing experiment (racy: wrong results): 64 = do not wait for the fragment reads before the barrier
            if (!(TSKIP(a) & 64))
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this tile's k-step 2-3 fragments are in registers
            if (kt + 2 < KT) WAIT_VMCNT(LD); else WAIT_VMCNT(0);        // tile kt+1 landed (kt+2 may still fly)
            __builtin_amdgcn_s_barrier();
            if (kt + 3 < KT) { advance(); issue(stage); }               // tile kt+3 -> the stage tile kt vacated
            read_half(nxt, 0);
            mfma_half(1);
            stage = nxt;
        }
        read_half(stage, 1);
        mfma_half(0);
        mfma_half(1);
    } else {
    issue(0);
#pragma unroll
    for (int p = 1; p < STAGES - 1; ++p)
        if (p < KT) { advance(); issue(p); }
    bf16x8 af[4][FI], bfr[4][FJ];
    for (int kt = 0; kt < KT; ++kt) {
        // tile kt has landed once only the loads of the (up to STAGES-2) younger tiles are outstanding
        const int younger = min(KT - 1 - kt, STAGES - 2);
        unsigned long long tw0 = 0;
        if (TDBG(a)) tw0 = __builtin_readcyclecounter();
        if (younger >= 2) WAIT_VMCNT(2 * LD); else if (younger == 1) WAIT_VMCNT(LD); else WAIT_VMCNT(0);
        if (TDBG(a)) { unsigned long long tw1 = __builtin_readcyclecounter(); twait += tw1 - tw0; tw0 = tw1; }
        __builtin_amdgcn_s_barrier();
        if (TDBG(a)) tbar += __builtin_readcyclecounter() - tw0;
        if (TDBG(a) && kt == 0) tq1 = __builtin_readcyclecounter();
        const bool more = kt + STAGES - 1 < KT && !(TSKIP(a) & 1);
        if (more && !(TSKIP(a) & 4)) {
            advance();
            issue(stage >= 1 ? stage - 1 : STAGES - 1);      // (kt + STAGES - 1) % STAGES
        }
        if (TSKIP(a) & 2) { stage = (stage == STAGES - 1) ? 0 : stage + 1; continue; }
        const unsigned char* wb = smem + stage * TILE;
        const unsigned char* xb = wb + BC * 128;
        // all fragments of the K tile are fetched up front (4 k-steps x (FI+FJ) x 16 B per lane), then the MFMAs
        // run back to back: the LDS latency is paid once per K tile instead of once per k-step
        if (!(TSKIP(a) & 8) || kt == 0) {       // (tuning: bit 8 keeps the first tile's fragments -> MFMA without LDS reads)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                int r = wc * (BC / WC) + i * 32 + lrow;
                af[kk][i] = *(const bf16x8*)(wb + r * 128 + (((kk * 2 + lk) ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                int r = wp * (BP / WP) + j * 32 + lrow;
                bfr[kk][j] = *(const bf16x8*)(xb + r * 128 + (((kk * 2 + lk) ^ ((r >> 1) & 7)) << 4));
            }
        }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][i], bfr[kk][j], acc[i][j], 0, 0, 0);
        if (more && (TSKIP(a) & 4)) {          // tuning: DMA issued behind the MFMAs instead of in front of the LDS reads
            advance();
            issue(stage >= 1 ? stage - 1 : STAGES - 1);
        }
        stage = (stage == STAGES - 1) ? 0 : stage + 1;
    }
    }
    __syncthreads();
    if (TDBG(a)) tq2 = __builtin_readcyclecounter();

    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    conv_epilogue<BC, BP, WC, WP>(a, acc, smem, m0, c0, s, q, true, blockIdx.x & (NREP - 1));
    if (TDBG(a) && t == 0) {
        unsigned long long tq3 = __builtin_readcyclecounter();
        TDBG(a)[blockIdx.x * 4 + 0] = tq0; TDBG(a)[blockIdx.x * 4 + 1] = tq1;
        TDBG(a)[blockIdx.x * 4 + 2] = tq2; TDBG(a)[blockIdx.x * 4 + 3] = tq3;
        TDBG(a)[(gridDim.x + blockIdx.x) * 4 + 0] = twait; TDBG(a)[(gridDim.x + blockIdx.x) * 4 + 1] = tbar;
    }
#endif
}

template <int BC, int BP, int STAGES = 3, int WC = 2, int WP = 2, bool PIPE = false, bool XF = false>
__global__ void __launch_bounds__(64 * WC * WP) conv_igemm_kernel(ConvArgs a) {
    conv_igemm_body<BC, BP, STAGES, WC, WP, PIPE, XF>(a, blockIdx.x);
}

// Several INDEPENDENT small convolutions in one launch (rgda_conv2d_grouped): the four scales of a PPM head's branch
// convolutions are 4 - 36 workgroups each and run 32 - 72 K tiles -- alone on the chip each is a 17 - 35 us latency chain,
// eight of them in a row 140 us of a head's forward.  The problems' argument blocks travel by value (kernarg, <= 8 x 352 B);
// a workgroup finds its problem from the prefix of workgroup counts.
constexpr int CONV_GROUP_MAX = 8;
struct ConvGroup {
    ConvArgs a[CONV_GROUP_MAX];
    int start[CONV_GROUP_MAX + 1];
    int n;
};
static_assert(sizeof(ConvGroup) <= 4096, "kernel arguments");
template <int BC, int BP, int STAGES, int WC, int WP, bool PIPE>
__global__ void __launch_bounds__(64 * WC * WP) conv_igemm_grouped_kernel(ConvGroup g) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < CONV_GROUP_MAX; ++i)
        if (i < g.n && (int)blockIdx.x >= g.start[i]) p = i;
    conv_igemm_body<BC, BP, STAGES, WC, WP, PIPE, false>(g.a[p], (int)blockIdx.x - g.start[p]);
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / "same" convolutions with a long K on 32-wide maps (the heads' 2048 -> 512, layer 4's 512 -> 512 with
// dilation 1 / 2; forward and data gradient): 128 x 256 tiles like conv_igemm_kernel<128, 256, ...>, but the pixel operand
// is loaded ONCE per 64-channel slab as the tile's halo -- (8 + 2 D) x (32 + 2 D) pixels -- and the nine taps read it at
// nine shifted row offsets.  Per slab a workgroup pulls 9 x 16 KB of weights + one 43.5 KB halo (54 KB for D = 2) instead
// of 9 x 48 KB: these launches are bound by the L2 -> LDS stream (with the pixel pieces made out-of-range, i.e. no
// traffic, the head convolution runs 339 -> 248 us; section 4.4), so they get the bytes they no longer move back.
//   LDS: two halo buffers (slab s + 1 lands while slab s is multiplied; its pieces are issued one per tap step) + a
//   3-stage ring of weight tiles = 144 KB (D = 1) / 160 KB (D = 2); one barrier per tap step; the counted vmcnt wait
//   lets exactly the pieces of the previous step stay in flight.
// XF (forward only): the halo holds the producing convolution's RAW output; every wave turns the pieces IT deposited into
// relu(BatchNorm(x)) in place -- slab s + 1's piece i two tap steps after it was issued (the counted wait at the top of a
// step leaves only the previous step's pieces in flight, and a wave needs no barrier to see its own DMA data); halo rows
// outside the image stay the zeros the DMA deposited.  One 16-byte vector per thread and tap step.
// NS = 4 (a fourth weight stage): the K loop is software-pipelined ACROSS its barriers.  The wait at the top of step q then
// covers the weight tile of step q + 1 (issued three steps ahead instead of two), so behind barrier q every wave may read
// stage q + 1 as well: the fragments of step q + 1's first two k-slices are read under the last MFMAs of step q, and no
// wave opens a step with an LDS round trip in front of its first MFMA (with one barrier per step all eight waves of the
// workgroup did -- both waves of every SIMD at the same time, matrix pipes idle).
// (Round 5, measured and not kept: "loader waves" -- the first four waves, one per SIMD, issue ALL of a step's DMA instructions
// so that a SIMD always has one wave multiplying.  Isolated launches on random data: head convolution 278 -> 239 us, layer 3's
// 22.2 -> 21.5; inside the step, by kernel trace on one box: 1756 -> 1745 us and 1059 -> 1064 us per step, and with the operand
// transform 584 -> 736.  The step runs these kernels on ReLU-sparse data at a higher clock; what an isolated launch on N(0, 1)
// operands gains in issue slots does not exist there.  Kernel changes are judged by the step's kernel trace.)
// WIDE (conv3x3_halo_wide_kernel): maps whose width is a MULTIPLE of 32 (1024 x 1024 tiles at stride 16: 64 columns) -- a tile
// is TR image rows x one 32-column band; its halo columns come from the neighbouring bands (zeros only at the map's edges) and
// its output rows are 32-pixel pieces W apart (epilogue_rows<.., WMAP>).  Everything else is the 32-wide kernel.
template <int D, int TR, bool XF, int NS, bool PF, bool WIDE>
static __device__ __forceinline__ void conv3x3_halo_body(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TC = 32;                                  // pixel tile: TR image rows x 32 columns (the whole map width unless WIDE)
    constexpr int BC = 128, BP = TR * TC, WC = 2, WP = 4, NW = 8, FI = 2, FJ = TR / WP;
    constexpr int HR = TR + 2 * D, HC = TC + 2 * D, NH = HR * HC;
    constexpr int PXW = (NH + 63) / 64;                     // halo pieces (8 rows of 128 B) per wave and slab
    constexpr int XS = PXW * 64 * 128, WS = BC * 128;       // bytes of a halo buffer / a weight stage
    constexpr int CSTR = BC * 2 + 16, EPI = BP * CSTR + NW * BC * 2 * 4;
    constexpr int SMEM = 2 * XS + NS * WS;
    // PF with NS = 3 (dilation 2: four stages do not fit 160 KB): the weight tile of step q + 1 is issued ONE step ahead and
    // waited for in full at the top of step q (only the previous step's halo piece stays in flight)
    constexpr int AH = NS - 1;                              // PF: weight tiles issued ahead
    static_assert(NS == 3 || NS == 4, "ring depth");
    static_assert(!PF || PXW <= 7, "the next slab's halo is complete behind the barrier of tap 8");
    static_assert(!(PF && XF) || PXW + 3 <= 9, "... and transformed");
    constexpr int XFTAB = XF ? RGDA_BNIN_MAX_C * 2 * 4 : 0;
    static_assert(EPI <= SMEM && SMEM + XFTAB <= 160 * 1024, "LDS budget");
    // the pipelined loop leaves 'dead' (out-of-range, zero-depositing) weight DMAs of tiles >= KT in flight when it exits;
    // they land in weight stages, which the epilogue's staging image must therefore never reach
    static_assert(!PF || EPI <= 2 * XS, "the epilogue image stays inside the two halo buffers (late weight DMAs land behind it)");
    static_assert(!XF || PXW + 2 <= 9, "every piece of the next slab is transformed before the slab ends");
    __shared__ __attribute__((aligned(256))) unsigned char smem[SMEM + XFTAB];
    unsigned char* const sxb = smem;
    unsigned char* const swb = smem + 2 * XS;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wc = wave % WC, wp = wave / WC;
    const int lrow = lane & 31, lk = lane >> 5, lrow8 = lane >> 3, lslot = lane & 7;
    const int logical = xcd_remap(blockIdx.x, a.tiles_c * a.tiles_p);
    const int c0 = (logical % a.tiles_c) * BC;
    const int pt = logical / a.tiles_c;
    const int bands = WIDE ? a.W / TC : 1, tpi = (a.H / TR) * bands;
    const int n = pt / tpi, y0 = ((pt % tpi) / bands) * TR, x0 = WIDE ? ((pt % tpi) % bands) * TC : 0;
    const int mapw = WIDE ? a.W : TC;
    const int m0 = WIDE ? (n * a.H + y0) * a.W + x0 : pt * BP;     // memory row of the tile's first pixel
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, (int)((size_t)a.Cout * 9 * a.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((((size_t)a.N * a.H * a.W - 1) * a.ldx + a.Cin) * 2), 0x00020000);
    int wvo[2], hvo[PXW];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (i * NW + wave) * 8 + lrow8, co = c0 + r;
        wvo[i] = (co < a.Cout) ? (co * 9 * a.Cin * 2 + (lslot ^ ((r >> 1) & 7)) * 16) : OOB;
    }
#pragma unroll
    for (int i = 0; i < PXW; ++i) {
        const int h = (i * NW + wave) * 8 + lrow8;
        const int y = y0 - D + h / HC, x = x0 + h % HC - D;
        const bool ok = h < NH && y >= 0 && y < a.H && x >= 0 && x < mapw;
        hvo[i] = ok ? (((n * a.H + y) * mapw + x) * a.ldx * 2 + (lslot ^ ((h >> 1) & 7)) * 16) : OOB;
    }
    const i32x4 rs_wa = dma_rsrc(a.w, (unsigned)((size_t)a.Cout * 9 * a.Cin * 2));
    // live = false: the same instructions with out-of-range offsets (no traffic, zeros into a stage nobody reads): the
    // pipelined loop issues a constant number of DMA instructions per step, so its vmcnt waits are compile-time constants
    auto issue_w2 = [&](int tap, int sl, int stage, bool live) {
        const int so = (tap * a.Cin + sl * 64) * 2;
        unsigned char* wb = swb + stage * WS + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int vo = live ? wvo[i] : OOB;
            if constexpr (XF) dma16_to_lds(rs_wa, wb + i * NW * 1024, vo, so);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(wb + i * NW * 1024), 16, vo, so, 0, 0);
        }
    };
    auto issue_w = [&](int q, int stage, bool live = true) {
        const int tap = q % 9, sl = q / 9;
        const int so = (tap * a.Cin + sl * 64) * 2;
        unsigned char* wb = swb + stage * WS + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int vo = live ? wvo[i] : OOB;
            if constexpr (XF) dma16_to_lds(rs_wa, wb + i * NW * 1024, vo, so);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(wb + i * NW * 1024), 16, vo, so, 0, 0);
        }
    };
    const i32x4 rs_xa = dma_rsrc(a.x, (unsigned)((((size_t)a.N * a.H * a.W - 1) * a.ldx + a.Cin) * 2));
    auto issue_h = [&](int sl, int i, bool live = true) {
        unsigned char* xb = sxb + (sl & 1) * XS + wave * 1024 + i * NW * 1024;
        const int vo = live ? hvo[i] : OOB;
        if constexpr (XF) dma16_to_lds(rs_xa, xb, vo, sl * 128);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, LDS_PTR(xb), 16, vo, sl * 128, 0, 0);
    };

    f32x16 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int S = a.Cin >> 6, KT = 9 * S;
    int hb[FJ];                                             // halo row of this lane's pixel (tap 0,0) per pixel block
#pragma unroll
    for (int j = 0; j < FJ; ++j) hb[j] = (wp * FJ + j) * HC + lrow;
    int wr[FI];                                             // weight-tile row byte offset and its swizzle key
#pragma unroll
    for (int i = 0; i < FI; ++i) wr[i] = wc * 64 + i * 32 + lrow;
    // one half (two of the four 16-wide K slices) of a tap step's fragments
    auto read_half = [&](const unsigned char* wb, const unsigned char* xb, int tp, int half, bf16x8 (&fa)[2][FI],
                         bf16x8 (&fb)[2][FJ]) {
        const int kh = tp / 3, kw = tp % 3;
        const int toff = (a.mode ? (2 - kh) : kh) * D * HC + (a.mode ? (2 - kw) : kw) * D;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int kk = half * 2 + k2;
#pragma unroll
            for (int i = 0; i < FI; ++i)
                fa[k2][i] = *(const bf16x8*)(wb + wr[i] * 128 + (((kk * 2 + lk) ^ ((wr[i] >> 1) & 7)) << 4));
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const int hr = hb[j] + toff;                // (the compiler hoists the nine taps' addresses out of the
                                                            // slab loop: ~190 registers, no spill; recomputing them per
                                                            // tap measured 5 % slower)
                fb[k2][j] = *(const bf16x8*)(xb + hr * 128 + (((kk * 2 + lk) ^ ((hr >> 1) & 7)) << 4));
            }
        }
    };
    auto mfma_half = [&](bf16x8 (&fa)[2][FI], bf16x8 (&fb)[2][FJ]) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k2][i], fb[k2][j], acc[i][j], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < PXW; ++i) issue_h(0, i);
    issue_w(0, 0);
    issue_w(1, 1);
    if constexpr (PF && NS == 4) issue_w(2, 2);
    // ---- XF: this thread's vector of halo piece k is row k * 64 + (t >> 3), LOGICAL slot lslot (the channels, hence the
    // scale / shift registers, are the same for every row); those are bytes of the wave's own DMA instructions
    float xsc[8], xsh[8];
    float* const xtab = (float*)(smem + SMEM);
    auto xf_load = [&](int sl) {
        const f32x4* ts = (const f32x4*)(xtab + sl * 64 + lslot * 8);
        const f32x4* th = (const f32x4*)(xtab + a.Cin + sl * 64 + lslot * 8);
        const f32x4 s0 = ts[0], s1 = ts[1], h0 = th[0], h1 = th[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) { xsc[e] = s0[e]; xsc[4 + e] = s1[e]; xsh[e] = h0[e]; xsh[4 + e] = h1[e]; }
    };
    auto xf_piece = [&](int sl, int k) {
        if (hvo[k] == OOB) return;                          // padding (or beyond the halo): stays zero
        const int h = k * 64 + (t >> 3);
        unsigned char* p = sxb + (sl & 1) * XS + h * 128 + ((lslot ^ ((h >> 1) & 7)) << 4);
        *(uint4*)p = bn_operand8(*(const uint4*)p, xsc, xsh, a.xf.relu != 0);
    };
    if constexpr (XF) {
        // (tuning builds: skip bit 256 = no statistics -> table prologue, 512 = no in-loop transform; timing only)
        if (!(TSKIP(a) & 256)) bn_operand_table<512>(a.xf, m0 / a.rows_per_group, logical == 0, xtab);
        WAIT_VMCNT(0);                                      // (the halo DMA is opaque to the compiler)
        __syncthreads();                                    // the table; also every DMA of the prologue has landed
        xf_load(0);
#pragma unroll
        for (int k = 0; k < PXW; ++k) xf_piece(0, k);
    }
    int pend = 2, wstage = 0;
    bf16x8 fa0[2][FI], fb0[2][FJ], fa1[2][FI], fb1[2][FJ];
    if constexpr (PF) {
        // the first step's first fragments: halo slab 0 and weight tile 0 have landed (tiles 1 and 2 stay in flight)
        WAIT_VMCNT(2 * (NS - 2));
        if constexpr (XF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_half(swb, sxb, 0, 0, fa0, fb0);
    }
    if constexpr (PF) {
      for (int sl = 0; sl < S; ++sl) {
        const bool more = sl + 1 < S;
        const unsigned char* xb = sxb + (sl & 1) * XS;
        const unsigned char* xbn = sxb + ((sl + 1) & 1) * XS;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int q = sl * 9 + tp;
            // everything but the previous step's DMA instructions has landed: weight tile q + 1 (and, with the barrier, every
            // wave's part of it); in flight: tile q + 2 and the halo piece of step tp - 1
            // (tuning builds, timing only: skip bit 1 = no DMA inside the loop, 2 = no vmcnt wait, 4 = no barrier)
            if (!(TSKIP(a) & 2)) { if (tp >= 1 && tp - 1 < PXW) WAIT_VMCNT(2 * (NS - 3) + 1); else WAIT_VMCNT(2 * (NS - 3)); }
            if constexpr (XF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the previous step's in-place transform
            if (!(TSKIP(a) & 4)) __builtin_amdgcn_s_barrier();
            if (!(TSKIP(a) & 1)) {
            issue_w2((tp + AH) % 9, sl + (tp + AH) / 9, wstage >= 1 ? wstage - 1 : NS - 1, q + AH < KT);
            if (tp < PXW) issue_h(sl + 1, tp, more);
            }
            if constexpr (XF) {
                if (more) {
                    if (tp == 1) xf_load(sl + 1);
                    if (tp >= 2 && tp - 2 < PXW) xf_piece(sl + 1, tp - 2);
                }
            }
            // k-slices 2, 3 of this step are read while 0, 1 (read under the previous step) multiply; then the next step's
            // k-slices 0, 1 -- the stage after this one, the other halo buffer behind tap 8 -- under 2, 3.  (Behind the last
            // step the prefetch reads a stage that holds nothing new: unconditional, so that the step stays ONE basic block.)
            read_half(swb + wstage * WS, xb, tp, 1, fa1, fb1);
            mfma_half(fa0, fb0);
            read_half(swb + (wstage == NS - 1 ? 0 : wstage + 1) * WS, tp == 8 ? xbn : xb, tp == 8 ? 0 : tp + 1, 0, fa0, fb0);
            mfma_half(fa1, fb1);
            // one fragment read behind every MFMA (the compiler's own order -- all of a half's reads in one burst, then a
            // wait that also covers part of the NEXT burst -- stalls the first MFMA of each half on LDS latency)
#pragma unroll
            for (int i = 0; i < 4 * FI * FJ; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, (FI + FJ + FI * FJ - 1) / (FI * FJ), 0);   // its share of the LDS reads
            }
            wstage = (wstage == NS - 1) ? 0 : wstage + 1;
        }
      }
    } else {
    for (int sl = 0; sl < S; ++sl) {
        const bool more = sl + 1 < S;
        const unsigned char* xb = sxb + (sl & 1) * XS;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int q = sl * 9 + tp;
            switch (pend) {
                case 0: WAIT_VMCNT(0); break;
                case 1: WAIT_VMCNT(1); break;
                case 2: WAIT_VMCNT(2); break;
                default: WAIT_VMCNT(3); break;
            }
            if constexpr (XF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the previous step's in-place transform
            __builtin_amdgcn_s_barrier();
            pend = 0;
            if (!(TSKIP(a) & 1)) {      // (tuning builds: 1 = no DMA inside the loop, 8 = fragments read once, 2 = no reads / MFMAs)
            if (q + 2 < KT) { issue_w(q + 2, wstage >= 1 ? wstage - 1 : 2); pend += 2; }
            if (more && tp < PXW) { issue_h(sl + 1, tp); pend += 1; }
            }
            if constexpr (XF) {
                if (more && !(TSKIP(a) & 512)) {
                    if (tp == 1) xf_load(sl + 1);
                    if (tp >= 2 && tp - 2 < PXW) xf_piece(sl + 1, tp - 2);
                }
            }
            const unsigned char* wb = swb + wstage * WS;
            if (!(TSKIP(a) & 8) || q == 0) {
            read_half(wb, xb, tp, 0, fa0, fb0);
            read_half(wb, xb, tp, 1, fa1, fb1);
            }
            mfma_half(fa0, fb0);
            mfma_half(fa1, fb1);
            wstage = (wstage == NS - 1) ? 0 : wstage + 1;
        }
    }
    }
    __syncthreads();
    float s[8], q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q8[e] = 0.f; }
    conv_epilogue<BC, BP, WC, WP, false, -1, WIDE ? 1 : 0>(a, acc, smem, m0, c0, s, q8, true, blockIdx.x & (NREP - 1));
#endif
}
template <int D, int TR, bool XF = false, int NS = 3, bool PF = (NS == 4)>
__global__ void __launch_bounds__(512) conv3x3_halo_kernel(ConvArgs a) {
    conv3x3_halo_body<D, TR, XF, NS, PF, false>(a);
}
template <int D, int TR, int NS = 3, bool PF = (NS == 4), bool XF = false>
__global__ void __launch_bounds__(512) conv3x3_halo_wide_kernel(ConvArgs a) {
    conv3x3_halo_body<D, TR, XF, NS, PF, true>(a);
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 / stride 1 convolutions with a SHORT K (64 or 128 input channels) on large maps (layer 1's 64 -> 256, layer 2's
// 128 -> 512: forward, the data gradients of the 256 -> 64 / 512 -> 128 convolutions, the teacher).  One or two K tiles per
// output tile: conv_igemm_kernel has no steady state there -- a workgroup is prologue (index arithmetic, the first DMA and
// its full memory latency) and epilogue, 2.7-3.2 TB/s of the traffic that is all such a convolution is (64 -> 256: 1 byte
// read per 4 written), and every 128-pixel tile sends its 256 statistics atomics.  Here a persistent workgroup (one per CU)
// owns ONE channel tile and a CHUNK of consecutive pixel tiles: its 128 x K weights stay in registers (A fragments straight
// from memory, 32 / 64 VGPRs), the pixel tiles stream through a two-stage LDS ring one tile AHEAD of the tile being
// multiplied and stored, and the statistics leave once per workgroup.  64 -> 256 on 16 x 128 x 128 with statistics:
// 58 -> 37 us (4.6 TB/s); 128 -> 512 on 16 x 64 x 64: 36 -> 27 us.
//   Ordering (vmcnt retires in order, loads AND stores on gfx9): a tile ends with its SC = 4 row stores per thread (FULL
//   epilogue: every row and channel exists, the host guarantees it) and `s_waitcnt vmcnt(SC)` -- everything older than those
//   stores is complete: the epilogue's own loads (so the compiler carries no pending-load state into the next tile: with
//   conditional loads in the loop it opens every tile with `s_waitcnt vmcnt(0)`) AND the DMA of the next tile, issued at
//   the end of the previous one.  The tile after next is issued behind that wait, into the stage the MFMAs have just left.
//   The DMA is issued from inline assembly: through the builtin the compiler would put `s_waitcnt vmcnt(0)` in front of
//   the epilogue's LDS stores.  One instantiation per fused epilogue KIND (with all seven in one loop: 193 registers
//   instead of 106-158).
//   Tiles: 128 channels x 128 pixels, 8 waves as 2 x 4, two accumulators each.  (The template also builds 128 x 64 tiles
//   with the waves as 4 x 2 for 256 input channels -- layer 3's 256 -> 1024, whose 128-pixel stage would be 64 KB.  Measured
//   and NOT dispatched: 20.0 us against conv_igemm_kernel's 20.3 plain, 22.0 against 25.3 with statistics, the step
//   unchanged.  With neither DMA nor row stores that variant still takes 13.5 us: eight lockstep waves per CU walk a chain
//   of barrier -> fragment reads -> 16 dependent MFMAs -> accumulators to LDS -> barrier -> row pass per 64-pixel unit,
//   ~1.3 us each, and nothing else is resident to fill it.  The 64 / 128 channel shapes are bound by their row stores.)
// Requires: 1x1, stride 1, no padding, Cin = 64 KC, Cout % 128 == 0, M % (BP a.tpw) == 0, a chunk inside one statistics
// group; a.tpw = pixel tiles per workgroup, grid = tiles_c * tiles_p / a.tpw.
template <int KC, int BP, int WC, int WP, int KIND>
__global__ void __launch_bounds__(512) conv1x1_stream_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BC = 128, NW = 8, FI = BC / WC / 32, NS = 2;
    static_assert(WC * WP == NW && BP / WP == 32, "one 32-pixel fragment column per wave");
    constexpr int XL = BP / (NW * 8);                      // DMA instructions per wave and 64-channel sub-tile (1 KiB each)
    constexpr int SUB = BP * 128;                          // bytes of one sub-tile: BP pixels x 64 channels
    constexpr int STAGE = KC * SUB;
    constexpr int CSTR = BC * 2 + 16;
    constexpr int EPI = BP * CSTR + NW * BC * 2 * 4;
    constexpr int SC = BP * BC * 2 / 16 / (64 * NW);       // row stores per thread and tile (4 / 2)
    __shared__ __attribute__((aligned(256))) unsigned char smem[NS * STAGE + EPI];
    unsigned char* se = smem + NS * STAGE;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wc = wave % WC, wp = wave / WC;
    const int lrow = lane & 31, lk = lane >> 5;
    const int lrow8 = lane >> 3, lslot = lane & 7;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int c0 = (logical % a.tiles_c) * BC;             // channel tiles of one chunk are neighbours: they share the pixels in L2
    const int T = a.tpw;
    const int mbase = (logical / a.tiles_c) * T * BP;

    // the wave's BC / WC weight rows stay in registers for the whole chunk
    bf16x8 fa[KC * 4][FI];
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        const bf16_t* wrow = a.w + (size_t)(c0 + wc * (BC / WC) + i * 32 + lrow) * a.Cin + lk * 8;
#pragma unroll
        for (int kk = 0; kk < KC * 4; ++kk) fa[kk][i] = *(const bf16x8*)(wrow + kk * 16);
    }
    // the weights are waited for HERE: left to the compiler, the wait lands at their first use inside the loop and, being
    // in a loop, becomes `s_waitcnt vmcnt(0)` on every tile
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int kk = 0; kk < KC * 4; ++kk) asm volatile("" ::"v"(fa[kk][i]));
    const i32x4 rs_x = dma_rsrc(a.x, (unsigned)((((size_t)a.M - 1) * a.ldx + a.Cin) * 2));
    int xvo[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int r = (i * NW + wave) * 8 + lrow8;
        xvo[i] = r * a.ldx * 2 + (lslot ^ ((r >> 1) & 7)) * 16;
    }
    auto issue = [&](int tile, int stage) {
        const int so = (mbase + tile * BP) * a.ldx * 2;
#pragma unroll
        for (int kt = 0; kt < KC; ++kt)
#pragma unroll
            for (int i = 0; i < XL; ++i)
                dma16_to_lds(rs_x, smem + stage * STAGE + kt * SUB + (i * NW + wave) * 1024, xvo[i], so + kt * 128);
    };
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    // the epilogue's per-channel constants (inference BatchNorm scale / shift, BatchNorm-backward mean / invstd ...): once,
    // not once per tile -- their loads would sit in front of every row pass and, retiring in order, pull the prefetch in
    EpiParams P;
    epilogue_params<KIND>(a, mbase, c0 + (t % (BC / 8)) * 8, true, P);
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(P.k0[e]), "v"(P.k1[e]), "v"(P.k2[e]), "v"(P.k3[e]));
    issue(0, 0);
    if (T > 1) { issue(1, 1); WAIT_VMCNT(XL * KC); } else WAIT_VMCNT(0);          // tile 0 has landed
    for (int i = 0; i < T; ++i) {
        const int stage = i & 1;
        __builtin_amdgcn_s_barrier();                      // every wave's pieces of tile i are in LDS; tile i-1's C image is consumed
        f32x16 acc[FI][1];
#pragma unroll
        for (int ii = 0; ii < FI; ++ii)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ii][0][r] = 0.f;
        const int rb = wp * 32 + lrow;
        const unsigned char* xb = smem + stage * STAGE + rb * 128;
#pragma unroll
        for (int kt = 0; kt < KC; ++kt) {
            bf16x8 fb[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fb[kk] = *(const bf16x8*)(xb + kt * SUB + (((kk * 2 + lk) ^ ((rb >> 1) & 7)) << 4));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ii = 0; ii < FI; ++ii)
                    acc[ii][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kt * 4 + kk][ii], fb[kk], acc[ii][0], 0, 0, 0);
        }
        // (the barrier inside the epilogue, behind its accumulator -> LDS writes, also says: every wave has read its fragments
        // of this stage)
        conv_epilogue<BC, BP, WC, WP, true, KIND>(a, acc, se, mbase + i * BP, c0, s, q, i + 1 == T, blockIdx.x & (NREP - 1), &P);
        // all but the SC youngest memory operations (this tile's row stores) are complete: tile i+1 is in LDS, and the
        // compiler knows the epilogue's loads are done.  vmcnt = SC, expcnt / lgkmcnt untouched (gfx9 encoding)
        __builtin_amdgcn_s_waitcnt(SC | (7 << 4) | (15 << 8));
        if (i + 2 < T && !(TSKIP(a) & 1)) issue(i + 2, stage);      // (tuning: bit 1 = no DMA inside the loop, timing only)
    }
#endif
}

// layer1's 3x3 convolutions: 64 -> 64 channels, stride 1, on 128-wide maps (forward, data gradient, teacher).
// With K = 576 and a 64 x 64 tile the implicit-GEMM kernel above spends its time loading operands: 4096 workgroups
// each pull all 72 KB of weights plus nine shifted copies of their pixels from L2 (600 MB into LDS for 67 MB of
// activations; scripts/dev/dev_conv_skip.py: 40 us with the loads, 27 us without).  Here a persistent workgroup (one per
// CU) keeps ALL weights in LDS, owns a strip of consecutive image rows and rolls a three-row input window through LDS:
// per 128-pixel output row it loads ONE new input row (16.6 KB) while the previous row's epilogue runs.
//   LDS: [9 taps x 64 x 128 B weights][3 row slots x 136 x 128 B][epilogue image]  = 147 KB
// Requires: Cin = Cout = 64, W = 128, pad = dil = 1, H % a.tpw == 0 (rows per workgroup), whole images per group.
// (round 6) The DMA is issued from inline assembly and the epilogue is the LDS-only-barrier form with one instantiation per
// fused variant: through the builtin the compiler put `s_waitcnt vmcnt(0)` in front of every row's accumulator -> LDS stores,
// i.e. each row waited for the input row issued in its middle AND for the previous row's stores.
template <int TW, int KIND>
__global__ void __launch_bounds__(512) conv3x3_c64_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BC = 64, BP = TW, WC = 2, WP = 4, NW = 8;
    constexpr int HP = TW + 2;                             // halo pixels of one input row
    constexpr int RI = (HP + 7) / 8;                       // DMA instructions per input row (8 pixels each): 17
    constexpr int SLOT = RI * 1024;                        // bytes of one row slot
    constexpr int WBYTES = 9 * BC * 128;
    constexpr int CSTR = BC * 2 + 16;
    constexpr int EPI = BP * CSTR + NW * BC * 2 * 4;
    constexpr int SC = BP / (64 * NW / (BC / 8));          // epilogue stores per thread and tile
    __shared__ __attribute__((aligned(256))) unsigned char smem[WBYTES + 3 * SLOT + EPI];
    unsigned char* sw = smem;
    unsigned char* sr = smem + WBYTES;
    unsigned char* se = sr + 3 * SLOT;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wc = wave % WC, wp = wave / WC;
    const int lrow = lane & 31, lk = lane >> 5;
    const int lrow8 = lane >> 3, lslot = lane & 7;
    const int rpw = a.tpw;                                  // image rows per workgroup
    const int strip = blockIdx.x;                           // strips never straddle an image: H % rpw == 0
    const int row0 = strip * rpw;                           // global row index n * H + y
    const int n = row0 / a.H, y0 = row0 % a.H;
    constexpr int OOB = (int)0x80000000;
    const i32x4 rs_w = dma_rsrc(a.w, 64 * 9 * 64 * 2);
    const i32x4 rs_x = dma_rsrc(a.x, (unsigned)((((size_t)a.N * a.H * a.W - 1) * a.ldx + a.Cin) * 2));
    // weights: LDS row j*8 + lrow8 = (tap, co); memory row (co, tap) is 128 contiguous bytes
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int r = (j * NW + wave) * 8 + lrow8;          // 0 .. 575
        const int tap = r / BC, co = r % BC;
        const int vo = (co * 9 + tap) * 128 + (lslot ^ ((r >> 1) & 7)) * 16;
        dma16_to_lds(rs_w, sw + (j * NW + wave) * 1024, vo, 0);
    }
    // one input row -> slot: halo pixel hp = column hp - 1
    int rvo[(RI + NW - 1) / NW];
#pragma unroll
    for (int i = 0; i < (RI + NW - 1) / NW; ++i) {
        const int q = i * NW + wave;                        // instruction index inside the row
        const int hp = q * 8 + lrow8, ix = hp - 1;
        rvo[i] = (q < RI && hp < HP && ix >= 0 && ix < a.W) ? (ix * a.ldx * 2 + (lslot ^ ((hp >> 1) & 7)) * 16) : OOB;
    }
    auto issue_row = [&](int y) {                           // image row y (may be outside: zeros) -> slot (y + 1) % 3
        const bool inside = y >= 0 && y < a.H;
        const int so = inside ? ((n * a.H + y) * a.W) * a.ldx * 2 : 0;
        unsigned char* rb = sr + ((y + 1) % 3) * SLOT;
#pragma unroll
        for (int i = 0; i < (RI + NW - 1) / NW; ++i) {
            const int q = i * NW + wave;
            if (q < RI)
                dma16_to_lds(rs_x, rb + q * 1024, inside ? rvo[i] : OOB, so);
        }
    };
    issue_row(y0 - 1);
    issue_row(y0);
    issue_row(y0 + 1);
    float s[8], q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q8[e] = 0.f; }
    EpiParams P;
    epilogue_params<KIND>(a, row0 * BP, (t % (BC / 8)) * 8, true, P);
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(P.k0[e]), "v"(P.k1[e]), "v"(P.k2[e]), "v"(P.k3[e]));
    WAIT_VMCNT(0);
    for (int ty = 0; ty < rpw; ++ty) {
        const int y = y0 + ty;
        __builtin_amdgcn_s_barrier();
        f32x16 acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
        const int px = wp * 32 + lrow;
        const int ra = wc * 32 + lrow;
        // window rows oldest first: once the three taps that read row y - 1 are done (every wave: one barrier), that
        // row's slot takes row y + 2, whose load then has the other six taps and the epilogue to land
#pragma unroll
        for (int wr = 0; wr < 3; ++wr) {
            const unsigned char* rb = sr + ((y + wr) % 3) * SLOT;         // slot of image row y - 1 + wr
            const int kh = (a.mode == 0) ? wr : 2 - wr;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int hp = px + ((a.mode == 0) ? kw : 2 - kw);
                const unsigned char* wb = sw + (kh * 3 + kw) * (BC * 128);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const bf16x8 fa = *(const bf16x8*)(wb + ra * 128 + (((kk * 2 + lk) ^ ((ra >> 1) & 7)) << 4));
                    const bf16x8 fb = *(const bf16x8*)(rb + hp * 128 + (((kk * 2 + lk) ^ ((hp >> 1) & 7)) << 4));
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[0][0], 0, 0, 0);
                }
            }
            if (wr == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's reads of row y - 1 have returned
                __builtin_amdgcn_s_barrier();
                if (ty + 1 < rpw) issue_row(y + 2);
            }
        }
        const int m0 = (row0 + ty) * BP;
        conv_epilogue<BC, BP, WC, WP, true, KIND>(a, acc, se, m0, 0, s, q8, ty + 1 == rpw, blockIdx.x & (NREP - 1), &P);
        // the newest row of the next window was issued in the middle of this row: every operation of the epilogue is younger,
        // and it ends with exactly SC row stores per thread (vmcnt retires in order): all but those are complete
        __builtin_amdgcn_s_waitcnt(SC | (7 << 4) | (15 << 8));
    }
#endif
}

// tile choice (measured on MI355X, scripts/dev/dev_conv_bench.py): the L2 -> LDS path sustains <= ~80 GB/s per CU, so the
// long-K head convolutions want the largest tile that still gives every CU a workgroup (128 x 256, 85 FLOP per
// byte, 8 waves); large-M layers run 128 x 128 tiles with 8 waves and a 2-stage ring (64 KiB: two workgroups =
// 16 waves per CU); everything else 128 x 64 tiles with 4 waves, two workgroups per CU.  `stages` 82 / 83 mean
// 8-wave workgroups with 2 / 3 stages.  rows_per_group != 0: a tile may not straddle two statistics groups.
// the cases conv3x3_halo_kernel serves: 3x3 / stride 1 / pad = dilation on 32-wide maps, tiles inside one statistics group
// -> image rows per tile (0 = not served):
//   8 (128 x 256 tiles): K >= 4096, dilation 1 or 2 (the heads, layer 4) -- where the 128 x 256 implicit-GEMM tile would be picked;
//   4 (128 x 128 tiles): K >= 2048, dilation 1 (layer 3's 256 -> 256: 27.4 -> 25.0 us against the pipelined 128 x 128 tile).
// From 100 tiles on: the EMA teacher's launches (M = 8192: 128 tiles) then occupy HALF the CUs with one fat workgroup
// each instead of all of them with 128 x 64 tiles -- no faster alone (20.4 vs 21.0 us), but they run beside the student's
// forward on another stream, which gets the other half: -0.33 ms per step in an interleaved A/B (240 -> 100 for both forms).
static int conv_use_halo(long long M, int Cout, int Cin, int kh, int kw, int stride, int pad, int dil, int H, int W, int Ho,
                         int Wo, int rows_per_group) {
    if (const char* e = TUNE_ENV("RGDA_HALO")) { if (!atoi(e)) return 0; }              // tuning experiments only
    if (kh != 3 || kw != 3 || stride != 1 || pad != dil || (dil != 1 && dil != 2)) return 0;
    if ((W & 31) || Wo != W || Ho != H || (Cin & 63)) return 0;        // (W > 32: conv3x3_halo_wide_kernel, 32-column bands)
    int min_tiles = 100;
    if (const char* e = TUNE_ENV("RGDA_HALO_MIN")) min_tiles = atoi(e);                 // tuning experiments only
    int min8 = 100;
    if (const char* e = TUNE_ENV("RGDA_HALO_MIN8")) min8 = atoi(e);                     // tuning experiments only
    if ((long long)9 * Cin >= 4096 && !(H & 7) && !(rows_per_group % 256) && (M / 256) * cdiv(Cout, 128) >= min8) return 8;
    if ((long long)9 * Cin >= 2048 && dil == 1 && !(H & 3) && !(rows_per_group % 128) && (M / 128) * cdiv(Cout, 128) >= min_tiles) return 4;
    return 0;
}

static int pick_tile(long long M, int Cout, long long ktot, int rows_per_group, int& bc, int& bp, int& stages) {
    bc = (Cout <= 64) ? 64 : 128;
    bp = 64;
    int waves = 4;
    int t82 = 512, t83 = 256;
    if (const char* e = TUNE_ENV("RGDA_T82")) t82 = atoi(e);     // tuning experiments only
    if (const char* e = TUNE_ENV("RGDA_T83")) t83 = atoi(e);     // tuning experiments only
    if (bc == 128 && ktot >= 4096 && (long long)cdiv(M, 256) * cdiv(Cout, bc) >= 240) { bp = 256; waves = 8; }
    else if (bc == 128 && (long long)cdiv(M, 128) * cdiv(Cout, bc) >= t82) { bp = 128; waves = 8; }
    else if (bc == 128 && (long long)cdiv(M, 128) * cdiv(Cout, bc) >= t83) { bp = 128; waves = 9; }
    if (rows_per_group) {
        while (bp > 64 && (rows_per_group % bp)) bp >>= 1;
        if (rows_per_group % bp) return 1;
    }
    stages = (bc == 64) ? 2 : 3;
    if (waves >= 8 && bp >= 128) stages = (bp == 256 || waves == 9) ? 83 : 82;
    return 0;
}

extern "C" int rgda_conv2d_tile(int64_t M, int Cout, int kh, int kw, int Cin, int rows_per_group) {
    int bc, bp, stages;
    if (pick_tile(M, Cout, (long long)kh * kw * Cin, rows_per_group, bc, bp, stages)) return RGDA_ERR_UNSUPPORTED;
    return bc | (bp << 10) | (stages << 20);
}

static int ilog2_exact(int v);
struct BnBwdFuse { const void* y; int ldy; const unsigned char* mask; const void* x; int ldx; const float* mi; const float* nscale; int rpi; int relu; const float* gamma; const float* beta; };
struct BnEvalFuse { const float* rm; const float* rv; const float* gamma; const float* beta; float eps; int relu; };

// which kernel serves a convolution whose operand is a BatchNorm (+ ReLU) on the operand path: 1 = conv3x3_halo_kernel
// <1, 4>, 2 = <1, 8>, 3 = conv_igemm_kernel<128, 128, 2, 2, 4, false, true>; 0 = none (the caller materialises the
// activation with rgda_bn_train_apply and runs the plain convolution)
static int conv_bnin_kind(long long M, int Cout, int Cin, int kh, int kw, int stride, int pad, int dil, int H, int W, int Ho,
                          int Wo, int groups) {
    if (groups < 1 || (M % groups) || Cin > RGDA_BNIN_MAX_C || (Cin & 63)) return 0;
    const int rpg = (int)(M / groups);
    if (const int tr = conv_use_halo(M, Cout, Cin, kh, kw, stride, pad, dil, H, W, Ho, Wo, rpg))
        return (tr == 4) ? 1 : (dil == 1 ? 2 : 0);          // (dilation 2: its 160 KB of LDS leave no room for the table)
    int bc, bp, stages;
    if (pick_tile(M, Cout, (long long)kh * kw * Cin, rpg, bc, bp, stages)) return 0;
    return (bc == 128 && bp == 128 && stages == 82) ? 3 : 0;
}

// 0 = not served, 1 = served, 2 = served and faster than the apply pass it replaces.  Where the transform pays
// (scripts/dev/xf_bench.py, isolated, against rgda_bn_train_apply + rgda_conv2d back to back): 3x3 256 -> 256 on 32 x 32 maps
// 36.2 -> 29.3 us, 3x3 128 -> 128 on 64 x 64 39.9 -> 36.9, 1x1 256 -> 1024 31.2 -> 29.5, 1x1 128 -> 512 42.0 -> 41.0; where it
// does not: 1x1 64 -> 256 on 128 x 128 maps 64.8 -> 70.5 and layer 4's 512-channel operands (3x3 81.5 -> 82.5, 1x1 512 -> 2048
// 68.8 -> 82.5: every one of 2048 workgroups rebuilds 512 channels' scale / shift from the accumulators).
extern "C" int rgda_conv2d_bnin_supported(int64_t M, int Cout, int Cin, int kh, int kw, int stride, int pad, int dil, int H,
                                          int W, int Ho, int Wo, int groups) {
    if (!conv_bnin_kind(M, Cout, Cin, kh, kw, stride, pad, dil, H, W, Ho, Wo, groups)) return 0;
    return (Cin > 256 || (kh == 1 && Cin < 128)) ? 1 : 2;
}

static int conv2d_launch(const void* x, int ldx, const void* wgt, void* y, int ldy, const void* res, int ldres,
                         const unsigned char* res_mask, rgda_stat_t* stats, int stat_groups, int N, int H, int W, int Cin, int Ho, int Wo, int Cout,
                         int kh, int kw, int stride, int pad, int dil, int mode, const BnBwdFuse* bnb,
                         rgda_stream_t stream, const BnEvalFuse* bne = nullptr, const rgda_bn_operand* bnin = nullptr,
                         const char** sel = nullptr, ConvArgs* out_args = nullptr) {
    // sel != nullptr: dry run -- *sel = the kernel instantiation that would serve the call (the name rocprofv3 reports),
    // nothing is launched (rgda_conv2d_kernel; bench.py labels its per-launch timings with it); out_args: the argument
    // block that launch would carry (rgda_conv2d_grouped packs several into one launch)
#define RGDA_LAUNCH(NAME, ...)                                                     \
    do {                                                                           \
        if (sel) { *sel = NAME; if (out_args) *out_args = a; return RGDA_OK; }     \
        __VA_ARGS__;                                                               \
    } while (0)
#define RGDA_IGEMM(BC, BP, ST, WC, WP, PIPE, XF)                                                                  \
    RGDA_LAUNCH("conv_igemm_kernel<" #BC ", " #BP ", " #ST ", " #WC ", " #WP ", " #PIPE ", " #XF ">",              \
                conv_igemm_kernel<BC, BP, ST, WC, WP, PIPE, XF><<<grid, 64 * WC * WP, 0, st>>>(a))
    if (!x || !wgt || !y) return RGDA_ERR_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 ||
        stride <= 0 || dil <= 0 || pad < 0 || (mode != 0 && mode != 1))
        return RGDA_ERR_ARG;
    if ((Cin & 63) || (Cout & 7) || (ldx & 7) || (ldy & 7) || (res && (ldres & 7)) || ldx < Cin || ldy < Cout)
        return RGDA_ERR_ARG;
    ConvArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)wgt; a.y = (bf16_t*)y; a.res = (const bf16_t*)res; a.stats = stats;
    if (res_mask && (!res || bne)) return RGDA_ERR_ARG;
    a.res_mask = res_mask;
    a.ldx = ldx; a.ldy = ldy; a.ldres = ldres;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.KH = kh; a.KW = kw;
    a.stride = stride; a.pad = pad; a.dil = dil; a.mode = mode;
    long long M = (long long)N * Ho * Wo;
    if (M > 0x7fffffffLL) return RGDA_ERR_ARG;
    a.M = (int)M;
    a.howo_shift = ilog2_exact(Ho * Wo);
    a.wo_shift = ilog2_exact(Wo);
    hipStream_t st = to_stream(stream);
    // tile choice: fill 256 CUs (2 workgroups each); prefer the big tile when it still gives >= 512 groups
    if (stat_groups < 1) stat_groups = 1;
    if (M % stat_groups) return RGDA_ERR_ARG;
    a.rows_per_group = (int)(M / stat_groups);
    a.bn_y = a.bn_x = nullptr; a.bn_mask = nullptr; a.bn_mi = a.bn_nscale = nullptr; a.bn_ldy = a.bn_ldx = a.bn_rpi = a.bn_relu = 0;
    a.bn_gamma = a.bn_beta = nullptr;
    if (bnb) {
        if (!stats || !bnb->x || !bnb->mi || (bnb->relu == 1 && !bnb->y && !bnb->mask) || (bnb->ldx & 7) ||
            (bnb->relu == 1 && bnb->y && (bnb->ldy & 7)) || (bnb->nscale && bnb->rpi <= 0) || bnb->relu < 0 || bnb->relu > 2)
            return RGDA_ERR_ARG;
        a.bn_mask = bnb->relu == 1 ? bnb->mask : nullptr;
        a.bn_y = (const bf16_t*)bnb->y; a.bn_x = (const bf16_t*)bnb->x; a.bn_mi = bnb->mi; a.bn_nscale = bnb->nscale;
        a.bn_ldy = bnb->ldy; a.bn_ldx = bnb->ldx; a.bn_rpi = bnb->rpi; a.bn_relu = bnb->relu;
        if (bnb->relu == 2 && (!bnb->gamma || !bnb->beta)) return RGDA_ERR_ARG;
        a.bn_gamma = bnb->gamma; a.bn_beta = bnb->beta;
    }
    a.ev_rm = a.ev_rv = a.ev_gamma = a.ev_beta = nullptr; a.ev_eps = 0.f; a.ev_relu = 0;
    if (bne) {
        if (!bne->rm || !bne->rv || !bne->gamma || !bne->beta || stats) return RGDA_ERR_ARG;
        a.ev_rm = bne->rm; a.ev_rv = bne->rv; a.ev_gamma = bne->gamma; a.ev_beta = bne->beta; a.ev_eps = bne->eps;
        a.ev_relu = bne->relu;
    }
    a.dbg = nullptr; a.skip = 0;
    if (const char* e = TUNE_ENV("RGDA_CONV_DBG")) a.dbg = (unsigned long long*)strtoull(e, nullptr, 0);   // tuning only
    if (const char* e = TUNE_ENV("RGDA_CONV_SKIP")) a.skip = atoi(e);                                      // tuning only
    a.xf.stats = nullptr;
    if (bnin) {
        // BatchNorm (+ ReLU) of the producer on this convolution's operand path: forward only, statistics groups = the
        // operand's groups (whole images), a kernel that carries the transform
        if (mode != 0 || bne || bnb || !bnin->stats || !bnin->gamma || !bnin->beta || bnin->groups != stat_groups ||
            ((long long)N * H * W) % bnin->groups || (bnin->running_mean == nullptr) != (bnin->running_var == nullptr))
            return RGDA_ERR_ARG;
        if ((long long)N * H * W / bnin->groups < 2) return RGDA_ERR_ARG;      // "Expected more than 1 value per channel"
        const int kind = conv_bnin_kind(M, Cout, Cin, kh, kw, stride, pad, dil, H, W, Ho, Wo, stat_groups);
        if (!kind) return RGDA_ERR_UNSUPPORTED;
        a.xf.stats = bnin->stats; a.xf.gamma = bnin->gamma; a.xf.beta = bnin->beta; a.xf.mi = bnin->mi;
        a.xf.rm = bnin->running_mean; a.xf.rv = bnin->running_var; a.xf.nbt = (long long*)bnin->num_batches_tracked;
        a.xf.eps = bnin->eps; a.xf.mom = bnin->momentum; a.xf.groups = bnin->groups; a.xf.relu = bnin->relu; a.xf.C = Cin;
        a.xf.rows_per_group = (int)((long long)N * H * W / bnin->groups);
        if (kind == 3) {
            a.tiles_c = cdiv(Cout, 128); a.tiles_p = cdiv(M, 128);
            const int grid = a.tiles_c * a.tiles_p;
            RGDA_IGEMM(128, 128, 2, 2, 4, false, true);
        } else {
            const int tr = kind == 1 ? 4 : 8;
            a.tiles_c = cdiv(Cout, 128); a.tiles_p = (int)(M / (tr * 32));
            const int grid = a.tiles_c * a.tiles_p;
            int ns = 4;
            if (const char* e = TUNE_ENV("RGDA_HALO_NS")) ns = atoi(e);                     // tuning experiments only
            if (W != 32 && kind == 1) RGDA_LAUNCH("conv3x3_halo_wide_kernel<1, 4, 4, true, true>", conv3x3_halo_wide_kernel<1, 4, 4, true, true><<<grid, 512, 0, st>>>(a));
            else if (W != 32) RGDA_LAUNCH("conv3x3_halo_wide_kernel<1, 8, 3, false, true>", conv3x3_halo_wide_kernel<1, 8, 3, false, true><<<grid, 512, 0, st>>>(a));
            else if (kind == 1 && ns == 4) RGDA_LAUNCH("conv3x3_halo_kernel<1, 4, true, 4, true>", conv3x3_halo_kernel<1, 4, true, 4><<<grid, 512, 0, st>>>(a));
            else if (kind == 1) RGDA_LAUNCH("conv3x3_halo_kernel<1, 4, true, 3, false>", conv3x3_halo_kernel<1, 4, true><<<grid, 512, 0, st>>>(a));
            // (the pipelined form of the 8-row variant with the transform spills 80 registers: it keeps the plain loop)
            else RGDA_LAUNCH("conv3x3_halo_kernel<1, 8, true, 3, false>", conv3x3_halo_kernel<1, 8, true><<<grid, 512, 0, st>>>(a));
        }
        RGDA_CHECK_LAUNCH();
        return RGDA_OK;
    }
    // layer1's 64 -> 64 3x3 on 128-wide maps: weights-resident rolling-window kernel, one workgroup per CU
    {
        int c64_on = 1;
        if (const char* e = TUNE_ENV("RGDA_C64")) c64_on = atoi(e);                         // tuning experiments only
        if (c64_on && kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1 && Cin == 64 && Cout == 64 && W == 128 &&
            Wo == W && Ho == H && !(a.rows_per_group % (H * W))) {
            int rpw = (int)((long long)N * H / 256);
            while (rpw > 1 && (H % rpw)) --rpw;
            if (rpw >= 2) {
                a.tpw = rpw;
                a.tiles_c = 1;
                a.tiles_p = N * H;
                const int kind = a.ev_rm ? EPI_EV : a.bn_x ? (a.bn_relu == 2 ? EPI_BNX2 : EPI_BNX)
                               : a.res ? (a.stats ? EPI_RES_STATS : EPI_RES) : (a.stats ? EPI_STATS : EPI_PLAIN);
                const int grid = N * H / rpw;
#define RGDA_C64(KIND) case KIND: RGDA_LAUNCH("conv3x3_c64_kernel<128, " #KIND ">", conv3x3_c64_kernel<128, KIND><<<grid, 512, 0, st>>>(a)); break
                switch (kind) { RGDA_C64(0); RGDA_C64(1); RGDA_C64(2); RGDA_C64(3); RGDA_C64(4); RGDA_C64(5); RGDA_C64(6); }
#undef RGDA_C64
                RGDA_CHECK_LAUNCH();
                return RGDA_OK;
            }
        }
    }
    // long-K 3x3 convolutions on 32-wide maps (heads, layer 4): the halo kernel (tiles of 8 image rows = 256 pixels)
    if (const int tr = conv_use_halo(M, Cout, Cin, kh, kw, stride, pad, dil, H, W, Ho, Wo, a.rows_per_group)) {
        a.tiles_c = cdiv(Cout, 128);
        a.tiles_p = (int)(M / (tr * 32));
        const int grid = a.tiles_c * a.tiles_p;
        int ns = 4;                                                                        // (dilation 2: 3 stages, 160 KB)
        if (const char* e = TUNE_ENV("RGDA_HALO_NS")) ns = atoi(e);                         // tuning experiments only
        if (W != 32) {
            if (tr == 4) RGDA_LAUNCH("conv3x3_halo_wide_kernel<1, 4, 4, true, false>", conv3x3_halo_wide_kernel<1, 4, 4><<<grid, 512, 0, st>>>(a));
            else if (dil == 1) RGDA_LAUNCH("conv3x3_halo_wide_kernel<1, 8, 4, true, false>", conv3x3_halo_wide_kernel<1, 8, 4><<<grid, 512, 0, st>>>(a));
            else RGDA_LAUNCH("conv3x3_halo_wide_kernel<2, 8, 3, true, false>", conv3x3_halo_wide_kernel<2, 8, 3, true><<<grid, 512, 0, st>>>(a));
        }
        else if (tr == 4 && ns == 4) RGDA_LAUNCH("conv3x3_halo_kernel<1, 4, false, 4, true>", conv3x3_halo_kernel<1, 4, false, 4><<<grid, 512, 0, st>>>(a));
        else if (tr == 4) RGDA_LAUNCH("conv3x3_halo_kernel<1, 4, false, 3, false>", conv3x3_halo_kernel<1, 4><<<grid, 512, 0, st>>>(a));
        else if (dil == 1 && ns == 4) RGDA_LAUNCH("conv3x3_halo_kernel<1, 8, false, 4, true>", conv3x3_halo_kernel<1, 8, false, 4><<<grid, 512, 0, st>>>(a));
        else if (dil == 1) RGDA_LAUNCH("conv3x3_halo_kernel<1, 8, false, 3, false>", conv3x3_halo_kernel<1, 8><<<grid, 512, 0, st>>>(a));
        else if (ns == 4) RGDA_LAUNCH("conv3x3_halo_kernel<2, 8, false, 3, true>", conv3x3_halo_kernel<2, 8, false, 3, true><<<grid, 512, 0, st>>>(a));
        else RGDA_LAUNCH("conv3x3_halo_kernel<2, 8, false, 3, false>", conv3x3_halo_kernel<2, 8><<<grid, 512, 0, st>>>(a));
        RGDA_CHECK_LAUNCH();
        return RGDA_OK;
    }
    // short-K 1x1 convolutions on large maps: the persistent streaming kernel, `wgs` workgroups of T pixel tiles each
    {
        int stream_on = 1, wpc = 1;
        if (const char* e = TUNE_ENV("RGDA_STREAM")) stream_on = atoi(e);                   // tuning experiments only
        if (const char* e = TUNE_ENV("RGDA_STREAM_WPC")) wpc = atoi(e);                     // tuning experiments only
        if (stream_on && kh == 1 && kw == 1 && stride == 1 && pad == 0 && Ho == H && Wo == W &&
            (Cin == 64 || Cin == 128) && !(Cout & 127) && M >= 4 * 128 && !(M & 127) &&
            (long long)M * ldx * 2 < (1ll << 31)) {            // (32-bit byte offsets into the pixel operand)
            const int bp = 128;
            const int tiles_c = Cout / 128;
            const long long tiles_p = M / bp;
            int chunks = (256 * wpc) / tiles_c;
            while (chunks > 1 && ((tiles_p % chunks) || (a.rows_per_group % (int)(tiles_p / chunks * bp)))) chunks >>= 1;
            const long long T = chunks >= 1 ? tiles_p / chunks : 0;
            if (!(M % bp) && chunks >= 1 && T >= 4 && T <= 4096 && !(tiles_p % chunks) && !(a.rows_per_group % (int)(T * bp))) {
                a.tiles_c = tiles_c; a.tiles_p = (int)tiles_p; a.tpw = (int)T;
                const int grid = tiles_c * chunks;
                // one instantiation per fused epilogue (the choice conv_epilogue makes from the arguments)
                const int kind = a.ev_rm ? EPI_EV : a.bn_x ? (a.bn_relu == 2 ? EPI_BNX2 : EPI_BNX)
                               : a.res ? (a.stats ? EPI_RES_STATS : EPI_RES) : (a.stats ? EPI_STATS : EPI_PLAIN);
#define RGDA_STREAM(KC, BP, WC, WP, KIND)                                                                      \
    case KIND: RGDA_LAUNCH("conv1x1_stream_kernel<" #KC ", " #BP ", " #WC ", " #WP ", " #KIND ">",              \
                           conv1x1_stream_kernel<KC, BP, WC, WP, KIND><<<grid, 512, 0, st>>>(a)); break
#define RGDA_STREAM_ALL(KC, BP, WC, WP)                                                                        \
    switch (kind) {                                                                                            \
        RGDA_STREAM(KC, BP, WC, WP, 0); RGDA_STREAM(KC, BP, WC, WP, 1); RGDA_STREAM(KC, BP, WC, WP, 2);         \
        RGDA_STREAM(KC, BP, WC, WP, 3); RGDA_STREAM(KC, BP, WC, WP, 4); RGDA_STREAM(KC, BP, WC, WP, 5);         \
        RGDA_STREAM(KC, BP, WC, WP, 6);                                                                         \
    }
                if (Cin == 64) { RGDA_STREAM_ALL(1, 128, 2, 4) }
                else { RGDA_STREAM_ALL(2, 128, 2, 4) }
#undef RGDA_STREAM_ALL
#undef RGDA_STREAM
                RGDA_CHECK_LAUNCH();
                return RGDA_OK;
            }
        }
    }
    int bc, bp, stages;
    if (pick_tile(M, Cout, (long long)kh * kw * Cin, (stats && stat_groups > 1) ? a.rows_per_group : 0, bc, bp, stages))
        return RGDA_ERR_UNSUPPORTED;
    if (const char* e = TUNE_ENV("RGDA_TILE")) sscanf(e, "%d,%d,%d", &bc, &bp, &stages);   // tuning experiments only
    a.tiles_c = cdiv(Cout, bc);
    a.tiles_p = cdiv(M, bp);
    int grid = a.tiles_c * a.tiles_p;
    static const bool pipe = TUNE_ENV("RGDA_NO_PIPE") == nullptr;                           // (off switch: tuning experiments only)
    if (pipe && bc == 128 && bp == 128 && stages == 83) RGDA_IGEMM(128, 128, 3, 2, 4, true, false);
    else if (pipe && bc == 128 && bp == 256 && stages == 83) RGDA_IGEMM(128, 256, 3, 2, 4, true, false);
    else if (pipe && bc == 128 && bp == 64 && stages == 3) RGDA_IGEMM(128, 64, 3, 2, 2, true, false);
    else if (bc == 128 && bp == 128 && stages == 83) RGDA_IGEMM(128, 128, 3, 2, 4, false, false);
    else if (bc == 128 && bp == 128 && stages == 82) RGDA_IGEMM(128, 128, 2, 2, 4, false, false);
    else if (bc == 128 && bp == 256 && stages == 83) RGDA_IGEMM(128, 256, 3, 2, 4, false, false);
    else if (bc == 128 && bp == 256) RGDA_IGEMM(128, 256, 3, 2, 2, false, false);
    else if (bc == 256 && bp == 128) RGDA_IGEMM(256, 128, 3, 2, 2, false, false);
    else if (bc == 128 && bp == 128 && stages == 2) RGDA_IGEMM(128, 128, 2, 2, 2, false, false);
    else if (bc == 128 && bp == 128 && stages == 4) RGDA_IGEMM(128, 128, 4, 2, 2, false, false);
    else if (bc == 128 && bp == 128) RGDA_IGEMM(128, 128, 3, 2, 2, false, false);
    else if (bc == 128 && bp == 64 && stages == 2) RGDA_IGEMM(128, 64, 2, 2, 2, false, false);
    else if (bc == 128 && bp == 64 && stages == 4) RGDA_IGEMM(128, 64, 4, 2, 2, false, false);
    else if (bc == 128 && bp == 64) RGDA_IGEMM(128, 64, 3, 2, 2, false, false);
    else if (bc == 64 && bp == 128) RGDA_IGEMM(64, 128, 3, 2, 2, false, false);
    else if (stages == 4) RGDA_IGEMM(64, 64, 4, 2, 2, false, false);
    else if (stages == 2) RGDA_IGEMM(64, 64, 2, 2, 2, false, false);
    else RGDA_IGEMM(64, 64, 3, 2, 2, false, false);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

#undef RGDA_IGEMM
#undef RGDA_LAUNCH

// The kernel instantiation that serves a convolution call, as rocprofv3 names it (no launch): `variant` 0 = rgda_conv2d
// (fused statistics with `stat_groups` groups when has_stats), 1 = rgda_conv2d_bneval, 2 = rgda_conv2d_bnbwd,
// 3 = rgda_conv2d_bnin.  NULL where the call would fail.
extern "C" const char* rgda_conv2d_kernel(int variant, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw,
                                          int stride, int pad, int dil, int mode, int has_stats, int stat_groups) {
    static rgda_stat_t dummy_stats;
    static const float dummy_f = 0.f;
    const char* name = nullptr;
    void* const p = (void*)&dummy_stats;           // never dereferenced: a dry run stops before any launch
    rgda_stat_t* st = (has_stats || variant >= 2) ? &dummy_stats : nullptr;
    BnEvalFuse e = {&dummy_f, &dummy_f, &dummy_f, &dummy_f, 1e-5f, 1};
    BnBwdFuse b = {nullptr, 0, (const unsigned char*)p, p, (Cout + 7) & ~7, &dummy_f, nullptr, 0, 1, nullptr, nullptr};
    rgda_bn_operand o = {&dummy_stats, &dummy_f, &dummy_f, nullptr, nullptr, nullptr, nullptr, 1e-5f, 0.1f, stat_groups < 1 ? 1 : stat_groups, 1};
    const int ld_in = (Cin + 7) & ~7, ld_out = (Cout + 7) & ~7;
    const int rc = conv2d_launch(p, ld_in, p, p, ld_out, nullptr, 0, nullptr, variant == 1 ? nullptr : st, stat_groups, N, H, W, Cin, Ho,
                                 Wo, Cout, kh, kw, stride, pad, dil, variant == 3 ? 0 : mode, variant == 2 ? &b : nullptr, nullptr,
                                 variant == 1 ? &e : nullptr, variant == 3 ? &o : nullptr, &name);
    return rc == RGDA_OK ? name : nullptr;
}

static const char* const GROUPED_KERNEL = "conv_igemm_kernel<128, 64, 3, 2, 2, true, false>";

extern "C" int rgda_conv2d_grouped(const rgda_conv2d_desc* descs, int n, rgda_stream_t stream) {
    if (!descs || n < 0) return RGDA_ERR_ARG;
    hipStream_t st = to_stream(stream);
    ConvGroup g;
    g.n = 0; g.start[0] = 0;
    auto flush = [&]() -> int {
        if (g.n == 1) {          // alone: the ordinary launch (same kernel, same argument block)
            conv_igemm_kernel<128, 64, 3, 2, 2, true, false><<<g.start[1], 256, 0, st>>>(g.a[0]);
        } else if (g.n > 1) {
            for (int i = g.n + 1; i <= CONV_GROUP_MAX; ++i) g.start[i] = g.start[g.n];
            conv_igemm_grouped_kernel<128, 64, 3, 2, 2, true><<<g.start[g.n], 256, 0, st>>>(g);
        }
        g.n = 0;
        RGDA_CHECK_LAUNCH();
        return RGDA_OK;
    };
    // validate everything first: nothing is launched when one descriptor is wrong
    for (int i = 0; i < n; ++i) {
        const rgda_conv2d_desc& d = descs[i];
        const char* name = nullptr;
        const int rc = conv2d_launch(d.x, d.ldx, d.wgt, d.y, d.ldy, d.res, d.ldres, d.res_relu_mask, (rgda_stat_t*)d.stats,
                                     d.stat_groups, d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.kh, d.kw, d.stride, d.pad, d.dil,
                                     d.mode, nullptr, stream, nullptr, nullptr, &name);
        if (rc != RGDA_OK) return rc;
    }
    for (int i = 0; i < n; ++i) {
        const rgda_conv2d_desc& d = descs[i];
        const char* name = nullptr;
        ConvArgs a;
        conv2d_launch(d.x, d.ldx, d.wgt, d.y, d.ldy, d.res, d.ldres, d.res_relu_mask, (rgda_stat_t*)d.stats, d.stat_groups, d.N, d.H,
                      d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.kh, d.kw, d.stride, d.pad, d.dil, d.mode, nullptr, stream, nullptr, nullptr,
                      &name, &a);
        if (name && !strcmp(name, GROUPED_KERNEL)) {
            g.a[g.n] = a;
            g.start[g.n + 1] = g.start[g.n] + a.tiles_c * a.tiles_p;
            if (++g.n == CONV_GROUP_MAX) { if (int rc = flush()) return rc; }
        } else {                 // another kernel serves it: its own launch
            const int rc = conv2d_launch(d.x, d.ldx, d.wgt, d.y, d.ldy, d.res, d.ldres, d.res_relu_mask, (rgda_stat_t*)d.stats,
                                         d.stat_groups, d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.kh, d.kw, d.stride, d.pad, d.dil,
                                         d.mode, nullptr, stream);
            if (rc != RGDA_OK) return rc;
        }
    }
    return flush();
}

// how many kernel launches rgda_conv2d_grouped makes of a list, and (name != NULL) whether they all share the grouped kernel
extern "C" int rgda_conv2d_grouped_launches(const rgda_conv2d_desc* descs, int n) {
    if (!descs || n < 0) return RGDA_ERR_ARG;
    int grouped = 0, single = 0;
    for (int i = 0; i < n; ++i) {
        const rgda_conv2d_desc& d = descs[i];
        const char* name = nullptr;
        const int rc = conv2d_launch(d.x, d.ldx, d.wgt, d.y, d.ldy, d.res, d.ldres, d.res_relu_mask, (rgda_stat_t*)d.stats,
                                     d.stat_groups, d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.kh, d.kw, d.stride, d.pad, d.dil,
                                     d.mode, nullptr, nullptr, nullptr, nullptr, &name);
        if (rc != RGDA_OK) return rc;
        if (!strcmp(name, GROUPED_KERNEL)) ++grouped; else ++single;
    }
    return single + (grouped + CONV_GROUP_MAX - 1) / CONV_GROUP_MAX;
}

extern "C" int rgda_conv2d(const void* x, int ldx, const void* wgt, void* y, int ldy, const void* res, int ldres,
                           const uint8_t* res_relu_mask, rgda_stat_t* stats, int stat_groups, int N, int H, int W, int Cin,
                           int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int mode,
                           rgda_stream_t stream) {
    return conv2d_launch(x, ldx, wgt, y, ldy, res, ldres, res_relu_mask, stats, stat_groups, N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride,
                         pad, dil, mode, nullptr, stream);
}

extern "C" int rgda_conv2d_bneval(const void* x, int ldx, const void* wgt, void* y, int ldy, const void* res, int ldres,
                                  const float* running_mean, const float* running_var, const float* gamma,
                                  const float* beta, float eps, int relu, int N, int H, int W, int Cin, int Ho, int Wo,
                                  int Cout, int kh, int kw, int stride, int pad, int dil, rgda_stream_t stream) {
    BnEvalFuse e = {running_mean, running_var, gamma, beta, eps, relu};
    return conv2d_launch(x, ldx, wgt, y, ldy, res, ldres, nullptr, nullptr, 1, N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride,
                         pad, dil, 0, nullptr, stream, &e);
}

extern "C" int rgda_conv2d_bnin(const rgda_bn_operand* bn_in, const void* x, int ldx, const void* wgt, void* y, int ldy,
                                const void* res, int ldres, rgda_stat_t* stats, int stat_groups, int N, int H, int W, int Cin,
                                int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, rgda_stream_t stream) {
    if (!bn_in) return RGDA_ERR_ARG;
    return conv2d_launch(x, ldx, wgt, y, ldy, res, ldres, nullptr, stats, stat_groups, N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride,
                         pad, dil, 0, nullptr, stream, nullptr, bn_in);
}

extern "C" int rgda_conv2d_bnbwd(const void* x, int ldx, const void* wgt, void* y, int ldy, const void* res, int ldres,
                                 const uint8_t* res_relu_mask, rgda_stat_t* sums, int groups, const void* bn_y, int bn_ldy, const uint8_t* bn_relu_mask,
                                 const void* bn_x, int bn_ldx,
                                 const float* bn_mi, const float* bn_nscale, int rows_per_image, int relu,
                                 const float* bn_gamma, const float* bn_beta, int N,
                                 int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                                 int dil, int mode, rgda_stream_t stream) {
    BnBwdFuse b = {bn_y, bn_ldy, bn_relu_mask, bn_x, bn_ldx, bn_mi, bn_nscale, rows_per_image, relu, bn_gamma, bn_beta};
    return conv2d_launch(x, ldx, wgt, y, ldy, res, ldres, res_relu_mask, sums, groups, N, H, W, Cin, Ho, Wo, Cout, kh, kw,
                         stride, pad, dil, mode, &b, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// The stem convolution (7x7 / stride 2 / pad 3, 3 -> 64 channels; regda/_resnets.py:150-151) straight from the NCHW
// fp32 image.  As an implicit GEMM over a patch matrix it has K = 147 and the 1x1 route -- rgda_stem_im2col + rgda_conv2d
// -- writes and re-reads a 403 MB patch matrix for 16 images of 512 x 512.  Rounds 3 - 5 built the 64 x 192 patch tile of a
// 64-pixel output-row segment in LDS instead (12 288 two-byte LDS gathers per segment: that, not the 12 MFMAs per wave, was
// the kernel: 67 us per 8 images, 1.4 TB/s).  Round 6: NO patch tile.  The image rows are staged in LDS as bf16 RGBX pixels
// (8 bytes each); output pixel x of a row reads input columns 2x .. 2x + 6, so with K ordered (kh, kw, c4) -- 7 filter rows
// of 8 columns x 4 channels = 32, the eighth column and the fourth channel carrying zero weights -- the MFMA B fragment of
// pixel x, filter row kh, K slice (kk, lk) is the 16 contiguous bytes at  row(kh) + 16 x + 32 kk + 16 lk : fragments are read
// straight from the staged rows, 16-byte aligned, 32 lanes x 16 bytes contiguous (conflict-free), 14 MFMAs per wave (K = 224
// instead of 192).  A workgroup owns 64 consecutive pixels of `rpw` consecutive output rows and rolls a seven-row window
// through a seven-slot ring: an output row costs TWO new input rows (fetched into registers under the previous row's MFMAs
// and epilogue) instead of seven.  The wave's weight fragments (14 x 16 B per lane) are gathered once from the [64][192]
// matrix.  19 KB of LDS, ~110 registers: four workgroups per CU.  Same epilogue function (bf16 rows, BatchNorm statistics or
// inference BatchNorm + ReLU); the result differs from the patch-tile kernel's only in fp32 summation order.
// Requires Wo % 64 == 0 (the 1x1 route serves everything else).
template <int KIND>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) stem_conv_kernel(ConvArgs a, const float* __restrict__ img, int rpw) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BC = 64, BP = 64, WC = 2, WP = 2, NW = 4;
    constexpr int SWP = 136;                            // staged pixels per input row: wi = 2 wo0 - 3 .. 2 wo0 + 132 (134 used)
    constexpr int SROW = SWP * 8;                       // bytes of a staged row
    constexpr int CSTR = BC * 2 + 16;
    constexpr int EPI = BP * CSTR + NW * BC * 2 * 4;
    __shared__ __attribute__((aligned(256))) unsigned char smem[EPI + 7 * SROW];
    unsigned char* se = smem;
    unsigned char* srow = smem + EPI;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wc = wave % WC, wp = wave / WC;
    const int lrow = lane & 31, lk = lane >> 5;
    const int H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo;
    const int wtiles = Wo / BP, rblocks = Ho / rpw;
    const int wo0 = (blockIdx.x % wtiles) * BP;
    const int ho0 = ((blockIdx.x / wtiles) % rblocks) * rpw, n = blockIdx.x / (wtiles * rblocks);
    const int hbase = ho0 * 2 - 3;                      // image row of window index j = 0

    // the wave's 32 weight rows as A fragments in (kh, kw, c4) order: element e of slice (kh, kk, lk) is filter column
    // kk * 4 + lk * 2 + e / 4, channel e % 4; memory order of a.w is [co][(kh * 7 + kw) * 3 + c]
    bf16x8 fa[7][2];
    {
        const bf16_t* wrow = a.w + (size_t)(wc * 32 + lrow) * 192;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int kw = kk * 4 + lk * 2 + (e >> 2), c = e & 3;
                    v[e] = (kw < 7 && c < 3) ? wrow[(kh * 7 + kw) * 3 + c] : (bf16_t)0;
                }
                fa[kh][kk] = __builtin_bit_cast(bf16x8, v);
            }
    }
    // staging: a thread converts pixels (window row j, staged column x) = three fp32 planes -> one 8-byte RGBX pixel
    const size_t plane = (size_t)H * W;
    const float* const img_n = img + (size_t)n * 3 * plane;
    auto load_px = [&](int j, int x, float (&v)[3]) {
        const int hi = hbase + j, wi = wo0 * 2 - 3 + x;
        const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W && x < 134;
        const size_t o = (size_t)(ok ? hi : 0) * W + (ok ? wi : 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = ok ? img_n[c * plane + o] : 0.f;
    };
    auto store_px = [&](int j, int x, const float (&v)[3]) {
        uint2 p;
        p.x = pack2bf(v[0], v[1]);
        p.y = pack2bf(v[2], 0.f);
        *(uint2*)(srow + (j % 7) * SROW + x * 8) = p;
    };
    // the first window: rows j = 0 .. 6 (952 pixels over 256 threads)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = t + 256 * i;
        if (idx < 7 * SWP) {
            float v[3];
            load_px(idx / SWP, idx % SWP, v);
            store_px(idx / SWP, idx % SWP, v);
        }
    }
    // the two rows an output-row step adds: thread t takes pixel t of the pair (272 pixels: threads 0 .. 15 take a second).
    // They are fetched TWO output rows ahead (registers nb*), handed over (na*) and stored one row ahead: a fetch has a whole
    // row -- MFMAs, epilogue, row stores -- to land, and no barrier of the loop waits for memory (LDS-only waits: the
    // epilogue's `__syncthreads()` form would drain the fetch and the row stores every row, 3.7 us per row and workgroup)
    const int nj0 = t / SWP, nx0 = t % SWP;             // t < 256: row 0 / 1 of the pair
    const int nj1 = (t + 256) / SWP, nx1 = (t + 256) % SWP;
    const bool second = t + 256 < 2 * SWP;
    float na0[3] = {0.f, 0.f, 0.f}, na1[3] = {0.f, 0.f, 0.f}, nb0[3] = {0.f, 0.f, 0.f}, nb1[3] = {0.f, 0.f, 0.f};
    if (1 < rpw) {
        load_px(7 + nj0, nx0, na0);
        if (second) load_px(7 + nj1, nx1, na1);
    }
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    EpiParams P;
    epilogue_params<KIND>(a, ((n * Ho + ho0) * Wo + wo0), (t % (BC / 8)) * 8, true, P);
    for (int r = 0; r < rpw; ++r) {
        const int ho = ho0 + r;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // the window of this row is complete; the previous C image is consumed
        if (r + 2 < rpw) {                              // rows j = 2 r + 9, 2 r + 10: needed two output rows from now
            load_px(2 * r + 9 + nj0, nx0, nb0);
            if (second) load_px(2 * r + 9 + nj1, nx1, nb1);
        }
        // two accumulation chains (K slices kk = 0 / 1), added at the end: a single chain of 14 dependent MFMAs idles the pipe
        f32x16 acc[1][1], acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[0][0][e] = 0.f; acc1[e] = 0.f; }
        const int px = wp * 32 + lrow;
        const unsigned char* fbase = srow + px * 16 + lk * 16;
        const int s0 = (2 * r) % 7;
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            const int slot = (s0 + kh >= 7) ? s0 + kh - 7 : s0 + kh;
            const bf16x8 fb0 = *(const bf16x8*)(fbase + slot * SROW);
            const bf16x8 fb1 = *(const bf16x8*)(fbase + slot * SROW + 32);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kh][0], fb0, acc[0][0], 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kh][1], fb1, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][0][e] += acc1[e];
        const int m0 = (n * Ho + ho) * Wo + wo0;
        // (the LDS-only barrier inside the epilogue, behind the accumulator -> LDS writes, also says: every wave has read this
        // row's window)
        conv_epilogue<BC, BP, WC, WP, true, KIND>(a, acc, se, m0, 0, s, q, r + 1 == rpw, blockIdx.x & (NREP - 1), &P);
        if (r + 1 < rpw) {                              // rows 2 r, 2 r + 1 are dead: their slots take rows 2 r + 7, 2 r + 8
            store_px(2 * r + 7 + nj0, nx0, na0);
            if (second) store_px(2 * r + 7 + nj1, nx1, na1);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { na0[c] = nb0[c]; na1[c] = nb1[c]; }
    }
#endif
}

static int stem_conv_launch(const float* img, const void* wgt, void* y, int ldy, rgda_stat_t* stats, int stat_groups,
                            const BnEvalFuse* bne, int N, int H, int W, int Ho, int Wo, rgda_stream_t stream) {
    if (!img || !wgt || !y || N <= 0 || H <= 0 || W <= 0 || (ldy & 7) || ldy < 64) return RGDA_ERR_ARG;
    if (Ho != (H + 6 - 7) / 2 + 1 || Wo != (W + 6 - 7) / 2 + 1) return RGDA_ERR_ARG;
    if (Wo % 64) return RGDA_ERR_UNSUPPORTED;           // the im2col + 1x1 route serves these
    // the kernel packs the filter row into bits 28-30 of a per-thread image offset: images of 2^28 elements and more
    // (~9.4 k x 9.4 k) take the im2col route as well
    if ((long long)3 * H * W >= (1ll << 28)) return RGDA_ERR_UNSUPPORTED;
    if (stat_groups < 1) stat_groups = 1;
    if ((N % stat_groups) || (bne && stats)) return RGDA_ERR_ARG;
    const long long M = (long long)N * Ho * Wo;
    if (M > 0x7fffffffLL) return RGDA_ERR_ARG;
    ConvArgs a;
    a.x = nullptr; a.w = (const bf16_t*)wgt; a.y = (bf16_t*)y; a.res = nullptr; a.stats = stats; a.res_mask = nullptr;
    a.ldx = 0; a.ldy = ldy; a.ldres = 0;
    a.N = N; a.H = H; a.W = W; a.Cin = 3; a.Ho = Ho; a.Wo = Wo; a.Cout = 64; a.KH = 7; a.KW = 7;
    a.stride = 2; a.pad = 3; a.dil = 1; a.mode = 0;
    a.M = (int)M; a.tiles_c = 1; a.tiles_p = 0;
    a.rows_per_group = (int)(M / stat_groups);
    a.howo_shift = a.wo_shift = -1; a.dbg = nullptr; a.tpw = 0; a.skip = 0;
    a.bn_y = a.bn_x = nullptr; a.bn_mask = nullptr; a.bn_mi = a.bn_nscale = nullptr; a.bn_ldy = a.bn_ldx = a.bn_rpi = a.bn_relu = 0;
    a.bn_gamma = a.bn_beta = nullptr; a.xf.stats = nullptr;
    a.ev_rm = a.ev_rv = a.ev_gamma = a.ev_beta = nullptr; a.ev_eps = 0.f; a.ev_relu = 0;
    if (bne) {
        if (!bne->rm || !bne->rv || !bne->gamma || !bne->beta) return RGDA_ERR_ARG;
        a.ev_rm = bne->rm; a.ev_rv = bne->rv; a.ev_gamma = bne->gamma; a.ev_beta = bne->beta; a.ev_eps = bne->eps; a.ev_relu = bne->relu;
    }
    // rows per workgroup: as many as still leave two workgroups per CU (their prologue -- the weight gather and the first
    // seven-row window -- is ~3 us of dependent loads), at least 8
    int rpw = 8;
    if (const char* e = TUNE_ENV("RGDA_STEM_RPW")) rpw = atoi(e);                          // tuning experiments only
    else while (rpw < 64 && (long long)N * (Ho / (2 * rpw)) * (Wo / 64) >= 512 && !(Ho % (2 * rpw))) rpw *= 2;
    while (rpw > 1 && (Ho % rpw)) rpw >>= 1;
    const long long blocks = (long long)N * (Ho / rpw) * (Wo / 64);
    if (blocks > 0x7fffffffLL) return RGDA_ERR_ARG;
    // one instantiation per fused epilogue: statistics (training), inference BatchNorm (+ ReLU; the teacher), plain
    if (bne) stem_conv_kernel<EPI_EV><<<(int)blocks, 256, 0, to_stream(stream)>>>(a, img, rpw);
    else if (stats) stem_conv_kernel<EPI_STATS><<<(int)blocks, 256, 0, to_stream(stream)>>>(a, img, rpw);
    else stem_conv_kernel<EPI_PLAIN><<<(int)blocks, 256, 0, to_stream(stream)>>>(a, img, rpw);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The stem's weight gradient straight from the image: dw[co][k] += sum_p dy[p][co] * patch(p)[k], k = (kh*7+kw)*3+c.
// The patch-matrix route (rgda_stem_im2col + a 1x1 weight gradient) writes and re-reads 403 MB for 16 images of
// 512 x 512 (186 + 184 us of kernel time per step) to feed 20 GFLOP.  Here a workgroup owns a 64-pixel column block of
// `rpw` output rows of one image: per row it stages the image patch (as stem_conv_kernel does), builds the 64 x 192
// patch tile in LDS -- [pixel][k] in the layout the transposing LDS reads of the weight-gradient kernels expect --, copies
// the row's 64 x 64 gradient tile next to it and multiplies: dy^T (64 co x 64 pixels) x patch (64 pixels x 192) with
// ds_read_b64_tr_b16 fragments for both operands, 12 MFMAs per wave (wave = 32 channels x 96 columns), accumulating over
// its rows in registers.  Its partial [64][192] goes to the workspace; stem_wgrad_reduce_kernel adds the partials in
// workgroup order (reproducible) into dw [64][147].
// (round 6) No patch tile here either: the image rows are staged as bf16 RGBX pixels exactly as stem_conv_kernel does, and
// the B operand of  dw[co][kh][kw, c] += sum_px dy[px][co] * row(kh)[2 px + kw][c]  -- [K = pixel][N = (kw, c4)] -- is read
// with the transposing LDS read straight from the staged row: element (px, col) sits at byte 16 px + 2 col, so the "rows" of
// the 4 x 16 blocks the instruction transposes are the 64-byte windows of consecutive pixels, 16 bytes apart (overlapping
// windows: every lane supplies its own address).  K = 64 pixels per output-row segment, result [64 co][7 kh][32]: wave =
// (channel half, filter rows 0 - 3 | 4 - 6), 16 MFMAs per wave and row.  Nine-slot ring of image rows (seven live + the two the
// next output row adds), the gradient tile double-buffered: ONE LDS-only barrier per row, fetches one / two rows ahead.
__global__ void __launch_bounds__(256) stem_wgrad_kernel(const float* __restrict__ img, const bf16_t* __restrict__ dy, int lddy,
                                                         float* __restrict__ part, int H, int W, int Ho, int Wo, int rpw) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BP = 64, KP = 192;
    constexpr int SWP = 136, SROW = SWP * 8, NSLOT = 9;
    constexpr int DT = 64 * 128;                        // gradient tile: [64 pixels][64 channels]
    __shared__ __attribute__((aligned(256))) unsigned char smem[2 * DT + NSLOT * SROW];
    unsigned char* const sd = smem;
    unsigned char* const srow = smem + 2 * DT;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wave & 1, wj = wave >> 1;             // 32 output channels x filter rows {0..3} / {4..6} per wave
    const int wtiles = Wo / BP, rblocks = Ho / rpw;
    const int wo0 = (blockIdx.x % wtiles) * BP;
    const int ho0 = ((blockIdx.x / wtiles) % rblocks) * rpw, n = blockIdx.x / (wtiles * rblocks);
    const int hbase = ho0 * 2 - 3;

    const size_t plane = (size_t)H * W;
    const float* const img_n = img + (size_t)n * 3 * plane;
    auto load_px = [&](int j, int x, float (&v)[3]) {
        const int hi = hbase + j, wcol = wo0 * 2 - 3 + x;
        const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wcol < (unsigned)W && x < 134;
        const size_t o = (size_t)(ok ? hi : 0) * W + (ok ? wcol : 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = ok ? img_n[c * plane + o] : 0.f;
    };
    auto store_px = [&](int j, int x, const float (&v)[3]) {
        uint2 p;
        p.x = pack2bf(v[0], v[1]);
        p.y = pack2bf(v[2], 0.f);
        *(uint2*)(srow + (j % NSLOT) * SROW + x * 8) = p;
    };
    // gradient tile: thread = (pixel, 32-byte quarter of its 64 channels)
    const int dpx = t >> 2, dq = t & 3;
    auto load_dy = [&](int ho, uint4& v0, uint4& v1) {
        const bf16_t* dp = dy + ((size_t)(n * Ho + ho) * Wo + wo0 + dpx) * lddy + dq * 16;
        v0 = *(const uint4*)dp;
        v1 = *(const uint4*)(dp + 8);
    };
    auto store_dy = [&](int buf, const uint4& v0, const uint4& v1) {
        const int f = (dpx >> 1) & 1, s0 = dq * 2, s1 = dq * 2 + 1;
        unsigned char* b = sd + buf * DT + dpx * 128;
        *(uint4*)(b + ((((s0 >> 2) ^ f) << 6) | ((s0 & 3) << 4))) = v0;
        *(uint4*)(b + ((((s1 >> 2) ^ f) << 6) | ((s1 & 3) << 4))) = v1;
    };
    // the first window (rows j = 0 .. 6) and the first gradient tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = t + 256 * i;
        if (idx < 7 * SWP) {
            float v[3];
            load_px(idx / SWP, idx % SWP, v);
            store_px(idx / SWP, idx % SWP, v);
        }
    }
    {
        uint4 v0, v1;
        load_dy(ho0, v0, v1);
        store_dy(0, v0, v1);
    }
    const int nj0 = t / SWP, nx0 = t % SWP;
    const int nj1 = (t + 256) / SWP, nx1 = (t + 256) % SWP;
    const bool second = t + 256 < 2 * SWP;
    float na0[3] = {0.f, 0.f, 0.f}, na1[3] = {0.f, 0.f, 0.f}, nb0[3] = {0.f, 0.f, 0.f}, nb1[3] = {0.f, 0.f, 0.f};
    if (1 < rpw) {
        load_px(7 + nj0, nx0, na0);
        if (second) load_px(7 + nj1, nx1, na1);
    }
    // transposing-read lane geometry (conv_wgrad_kernel): 16-lane group g reads a [4 k][16 col] block
    const int g = lane >> 4, la = lane & 15;
    const int krow = (g >> 1) * 8 + (la >> 2);
    const int kcol2 = ((g & 1) * 16 + (la & 3) * 4) * 2;
    auto pack_tr = [](s16x4 lo, s16x4 hi) {
        u16x8 v8 = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                    (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
        return __builtin_bit_cast(bf16x8, v8);
    };
    auto tr_dy = [&](const unsigned char* base, int byte, int r0) {        // 128-byte rows, 64-byte granules ^ ((row >> 1) & 1)
        const int r1 = r0 + 4;
        const unsigned char* p0 = base + r0 * 128 + ((((byte >> 6) ^ ((r0 >> 1) & 1)) << 6) | (byte & 63));
        const unsigned char* p1 = base + r1 * 128 + ((((byte >> 6) ^ ((r1 >> 1) & 1)) << 6) | (byte & 63));
        return pack_tr(__builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0)),
                       __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1)));
    };
    auto tr_img = [&](const unsigned char* rowbase, int r0) {              // pixel windows 16 bytes apart, no swizzle
        const unsigned char* p0 = rowbase + r0 * 16 + kcol2;
        return pack_tr(__builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0)),
                       __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0 + 64)));
    };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const int kh0 = wj * 4, nkh = wj ? 3 : 4;

    for (int r = 0; r < rpw; ++r) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // window rows and gradient tile of this row are complete
        uint4 dv0 = uint4{0u, 0u, 0u, 0u}, dv1 = dv0;
        if (r + 1 < rpw) load_dy(ho0 + r + 1, dv0, dv1);
        if (r + 2 < rpw) {
            load_px(2 * r + 9 + nj0, nx0, nb0);
            if (second) load_px(2 * r + 9 + nj1, nx1, nb1);
        }
        const unsigned char* sdc = sd + (r & 1) * DT;
        const int s0 = (2 * r + kh0) % NSLOT;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int r0 = kk * 16 + krow;
            const bf16x8 af = tr_dy(sdc, wi * 64 + kcol2, r0);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (a < nkh) {
                    const int slot = (s0 + a >= NSLOT) ? s0 + a - NSLOT : s0 + a;
                    const bf16x8 bf = tr_img(srow + slot * SROW, r0);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[a], 0, 0, 0);
                }
            }
        }
        if (r + 1 < rpw) {
            store_dy((r + 1) & 1, dv0, dv1);            // (the other buffer: nobody reads it before the next barrier)
            store_px(2 * r + 7 + nj0, nx0, na0);        // slots of rows 2 r + 7, 2 r + 8: free since the barrier above
            if (second) store_px(2 * r + 7 + nj1, nx1, na1);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { na0[c] = nb0[c]; na1[c] = nb1[c]; }
    }
    float* out = part + (size_t)blockIdx.x * 64 * KP;
    const int lrow = lane & 31, lk = lane >> 5;
    const int kw = lrow >> 2, c = lrow & 3;
    if (kw < 7 && c < 3) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a < nkh) {
                const int k = ((kh0 + a) * 7 + kw) * 3 + c;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        out[(wi * 32 + 8 * g4 + 4 * lk + e) * KP + k] = acc[a][4 * g4 + e];
            }
    }
#endif
}

// dw[co][k] += sum over the workgroups' partials [nwg][64][192], k < 147: 32 outputs x 8 slices per workgroup, every slice
// adds its partials (wg = slice, slice + 8, ...) in order, the slices are added in order
__global__ void __launch_bounds__(256) stem_wgrad_reduce_kernel(const float* __restrict__ part, int nwg, float* __restrict__ dw) {
    __shared__ float red[8][32];
    const int o = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
    float a0 = 0.f, a1 = 0.f;
    if (o < 64 * 147) {
        const float* p = part + (size_t)(o / 147) * 192 + o % 147;
        int wg = sl;
        for (; wg + 8 < nwg; wg += 16) { a0 += p[(size_t)wg * 64 * 192]; a1 += p[(size_t)(wg + 8) * 64 * 192]; }
        if (wg < nwg) a0 += p[(size_t)wg * 64 * 192];
    }
    red[sl][threadIdx.x & 31] = a0 + a1;
    __syncthreads();
    if (threadIdx.x < 32 && o < 64 * 147) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) tot += red[i][threadIdx.x];
        dw[o] += tot;
    }
}

static int stem_wgrad_rows(int N, int Ho, int Wo) {
    int rpw = 32;
    while (rpw > 1 && ((Ho % rpw) || (long long)N * (Ho / rpw) * (Wo / 64) < 512)) rpw >>= 1;
    return rpw;
}

extern "C" size_t rgda_stem_wgrad_workspace(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    if ((Wo % 64) || (long long)3 * H * W >= (1ll << 28)) return 0;
    return (size_t)N * (Ho / stem_wgrad_rows(N, Ho, Wo)) * (Wo / 64) * 64 * 192 * sizeof(float);
}

extern "C" int rgda_stem_wgrad(const float* img, const void* dy, int lddy, float* dw, void* ws, size_t ws_bytes, int N,
                               int H, int W, int Ho, int Wo, rgda_stream_t stream) {
    if (!img || !dy || !dw || !ws || N <= 0 || H <= 0 || W <= 0 || (lddy & 7) || lddy < 64) return RGDA_ERR_ARG;
    if (Ho != (H + 6 - 7) / 2 + 1 || Wo != (W + 6 - 7) / 2 + 1) return RGDA_ERR_ARG;
    if (Wo % 64) return RGDA_ERR_UNSUPPORTED;           // rgda_stem_im2col + rgda_conv2d_wgrad serve these
    if ((long long)3 * H * W >= (1ll << 28)) return RGDA_ERR_UNSUPPORTED;      // (offset packing, see stem_conv_launch)
    if (((uintptr_t)ws & 15) || ws_bytes < rgda_stem_wgrad_workspace(N, H, W)) return RGDA_ERR_WORKSPACE;
    const int rpw = stem_wgrad_rows(N, Ho, Wo);
    const long long blocks = (long long)N * (Ho / rpw) * (Wo / 64);
    if (blocks > 0x7fffffffLL || (long long)N * Ho * Wo > 0x7fffffffLL) return RGDA_ERR_ARG;
    hipStream_t st = to_stream(stream);
    stem_wgrad_kernel<<<(int)blocks, 256, 0, st>>>(img, (const bf16_t*)dy, lddy, (float*)ws, H, W, Ho, Wo, rpw);
    RGDA_CHECK_LAUNCH();
    stem_wgrad_reduce_kernel<<<cdiv(64 * 147, 32), 256, 0, st>>>((const float*)ws, (int)blocks, dw);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

extern "C" int rgda_stem_conv(const float* img, const void* wgt, void* y, int ldy, rgda_stat_t* stats, int stat_groups,
                              int N, int H, int W, int Ho, int Wo, rgda_stream_t stream) {
    return stem_conv_launch(img, wgt, y, ldy, stats, stat_groups, nullptr, N, H, W, Ho, Wo, stream);
}

extern "C" int rgda_stem_conv_bneval(const float* img, const void* wgt, void* y, int ldy, const float* running_mean,
                                     const float* running_var, const float* gamma, const float* beta, float eps, int relu,
                                     int N, int H, int W, int Ho, int Wo, rgda_stream_t stream) {
    BnEvalFuse e = {running_mean, running_var, gamma, beta, eps, relu};
    return stem_conv_launch(img, wgt, y, ldy, nullptr, 1, &e, N, H, W, Ho, Wo, stream);
}

// ======================================================================================
// weight gradient
// ======================================================================================
struct WgradArgs {
    const bf16_t* x;
    const bf16_t* dy;
    float* dw;
    int ldx, lddy;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, dil;
    int M;
    int tiles_co, tiles_ci, splits, kt_per_split;
    int howo_shift, wo_shift;   // log2 when powers of two, else -1
    int tap_fused;              // host side: which kernel family the layer maps to
    int lddw;                   // elements between consecutive (co, tap) rows of dw
    int co_shift, co_nsub;      // co_shift >= 0: output row r -> dw row (r & ((1 << co_shift) - 1)) * co_nsub + (r >> co_shift)
    float* ws_part;             // splits > 1: this layer's partial tiles [tile][split][accumulators of the workgroup]
    int* ws_cnt;                //             and its per-tile arrival counters (zero before and after the launch)
    unsigned long long* dbg;    // tuning builds only: per-workgroup phase timestamps
};

// ---- reproducible split-K: the workgroups that share a result tile leave their accumulators in the workspace; the
// last one to arrive (one atomic counter per tile) sums the partials in split order and returns true -- it alone then
// adds the total to dW.  No workgroup ever waits for another one.
// The partials cross XCDs (each XCD has its own L2).  They are written and read with device-scope accesses
// (`sc1`: write-through / coherent read), the stores are acknowledged (vmcnt 0) before the workgroup's arrival is
// counted with a device-scope atomic, and the reader issues its loads after it has seen the count: no cache-wide
// write-back / invalidate.  (`__threadfence()` -- buffer_wbl2 + buffer_inv on every wave -- in these two places cost
// 2.5 ms per step.)  NACC = f32x16 accumulators per thread, NT = threads per workgroup.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define RGDA_AUX_SC1 16     /* cache-policy operand of the raw buffer builtins: bit 4 = sc1 (device scope) on gfx94x/95x */
// Many splits (a 64 x 64 result over 262 144 pixels wants hundreds of workgroups) are combined on TWO levels so that no
// workgroup walks hundreds of partials: splits form groups of RGDA_SPLIT_GROUP consecutive splits, the last arriver of
// a group sums the group's partials in split order and leaves ONE level-2 partial, the last group to finish sums the
// level-2 partials in group order.  The association ((p0 + .. + p7) + (p8 + .. + p15) + ...) is fixed by the split
// numbers alone.  Workspace per tile: splits + groups slices, 1 + groups counters (all left zero).
#define RGDA_SPLIT_GROUP 8
static inline __host__ __device__ int split_groups(int splits) { return splits > 2 * RGDA_SPLIT_GROUP ? (splits + RGDA_SPLIT_GROUP - 1) / RGDA_SPLIT_GROUP : 1; }
static inline __host__ __device__ int split_slots(int splits) { const int g = split_groups(splits); return splits + (g > 1 ? g : 0); }

// Register discipline: the tap-fused kernels are capped at 256 registers with 144 of them accumulators, and a SPILL inside
// their K loop breaks the counted `s_waitcnt vmcnt(N)` the LDS-DMA ring is ordered by (a scratch access is one more
// vector-memory operation in flight: seen as racy 1e-3 errors in multi-layer launches when this function staged 36
// vectors).  So the partials are accumulated straight into the (dead) accumulators, 12 staging vectors at a time.
template <int NACC, int NT>
static __device__ __forceinline__ bool split_k_combine(const WgradArgs& a, int tile, int split, f32x16 (&acc)[NACC],
                                                       unsigned char* smem) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int t = threadIdx.x;
    constexpr int SLICE_B = NACC * 16 * NT * 4;                        // bytes per partial tile
    const int S = a.splits, NG = split_groups(S), slots = split_slots(S);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.ws_part + (size_t)tile * slots * (SLICE_B / 4)), 0, slots * SLICE_B, 0x00020000);
    int* cnt = a.ws_cnt + tile * (NG + 1);                            // [0]: level 2, [1 + g]: level-1 group g
    int* flag = (int*)smem;
    // leave the accumulators in slice `slot`, make them visible device-wide, count the arrival: -> arrivals before this one
    auto publish = [&](int slot, int* counter) {
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 v = {__float_as_uint(acc[q][4 * g]), __float_as_uint(acc[q][4 * g + 1]),
                                 __float_as_uint(acc[q][4 * g + 2]), __float_as_uint(acc[q][4 * g + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((q * 4 + g) * NT + t) * 16, slot * SLICE_B, RGDA_AUX_SC1);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's partial values have reached memory ...
        __syncthreads();                                    // ... every thread's have, before the arrival is counted
        if (t == 0) *flag = atomicAdd(counter, 1);
        __syncthreads();
        const int before = *flag;
        __syncthreads();                                    // (the flag word is reused by a second publish)
        return before;
    };
    // acc = slices [first, first + count) added in that order, in chunks of <= 4 accumulators whose 12-16 loads of a
    // slice are in flight together (an element-by-element loop is a dependent memory round trip per 16 bytes); kernels
    // with few accumulators fetch two slices per round trip
    auto gather = [&](int first, int count) {
        constexpr int CH = (NACC <= 4) ? NACC : ((NACC % 3 == 0) ? 3 : 4);
        constexpr bool PAIR = NACC <= 4;
        static_assert(NACC % CH == 0, "accumulator count must divide into chunks");
#pragma unroll
        for (int q0 = 0; q0 < NACC; q0 += CH) {
#pragma unroll
            for (int q = q0; q < q0 + CH; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
            for (int sp = first; sp < first + count; sp += (PAIR ? 2 : 1)) {
                u32x4 v[CH * 4], w[PAIR ? CH * 4 : 1];
                const bool two = PAIR && sp + 1 < first + count;
#pragma unroll
                for (int i = 0; i < CH * 4; ++i)
                    v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((q0 * 4 + i) * NT + t) * 16, sp * SLICE_B, RGDA_AUX_SC1);
                if constexpr (PAIR) {
                    if (two) {
#pragma unroll
                        for (int i = 0; i < CH * 4; ++i)
                            w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((q0 * 4 + i) * NT + t) * 16, (sp + 1) * SLICE_B, RGDA_AUX_SC1);
                    }
                }
#pragma unroll
                for (int i = 0; i < CH * 4; ++i) {
                    const int q = q0 + i / 4, g = i % 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q][4 * g + e] += __uint_as_float(v[i][e]);
                }
                if constexpr (PAIR) {
                    if (two) {
#pragma unroll
                        for (int i = 0; i < CH * 4; ++i) {
                            const int q = q0 + i / 4, g = i % 4;
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[q][4 * g + e] += __uint_as_float(w[i][e]);
                        }
                    }
                }
            }
        }
    };
    const int G1 = (NG > 1) ? RGDA_SPLIT_GROUP : S;
    const int grp = split / G1, gfirst = grp * G1, gsize = min(G1, S - gfirst);
    if (publish(split, cnt + 1 + grp) != gsize - 1) return false;
    gather(gfirst, gsize);
    if (t == 0) cnt[1 + grp] = 0;          // ready for the next launch (nobody else looks at this counter any more)
    if (NG == 1) return true;
    if (publish(S + grp, cnt) != NG - 1) return false;
    gather(S, NG);
    if (t == 0) cnt[0] = 0;
    return true;
#else
    return false;
#endif
}

// WI x WJ waves tile the [BCO][BCI] result; 8 waves (two per SIMD) keep the 8-byte transposing LDS reads and the
// MFMA pipe busy while the other wave of the SIMD waits (one wave per SIMD reaches a fraction of the LDS rate).
// One launch covers the weight gradients of up to RGDA_WGRAD_MAXG layers that map to the same kernel
// instantiation: workgroup ids [first[l], first[l+1]) belong to layer l.  A layer-3 bottleneck of ResNet-101 has
// only 16 result tiles, so a launch per layer must split K 16 ways to fill 256 CUs -- 16x the fp32 atomics, and a
// launch ramp, a first-tile miss and an atomic drain per 9 GFLOP.  Grouped, the tiles of many layers fill the
// chip with 1-2 K splits and those fixed costs are paid once per group.
#define RGDA_WGRAD_MAXG 16
struct WgradGroup {
    int n;
    int remap;      // 1: XCD-contiguous work-item ranges (see the kernels); measured slower for grouped 1x1 layers
    int first[RGDA_WGRAD_MAXG + 1];
    WgradArgs a[RGDA_WGRAD_MAXG];
};

// wait until at most `tiles` of the most recently issued K tiles (LD DMA instructions each) are still in flight
template <int LD>
static __device__ __forceinline__ void wait_tiles_in_flight(int tiles) {
    switch (tiles) {
        case 0: WAIT_VMCNT(0); break;
        case 1: WAIT_VMCNT(LD); break;
        case 2: WAIT_VMCNT(2 * LD); break;
        case 3: WAIT_VMCNT(3 * LD); break;
        default: WAIT_VMCNT(4 * LD); break;
    }
}

template <int BCO, int BCI, int WI = 2, int WJ = 2, int STAGES = 3>
__global__ void __launch_bounds__(64 * WI * WJ) conv_wgrad_kernel(WgradGroup g) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WI * WJ;
    constexpr int FI = BCO / (32 * WI), FJ = BCI / (32 * WJ);
    static_assert(FI >= 1 && FJ >= 1, "wave grid too fine for the tile");
    constexpr int RA = BCO * 2, RB = BCI * 2;             // LDS row bytes (64 pixel rows per K tile)
    constexpr int TA = 64 * RA, TB = 64 * RB, TILE = TA + TB;
    constexpr int LA = TA / (1024 * NW), LB = TB / (1024 * NW);   // LDS-DMA instructions per wave per tile
    static_assert(LA >= 1 && LB >= 1, "too many waves for the tile");
    constexpr int LD = LA + LB;
    static_assert(STAGES >= 3 && STAGES <= 6 && LD * (STAGES - 2) < 64, "vmcnt is 6 bits");
    __shared__ __attribute__((aligned(256))) unsigned char smem[STAGES * TILE];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wave % WI, wj = wave / WI;
    // workgroup b runs on XCD b % 8: with `remap` each XCD gets a contiguous range of work items = whole layers
    int bid = g.remap ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x, layer = 0;
    while (layer + 1 < g.n && bid >= g.first[layer + 1]) ++layer;
    bid -= g.first[layer];
    const WgradArgs& a = g.a[layer];
    const int taps = a.KH * a.KW;
    const int split = bid % a.splits; bid /= a.splits;
    const int tco = bid % a.tiles_co; bid /= a.tiles_co;
    const int tci = bid % a.tiles_ci;
    const int tap = bid / a.tiles_ci;
    const int kh = tap / a.KW, kw = tap % a.KW;
    const int co0 = tco * BCO, ci0 = tci * BCI;
    const int kt_beg = split * a.kt_per_split;
    const int kt_end = min(kt_beg + a.kt_per_split, (a.M + 63) >> 6);
    const bool pointwise = (taps == 1 && a.stride == 1 && a.pad == 0);

    // LDS image: [64 pixel rows][RA bytes], lane-linear per 1 KiB instruction; the 64-byte granules of a
    // row are XOR-swizzled with the row index so that the 4 pixel rows a transposing read touches fall
    // into 4 different 64-byte bank slots (the permutation is applied to the SOURCE address).
    //   512-byte rows: 2 rows / instruction, slot p = lane & 31, granule' = granule ^ (row & 3)
    //   256-byte rows: 4 rows / instruction, slot p = lane & 15, granule' = granule ^ (row & 3)
    //   128-byte rows: 8 rows / instruction, slot p = lane & 7,  granule' = granule ^ ((row >> 1) & 1)
    constexpr int RPA = 1024 / RA, RPB = 1024 / RB;       // rows per instruction
    const int ra = lane / (RA / 16), pa = lane % (RA / 16);
    const int rb = lane / (RB / 16), pb = lane % (RB / 16);
    constexpr int OOB = (int)0x80000000;
    // buffer descriptors sized to the last valid byte: pixel rows >= M fall out of range -> zeros
    const i32x4 rs_a = dma_rsrc(a.dy, (unsigned)((((size_t)a.M - 1) * a.lddy + a.Cout) * 2));
    const i32x4 rs_b = dma_rsrc(a.x, (unsigned)((((size_t)a.N * a.H * a.W - 1) * a.ldx + a.Cin) * 2));
    auto swz = [](int row, int slot, int rowbytes) {
        int f = (rowbytes >= 256) ? (row & 3) : ((row >> 1) & 1);
        return (((slot >> 2) ^ f) << 2) | (slot & 3);
    };
    int avo[LA], brow[LB], bcolb[LB], bvo[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int row = (i * NW + wave) * RPA + ra;
        int col = co0 + swz(row, pa, RA) * 8;
        avo[i] = (col < a.Cout) ? ((row * a.lddy + col) * 2) : OOB;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        brow[i] = (i * NW + wave) * RPB + rb;
        int col = ci0 + swz(brow[i], pb, RB) * 8;
        bcolb[i] = (col < a.Cin) ? col * 2 : OOB;
        bvo[i] = (col < a.Cin) ? ((brow[i] * a.ldx + col) * 2) : OOB;
    }
    auto issue = [&](int kt, int stage) {
        const int mb = kt << 6;
        unsigned char* ab = smem + stage * TILE + wave * 1024;
        const int so_a = mb * a.lddy * 2;
#pragma unroll
        for (int i = 0; i < LA; ++i)
            dma16_to_lds(rs_a, ab + i * 1024 * NW, avo[i], so_a);
        unsigned char* bb = ab + TA;
        if (pointwise) {
            const int so_b = mb * a.ldx * 2;
#pragma unroll
            for (int i = 0; i < LB; ++i)
                dma16_to_lds(rs_b, bb + i * 1024 * NW, bvo[i], so_b);
        } else {
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                int m = mb + brow[i];
                int n, rem, ho, wo;
                if (a.howo_shift >= 0) { n = m >> a.howo_shift; rem = m & ((1 << a.howo_shift) - 1); }
                else { n = m / (a.Ho * a.Wo); rem = m % (a.Ho * a.Wo); }
                if (a.wo_shift >= 0) { ho = rem >> a.wo_shift; wo = rem & ((1 << a.wo_shift) - 1); }
                else { ho = rem / a.Wo; wo = rem % a.Wo; }
                int hi = ho * a.stride - a.pad + kh * a.dil, wi2 = wo * a.stride - a.pad + kw * a.dil;
                bool ok = (m < a.M) && (bcolb[i] != OOB) && hi >= 0 && hi < a.H && wi2 >= 0 && wi2 < a.W;
                int vo = ok ? (((n * a.H + hi) * a.W + wi2) * a.ldx * 2 + bcolb[i]) : OOB;
                dma16_to_lds(rs_b, bb + i * 1024 * NW, vo, 0);
            }
        }
    };

    f32x16 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned long long tq0 = 0, tq1 = 0, tq2 = 0;
    if (TDBG(a)) tq0 = __builtin_readcyclecounter();
    if (kt_beg < kt_end) {
        // transposing-read lane geometry: 16-lane group g reads a [4 k][16 col] block
        const int g = lane >> 4, la = lane & 15;
        const int krow = (g >> 1) * 8 + (la >> 2);          // + kk*16 (+4 for the second half)
        const int kcol2 = ((g & 1) * 16 + (la & 3) * 4) * 2;  // byte offset inside a 64-byte granule pair
        bf16x8 af[4][FI], bfr[4][FJ];
        auto tr_pair = [&](const unsigned char* base, int rowbytes, int byte, int r0) {
            const int r1 = r0 + 4;
            int f0 = (rowbytes >= 256) ? (r0 & 3) : ((r0 >> 1) & 1), f1 = (rowbytes >= 256) ? (r1 & 3) : ((r1 >> 1) & 1);
            const unsigned char* p0 = base + r0 * rowbytes + ((((byte >> 6) ^ f0) << 6) | (byte & 63));
            const unsigned char* p1 = base + r1 * rowbytes + ((((byte >> 6) ^ f1) << 6) | (byte & 63));
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
            u16x8 v8 = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                        (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
            return __builtin_bit_cast(bf16x8, v8);
        };
        auto read_half = [&](int st, int h) {
            const unsigned char* ab = smem + st * TILE;
            const unsigned char* bb = ab + TA;
#pragma unroll
            for (int kk = 2 * h; kk < 2 * h + 2; ++kk) {
                const int r0 = kk * 16 + krow;
#pragma unroll
                for (int i = 0; i < FI; ++i) af[kk][i] = tr_pair(ab, RA, (wi * (BCO / WI) + i * 32) * 2 + kcol2, r0);
#pragma unroll
                for (int j = 0; j < FJ; ++j) bfr[kk][j] = tr_pair(bb, RB, (wj * (BCI / WJ) + j * 32) * 2 + kcol2, r0);
            }
        };
        auto mfma_half = [&](int h) {
#pragma unroll
            for (int kk = 2 * h; kk < 2 * h + 2; ++kk)
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < FJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][i], bfr[kk][j], acc[i][j], 0, 0, 0);
        };
        int stage = 0;
        if constexpr (STAGES == 3) {
            // software-pipelined like conv_igemm_kernel<..., PIPE>: all three stages in flight, the barrier of tile
            // kt+1 between the two halves of tile kt's MFMAs, the last tile peeled
#pragma unroll
            for (int s0 = 0; s0 < 3; ++s0)
                if (kt_beg + s0 < kt_end) issue(kt_beg + s0, s0);
            wait_tiles_in_flight<LD>(min(2, kt_end - 1 - kt_beg));
            __builtin_amdgcn_s_barrier();
            if (TDBG(a)) tq1 = __builtin_readcyclecounter();
            read_half(0, 0);
            for (int kt = kt_beg; kt + 1 < kt_end; ++kt) {
                const int nxt = (stage == 2) ? 0 : stage + 1;
                read_half(stage, 1);
                mfma_half(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every wave's reads of this tile are complete
                wait_tiles_in_flight<LD>(min(1, kt_end - 2 - kt));       // tile kt+1 landed (kt+2 may still fly)
                __builtin_amdgcn_s_barrier();
                if (kt + 3 < kt_end) issue(kt + 3, stage);
                read_half(nxt, 0);
                mfma_half(1);
                stage = nxt;
            }
            read_half(stage, 1);
            mfma_half(0);
            mfma_half(1);
        } else {
            // the operands stream from HBM / the Infinity Cache (~2300 cycles away): STAGES - 1 tiles stay in flight
#pragma unroll
            for (int s0 = 0; s0 < STAGES - 1; ++s0)
                if (kt_beg + s0 < kt_end) issue(kt_beg + s0, s0);
            for (int kt = kt_beg; kt < kt_end; ++kt) {
                wait_tiles_in_flight<LD>(min(STAGES - 2, kt_end - 1 - kt));
                __builtin_amdgcn_s_barrier();
                if (TDBG(a) && kt == kt_beg) tq1 = __builtin_readcyclecounter();
                if (kt + STAGES - 1 < kt_end) issue(kt + STAGES - 1, stage >= 1 ? stage - 1 : STAGES - 1);
                read_half(stage, 0);
                read_half(stage, 1);
                mfma_half(0);
                mfma_half(1);
                stage = (stage == STAGES - 1) ? 0 : stage + 1;
            }
        }
    }
    // ---- dW += the tile: ONE workgroup per tile adds (the only split, or the last of the splits to arrive, with the
    // ordered sum of all partials), so the result does not depend on any arrival order.  The add itself is still the
    // fire-and-forget fp32 atomic (a plain read-modify-write is a dependent memory round trip per element: +3 ms/step).
    // 32-bit element offsets off one base (a layer's dW is far below 2^31 elements): no 64-bit multiplies here.
    if (TDBG(a)) tq2 = __builtin_readcyclecounter();
    if (a.splits > 1 && a.ws_part) {
        __syncthreads();                   // every wave is done with the LDS ring
        f32x16 (&flat)[FI * FJ] = *reinterpret_cast<f32x16 (*)[FI * FJ]>(&acc[0][0]);
        if (!split_k_combine<FI * FJ, 64 * NW>(a, (tap * a.tiles_ci + tci) * a.tiles_co + tco, split, flat, smem)) return;
    }
    const int lcol = lane & 31, lk = lane >> 5;
    // dw row of output channel co: co itself, or (stacked filters, co_shift >= 0) its channel-major position; consecutive
    // channels of one 32-channel block are `rstep` rows apart either way (a block never straddles a filter)
    const unsigned rstep = a.co_shift >= 0 ? (unsigned)a.co_nsub : 1u;
    const unsigned rs = rstep * (unsigned)(taps * a.lddw);
    const bool full = (co0 + BCO <= a.Cout) && (ci0 + BCI <= a.Cin);
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int cob = co0 + wi * (BCO / WI) + i * 32 + 4 * lk;
            const int ci = ci0 + wj * (BCI / WJ) + j * 32 + lcol;
            const unsigned rowb = a.co_shift >= 0 ? (unsigned)((cob & ((1 << a.co_shift) - 1)) * a.co_nsub + (cob >> a.co_shift))
                                                  : (unsigned)cob;
            unsigned o = rowb * (unsigned)(taps * a.lddw) + (unsigned)(tap * a.lddw + ci);
            if (full) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) atomicAdd(a.dw + (o + q * rs), acc[i][j][g4 * 4 + q]);
                    o += 8 * rs;
                }
            } else if (ci < a.Cin) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int co = cob + (r & 3) + 8 * (r >> 2);
                    if (co < a.Cout) atomicAdd(a.dw + (o + (unsigned)((r & 3) + 8 * (r >> 2)) * rs), acc[i][j][r]);
                }
            }
        }
    if (TDBG(a) && t == 0) {
        TDBG(a)[blockIdx.x * 4 + 0] = tq0; TDBG(a)[blockIdx.x * 4 + 1] = tq1;
        TDBG(a)[blockIdx.x * 4 + 2] = tq2; TDBG(a)[blockIdx.x * 4 + 3] = __builtin_readcyclecounter();
    }
#endif
}

// ======================================================================================
// weight gradient of 3x3 / stride-1 / "same" convolutions, all nine taps fused
//   One workgroup owns dW[64 cout][9 taps][64 cin].  A K tile is 64 pixels = R image rows x WT columns; its
//   dY rows [64][64 cout] and ONE halo tile of the input [(R+2D) x (WT+2D) pixels][64 cin] are DMA'd to LDS and
//   the nine taps read the halo tile at nine shifted row offsets: 25 KB per 4.7 MFLOP = 188 FLOP per byte pulled
//   from L2 (the generic kernel: 64), and the dY tile is read once instead of nine times.
// ======================================================================================
// Occupancy: nine 32 x 32 accumulators are 144 VGPRs; left alone the compiler takes 400 registers (double-buffered
// fragments), i.e. ONE wave per SIMD, and the matrix pipe idles whenever that wave waits for LDS (27-34 % MFMA busy).
// Capped at 256 registers (two waves per SIMD, no spills to speak of) and with a 2-stage ring wherever two rings fit
// the 160 KB of a CU, two workgroups share a CU and one's LDS / barrier waits hide under the other's MFMAs:
// layer-3 group launches 366 -> 291 us, -0.22 ms per step in an interleaved A/B (round 3).
template <int WT, int D, int STAGES>
static __device__ __forceinline__ void conv_wgrad3x3_body(const WgradGroup& g) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int R = 64 / WT;                       // image rows per K tile
    constexpr int HR = R + 2 * D, HC = WT + 2 * D;   // halo tile (pixels)
    // the halo tile is padded to whole 8-row DMA pieces only (NG of them, 1 KiB each); the four waves take pieces i * 4 + wave,
    // and where the last round has fewer than four pieces the surplus waves repeat the last one (same source, same
    // destination, same bytes): every wave issues LD instructions per tile -- the counted vmcnt waits need that -- and the
    // tile is 25 KB instead of 28 (WT = 32, D = 1: three stages of it fit a CU twice)
    constexpr int NH = HR * HC, NG = (NH + 7) / 8, NHP = NG * 8;
    constexpr int TA = 64 * 128, TB = (NHP * 128 + 1023) / 1024 * 1024, TILE = TA + TB;
    constexpr int LA = 2, LB = (NG + 3) / 4;         // LDS-DMA instructions per wave per tile (8 rows each)
    constexpr int LD = LA + LB;
    static_assert(STAGES == 2 || STAGES == 3, "ring of 2 (two workgroups per CU) or 3 stages");
    __shared__ __attribute__((aligned(256))) unsigned char smem[STAGES * TILE];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wave & 1, wj = wave >> 1;
    // workgroup b runs on XCD b % 8: with `remap` each XCD gets a contiguous range of work items = whole layers
    int bid = g.remap ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x, layer = 0;
    while (layer + 1 < g.n && bid >= g.first[layer + 1]) ++layer;
    bid -= g.first[layer];
    const WgradArgs& a = g.a[layer];
    const int split = bid % a.splits; bid /= a.splits;
    const int tco = bid % a.tiles_co;
    const int tci = bid / a.tiles_co;
    const int co0 = tco * 64, ci0 = tci * 64;
    const int tpr = a.W / WT, tpi = tpr * (a.H / R);        // K tiles per image row-band / per image
    const int KT = a.N * tpi;
    const int kt_beg = split * a.kt_per_split, kt_end = min(kt_beg + a.kt_per_split, KT);
    constexpr int OOB = (int)0x80000000;
    const i32x4 rs_a = dma_rsrc(a.dy, (unsigned)((((size_t)a.M - 1) * a.lddy + a.Cout) * 2));
    const i32x4 rs_b = dma_rsrc(a.x, (unsigned)((((size_t)a.N * a.H * a.W - 1) * a.ldx + a.Cin) * 2));

    // lane geometry of a DMA instruction: 8 rows x 8 slots of 16 B; 128-byte rows, granule swizzle (row>>1)&1
    const int lr = lane >> 3, ls = lane & 7;
    auto swz = [](int row, int slot) { return (((slot >> 2) ^ ((row >> 1) & 1)) << 2) | (slot & 3); };
    int avo[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int row = (i * 4 + wave) * 8 + lr;
        int col = co0 + swz(row, ls) * 8;
        avo[i] = (col < a.Cout) ? ((row * a.lddy + col) * 2) : OOB;
    }
    int bhr[LB], bhc[LB], bcol[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        int h = min(i * 4 + wave, NG - 1) * 8 + lr;
        int col = ci0 + swz(h, ls) * 8;
        bool ok = (h < NH) && (col < a.Cin);
        bhr[i] = ok ? (h / HC - D) : -100000;             // row / column of the halo pixel relative to the tile origin
        bhc[i] = h % HC - D;
        bcol[i] = col * 2;
    }
    auto issue = [&](int kt, int stage) {
        const int n = kt / tpi, rem = kt % tpi;
        const int y0 = (rem / tpr) * R, x0 = (rem % tpr) * WT;
        unsigned char* ab = smem + stage * TILE + wave * 1024;
        const int so_a = ((n * a.H + y0) * a.W + x0) * a.lddy * 2;
#pragma unroll
        for (int i = 0; i < LA; ++i)
            dma16_to_lds(rs_a, ab + i * 4096, avo[i], so_a);
        unsigned char* bb = smem + stage * TILE + TA;
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            int y = y0 + bhr[i], x = x0 + bhc[i];
            bool ok = (y >= 0) && (y < a.H) && (x >= 0) && (x < a.W);
            int vo = ok ? (((n * a.H + y) * a.W + x) * a.ldx * 2 + bcol[i]) : OOB;
            dma16_to_lds(rs_b, bb + min(i * 4 + wave, NG - 1) * 1024, vo, 0);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    if (kt_beg < kt_end) {
        issue(kt_beg, 0);
        if (STAGES == 3 && kt_beg + 1 < kt_end) issue(kt_beg + 1, 1);
        const int g = lane >> 4, la = lane & 15;
        const int klane = (g >> 1) * 8 + (la >> 2);                    // pixel of this lane inside a 16-pixel k step
        const int cbyte = ((g & 1) * 16 + (la & 3) * 4) * 2;           // column byte inside the wave's 32 columns
        const int abyte = wi * 64 + cbyte, bbyte = wj * 64 + cbyte;    // 32 columns = 64 bytes per wave half
        auto tr = [&](const unsigned char* base, int row, int byte) {
            const unsigned char* p = base + row * 128 + ((((byte >> 6) ^ ((row >> 1) & 1)) << 6) | (byte & 63));
            return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
        };
        auto pack = [](s16x4 lo, s16x4 hi) {
            u16x8 v8 = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                        (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
            return __builtin_bit_cast(bf16x8, v8);
        };
        // (a software-pipelined variant of this loop, like conv_igemm_kernel<..., PIPE>, measured 8-10 % SLOWER here:
        // with 9 MFMAs per 20 transposing reads the compiler's own interleaving already hides the LDS latency)
        int stage = 0;
        for (int kt = kt_beg; kt < kt_end; ++kt) {
            if (STAGES == 3 && kt + 1 < kt_end) WAIT_VMCNT(LD); else WAIT_VMCNT(0);
            __builtin_amdgcn_s_barrier();
            if (kt + STAGES - 1 < kt_end) issue(kt + STAGES - 1, stage >= 1 ? stage - 1 : STAGES - 1);
            const unsigned char* ab = smem + stage * TILE;
            const unsigned char* bb = ab + TA;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                // the 16 pixels of this k step lie in one image row of the tile (WT >= 16)
                constexpr int dummy = 0; (void)dummy;
                const int r_kk = (kk * 16) / WT, c_kk = (kk * 16) % WT;
                const int k0 = kk * 16 + klane;
                bf16x8 af = pack(tr(ab, k0, abyte), tr(ab, k0 + 4, abyte));
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int h0 = (r_kk + kh * D) * HC + c_kk + kw * D + klane;
                        bf16x8 bfr = pack(tr(bb, h0, bbyte), tr(bb, h0 + 4, bbyte));
                        acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[kh * 3 + kw], 0, 0, 0);
                    }
            }
            // (the tile's 80 transposing reads spread between its 36 MFMAs by sched_group_barrier: measured no faster, and
            // 12 - 20 % slower for the dilated layers -- round 5)
            stage = (stage == STAGES - 1) ? 0 : stage + 1;
        }
    }
    if (a.splits > 1 && a.ws_part) {
        __syncthreads();                   // every wave is done with the LDS ring
        if (!split_k_combine<9, 256>(a, tci * a.tiles_co + tco, split, acc, smem)) return;
    }
    const int lcol = lane & 31, lk = lane >> 5;
    const int ci = ci0 + wj * 32 + lcol;
    if (ci < a.Cin) {
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co < a.Cout) atomicAdd(a.dw + ((size_t)co * 9 + q) * a.lddw + ci, acc[q][r]);
            }
    }
#endif
}
template <int WT, int D, int STAGES = 3>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_wgrad3x3_kernel(WgradGroup g) {
    conv_wgrad3x3_body<WT, D, STAGES>(g);
}
// the variants that do not fit 256 registers without spilling (a spill inside the K loop would break the counted vmcnt
// waits: tests/test_host_logic.py checks the compiler's report) or two rings per CU: one wave per SIMD, 3-stage ring
template <int WT, int D, int STAGES = 3>
__global__ void __launch_bounds__(256) conv_wgrad3x3_wide_kernel(WgradGroup g) {
    conv_wgrad3x3_body<WT, D, STAGES>(g);
}

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

// kernel families: 0..4 generic <128,128> <128,64> <64,128> <64,64> <256,128>; 5..10 tap-fused 3x3 <WT,D>
enum { WK_G128_128 = 0, WK_G128_64, WK_G64_128, WK_G64_64, WK_G256_128, WK_F64_1, WK_F64_2, WK_F32_1, WK_F32_2, WK_F16_1, WK_F16_2, WK_COUNT };

#ifndef RGDA_TAPFUSED_MIN_TILES
#define RGDA_TAPFUSED_MIN_TILES 1
#endif
static int wgrad_prepare(const rgda_wgrad_desc& d, WgradArgs& a) {
    if (!d.x || !d.dy || !d.dw) return RGDA_ERR_ARG;
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || d.Ho <= 0 || d.Wo <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.kh <= 0 || d.kw <= 0 ||
        d.stride <= 0 || d.dil <= 0 || d.pad < 0)
        return RGDA_ERR_ARG;
    if ((d.Cin & 7) || (d.Cout & 7) || (d.ldx & 7) || (d.lddy & 7) || d.ldx < d.Cin || d.lddy < d.Cout) return RGDA_ERR_ARG;
    a.x = (const bf16_t*)d.x; a.dy = (const bf16_t*)d.dy; a.dw = d.dw; a.ldx = d.ldx; a.lddy = d.lddy;
    a.N = d.N; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.Ho = d.Ho; a.Wo = d.Wo; a.Cout = d.Cout; a.KH = d.kh; a.KW = d.kw;
    a.stride = d.stride; a.pad = d.pad; a.dil = d.dil;
    a.lddw = d.lddw ? d.lddw : d.Cin;
    a.co_shift = -1; a.co_nsub = 1;
    if (a.lddw < d.Cin) return RGDA_ERR_ARG;
    if (d.co_split) {           // stacked 1x1 filters written channel-major
        const int sh = ilog2_exact(d.co_split);
        if (sh < 5 || (d.Cout % d.co_split) || d.kh != 1 || d.kw != 1) return RGDA_ERR_ARG;    // (32-channel blocks stay inside a filter)
        a.co_shift = sh; a.co_nsub = d.Cout / d.co_split;
    }
    if ((long long)d.Cout * d.kh * d.kw * a.lddw > 0xffffffffLL) return RGDA_ERR_ARG;          // 32-bit element offsets
    long long M = (long long)d.N * d.Ho * d.Wo;
    if (M > 0x7fffffffLL - 64) return RGDA_ERR_ARG;
    a.M = (int)M;
    a.howo_shift = ilog2_exact(d.Ho * d.Wo);
    a.wo_shift = ilog2_exact(d.Wo);
    a.dbg = nullptr;
    if (const char* e = TUNE_ENV("RGDA_CONV_DBG")) a.dbg = (unsigned long long*)strtoull(e, nullptr, 0);   // tuning only
    a.splits = 1; a.kt_per_split = 0; a.ws_part = nullptr; a.ws_cnt = nullptr;
    // tap-fused path: 3x3, stride 1, "same" padding, the map tiles into 64-pixel row blocks
    a.tap_fused = 0;
    if (d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad == d.dil && d.Ho == d.H && d.Wo == d.W &&
        cdiv(d.Cout, 64) * cdiv(d.Cin, 64) >= RGDA_TAPFUSED_MIN_TILES && !TUNE_ENV("RGDA_WGRAD_GENERIC")) {
        int wt = (d.W >= 64) ? 64 : d.W;
        if ((wt == 64 || wt == 32 || wt == 16) && (d.W % wt) == 0 && (d.H % (64 / wt)) == 0 && (d.dil == 1 || d.dil == 2)) {
            a.tap_fused = 1;
            a.tiles_co = cdiv(d.Cout, 64);
            a.tiles_ci = cdiv(d.Cin, 64);
            return (wt == 64 ? WK_F64_1 : wt == 32 ? WK_F32_1 : WK_F16_1) + (d.dil == 2 ? 1 : 0);
        }
    }
    int bco = (d.Cout <= 64) ? 64 : 128, bci = (d.Cin <= 64) ? 64 : 128;
    // wide layers: 256 x 128 result tiles (8 waves of 64 x 64): 1.0 LDS fragment read per MFMA instead of 1.5 and
    // 48 KB instead of 64 KB of operands through LDS per 4.2 MFLOP -- the 128 x 128 loop is bound by LDS cycles
    int big = 1;
    if (const char* e = TUNE_ENV("RGDA_WGRAD_BIG")) big = atoi(e);                        // tuning experiments only
    // (a 2-stage 256 x 256 tile, MFMA-bound on paper, measured the same step time: 21.71 vs 21.72 ms)
    if (big && bci == 128 && d.Cout >= 256 && (d.Cout % 256) == 0) bco = 256;
    a.tiles_co = cdiv(d.Cout, bco);
    a.tiles_ci = cdiv(d.Cin, bci);
    if (bco == 256) return WK_G256_128;
    return (bco == 128) ? (bci == 128 ? WK_G128_128 : WK_G128_64) : (bci == 128 ? WK_G64_128 : WK_G64_64);
}

static inline int wgrad_tiles(const WgradArgs& a) {
    return a.tiles_co * a.tiles_ci * (a.tap_fused ? 1 : a.KH * a.KW);
}
static inline int wgrad_ktiles(const WgradArgs& a) {
    if (!a.tap_fused) return cdiv(a.M, 64);
    int wt = (a.W >= 64) ? 64 : a.W;
    return a.N * (a.H / (64 / wt)) * (a.W / wt);
}

// floats a workgroup of kernel family `kind` leaves in the workspace per split (its accumulators)
static inline size_t wgrad_slice_floats(int kind) {
    switch (kind) {
        case WK_G128_128: return 128 * 128;
        case WK_G128_64: return 128 * 64;
        case WK_G64_128: return 64 * 128;
        case WK_G64_64: return 64 * 64;
        case WK_G256_128: return 256 * 128;
        default: return 9 * 64 * 64;            // tap-fused: nine 64 x 64 taps
    }
}
#define RGDA_WGRAD_WS_COUNTERS (64 * 1024)      // bytes of tile counters at the head of the workspace (16 K tiles per launch)

// K splits of the layers of one launch (g.a[0..n)); 1 without a workspace.  One split count S for the launch (a layer
// with few K tiles gets fewer: never fewer than 16 K tiles per split), chosen by a round model of the launch: its
// workgroups run in ceil(workgroups / capacity) rounds (capacity = 256, one workgroup per CU, for the 144 KB tiles;
// 512 for the others) of ceil(K tiles / S) tile steps each, plus 8 + 2 x (partials one workgroup sums) tile steps when
// S > 1: the partial tiles' trip through the workspace and the last arrivers' walks (S partials on one level, 8 + S / 8
// on two, split_k_combine).  (Splitting "until the chip is full" -- the rule before ABI 4 -- made
// 336 workgroups out of 112 tiles: two rounds of 86 steps where S = 2 gives one round of 128.)
// Returns the bytes of partial tiles the launch needs behind the counters.
static size_t wgrad_plan_splits(int kind, WgradGroup& g, bool have_ws) {
    // workgroups the chip holds at once: one per CU for the 144 KB generic tile and the dilated 64-wide tap-fused kernel,
    // two per CU for everything else (the tap-fused kernels run 2-stage rings of 48-72 KB)
    const int capacity = (kind == WK_G256_128 || kind == WK_F64_1 || kind == WK_F64_2 || kind == WK_F32_2) ? 256 : 512;
    int minkt = 16;
    if (const char* e = TUNE_ENV("RGDA_WGRAD_MINKT")) minkt = atoi(e);                     // tuning experiments only
    int best = 1;
    int rule = 1;
    if (const char* e = TUNE_ENV("RGDA_WGRAD_RULE")) rule = atoi(e);                       // tuning experiments only
    if (have_ws && rule == 0) {            // the pre-ABI-4 rule: split until the launch fills the chip
        int total = 0;
        for (int l = 0; l < g.n; ++l) total += wgrad_tiles(g.a[l]);
        best = cdiv(capacity, total);
    } else if (have_ws) {
        long long best_cost = -1;
        for (int S = 1; S <= 512; S += (S < 16 ? 1 : 8)) {
            long long wgs = 0, steps = 0;
            for (int l = 0; l < g.n; ++l) {
                const int KT = wgrad_ktiles(g.a[l]);
                int sl = S;
                if (sl > KT / minkt) sl = KT / minkt;
                if (sl < 1) sl = 1;
                const int per = cdiv(KT, sl);
                wgs += (long long)wgrad_tiles(g.a[l]) * cdiv(KT, per);
                if (per > steps) steps = per;
            }
            const int walk = split_groups(S) > 1 ? RGDA_SPLIT_GROUP + split_groups(S) + 8 : S;   // partials one workgroup sums
            const long long cost = (long long)cdiv(wgs, capacity) * (steps + (S > 1 ? 8 + 2 * walk : 0));
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = S; }
        }
    }
    if (const char* e = TUNE_ENV("RGDA_WGRAD_SPLITS")) best = atoi(e);                     // tuning experiments only
    size_t floats = 0;
    int items = 0, tiles_before = 0;
    for (int l = 0; l < g.n; ++l) {
        WgradArgs& a = g.a[l];
        const int KT = wgrad_ktiles(a);
        int splits = best;
        if (splits > KT / minkt) splits = KT / minkt;
        if (splits < 1) splits = 1;
        a.kt_per_split = cdiv(KT, splits);
        a.splits = cdiv(KT, a.kt_per_split);
        g.first[l] = items;
        items += wgrad_tiles(a) * a.splits;
        // workspace slots of this layer (offsets here; the launch turns them into pointers)
        a.ws_part = (float*)(uintptr_t)floats;
        a.ws_cnt = (int*)(uintptr_t)tiles_before;
        if (a.splits > 1) {
            floats += (size_t)wgrad_tiles(a) * split_slots(a.splits) * wgrad_slice_floats(kind);
            tiles_before += wgrad_tiles(a) * (split_groups(a.splits) + 1);      // counters of this layer
        }
    }
    for (int l = g.n; l <= RGDA_WGRAD_MAXG; ++l) g.first[l] = items;
    return floats * 4;
}

static int wgrad_launch(int kind, WgradGroup& g, void* ws, size_t ws_bytes, hipStream_t st) {
    const size_t need = wgrad_plan_splits(kind, g, ws != nullptr);
    int counters = 0;
    for (int l = 0; l < g.n; ++l)
        if (g.a[l].splits > 1) counters += wgrad_tiles(g.a[l]) * (split_groups(g.a[l].splits) + 1);
    if (need && (size_t)counters * 4 > RGDA_WGRAD_WS_COUNTERS) {
        // more result tiles than tile counters: this launch runs unsplit (slower, never wrong) instead of failing
        wgrad_plan_splits(kind, g, false);
        return wgrad_launch(kind, g, nullptr, 0, st);
    }
    if (need && ws_bytes < RGDA_WGRAD_WS_COUNTERS + need) return RGDA_ERR_WORKSPACE;
    for (int l = 0; l < g.n; ++l) {
        WgradArgs& a = g.a[l];
        a.ws_part = (float*)((char*)ws + RGDA_WGRAD_WS_COUNTERS) + (size_t)(uintptr_t)a.ws_part;
        a.ws_cnt = (int*)ws + (size_t)(uintptr_t)a.ws_cnt;
        if (TUNE_ENV("RGDA_WGRAD_ATOMIC")) a.ws_part = nullptr;    // tuning only: every split adds to dW itself (not reproducible)
    }
    const int items = g.first[RGDA_WGRAD_MAXG];
    // keeping a layer's work items on one XCD (one L2) pays where the items of a layer re-read the same rows many
    // times (the taps of the small-channel 3x3 layers in the 64x64 kernel: 297 -> 220 us; tap-fused: 2 %); the
    // grouped 1x1 layers of the 128x128 kernel measured 9 % SLOWER with it (207 -> 225 us), so they keep b -> item b.
    // (Round 5, measured and not kept: a finer mapping for those -- the result tiles of one (layer, K split), which stream the
    // same pixel rows, in consecutive slots of ONE XCD, the units round-robin over the XCDs, so that the 1.5 x over-fetch of
    // b -> item b (FETCH_SIZE 1.0 GB against 0.67 GB unique per 16-layer launch) would hit one L2: conv_wgrad_kernel<256, 128>
    // 194 -> 226 us, step +0.28 ms (interleaved A/B).  32 workgroups x 3 stages x 48 KB in flight per XCD exceed its 4 MB
    // L2; the launches run at 5.2 TB/s of fabric reads either way and spreading a unit over four L2s is what reaches it.)
    g.remap = (kind != WK_G128_128 && kind != WK_G256_128);
    if (const char* e = TUNE_ENV("RGDA_WGRAD_REMAP")) g.remap = atoi(e);                   // tuning experiments only
    switch (kind) {
        case WK_G128_128: conv_wgrad_kernel<128, 128, 2, 4><<<items, 512, 0, st>>>(g); break;
        case WK_G128_64: conv_wgrad_kernel<128, 64, 4, 2><<<items, 512, 0, st>>>(g); break;
        case WK_G64_128: conv_wgrad_kernel<64, 128, 2, 4><<<items, 512, 0, st>>>(g); break;
        case WK_G64_64: conv_wgrad_kernel<64, 64><<<items, 256, 0, st>>>(g); break;
        case WK_G256_128: conv_wgrad_kernel<256, 128, 4, 2><<<items, 512, 0, st>>>(g); break;
        case WK_F64_1: conv_wgrad3x3_wide_kernel<64, 1, 3><<<items, 256, 0, st>>>(g); break;  // (capped at 256 registers it spills)
        case WK_F64_2: conv_wgrad3x3_wide_kernel<64, 2, 3><<<items, 256, 0, st>>>(g); break;  // 52 KB per stage: one per CU either way
        // (round 5: with the compact halo tile THREE stages fit a CU twice (75 KB); measured 296 against 294 us per launch and
        // +0.2 ms on the step: the loop does not wait for its DMA -- the other workgroup of the CU covers it -- and the
        // larger footprint costs the neighbouring streams' kernels their place)
        case WK_F32_1: conv_wgrad3x3_kernel<32, 1, 2><<<items, 256, 0, st>>>(g); break;       // 50 KB
        case WK_F32_2: conv_wgrad3x3_wide_kernel<32, 2, 3><<<items, 256, 0, st>>>(g); break;  // (capped at 256 registers it spills)
        case WK_F16_1: conv_wgrad3x3_kernel<16, 1, 3><<<items, 256, 0, st>>>(g); break;       // 72 KB
        default: conv_wgrad3x3_kernel<16, 2, 2><<<items, 256, 0, st>>>(g); break;              // 56 KB
    }
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// walks the list the way the launches are cut (per kernel family, RGDA_WGRAD_MAXG layers per launch) and calls
// fn(kind, group) for every launch, in issue order
template <typename F>
static int wgrad_for_each_launch(const rgda_wgrad_desc* descs, int n, F fn) {
    static thread_local WgradGroup groups[WK_COUNT];
    for (int k = 0; k < WK_COUNT; ++k) groups[k].n = 0;
    // validate everything first: nothing is launched for a list with a bad entry
    for (int i = 0; i < n; ++i) {
        WgradArgs a;
        int kind = wgrad_prepare(descs[i], a);
        if (kind < 0) return kind;
    }
    for (int i = 0; i < n; ++i) {
        WgradArgs a;
        int kind = wgrad_prepare(descs[i], a);
        WgradGroup& g = groups[kind];
        g.a[g.n++] = a;
        if (g.n == RGDA_WGRAD_MAXG) {
            int rc = fn(kind, g);
            if (rc != RGDA_OK) return rc;
            g.n = 0;
        }
    }
    for (int k = 0; k < WK_COUNT; ++k)
        if (groups[k].n) {
            int rc = fn(k, groups[k]);
            if (rc != RGDA_OK) return rc;
        }
    return RGDA_OK;
}

// the kernel instantiation (as rocprofv3 names it) a layer's weight gradient maps to; layers with the same name share launches
extern "C" const char* rgda_conv2d_wgrad_kernel(const rgda_wgrad_desc* d) {
    static const char* const names[WK_COUNT] = {
        "conv_wgrad_kernel<128, 128, 2, 4, 3>", "conv_wgrad_kernel<128, 64, 4, 2, 3>", "conv_wgrad_kernel<64, 128, 2, 4, 3>",
        "conv_wgrad_kernel<64, 64, 2, 2, 3>", "conv_wgrad_kernel<256, 128, 4, 2, 3>",
        "conv_wgrad3x3_wide_kernel<64, 1, 3>", "conv_wgrad3x3_wide_kernel<64, 2, 3>", "conv_wgrad3x3_kernel<32, 1, 2>",
        "conv_wgrad3x3_wide_kernel<32, 2, 3>", "conv_wgrad3x3_kernel<16, 1, 3>", "conv_wgrad3x3_kernel<16, 2, 2>"};
    if (!d) return nullptr;
    WgradArgs a;
    const int kind = wgrad_prepare(*d, a);
    return (kind >= 0 && kind < WK_COUNT) ? names[kind] : nullptr;
}

extern "C" size_t rgda_conv2d_wgrad_workspace(const rgda_wgrad_desc* descs, int n) {
    if (n <= 0 || !descs) return RGDA_WGRAD_WS_COUNTERS;
    size_t most = 0;
    wgrad_for_each_launch(descs, n, [&](int kind, WgradGroup& g) {
        const size_t b = wgrad_plan_splits(kind, g, true);
        if (b > most) most = b;
        return (int)RGDA_OK;
    });
    return RGDA_WGRAD_WS_COUNTERS + most;       // the launches of one call run one after the other: they share the space
}

extern "C" int rgda_conv2d_wgrad_grouped(const rgda_wgrad_desc* descs, int n, void* ws, size_t ws_bytes, rgda_stream_t stream) {
    if (n < 0 || (n > 0 && !descs)) return RGDA_ERR_ARG;
    if (ws && (ws_bytes < RGDA_WGRAD_WS_COUNTERS || ((uintptr_t)ws & 15))) return RGDA_ERR_WORKSPACE;
    hipStream_t st = to_stream(stream);
    const int rc = wgrad_for_each_launch(descs, n, [&](int kind, WgradGroup& g) { return wgrad_launch(kind, g, ws, ws_bytes, st); });
    // the split-K protocol relies on the tile counters being zero between launches: after a failed call (some launches of
    // the list may have run, one did not) they are cleared behind whatever is in flight, so later calls start clean
    if (rc != RGDA_OK && ws) (void)hipMemsetAsync(ws, 0, RGDA_WGRAD_WS_COUNTERS, st);
    return rc;
}

extern "C" int rgda_conv2d_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, int N, int H, int W,
                                 int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil,
                                 void* ws, size_t ws_bytes, rgda_stream_t stream) {
    rgda_wgrad_desc d;
    d.x = x; d.dy = dy; d.dw = dw; d.ldx = ldx; d.lddy = lddy; d.lddw = 0; d.co_split = 0;
    d.N = N; d.H = H; d.W = W; d.Cin = Cin; d.Ho = Ho; d.Wo = Wo; d.Cout = Cout; d.kh = kh; d.kw = kw;
    d.stride = stride; d.pad = pad; d.dil = dil;
    return rgda_conv2d_wgrad_grouped(&d, 1, ws, ws_bytes, stream);
}
