// Implicit-GEMM convolution for gfx950 on pixel-major bf16 activations ("PxC": [N*H*W][C]).
//
//   forward / data-gradient : D[co][pixel] = sum_{tap,ci} W[co][tap][ci] * X[src(pixel,tap)][ci]
//   weight-gradient         : dW[co][tap][ci] += sum_pixel dY[pixel][co] * X[src(pixel,tap)][ci]
//
// v_mfma_f32_32x32x16_bf16, fp32 accumulate.  256-thread workgroups = 2x2 wavefronts, every
// wavefront owns a (BC/2)x(BP/2) block of 32x32 accumulators.  Operand tiles are staged
// HBM -> registers -> LDS (issue-early / write-late, double buffered, one barrier per K tile):
// the gathered, zero-padded pixel rows of an implicit GEMM cannot be expressed as the
// lane-linear image global_load_lds wants without a second pass.
//   * forward: both operands are K(channel)-contiguous -> 128-byte LDS rows with a 16-byte-slot
//     XOR swizzle (slot ^= (row>>1)&7): conflict-free for the 16-lane groups of ds_read_b128.
//   * wgrad:   both operands are K(pixel)-STRIDED -> rows of pixels with a +64 B pad and
//     ds_read_b64_tr_b16 transposing reads (4 k-rows x 16 columns per 16-lane group).
// Epilogue (forward): accumulators -> bf16 -> LDS -> 16-byte coalesced row stores, with the
// optional residual add and the per-channel sum / sum-of-squares of BatchNorm folded in.
#include "common.h"

struct ConvArgs {
    const bf16_t* x;
    const bf16_t* w;
    bf16_t* y;
    const bf16_t* res;
    float* stats;
    int ldx, ldy, ldres;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, dil, mode;
    int M;
    int tiles_c, tiles_p;
};

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

template <int BC, int BP>
__global__ void __launch_bounds__(256, 2) conv_igemm_kernel(ConvArgs a) {
    constexpr int FI = BC / 64, FJ = BP / 64;
    constexpr int WL = BC / 32, XL = BP / 32;          // 16-byte loads per thread per tile
    constexpr int TILE = (BC + BP) * 128;               // bytes of one K tile (64 channels)
    constexpr int CSTR = BC * 2 + 16;                   // epilogue row stride (bytes)
    constexpr int EPI = BP * CSTR + 4 * BC * 2 * 4;
    constexpr int SMEM = (2 * TILE > EPI) ? 2 * TILE : EPI;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wc = wave & 1, wp = wave >> 1;
    const int logical = xcd_remap(blockIdx.x, a.tiles_c * a.tiles_p);
    const int c0 = (logical % a.tiles_c) * BC;
    const int m0 = (logical / a.tiles_c) * BP;
    const int taps = a.KH * a.KW;
    const int v = t & 7, r0 = t >> 3;

    // ---- per-thread pixel rows (fixed for the whole K loop)
    int xn[XL], xh[XL], xw[XL];
    bool xm[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        int m = m0 + r0 + 32 * i;
        xm[i] = m < a.M;
        int mm = xm[i] ? m : 0;
        int n = mm / (a.Ho * a.Wo), rem = mm % (a.Ho * a.Wo);
        int ho = rem / a.Wo, wo = rem % a.Wo;
        xn[i] = n * a.H * a.W;
        if (a.mode == 0) { xh[i] = ho * a.stride - a.pad; xw[i] = wo * a.stride - a.pad; }
        else             { xh[i] = ho + a.pad;            xw[i] = wo + a.pad; }
    }
    const bf16_t* wrow[WL];
    bool wm[WL];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        int co = c0 + r0 + 32 * i;
        wm[i] = co < a.Cout;
        wrow[i] = a.w + (size_t)(wm[i] ? co : 0) * taps * a.Cin + v * 8;
    }

    // ---- loader state
    int tap = 0, kh = 0, kw = 0, ci0 = 0;
    size_t xoff[XL];
    bool xok[XL];
    auto retap = [&]() {
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            int hi, wi;
            bool okh, okw;
            src_coord(a.mode, xh[i], kh, a.dil, a.stride, a.H, hi, okh);
            src_coord(a.mode, xw[i], kw, a.dil, a.stride, a.W, wi, okw);
            xok[i] = xm[i] && okh && okw;
            xoff[i] = xok[i] ? ((size_t)(xn[i] + hi * a.W + wi) * a.ldx + v * 8) : 0;
        }
    };
    retap();
    u16x8 wreg[WL], xreg[XL];
    auto gload = [&]() {
        const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < WL; ++i)
            wreg[i] = wm[i] ? *(const u16x8*)(wrow[i] + (size_t)tap * a.Cin + ci0) : zero;
#pragma unroll
        for (int i = 0; i < XL; ++i) xreg[i] = xok[i] ? *(const u16x8*)(a.x + xoff[i] + ci0) : zero;
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
    auto lstore = [&](int buf) {
        unsigned char* wb = smem + buf * TILE;
        unsigned char* xb = wb + BC * 128;
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            int r = r0 + 32 * i;
            *(u16x8*)(wb + r * 128 + ((v ^ ((r >> 1) & 7)) << 4)) = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            int r = r0 + 32 * i;
            *(u16x8*)(xb + r * 128 + ((v ^ ((r >> 1) & 7)) << 4)) = xreg[i];
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
    gload();
    lstore(0);
    __syncthreads();
    const int lrow = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) { advance(); gload(); }
        const unsigned char* wb = smem + (kt & 1) * TILE;
        const unsigned char* xb = wb + BC * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 af[FI], bfr[FJ];
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                int r = wc * (BC / 2) + i * 32 + lrow;
                af[i] = *(const bf16x8*)(wb + r * 128 + (((kk * 2 + lk) ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                int r = wp * (BP / 2) + j * 32 + lrow;
                bfr[j] = *(const bf16x8*)(xb + r * 128 + (((kk * 2 + lk) ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) lstore((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> bf16 C tile [pixel][cout] in LDS
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            int px = wp * (BP / 2) + j * 32 + lrow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int co = wc * (BC / 2) + i * 32 + 8 * g + 4 * lk;
                uint2 pk;
                pk.x = pack2bf(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1]);
                pk.y = pack2bf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                *(uint2*)(smem + px * CSTR + co * 2) = pk;
            }
        }
    __syncthreads();
    constexpr int VPR = BC / 8;              // 16-byte vectors per C row
    constexpr int RPP = 256 / VPR;           // rows per pass
    const int cv = t % VPR, rr = t / VPR;
    const int co = c0 + cv * 8;
    const bool cok = co < a.Cout;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
#pragma unroll 2
    for (int p = 0; p < BP / RPP; ++p) {
        int row = rr + p * RPP;
        int m = m0 + row;
        if (m < a.M && cok) {
            u16x8 val = *(const u16x8*)(smem + row * CSTR + cv * 16);
            if (a.res) {
                u16x8 rv = *(const u16x8*)(a.res + (size_t)m * a.ldres + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) val[e] = f2bf(bf2f(val[e]) + bf2f(rv[e]));
            }
            if (a.stats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { float f = bf2f(val[e]); s[e] += f; q[e] += f * f; }
            }
            *(u16x8*)(a.y + (size_t)m * a.ldy + co) = val;
        }
    }
    if (a.stats) {
        // lanes with equal cv inside a wave: strides VPR, 2*VPR, ... < 64
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = VPR; o < 64; o <<= 1) {
                s[e] += __shfl_xor(s[e], o, 64);
                q[e] += __shfl_xor(q[e], o, 64);
            }
        }
        float* red = (float*)(smem + BP * CSTR);     // [4 waves][2][BC]
        if (lane < VPR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 2 + 0) * BC + cv * 8 + e] = s[e];
                red[(wave * 2 + 1) * BC + cv * 8 + e] = q[e];
            }
        }
        __syncthreads();
        if (t < 2 * BC) {
            int which = t / BC, c = t % BC;
            float tot = red[(0 * 2 + which) * BC + c] + red[(1 * 2 + which) * BC + c] +
                        red[(2 * 2 + which) * BC + c] + red[(3 * 2 + which) * BC + c];
            if (c0 + c < a.Cout) atomicAdd(&a.stats[((blockIdx.x & (NREP - 1)) * 2 + which) * a.Cout + c0 + c], tot);
        }
    }
}

extern "C" int rgda_conv2d(const void* x, int ldx, const void* wgt, void* y, int ldy, const void* res, int ldres,
                           float* stats, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw,
                           int stride, int pad, int dil, int mode, rgda_stream_t stream) {
    if (!x || !wgt || !y) return RGDA_ERR_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 ||
        stride <= 0 || dil <= 0 || pad < 0 || (mode != 0 && mode != 1))
        return RGDA_ERR_ARG;
    if ((Cin & 63) || (Cout & 7) || (ldx & 7) || (ldy & 7) || (res && (ldres & 7)) || ldx < Cin || ldy < Cout)
        return RGDA_ERR_ARG;
    ConvArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)wgt; a.y = (bf16_t*)y; a.res = (const bf16_t*)res; a.stats = stats;
    a.ldx = ldx; a.ldy = ldy; a.ldres = ldres;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.KH = kh; a.KW = kw;
    a.stride = stride; a.pad = pad; a.dil = dil; a.mode = mode;
    long long M = (long long)N * Ho * Wo;
    if (M > 0x7fffffffLL) return RGDA_ERR_ARG;
    a.M = (int)M;
    hipStream_t st = to_stream(stream);
    // tile choice: fill 256 CUs (2 workgroups each); prefer the big tile when it still gives >= 512 groups
    int bc = (Cout <= 64) ? 64 : 128;
    long long big = (long long)cdiv(M, 128) * cdiv(Cout, bc);
    int bp = (big >= 384) ? 128 : 64;
    a.tiles_c = cdiv(Cout, bc);
    a.tiles_p = cdiv(M, bp);
    int grid = a.tiles_c * a.tiles_p;
    if (bc == 128 && bp == 128) conv_igemm_kernel<128, 128><<<grid, 256, 0, st>>>(a);
    else if (bc == 128 && bp == 64) conv_igemm_kernel<128, 64><<<grid, 256, 0, st>>>(a);
    else if (bc == 64 && bp == 128) conv_igemm_kernel<64, 128><<<grid, 256, 0, st>>>(a);
    else conv_igemm_kernel<64, 64><<<grid, 256, 0, st>>>(a);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
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
};

template <int BCO, int BCI>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(WgradArgs a) {
    constexpr int FI = BCO / 64, FJ = BCI / 64;
    constexpr int SA = BCO * 2 + 64, SB = BCI * 2 + 64;   // LDS row strides (bytes): +64 B pad
    constexpr int TILE = 64 * (SA + SB);
    constexpr int VA = BCO / 8, VB = BCI / 8;             // 16-byte vectors per row
    constexpr int LA = 64 * VA / 256, LB = 64 * VB / 256; // loads per thread per tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int taps = a.KH * a.KW;
    int bid = blockIdx.x;
    const int split = bid % a.splits; bid /= a.splits;
    const int tco = bid % a.tiles_co; bid /= a.tiles_co;
    const int tci = bid % a.tiles_ci;
    const int tap = bid / a.tiles_ci;
    const int kh = tap / a.KW, kw = tap % a.KW;
    const int co0 = tco * BCO, ci0 = tci * BCI;
    const int kt_beg = split * a.kt_per_split;
    const int kt_end = min(kt_beg + a.kt_per_split, (a.M + 63) >> 6);
    const bool pointwise = (taps == 1 && a.stride == 1 && a.pad == 0);

    const int va = t % VA, ra = t / VA;     // dY tile: rows ra + (256/VA)*i
    const int vb = t % VB, rb = t / VB;
    const bool aok = (co0 + va * 8) < a.Cout;
    const bool bok = (ci0 + vb * 8) < a.Cin;

    u16x8 areg[LA], breg[LB];
    auto gload = [&](int kt) {
        const u16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        const int mb = kt << 6;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            int m = mb + ra + (256 / VA) * i;
            areg[i] = (aok && m < a.M) ? *(const u16x8*)(a.dy + (size_t)m * a.lddy + co0 + va * 8) : zero;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            int m = mb + rb + (256 / VB) * i;
            bool ok = bok && m < a.M;
            size_t src = (size_t)m;
            if (!pointwise && ok) {
                int n, rem, ho, wo;
                if (a.howo_shift >= 0) { n = m >> a.howo_shift; rem = m & ((1 << a.howo_shift) - 1); }
                else { n = m / (a.Ho * a.Wo); rem = m % (a.Ho * a.Wo); }
                if (a.wo_shift >= 0) { ho = rem >> a.wo_shift; wo = rem & ((1 << a.wo_shift) - 1); }
                else { ho = rem / a.Wo; wo = rem % a.Wo; }
                int hi = ho * a.stride - a.pad + kh * a.dil, wi2 = wo * a.stride - a.pad + kw * a.dil;
                ok = hi >= 0 && hi < a.H && wi2 >= 0 && wi2 < a.W;
                src = (size_t)(n * a.H + hi) * a.W + wi2;
            }
            breg[i] = ok ? *(const u16x8*)(a.x + src * a.ldx + ci0 + vb * 8) : zero;
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* ab = smem + buf * TILE;
        unsigned char* bb = ab + 64 * SA;
#pragma unroll
        for (int i = 0; i < LA; ++i) *(u16x8*)(ab + (ra + (256 / VA) * i) * SA + va * 16) = areg[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *(u16x8*)(bb + (rb + (256 / VB) * i) * SB + vb * 16) = breg[i];
    };

    f32x16 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt_beg < kt_end) {
        gload(kt_beg);
        lstore(0);
        __syncthreads();
        // transposing-read lane geometry: 16-lane group g reads a [4 k][16 col] block
        const int g = lane >> 4, la = lane & 15;
        const int krow = (g >> 1) * 8 + (la >> 2);
        const int kcol = (g & 1) * 16 + (la & 3) * 4;
        for (int kt = kt_beg; kt < kt_end; ++kt) {
            const bool more = kt + 1 < kt_end;
            if (more) gload(kt + 1);
            const int buf = (kt - kt_beg) & 1;
            const unsigned char* ab = smem + buf * TILE;
            const unsigned char* bb = ab + 64 * SA;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 af[FI], bfr[FJ];
#pragma unroll
                for (int i = 0; i < FI; ++i) {
                    const unsigned char* p = ab + (kk * 16 + krow) * SA + (wi * (BCO / 2) + i * 32 + kcol) * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * SA));
                    u16x8 v8 = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                                (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
                    af[i] = __builtin_bit_cast(bf16x8, v8);
                }
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    const unsigned char* p = bb + (kk * 16 + krow) * SB + (wj * (BCI / 2) + j * 32 + kcol) * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * SB));
                    u16x8 v8 = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                                (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
                    bfr[j] = __builtin_bit_cast(bf16x8, v8);
                }
#pragma unroll
                for (int i = 0; i < FI; ++i)
#pragma unroll
                    for (int j = 0; j < FJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            if (more) lstore(buf ^ 1);
            __syncthreads();
        }
    }
    // ---- split-K reduction: fp32 atomics straight into the gradient buffer (128 B per half-wave)
    const int lcol = lane & 31, lk = lane >> 5;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            int ci = ci0 + wj * (BCI / 2) + j * 32 + lcol;
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = co0 + wi * (BCO / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (co < a.Cout) atomicAdd(a.dw + ((size_t)co * taps + tap) * a.Cin + ci, acc[i][j][r]);
            }
        }
}

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

extern "C" int rgda_conv2d_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, int N, int H, int W,
                                 int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil,
                                 rgda_stream_t stream) {
    if (!x || !dy || !dw) return RGDA_ERR_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 ||
        stride <= 0 || dil <= 0 || pad < 0)
        return RGDA_ERR_ARG;
    if ((Cin & 7) || (Cout & 7) || (ldx & 7) || (lddy & 7) || ldx < Cin || lddy < Cout) return RGDA_ERR_ARG;
    WgradArgs a;
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.dw = dw; a.ldx = ldx; a.lddy = lddy;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.KH = kh; a.KW = kw;
    a.stride = stride; a.pad = pad; a.dil = dil;
    long long M = (long long)N * Ho * Wo;
    if (M > 0x7fffffffLL - 64) return RGDA_ERR_ARG;
    a.M = (int)M;
    a.howo_shift = ilog2_exact(Ho * Wo);
    a.wo_shift = ilog2_exact(Wo);
    int bco = (Cout <= 64) ? 64 : 128, bci = (Cin <= 64) ? 64 : 128;
    a.tiles_co = cdiv(Cout, bco);
    a.tiles_ci = cdiv(Cin, bci);
    int tiles = a.tiles_co * a.tiles_ci * kh * kw;
    int KT = cdiv(M, 64);
    int splits = cdiv(1024, tiles);
    if (splits > KT / 4) splits = KT / 4;
    if (splits < 1) splits = 1;
    a.kt_per_split = cdiv(KT, splits);
    a.splits = cdiv(KT, a.kt_per_split);
    int grid = tiles * a.splits;
    hipStream_t st = to_stream(stream);
    if (bco == 128 && bci == 128) conv_wgrad_kernel<128, 128><<<grid, 256, 0, st>>>(a);
    else if (bco == 128 && bci == 64) conv_wgrad_kernel<128, 64><<<grid, 256, 0, st>>>(a);
    else if (bco == 64 && bci == 128) conv_wgrad_kernel<64, 128><<<grid, 256, 0, st>>>(a);
    else conv_wgrad_kernel<64, 64><<<grid, 256, 0, st>>>(a);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}
