// Stage-2 ("align") kernels, SURVEY.md 8f rank 2: PrototypeContrastiveLoss of regda/loss.py:10-47 -- forward and the
// gradient w.r.t. the features in one pass over the feature map.
#include "common.h"
#include <math.h>

// ws layout: float pn[C][K] (L2-normalised prototypes) | int count (valid pixels) | pad
// one workgroup per class: pn[c] = protos[c] / max(||protos[c]||, 1e-12)   (tnf.normalize, loss.py:41)
__global__ void __launch_bounds__(256) pcl_prep_kernel(const float* __restrict__ protos, float* __restrict__ pn, int K) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const float* p = protos + (size_t)c * K;
    float q = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) q += p[k] * p[k];
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
    for (int k = threadIdx.x; k < K; k += 256) pn[(size_t)c * K + k] = p[k] / nrm;
}

__global__ void __launch_bounds__(256) pcl_count_kernel(const int64_t* __restrict__ labels, long long n, int ignore_label,
                                                        int C, int* count, int* flag) {
    int c = 0, bad = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long l = labels[i];
        if (l == ignore_label) continue;
        if (l < 0 || l >= C) { bad = 1; continue; }
        ++c;
    }
    c = (int)wave_sum((float)c);            // <= 64 * (n / grid) per wave: exact in fp32 for the maps we see (< 2^24)
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
    if (bad) atomicOr(flag, 4);
}

// Workgroup = PX pixels x SL k-slices.  Pass 1: ||f||^2 and the C dot products with the normalised prototypes
// (prototypes in LDS); per pixel: logits z_c = d_c / (max(||f||, 1e-12) * T), CE against the label, and the
// coefficients of dL/df = sum_c alpha_c * pn_c - beta * f.  Pass 2 re-reads the pixel's features (L2), forms the
// gradient 128 channels at a time, transposes it through LDS and stores pixel-major bf16 rows (the layout the
// instance-norm backward consumes) with 16-byte vectors.
template <int C, int PX, int SL>
__global__ void __launch_bounds__(PX * SL) pcl_kernel(const float* __restrict__ feat, const int64_t* __restrict__ labels,
                                                      const float* __restrict__ pn, const int* __restrict__ count,
                                                      rgda_stat_t* loss_acc, bf16_t* __restrict__ dfeat, int lddf, int accumulate,
                                                      int K, int hw, int ignore_label, float inv_temp, float weight) {
    extern __shared__ float lds[];          // pn[C][K] | red[SL][PX][C+1] | coef[PX][C+1] | tile (bf16 [PX][KC+8])
    constexpr int KC = 128;                 // channels per transposed store chunk
    float* lpn = lds;
    float* red = lds + (size_t)C * K;
    float* coef = red + SL * PX * (C + 1);
    bf16_t* tile = (bf16_t*)(coef + PX * (C + 1));
    const int b = blockIdx.y;
    const int lane = threadIdx.x % PX, slice = threadIdx.x / PX;
    const int p = blockIdx.x * PX + lane;
    const bool ok = p < hw;
    for (int i = threadIdx.x; i < C * K; i += PX * SL) lpn[i] = pn[i];
    const float* f = feat + (size_t)b * K * hw + (ok ? p : 0);
    const int kper = (K + SL - 1) / SL;
    const int k0 = min(slice * kper, K), k1 = min(k0 + kper, K);
    __syncthreads();
    float q = 0.f, d[C];
#pragma unroll
    for (int c = 0; c < C; ++c) d[c] = 0.f;
#pragma unroll 4
    for (int k = k0; k < k1; ++k) {
        const float v = f[(size_t)k * hw];
        q += v * v;
#pragma unroll
        for (int c = 0; c < C; ++c) d[c] += v * lpn[c * K + k];
    }
    float* r = red + (slice * PX + lane) * (C + 1);
    r[0] = q;
#pragma unroll
    for (int c = 0; c < C; ++c) r[1 + c] = d[c];
    __syncthreads();
    const int n = *count;
    if (slice == 0) {
        float* cf = coef + lane * (C + 1);
        const long long lab = ok ? labels[(size_t)b * hw + p] : (long long)ignore_label;
        const bool valid = ok && lab != ignore_label && lab >= 0 && lab < C;
        float contrib = 0.f;
#pragma unroll
        for (int c = 0; c <= C; ++c) cf[c] = 0.f;
        if (valid) {
            float qq = 0.f, dd[C];
            for (int s = 0; s < SL; ++s) qq += red[(s * PX + lane) * (C + 1)];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float a = 0.f;
                for (int s = 0; s < SL; ++s) a += red[(s * PX + lane) * (C + 1) + 1 + c];
                dd[c] = a;
            }
            const float nrm = fmaxf(sqrtf(qq), 1e-12f);
            float z[C], zmax = -INFINITY;
#pragma unroll
            for (int c = 0; c < C; ++c) { z[c] = dd[c] / nrm * inv_temp; zmax = fmaxf(zmax, z[c]); }
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) se += expf(z[c] - zmax);
            const float lse = zmax + logf(se);
            const float scale = weight / (float)n;               // mean over the valid pixels (nn.CrossEntropyLoss)
            float gd = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float g = (expf(z[c] - lse) - ((int)lab == c ? 1.f : 0.f)) * scale;       // dL/dz_c
                cf[1 + c] = g * inv_temp / nrm;                   // alpha_c
                gd += g * dd[c];
                if ((int)lab == c) contrib = (lse - z[c]) * scale;
            }
            cf[0] = gd * inv_temp / (nrm * nrm * nrm);            // beta
        }
        // loss: one fixed-point integer atomic per wave (order-independent total, common.h: stat_add)
        float tot = contrib;
        for (int o = PX / 2; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
        if (lane == 0 && tot != 0.f) stat_add(loss_acc, tot, RGDA_STAT_FRAC_BWD);
    }
    __syncthreads();
    if (!dfeat) return;
    // ---- pass 2: gradient, KC channels per round; thread = (pixel lane, KC/SL consecutive channels)
    constexpr int PER = KC / SL;
    constexpr int TS = KC + 8;              // padded tile row (bf16 elements)
    const float* cf = coef + lane * (C + 1);
    float alpha[C];
#pragma unroll
    for (int c = 0; c < C; ++c) alpha[c] = cf[1 + c];
    const float beta = cf[0];
    for (int kc = 0; kc < K; kc += KC) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int k = kc + slice * PER + j;
            float gk = 0.f;
            if (k < K) {
                const float v = f[(size_t)k * hw];
                gk = -beta * v;
#pragma unroll
                for (int c = 0; c < C; ++c) gk += alpha[c] * lpn[c * K + k];
            }
            tile[lane * TS + slice * PER + j] = f2bf(gk);
        }
        __syncthreads();
        // PX rows x KC/8 vectors
        for (int v = threadIdx.x; v < PX * (KC / 8); v += PX * SL) {
            const int row = v / (KC / 8), cv = v % (KC / 8);
            const int pp = blockIdx.x * PX + row, k = kc + cv * 8;
            if (pp < hw && k < K) {
                u16x8 val = *(const u16x8*)(tile + row * TS + cv * 8);
                bf16_t* dst = dfeat + ((size_t)b * hw + pp) * lddf + k;
                if (accumulate) {
                    u16x8 old = *(const u16x8*)dst;
#pragma unroll
                    for (int e = 0; e < 8; ++e) val[e] = f2bf(bf2f(val[e]) + bf2f(old[e]));
                }
                *(u16x8*)dst = val;
            }
        }
        __syncthreads();
    }
}

// *loss += the accumulated total; no pixel kept: nn.CrossEntropyLoss averages over zero elements -> NaN loss (and zero
// gradients), like the reference
__global__ void pcl_loss_finish_kernel(const rgda_stat_t* acc, const int* count, float* loss) {
    *loss += (*count == 0) ? __builtin_nanf("") : (float)((double)*acc * (1.0 / (double)(1ll << RGDA_STAT_FRAC_BWD)));
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t rgda_pcl_loss_workspace(int C, int K) {
    return align256((size_t)C * K * 4) + 256;
}

extern "C" int rgda_pcl_loss(const float* feat, const int64_t* labels, const float* protos, float* loss, void* dfeat,
                             int lddf, int accumulate, int b, int K, int C, int h, int w, int ignore_label,
                             float temperature, float weight, void* ws, size_t ws_bytes, rgda_stream_t stream) {
    if (!feat || !labels || !protos || !loss || !ws) return RGDA_ERR_ARG;
    if (C != 6) return RGDA_ERR_UNSUPPORTED;       // ISPRS: 6 classes (regda/datasets/isprsda.py:18-26)
    if (b <= 0 || K < 8 || K > 4096 || (K & 7) || h <= 0 || w <= 0 || !(temperature > 0.f)) return RGDA_ERR_ARG;
    if (dfeat && ((lddf & 7) || lddf < K)) return RGDA_ERR_ARG;
    if (ws_bytes < rgda_pcl_loss_workspace(C, K)) return RGDA_ERR_WORKSPACE;
    hipStream_t st = to_stream(stream);
    float* pn = (float*)ws;
    int* count = (int*)((char*)ws + align256((size_t)C * K * 4));
    int* flag = count + 1;
    rgda_stat_t* lacc = (rgda_stat_t*)(count + 2);           // the loss total of this call, fixed point
    if (hipMemsetAsync(count, 0, 16, st) != hipSuccess) return RGDA_ERR_LAUNCH;
    pcl_prep_kernel<<<C, 256, 0, st>>>(protos, pn, K);
    RGDA_CHECK_LAUNCH();
    const long long n = (long long)b * h * w;
    long long g = (n + 255) / 256;
    pcl_count_kernel<<<(int)(g > 1024 ? 1024 : g), 256, 0, st>>>(labels, n, ignore_label, C, count, flag);
    RGDA_CHECK_LAUNCH();
    constexpr int PX = 32, SL = 16;
    const size_t lds = ((size_t)C * K + SL * PX * (C + 1) + PX * (C + 1)) * 4 + (size_t)PX * (128 + 8) * 2;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)pcl_kernel<6, PX, SL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RGDA_ERR_LAUNCH;
    dim3 grid(cdiv(h * w, PX), b);
    pcl_kernel<6, PX, SL><<<grid, PX * SL, lds, st>>>(feat, labels, pn, count, lacc, (bf16_t*)dfeat, lddf, accumulate, K, h * w,
                                                      ignore_label, 1.f / temperature, weight);
    RGDA_CHECK_LAUNCH();
    pcl_loss_finish_kernel<<<1, 1, 0, st>>>(lacc, count, loss);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}
