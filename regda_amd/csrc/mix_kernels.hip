// Factored spatial maps of the PPM heads.
//
// The head's 3x3 conv applied to the bilinearly upsampled s x s branch maps is  V @ Z  with
// V[(y,x)][(jy,jx),(ky,kx)] = Uy[y+ky-1][jy] * Ux[x+kx-1][jx]   (Encoder.py:30-51 re-associated, see ppm_tap_matrix).
// Applying V (forward) or V^T (backward) as one sparse map costs ~117 row gathers per pixel, all four scales
// together ~100 K gathers of 1 KB rows per image: L2-bandwidth-bound (340 us per head and pass).  V is separable,
// so both directions run as two short maps through a small intermediate (36 rows per image row: one per
// (scale, jx, kx)):
//   forward   B[(n,y)][r]  = sum_{jy,ky} Uy[y+ky-1][jy] Z_s[n][(jy,jx),(ky,kx)]     rgda_sparse_mix (<= 6 gathers / row)
//             out[(n,y)][x] = sum_r Wx[r][x] B[(n,y)][r]                             rgda_group_mix  (rows staged in LDS)
//   backward  A[(n,y)][r]  = sum_x Wx[r][x] dc[(n,y)][x]                             rgda_group_mix
//             dZ_s[n][..]  = sum_y Uy[y+ky-1][jy] A[(n,y)][r]                        rgda_sparse_mix (<= h gathers / row)
// which is ~4x fewer row gathers, most of them from LDS.  The intermediates are fp32.
#include "common.h"

static __device__ __forceinline__ void ld8(const bf16_t* p, float (&f)[8]) {
    const u16x8 v = *(const u16x8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = bf2f(v[e]);
}
static __device__ __forceinline__ void ld8(const float* p, float (&f)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
static __device__ __forceinline__ void st8(bf16_t* p, const float (&f)[8]) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    *(uint4*)p = v;
}
static __device__ __forceinline__ void st8(float* p, const float (&f)[8]) {
    *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
    *(float4*)(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}

// ------------------------------------------------------------------ dense map inside small row groups
// out[g][i][c] = sum_j W[i][j] * in[g][j][c]   for G groups of J consecutive rows (W f32 [I][J], shared by all groups).
// Workgroup = (group g, chunk of CV channel vectors): the group's J x (CV*8) input slab and W are staged in LDS once;
// thread = (channel vector, slice of the output rows), two output rows per pass share every LDS read; zero weights
// are skipped.  Deterministic (no atomics, fixed summation order).
template <typename TI, typename TO, int CV>
__global__ void __launch_bounds__(256) group_mix_kernel(const TI* __restrict__ in, int ldin, const float* __restrict__ W,
                                                        TO* __restrict__ out, int ldout, int I, int J, int C) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int SL = 256 / CV;
    TI* s_in = (TI*)smem_raw;                                   // [J][CV*8]
    float* s_w = (float*)(smem_raw + (size_t)J * CV * 8 * sizeof(TI));    // [I][J]
    const int g = blockIdx.x, cv0 = blockIdx.y * CV;
    const int ncv = min(CV, C / 8 - cv0);
    for (int t = threadIdx.x; t < I * J; t += 256) s_w[t] = W[t];
    constexpr int VEC = 16 / sizeof(TI);                         // elements per 16-byte transfer
    const int per_row = ncv * 8 / VEC;
    for (int t = threadIdx.x; t < J * per_row; t += 256) {
        const int j = t / per_row, v = t % per_row;
        *(uint4*)(s_in + (size_t)j * CV * 8 + v * VEC) = *(const uint4*)(in + ((size_t)g * J + j) * ldin + cv0 * 8 + v * VEC);
    }
    __syncthreads();
    const int cvl = threadIdx.x % CV, sl = threadIdx.x / CV;
    if (cvl >= ncv) return;
    for (int i0 = 2 * sl; i0 < I; i0 += 2 * SL) {
        const bool two = i0 + 1 < I;
        const float* w0 = s_w + (size_t)i0 * J;
        const float* w1 = w0 + (two ? J : 0);
        float a0[8] = {0}, a1[8] = {0};
        for (int j = 0; j < J; ++j) {
            const float u0 = w0[j], u1 = w1[j];
            if (u0 == 0.f && u1 == 0.f) continue;
            float f[8];
            ld8(s_in + (size_t)j * CV * 8 + cvl * 8, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0[e] += u0 * f[e]; a1[e] += u1 * f[e]; }
        }
        st8(out + ((size_t)g * I + i0) * ldout + (cv0 + cvl) * 8, a0);
        if (two) st8(out + ((size_t)g * I + i0 + 1) * ldout + (cv0 + cvl) * 8, a1);
    }
}

template <typename TI, typename TO>
static int launch_group_mix(const void* in, int ldin, const float* W, void* out, int ldout, int G, int I, int J, int C,
                            hipStream_t st) {
    constexpr int CV = 64;
    const size_t lds = (size_t)J * CV * 8 * sizeof(TI) + (size_t)I * J * 4;
    if (lds > 150 * 1024) return RGDA_ERR_UNSUPPORTED;
    auto kern = group_mix_kernel<TI, TO, CV>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RGDA_ERR_LAUNCH;
    kern<<<dim3(G, cdiv(C / 8, CV)), 256, lds, st>>>((const TI*)in, ldin, W, (TO*)out, ldout, I, J, C);
    return hipGetLastError() == hipSuccess ? RGDA_OK : RGDA_ERR_LAUNCH;
}

extern "C" int rgda_group_mix(const void* in, int ldin, int in_f32, const float* W, void* out, int ldout, int out_f32,
                              int G, int I, int J, int C, rgda_stream_t stream) {
    if (!in || !W || !out || G <= 0 || I <= 0 || J <= 0 || C <= 0 || (C & 7) || (ldin & 7) || (ldout & 7)) return RGDA_ERR_ARG;
    hipStream_t st = to_stream(stream);
    if (in_f32 && !out_f32) return launch_group_mix<float, bf16_t>(in, ldin, W, out, ldout, G, I, J, C, st);
    if (!in_f32 && out_f32) return launch_group_mix<bf16_t, float>(in, ldin, W, out, ldout, G, I, J, C, st);
    if (!in_f32 && !out_f32) return launch_group_mix<bf16_t, bf16_t>(in, ldin, W, out, ldout, G, I, J, C, st);
    return launch_group_mix<float, float>(in, ldin, W, out, ldout, G, I, J, C, st);
}

// ------------------------------------------------------------------ sparse (CSR) map per image, up to 4 sources
// out[n][i][c] = sum_{k in row i} vals[k] * ins[src(k)][n][col(k)][c],   cols[k] = (src << 24) | col.
// Wave = one output row (64 channel vectors per pass), four rows per workgroup; the row's (col, val) list is read
// wave-uniformly.  Deterministic.
struct SparseSrc { const void* in[4]; int ldin[4]; int J[4]; };

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) sparse_mix_kernel(SparseSrc S, const int* __restrict__ rowptr,
                                                         const int* __restrict__ cols, const float* __restrict__ vals,
                                                         TO* __restrict__ out, int ldout, int I, int C) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), n = blockIdx.y;
    if (i >= I) return;
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    for (int cv = threadIdx.x & 63; cv < C / 8; cv += 64) {
        float acc[8] = {0};
        for (int k = k0; k < k1; ++k) {
            const int cj = cols[k];
            const int q = cj >> 24, j = cj & 0xffffff;
            const float v = vals[k];
            float f[8];
            ld8((const TI*)S.in[q] + ((size_t)n * S.J[q] + j) * S.ldin[q] + cv * 8, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v * f[e];
        }
        st8(out + ((size_t)n * I + i) * ldout + cv * 8, acc);
    }
}

extern "C" int rgda_sparse_mix(int nsrc, const void* const* ins, const int* ldins, const int* Js, int in_f32,
                               const int* rowptr, const int* cols, const float* vals, void* out, int ldout, int out_f32,
                               int N, int I, int C, rgda_stream_t stream) {
    if (nsrc < 1 || nsrc > 4 || !ins || !ldins || !Js || !rowptr || !cols || !vals || !out || N <= 0 || I <= 0 ||
        C <= 0 || (C & 7) || (ldout & 7))
        return RGDA_ERR_ARG;
    SparseSrc S;
    for (int q = 0; q < 4; ++q) {
        const int p = q < nsrc ? q : 0;
        if (!ins[p] || Js[p] <= 0 || Js[p] >= (1 << 24) || (ldins[p] & 7)) return RGDA_ERR_ARG;
        S.in[q] = ins[p]; S.ldin[q] = ldins[p]; S.J[q] = Js[p];
    }
    dim3 grid(cdiv(I, 4), N);
    hipStream_t st = to_stream(stream);
    if (in_f32 && !out_f32) sparse_mix_kernel<float, bf16_t><<<grid, 256, 0, st>>>(S, rowptr, cols, vals, (bf16_t*)out, ldout, I, C);
    else if (!in_f32 && out_f32) sparse_mix_kernel<bf16_t, float><<<grid, 256, 0, st>>>(S, rowptr, cols, vals, (float*)out, ldout, I, C);
    else if (!in_f32 && !out_f32) sparse_mix_kernel<bf16_t, bf16_t><<<grid, 256, 0, st>>>(S, rowptr, cols, vals, (bf16_t*)out, ldout, I, C);
    else sparse_mix_kernel<float, float><<<grid, 256, 0, st>>>(S, rowptr, cols, vals, (float*)out, ldout, I, C);
    return hipGetLastError() == hipSuccess ? RGDA_OK : RGDA_ERR_LAUNCH;
}
