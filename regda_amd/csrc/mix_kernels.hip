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
// Workgroup = (group g, chunk of 64 channel vectors): the group's J x 512-channel input slab is staged in LDS once;
// wave (8 per workgroup) = a slice of the output rows (two rows per pass share every LDS read), lane = channel vector.  The weights of a
// wave's rows are wave-uniform: they come through the scalar cache (s_load), not through LDS or VGPRs.
// Deterministic (no atomics, fixed summation order).
template <typename TI, typename TO, int CV>
__global__ void __launch_bounds__(512) group_mix_kernel(const TI* __restrict__ in, int ldin, const float* __restrict__ W,
                                                        TO* __restrict__ out, int ldout, int I, int J, int C) {
    static_assert(CV == 64, "one wave per output-row slice");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int NT = 512, SL = NT / CV;                        // 8 waves: enough of them per SIMD to hide the LDS latency
    TI* s_in = (TI*)smem_raw;                                   // [J][CV*8]
    const int g = blockIdx.x, cv0 = blockIdx.y * CV;
    const int ncv = min(CV, C / 8 - cv0);
    constexpr int VEC = 16 / sizeof(TI);                         // elements per 16-byte transfer
    const int per_row = ncv * 8 / VEC;
    for (int t = threadIdx.x; t < J * per_row; t += NT) {
        const int j = t / per_row, v = t % per_row;
        *(uint4*)(s_in + (size_t)j * CV * 8 + v * VEC) = *(const uint4*)(in + ((size_t)g * J + j) * ldin + cv0 * 8 + v * VEC);
    }
    __syncthreads();
    const int cvl = threadIdx.x % CV;
    const int sl = __builtin_amdgcn_readfirstlane(threadIdx.x / CV);
    if (cvl >= ncv) return;
    const TI* col = s_in + cvl * 8;
    for (int i0 = 2 * sl; i0 < I; i0 += 2 * SL) {
        const bool two = i0 + 1 < I;
        const float* w0 = W + (size_t)i0 * J;
        const float* w1 = w0 + (two ? J : 0);
        float a0[8] = {0}, a1[8] = {0};
        int j = 0;
        for (; j + 4 <= J; j += 4) {
            float f[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) ld8(col + (size_t)(j + u) * CV * 8, f[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float u0 = w0[j + u], u1 = w1[j + u];
#pragma unroll
                for (int e = 0; e < 8; ++e) { a0[e] += u0 * f[u][e]; a1[e] += u1 * f[u][e]; }
            }
        }
        for (; j < J; ++j) {
            float f[8];
            ld8(col + (size_t)j * CV * 8, f);
            const float u0 = w0[j], u1 = w1[j];
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
    const size_t lds = (size_t)J * CV * 8 * sizeof(TI);
    if (lds > 150 * 1024) return RGDA_ERR_UNSUPPORTED;
    auto kern = group_mix_kernel<TI, TO, CV>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RGDA_ERR_LAUNCH;
    kern<<<dim3(G, cdiv(C / 8, CV)), 512, lds, st>>>((const TI*)in, ldin, W, (TO*)out, ldout, I, J, C);
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

// ------------------------------------------------------------------ sparse (CSR) map per image, <= 4 sources / outputs
// out[n][i][c] = sum_{k in row i} vals[k] * ins[src(k)][n][col(k)][c],   cols[k] = (src << 24) | col.
// The I rows may be split over up to four output tensors (rows [first[q], first[q+1]) -> outs[q], image-major with
// first[q+1] - first[q] rows per image): the four scales' gradients of one head leave in one launch.
// Wave = one output row (64 channel vectors per pass), four rows per workgroup; the row's (col, val) list is read
// wave-uniformly and up to eight row loads are kept in flight.  Deterministic.
struct SparseSrc { const void* in[4]; int ldin[4]; int J[4]; };
struct SparseDst { void* out[4]; int ldout[4]; int first[5]; };

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) sparse_mix_kernel(SparseSrc S, const int* __restrict__ rowptr,
                                                         const int* __restrict__ cols, const float* __restrict__ vals,
                                                         SparseDst D, int I, int C) {
    const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), n = blockIdx.y;
    if (i >= I) return;
    int d = 0;
    while (d < 3 && i >= D.first[d + 1]) ++d;
    const int Id = D.first[d + 1] - D.first[d];
    TO* orow = (TO*)D.out[d] + ((size_t)n * Id + (i - D.first[d])) * D.ldout[d];
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    auto src = [&](int cj, int cv) {
        const int q = cj >> 24, j = cj & 0xffffff;
        return (const TI*)S.in[q] + ((size_t)n * S.J[q] + j) * S.ldin[q] + cv * 8;
    };
    for (int cv = blockIdx.z * 64 + (threadIdx.x & 63); cv < C / 8; cv += 64 * gridDim.z) {
        float acc[8] = {0};
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float f[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ld8(src(cols[k + u], cv), f[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float v = vals[k + u];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v * f[u][e];
            }
        }
        for (; k + 4 <= k1; k += 4) {
            float f[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) ld8(src(cols[k + u], cv), f[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = vals[k + u];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v * f[u][e];
            }
        }
        for (; k < k1; ++k) {
            float f[8];
            ld8(src(cols[k], cv), f);
            const float v = vals[k];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v * f[e];
        }
        st8(orow + cv * 8, acc);
    }
}

extern "C" int rgda_sparse_mix(int nsrc, const void* const* ins, const int* ldins, const int* Js, int in_f32,
                               const int* rowptr, const int* cols, const float* vals, int ndst, void* const* outs,
                               const int* ldouts, const int* out_rows, int out_f32, int N, int C,
                               rgda_stream_t stream) {
    if (nsrc < 1 || nsrc > 4 || ndst < 1 || ndst > 4 || !ins || !ldins || !Js || !rowptr || !cols || !vals || !outs ||
        !ldouts || !out_rows || N <= 0 || C <= 0 || (C & 7))
        return RGDA_ERR_ARG;
    SparseSrc S;
    for (int q = 0; q < 4; ++q) {
        const int p = q < nsrc ? q : 0;
        if (!ins[p] || Js[p] <= 0 || Js[p] >= (1 << 24) || (ldins[p] & 7)) return RGDA_ERR_ARG;
        S.in[q] = ins[p]; S.ldin[q] = ldins[p]; S.J[q] = Js[p];
    }
    SparseDst D;
    D.first[0] = 0;
    for (int q = 0; q < 4; ++q) {
        const int p = q < ndst ? q : ndst - 1;
        if (!outs[p] || out_rows[p] <= 0 || (ldouts[p] & 7)) return RGDA_ERR_ARG;
        D.out[q] = outs[p]; D.ldout[q] = ldouts[p];
        D.first[q + 1] = D.first[q] + (q < ndst ? out_rows[q] : 0);
    }
    const int I = D.first[4];
    dim3 grid(cdiv(I, 4), N, cdiv(C / 8, 64));
    hipStream_t st = to_stream(stream);
    if (in_f32 && !out_f32) sparse_mix_kernel<float, bf16_t><<<grid, 256, 0, st>>>(S, rowptr, cols, vals, D, I, C);
    else if (!in_f32 && out_f32) sparse_mix_kernel<bf16_t, float><<<grid, 256, 0, st>>>(S, rowptr, cols, vals, D, I, C);
    else if (!in_f32 && !out_f32) sparse_mix_kernel<bf16_t, bf16_t><<<grid, 256, 0, st>>>(S, rowptr, cols, vals, D, I, C);
    else sparse_mix_kernel<float, float><<<grid, 256, 0, st>>>(S, rowptr, cols, vals, D, I, C);
    return hipGetLastError() == hipSuccess ? RGDA_OK : RGDA_ERR_LAUNCH;
}
