// ASPP head, Classifier_Module (regda/models/Encoder.py:68-84): out = sum over four dilations d of
// Conv2d(K -> classes, 3x3, padding = dilation = d, bias)(x).
//
// MI355X form: the two heads' eight convolutions share the input, so all their taps are ONE 1x1 convolution on the
// matrix cores,  Z = x @ Wstack^T  with one output column per (head, dilation, class, tap)  (2*4*C*9 columns, padded
// to a multiple of 64; column order = the reference's own weight layout [C][3][3] per conv, so the stacked filter
// is the eight master weights back to back and the weight gradient lands in that layout too).  What is left of the
// dilated 3x3 structure is a shifted gather of Z (forward) and the mirrored scatter of the logit gradient (backward),
// the two small kernels below.  A direct dilated conv with 6 output channels would waste > 90 % of a 64-wide tile.
#include "common.h"

struct AsppPtrs {
    const float* bias[8];    // [head][dilation] -> f32 [C]
    float* dbias[8];
    int dil[4];
};

// column of Z for (head, dilation index, class, tap)
static __device__ __forceinline__ int zcol(int head, int d, int c, int tap, int C) { return ((head * 4 + d) * C + c) * 9 + tap; }

// out[head][n][c][y][x] = sum_d bias_d[c] + sum_d sum_tap Z[(n, y + dy*d, x + dx*d)][zcol]      (zero outside the map)
__global__ void __launch_bounds__(256) aspp_gather_kernel(const bf16_t* __restrict__ z, int ldz, AsppPtrs P,
                                                          float* __restrict__ out1, float* __restrict__ out2, int N,
                                                          int h, int w, int C) {
    const int head = blockIdx.y;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * h * w * C) return;
    const int c = (int)(idx % C);
    const long long m = idx / C;
    const int x = (int)(m % w), y = (int)((m / w) % h), n = (int)(m / ((long long)w * h));
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        acc += P.bias[head * 4 + d][c];
        const int dl = P.dil[d];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + (tap / 3 - 1) * dl, xx = x + (tap % 3 - 1) * dl;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w)
                acc += bf2f(z[((long long)(n * h + yy) * w + xx) * ldz + zcol(head, d, c, tap, C)]);
        }
    }
    float* out = head ? out2 : out1;
    out[((long long)(n * C + c) * h + y) * w + x] = acc;
}

// dZ[(n,y,x)][zcol(head,d,c,tap)] = g_head[n][c][y - dy*d][x - dx*d]   (0 outside the map); pad columns zeroed.
__global__ void __launch_bounds__(256) aspp_scatter_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                           bf16_t* __restrict__ dz, int lddz, int zc, AsppPtrs P, int N,
                                                           int h, int w, int C) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int per_px = 8 * C;                                  // (head, d, c) triples
    if (idx >= (long long)N * h * w * per_px) return;
    const int t = (int)(idx % per_px);
    const long long m = idx / per_px;
    const int c = t % C, d = (t / C) & 3, head = t / (4 * C);
    const int x = (int)(m % w), y = (int)((m / w) % h), n = (int)(m / ((long long)w * h));
    const float* g = (head ? g2 : g1) + (long long)(n * C + c) * h * w;
    const int dl = P.dil[d];
    bf16_t* row = dz + m * lddz + zcol(head, d, c, 0, C);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y - (tap / 3 - 1) * dl, xx = x - (tap % 3 - 1) * dl;
        row[tap] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? f2bf(g[yy * w + xx]) : (bf16_t)0;
    }
    if (t == 0)
        for (int col = 72 * C; col < zc; ++col) dz[m * lddz + col] = (bf16_t)0;
}

// dbias_d[c] += sum over (n, y, x) of g_head[n][c][y][x], the same for the four dilations.  One block per (class, head):
// deterministic.
__global__ void __launch_bounds__(256) aspp_dbias_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                         AsppPtrs P, int N, int hw, int C) {
    __shared__ float part[4];
    const int c = blockIdx.x, head = blockIdx.y;
    const float* g = head ? g2 : g1;
    float s = 0.f;
    for (long long i = threadIdx.x; i < (long long)N * hw; i += 256) {
        const long long n = i / hw, p = i % hw;
        s += g[(n * C + c) * hw + p];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = part[0] + part[1] + part[2] + part[3];
        for (int d = 0; d < 4; ++d) P.dbias[head * 4 + d][c] += tot;
    }
}

static int fill_ptrs(AsppPtrs& P, const float* const* bias, float* const* dbias, const int* dil) {
    for (int i = 0; i < 8; ++i) {
        P.bias[i] = bias ? bias[i] : nullptr;
        P.dbias[i] = dbias ? dbias[i] : nullptr;
    }
    for (int d = 0; d < 4; ++d) {
        if (dil[d] < 1) return RGDA_ERR_ARG;
        P.dil[d] = dil[d];
    }
    return RGDA_OK;
}

extern "C" int rgda_aspp_gather(const void* z, int ldz, const float* const* bias, float* out1, float* out2, int N, int h,
                                int w, int C, const int* dil, rgda_stream_t stream) {
    if (!z || !bias || !out1 || !out2 || !dil || N < 1 || h < 1 || w < 1 || C < 1 || ldz < 72 * C) return RGDA_ERR_ARG;
    AsppPtrs P;
    if (int e = fill_ptrs(P, bias, nullptr, dil)) return e;
    for (int i = 0; i < 8; ++i)
        if (!P.bias[i]) return RGDA_ERR_ARG;
    const long long n = (long long)N * h * w * C;
    aspp_gather_kernel<<<dim3(cdiv(n, 256), 2), 256, 0, to_stream(stream)>>>((const bf16_t*)z, ldz, P, out1, out2, N, h, w, C);
    return hipGetLastError() == hipSuccess ? RGDA_OK : RGDA_ERR_LAUNCH;
}

extern "C" int rgda_aspp_scatter(const float* g1, const float* g2, void* dz, int lddz, int zc, float* const* dbias, int N,
                                 int h, int w, int C, const int* dil, rgda_stream_t stream) {
    if (!g1 || !g2 || !dz || !dbias || !dil || N < 1 || h < 1 || w < 1 || C < 1 || zc < 72 * C || lddz < zc)
        return RGDA_ERR_ARG;
    AsppPtrs P;
    if (int e = fill_ptrs(P, nullptr, dbias, dil)) return e;
    for (int i = 0; i < 8; ++i)
        if (!P.dbias[i]) return RGDA_ERR_ARG;
    const long long n = (long long)N * h * w * 8 * C;
    aspp_scatter_kernel<<<cdiv(n, 256), 256, 0, to_stream(stream)>>>(g1, g2, (bf16_t*)dz, lddz, zc, P, N, h, w, C);
    aspp_dbias_kernel<<<dim3(C, 2), 256, 0, to_stream(stream)>>>(g1, g2, P, N, h * w, C);
    return hipGetLastError() == hipSuccess ? RGDA_OK : RGDA_ERR_LAUNCH;
}
