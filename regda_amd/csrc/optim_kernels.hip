// Optimizer side of the SSL step on flat fp32 parameter / gradient / momentum / EMA buffers
// (tools/train_ssl_reg.py:174-175,239-241; regda/utils/ema.py:46-51) and the weight-format
// passes that feed the conv kernels.  All HBM-bound, float4 per lane.
#include "common.h"

__global__ void __launch_bounds__(256) sumsq_partial_kernel(const float* __restrict__ g, long long n, float* part) {
    double acc = 0.0;
    long long n4 = n >> 2;
    const float4* g4 = (const float4*)g;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = g4[i];
        acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[(n4 << 2) + threadIdx.x]; acc += (double)v * v; }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = (float)red[0];
}
__global__ void __launch_bounds__(256) sumsq_final_kernel(const float* __restrict__ part, int n, float* out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

extern "C" int rgda_sumsq(const float* g, int64_t n, float* out, float* ws, rgda_stream_t stream) {
    if (!g || !out || !ws || n <= 0) return RGDA_ERR_ARG;
    hipStream_t st = to_stream(stream);
    int blocks = min(cdiv(n, 256 * 16), 1024);
    sumsq_partial_kernel<<<blocks, 256, 0, st>>>(g, n, ws);
    RGDA_CHECK_LAUNCH();
    sumsq_final_kernel<<<1, 256, 0, st>>>(ws, blocks, out);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

__global__ void __launch_bounds__(256) sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ v, float* __restrict__ shadow,
                                                       bf16_t* __restrict__ pb, bf16_t* __restrict__ sb,
                                                       const float* __restrict__ gnorm_sq,
                                                       const float* __restrict__ lr_dev, long long n4, float momentum,
                                                       float wd, float max_norm, float gscale, float ema_d,
                                                       int first_step) {
    // torch clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    const float total = sqrtf(gnorm_sq[0]) * gscale;
    const float coef = fminf(max_norm / (total + 1e-6f), 1.f) * gscale;
    const float lr = lr_dev[0];
    float4* p4 = (float4*)p;
    const float4* g4 = (const float4*)g;
    float4* v4 = (float4*)v;
    float4* s4 = (float4*)shadow;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 pp = p4[i], gg = g4[i], vv;
        float d0 = gg.x * coef + wd * pp.x, d1 = gg.y * coef + wd * pp.y, d2 = gg.z * coef + wd * pp.z,
              d3 = gg.w * coef + wd * pp.w;
        if (first_step) { vv = make_float4(d0, d1, d2, d3); }       // torch SGD: buf = d_p on the first step
        else {
            vv = v4[i];
            vv.x = momentum * vv.x + d0; vv.y = momentum * vv.y + d1; vv.z = momentum * vv.z + d2; vv.w = momentum * vv.w + d3;
        }
        pp.x -= lr * vv.x; pp.y -= lr * vv.y; pp.z -= lr * vv.z; pp.w -= lr * vv.w;
        v4[i] = vv;
        p4[i] = pp;
        if (shadow) {
            float4 ss = s4[i];
            const float a = 1.f - ema_d;
            ss.x = a * pp.x + ema_d * ss.x; ss.y = a * pp.y + ema_d * ss.y; ss.z = a * pp.z + ema_d * ss.z; ss.w = a * pp.w + ema_d * ss.w;
            s4[i] = ss;
            if (sb) {           // the teacher's bf16 mirror, while the shadow weights are in registers
                uint2 pk;
                pk.x = pack2bf(ss.x, ss.y);
                pk.y = pack2bf(ss.z, ss.w);
                *(uint2*)(sb + (i << 2)) = pk;
            }
        }
        if (pb) {
            uint2 pk;
            pk.x = pack2bf(pp.x, pp.y);
            pk.y = pack2bf(pp.z, pp.w);
            *(uint2*)(pb + (i << 2)) = pk;
        }
    }
}

extern "C" int rgda_sgd_step(float* p, const float* g, float* v, float* shadow, void* p_bf16, void* shadow_bf16, const float* gnorm_sq,
                             const float* lr_dev, int64_t n, float momentum, float weight_decay, float max_norm,
                             float gscale, float ema_decay, int first_step, rgda_stream_t stream) {
    if (!p || !g || !v || !gnorm_sq || !lr_dev || n <= 0 || (n & 3)) return RGDA_ERR_ARG;
    long long n4 = n >> 2;
    int blocks = min(cdiv(n4, 256 * 4), 4096);
    sgd_step_kernel<<<blocks, 256, 0, to_stream(stream)>>>(p, g, v, shadow, (bf16_t*)p_bf16, (bf16_t*)shadow_bf16, gnorm_sq, lr_dev, n4,
                                                            momentum, weight_decay, max_norm, gscale, ema_decay,
                                                            first_step);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

__global__ void __launch_bounds__(256) cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n) {
    long long n4 = n >> 2;
    const float4* s4 = (const float4*)src;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = s4[i];
        uint2 pk;
        pk.x = pack2bf(v.x, v.y);
        pk.y = pack2bf(v.z, v.w);
        *(uint2*)(dst + (i << 2)) = pk;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = f2bf(src[(n4 << 2) + threadIdx.x]);
}

extern "C" int rgda_cast_bf16(const float* src, void* dst, int64_t n, rgda_stream_t stream) {
    if (!src || !dst || n <= 0) return RGDA_ERR_ARG;
    int blocks = min(cdiv(n, 256 * 16), 4096);
    cast_bf16_kernel<<<blocks, 256, 0, to_stream(stream)>>>(src, (bf16_t*)dst, n);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// ---- gradient exchange with bf16 payloads (regda_amd/ddp.py, payload = 'bf16'): every rank receives its shard of the
// bucket from every rank as bf16 (all-to-all), adds the `world` copies in fp32 IN RANK ORDER (the same order on every rank,
// whatever arrived first) and rounds the sum to bf16 once; the reduced shards are all-gathered as bf16 and widened back
// into the fp32 gradient buffer the optimizer reads.  Half the bytes of an fp32 all-reduce on every link, fp32 accumulation.
__global__ void __launch_bounds__(256) ddp_accumulate_bf16_kernel(const bf16_t* __restrict__ recv, int world, bf16_t* __restrict__ out,
                                                                  long long s) {
    const long long nv = s >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < world; ++r) {
            const u16x8 v = *(const u16x8*)(recv + (size_t)r * s + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
        }
        uint4 pk;
        pk.x = pack2bf(acc[0], acc[1]); pk.y = pack2bf(acc[2], acc[3]); pk.z = pack2bf(acc[4], acc[5]); pk.w = pack2bf(acc[6], acc[7]);
        *(uint4*)(out + i * 8) = pk;
    }
}

extern "C" int rgda_ddp_accumulate_bf16(const void* recv, int world, void* out, int64_t shard_elems, rgda_stream_t stream) {
    if (!recv || !out || world < 1 || shard_elems <= 0 || (shard_elems & 7)) return RGDA_ERR_ARG;
    const int blocks = min(cdiv(shard_elems >> 3, 256), 2048);
    ddp_accumulate_bf16_kernel<<<blocks, 256, 0, to_stream(stream)>>>((const bf16_t*)recv, world, (bf16_t*)out, shard_elems);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

__global__ void __launch_bounds__(256) cast_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = bf2f(src[i]);
}

extern "C" int rgda_cast_f32(const void* src, float* dst, int64_t n, rgda_stream_t stream) {
    if (!src || !dst || n <= 0) return RGDA_ERR_ARG;
    cast_f32_kernel<<<min(cdiv(n, 256 * 8), 4096), 256, 0, to_stream(stream)>>>((const bf16_t*)src, dst, n);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// w [Co][T][Ci] f32 -> wt [Ci][T][Co] bf16, 32x32 LDS tiles per tap
__global__ void __launch_bounds__(256) weight_transpose_kernel(const float* __restrict__ w, bf16_t* __restrict__ wt, int Co,
                                                               int T, int Ci) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z, co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Co && ci < Ci) ? w[((size_t)co * T + tap) * Ci + ci] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int ci = ci0 + r, co = co0 + tx;
        if (ci < Ci && co < Co) wt[((size_t)ci * T + tap) * Co + co] = f2bf(tile[tx][r]);
    }
}

// all derived bf16 weight layouts of a model in ONE launch: table[n][8] int64 =
//   {src*, dst bf16*, Co, T, Ci, first_block, src_ld, mode}; src is f32, or bf16 when mode has bit 4 (16) set -- the bf16
//   mirror of the master weights, written by the optimizer kernel with the same rounding: half the bytes to read
// src element (co, tap, ci) lives at src[(co*T + tap)*src_ld + ci] (src_ld = Ci for a whole tensor, larger for a
// channel slice of a wider one); mode 0: dst[ci][tap][co] (data-gradient operand), mode 1: dst[tap][co][ci]
// (nine 1x1 filters stacked), mode 2: dst[co][tap][ci] (plain slice).  A row owns ceil(Ci/32)*ceil(Co/32)*T
// consecutive blocks starting at first_block (ascending).
__global__ void __launch_bounds__(256) weight_transpose_batched_kernel(const long long* __restrict__ table, int n) {
    // 64 x 64 tiles: 256-byte fp32 row reads, 128-byte bf16 row writes in every mode (32-wide tiles wrote half lines)
    constexpr int TS = RGDA_LAYOUT_TILE;
    __shared__ float tile[TS][TS + 1];
    // which table row owns this block: last entry with first_block <= b.  The first_block column goes to LDS in ONE
    // coalesced round trip and is searched there: eight DEPENDENT global loads per block (a binary search in memory) were
    // most of a block's life (21 600 blocks of a few microseconds each)
    constexpr int NFB = 1024;
    __shared__ long long fb[NFB];
    const long long b = blockIdx.x;
    int lo = 0, hi = n - 1;
    if (n <= NFB) {
        for (int i = threadIdx.x; i < n; i += 256) fb[i] = table[i * 8 + 5];
        __syncthreads();
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (fb[mid] <= b) lo = mid; else hi = mid - 1;
        }
    } else {
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (table[mid * 8 + 5] <= b) lo = mid; else hi = mid - 1;
        }
    }
    const long long* e = table + lo * 8;
    const float* w = (const float*)e[0];
    const bf16_t* wh = (const bf16_t*)e[0];
    bf16_t* wt = (bf16_t*)e[1];
    const int Co = (int)e[2], T = (int)e[3], Ci = (int)e[4];
    const long long sld = e[6];
    const int mode = (int)e[7] & 15;
    const bool src16 = ((int)e[7] & 16) != 0;
    int rel = (int)(b - e[5]);
    const int nci = (Ci + TS - 1) / TS, nco = (Co + TS - 1) / TS;
    const int bx = rel % nci; rel /= nci;
    const int by = rel % nco;
    const int tap = rel / nco;
    const int co0 = by * TS, ci0 = bx * TS;
    // fast path: 16-byte reads (4 floats along ci), 8-byte writes (4 bf16 along the destination's contiguous index)
    const bool vec = !(Ci & 3) && !(sld & 3) && !((size_t)w & (src16 ? 7 : 15)) && !((size_t)wt & 7) && (mode != 0 || !(Co & 3));
    if (vec) {
        const int q = threadIdx.x & 15, rq = threadIdx.x >> 4;          // 16 quads per row, 16 rows per pass
        for (int r = rq; r < TS; r += 16) {
            const int co = co0 + r, ci = ci0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < Co && ci < Ci) {
                if (src16) {
                    const uint2 h = *(const uint2*)(wh + ((size_t)co * T + tap) * sld + ci);
                    v = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u),
                                    __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xffff0000u));
                } else {
                    v = *(const float4*)(w + ((size_t)co * T + tap) * sld + ci);
                }
            }
            tile[r][q * 4 + 0] = v.x; tile[r][q * 4 + 1] = v.y; tile[r][q * 4 + 2] = v.z; tile[r][q * 4 + 3] = v.w;
        }
        __syncthreads();
        for (int r = rq; r < TS; r += 16) {
            uint2 pk;
            if (mode == 0) {
                const int ci = ci0 + r, co = co0 + q * 4;
                if (ci >= Ci || co >= Co) continue;
                pk.x = pack2bf(tile[q * 4 + 0][r], tile[q * 4 + 1][r]);
                pk.y = pack2bf(tile[q * 4 + 2][r], tile[q * 4 + 3][r]);
                *(uint2*)(wt + ((size_t)ci * T + tap) * Co + co) = pk;
            } else {
                const int co = co0 + r, ci = ci0 + q * 4;
                if (ci >= Ci || co >= Co) continue;
                pk.x = pack2bf(tile[r][q * 4 + 0], tile[r][q * 4 + 1]);
                pk.y = pack2bf(tile[r][q * 4 + 2], tile[r][q * 4 + 3]);
                *(uint2*)(wt + (mode == 1 ? ((size_t)tap * Co + co) : ((size_t)co * T + tap)) * Ci + ci) = pk;
            }
        }
        return;
    }
    const int tx = threadIdx.x & (TS - 1), ty = threadIdx.x / TS;
    constexpr int RS = 256 / TS;            // rows per pass
    for (int r = ty; r < TS; r += RS) {
        int co = co0 + r, ci = ci0 + tx;
        const size_t si = ((size_t)co * T + tap) * sld + ci;
        tile[r][tx] = (co < Co && ci < Ci) ? (src16 ? bf2f(wh[si]) : w[si]) : 0.f;
    }
    __syncthreads();
    if (mode == 0) {
        for (int r = ty; r < TS; r += RS) {
            int ci = ci0 + r, co = co0 + tx;
            if (ci < Ci && co < Co) wt[((size_t)ci * T + tap) * Co + co] = f2bf(tile[tx][r]);
        }
    } else {
        for (int r = ty; r < TS; r += RS) {
            int co = co0 + r, ci = ci0 + tx;
            if (ci < Ci && co < Co)
                wt[(mode == 1 ? ((size_t)tap * Co + co) : ((size_t)co * T + tap)) * Ci + ci] = f2bf(tile[r][tx]);
        }
    }
}

extern "C" int rgda_weight_transpose_batched(const int64_t* table, int n, int64_t total_blocks, rgda_stream_t stream) {
    if (!table || n <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) return RGDA_ERR_ARG;
    weight_transpose_batched_kernel<<<(int)total_blocks, 256, 0, to_stream(stream)>>>((const long long*)table, n);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

extern "C" int rgda_weight_transpose_bf16(const float* w, void* wt, int Co, int T, int Ci, rgda_stream_t stream) {
    if (!w || !wt || Co <= 0 || T <= 0 || Ci <= 0) return RGDA_ERR_ARG;
    dim3 grid(cdiv(Ci, 32), cdiv(Co, 32), T);
    weight_transpose_kernel<<<grid, 256, 0, to_stream(stream)>>>(w, (bf16_t*)wt, Co, T, Ci);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// rows of K f32 -> rows of Kp bf16, zero padded (stem weights [64][147] -> [64][192])
__global__ void __launch_bounds__(256) pad_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int K, int Kp) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R * Kp) return;
    int r = i / Kp, k = i % Kp;
    dst[i] = (k < K) ? f2bf(src[(size_t)r * K + k]) : (bf16_t)0;
}
// dst[R][K] f32 += src[R][Kp] f32 (first K columns)
__global__ void __launch_bounds__(256) unpad_acc_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int K, int Kp) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R * K) return;
    int r = i / K, k = i % K;
    dst[i] += src[(size_t)r * Kp + k];
}

extern "C" int rgda_pad_cast_bf16(const float* src, void* dst, int R, int K, int Kp, rgda_stream_t stream) {
    if (!src || !dst || R <= 0 || K <= 0 || Kp < K) return RGDA_ERR_ARG;
    pad_cast_kernel<<<cdiv((long long)R * Kp, 256), 256, 0, to_stream(stream)>>>(src, (bf16_t*)dst, R, K, Kp);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}
extern "C" int rgda_unpad_acc_f32(const float* src, float* dst, int R, int K, int Kp, rgda_stream_t stream) {
    if (!src || !dst || R <= 0 || K <= 0 || Kp < K) return RGDA_ERR_ARG;
    unpad_acc_kernel<<<cdiv((long long)R * K, 256), 256, 0, to_stream(stream)>>>(src, dst, R, K, Kp);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}


// ---- the small host-side chores of a step as kernels of this library (a step then launches nothing from torch or the
// runtime's blit kernels: DESIGN.md 3): buffer clears, device -> device copies of the BatchNorm buffers / the stem's
// image copies, the learning rate word, the Dropout2d keep masks.
__global__ void __launch_bounds__(256) fill_zero_kernel(uint4* __restrict__ p, long long n16, unsigned char* tail, int ntail) {
    const uint4 z = {0u, 0u, 0u, 0u};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

extern "C" int rgda_fill_zero(void* p, size_t bytes, rgda_stream_t stream) {
    if (!p || ((uintptr_t)p & 15)) return RGDA_ERR_ARG;
    if (bytes == 0) return RGDA_OK;
    const long long n16 = (long long)(bytes >> 4);
    fill_zero_kernel<<<(int)min((long long)cdiv(n16 > 0 ? n16 : 1, 256 * 4), 4096ll), 256, 0, to_stream(stream)>>>(
        (uint4*)p, n16, (unsigned char*)p + (n16 << 4), (int)(bytes & 15));
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

struct CopyJobs { const uint4* src[4]; uint4* dst[4]; long long n16[4]; int first[5]; };
__global__ void __launch_bounds__(256) copy_multi_kernel(CopyJobs j, int njobs) {
    int k = 0;
    while (k + 1 < njobs && (int)blockIdx.x >= j.first[k + 1]) ++k;
    const int nb = j.first[k + 1] - j.first[k], bid = blockIdx.x - j.first[k];
    for (long long i = (long long)bid * 256 + threadIdx.x; i < j.n16[k]; i += (long long)nb * 256) j.dst[k][i] = j.src[k][i];
}

// up to four device -> device copies in one launch (16-byte aligned, sizes multiples of 16): dsts / srcs / bytes HOST arrays
extern "C" int rgda_copy_multi(int n, void* const* dsts, const void* const* srcs, const size_t* bytes, rgda_stream_t stream) {
    if (n < 1 || n > 4 || !dsts || !srcs || !bytes) return RGDA_ERR_ARG;
    CopyJobs j;
    int blocks = 0;
    for (int k = 0; k < n; ++k) {
        if (!dsts[k] || !srcs[k] || ((uintptr_t)dsts[k] & 15) || ((uintptr_t)srcs[k] & 15) || (bytes[k] & 15)) return RGDA_ERR_ARG;
        j.src[k] = (const uint4*)srcs[k]; j.dst[k] = (uint4*)dsts[k]; j.n16[k] = (long long)(bytes[k] >> 4);
        j.first[k] = blocks;
        blocks += (int)min((long long)cdiv(j.n16[k] > 0 ? j.n16[k] : 1, 256 * 4), 2048ll);
    }
    for (int k = n; k <= 4; ++k) j.first[k] = blocks;
    copy_multi_kernel<<<blocks, 256, 0, to_stream(stream)>>>(j, n);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

__global__ void set_f32_kernel(float* p, float v) { *p = v; }
extern "C" int rgda_set_f32(float* p, float value, rgda_stream_t stream) {
    if (!p) return RGDA_ERR_ARG;
    set_f32_kernel<<<1, 1, 0, to_stream(stream)>>>(p, value);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// Dropout2d(p) keep masks, already scaled (regda/models/Encoder.py:39): out[i] = (u_i >= p) / (1 - p), u_i uniform in [0, 1)
// from a counter-based generator (SplitMix64 of seed and index: every element independent of the launch geometry)
__global__ void __launch_bounds__(256) dropout_mask_kernel(float* __restrict__ out, long long n, float p, unsigned long long seed) {
    const float keep = 1.f / (1.f - p);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const float u = (float)(z >> 40) * 0x1p-24f;
        out[i] = (u >= p) ? keep : 0.f;
    }
}
extern "C" int rgda_dropout_mask(float* out, int64_t n, float p, uint64_t seed, rgda_stream_t stream) {
    if (!out || n <= 0 || !(p >= 0.f) || !(p < 1.f)) return RGDA_ERR_ARG;
    dropout_mask_kernel<<<min(cdiv(n, 256), 1024), 256, 0, to_stream(stream)>>>(out, n, p, seed);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

__global__ void __launch_bounds__(256) add_bf16_kernel(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ b,
                                                       int ldb, bf16_t* __restrict__ o, int ldo, long long M, int vpr) {
    long long total = M * vpr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i / vpr;
        int c = (int)(i % vpr) * 8;
        u16x8 x = *(const u16x8*)(a + r * lda + c), y = *(const u16x8*)(b + r * ldb + c), z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = f2bf(bf2f(x[e]) + bf2f(y[e]));
        *(u16x8*)(o + r * ldo + c) = z;
    }
}

extern "C" int rgda_add_bf16(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int64_t M, int C,
                             rgda_stream_t stream) {
    if (!a || !b || !out || M <= 0 || C <= 0 || (C & 7) || (lda & 7) || (ldb & 7) || (ldo & 7)) return RGDA_ERR_ARG;
    long long total = M * (C / 8);
    add_bf16_kernel<<<min(cdiv(total, 256), 8192), 256, 0, to_stream(stream)>>>((const bf16_t*)a, lda, (const bf16_t*)b,
                                                                                ldb, (bf16_t*)out, ldo, M, C / 8);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}
