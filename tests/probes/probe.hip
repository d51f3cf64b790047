// Hardware-semantics probes (test-only; NOT part of librgda_hip.so).
// They pin the gfx950 facts the conv kernels rely on:
//   * operand / accumulator lane layout of v_mfma_f32_32x32x16_bf16
//   * what ds_read_b64_tr_b16 returns for the per-lane addresses conv_wgrad uses
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// A [32][16] row-major bf16 bits, B [16][32] row-major, D [32][32] f32
__global__ void probe_mfma(const unsigned short* A, const unsigned short* B, float* D) {
    int l = threadIdx.x;
    u16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[(l & 31) * 16 + (l >> 5) * 8 + j];
        b[j] = B[((l >> 5) * 8 + j) * 32 + (l & 31)];
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        D[row * 32 + (l & 31)] = acc[r];
    }
}

// T: [16 rows(k)][stride elements] in LDS (copied from global); every lane reads with the address
// formula of conv_wgrad; out[lane][second][4]
__global__ void probe_tr(const unsigned short* T, int stride, unsigned short* out) {
    extern __shared__ unsigned short lds[];
    for (int i = threadIdx.x; i < 16 * stride; i += 64) lds[i] = T[i];
    __syncthreads();
    int l = threadIdx.x, g = l >> 4, a = l & 15;
    for (int second = 0; second < 2; ++second) {
        int row = (g >> 1) * 8 + (a >> 2) + 4 * second;
        int col = (g & 1) * 16 + (a & 3) * 4;
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(lds + row * stride + col));
        for (int j = 0; j < 4; ++j) out[(l * 2 + second) * 4 + j] = (unsigned short)v[j];
    }
}

extern "C" int probe_run_mfma(const void* A, const void* B, void* D, void* stream) {
    probe_mfma<<<1, 64, 0, (hipStream_t)stream>>>((const unsigned short*)A, (const unsigned short*)B, (float*)D);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
extern "C" int probe_run_tr(const void* T, int stride, void* out, void* stream) {
    probe_tr<<<1, 64, 16 * stride * 2, (hipStream_t)stream>>>((const unsigned short*)T, stride, (unsigned short*)out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// buffer_load_dwordx4 ... lds with an out-of-range voffset: does the LDS slot receive zeros?
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
__global__ void probe_buflds(const float* g, int nbytes, float4* out, int oob_lane) {
    __shared__ float4 lds[64];
    lds[threadIdx.x] = make_float4(-1.f, -1.f, -1.f, -1.f);
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
    int voff = (int)threadIdx.x * 16;
    if ((int)threadIdx.x == oob_lane) voff = 0x7fffff00;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}
extern "C" int probe_run_buflds(const void* g, int nbytes, void* out, int oob_lane, void* stream) {
    probe_buflds<<<1, 64, 0, (hipStream_t)stream>>>((const float*)g, nbytes, (float4*)out, oob_lane);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- bandwidth probes: how fast can ONE CU pull L2-resident data (a) into LDS with LDS-DMA, (b) into VGPRs
template <int MODE, int DEPTH>
__global__ void __launch_bounds__(256) probe_bw(const float4* __restrict__ src, int tile_bytes, int iters, float* sink) {
    __shared__ __attribute__((aligned(256))) unsigned char lds[96 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    // every block reads the same `tile_bytes` window again and again (L2 resident), 1 KiB per wave-instruction
    const int per_wave = tile_bytes / nwaves / 1024;      // instructions per wave per tile
    float4 acc = make_float4(0, 0, 0, 0);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 30, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        int base = ((it * 7 + blockIdx.x) & 63) * tile_bytes;     // walk over 64 tiles (stays in L2)
        if (MODE == 0) {
            for (int q = 0; q < per_wave; ++q) {
                int off = base + (q * nwaves + wave) * 1024 + lane * 16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + ((it % DEPTH) * tile_bytes + (q * nwaves + wave) * 1024) % (96 * 1024)), 16, off, 0, 0, 0);
            }
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * 4) : "memory");
        } else {
            for (int q = 0; q < per_wave; ++q) {
                float4 v = src[(base + (q * nwaves + wave) * 1024 + lane * 16) / 16];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE == 0) acc.x = ((float*)lds)[threadIdx.x];
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = acc.x;
}
extern "C" int probe_run_bw(int mode, int depth, int blocks, int threads, const void* src, int tile_bytes, int iters,
                            void* sink, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0 && depth == 1) probe_bw<0, 1><<<blocks, threads, 0, st>>>((const float4*)src, tile_bytes, iters, (float*)sink);
    else if (mode == 0 && depth == 2) probe_bw<0, 2><<<blocks, threads, 0, st>>>((const float4*)src, tile_bytes, iters, (float*)sink);
    else if (mode == 0 && depth == 4) probe_bw<0, 4><<<blocks, threads, 0, st>>>((const float4*)src, tile_bytes, iters, (float*)sink);
    else if (mode == 0) probe_bw<0, 8><<<blocks, threads, 0, st>>>((const float4*)src, tile_bytes, iters, (float*)sink);
    else probe_bw<1, 1><<<blocks, threads, 0, st>>>((const float4*)src, tile_bytes, iters, (float*)sink);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// store-bandwidth probe: every thread writes (mode 0: plain, 1: nontemporal) or copies (mode 2: plain, 3: nontemporal
// store) 16-byte vectors, grid-stride.
typedef float pf4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void probe_store(pf4* __restrict__ dst, const pf4* __restrict__ src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    pf4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (; i < n; i += stride) {
        if (MODE >= 2) v = src[i];
        if (MODE & 1) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
extern "C" int probe_run_store(int mode, int blocks, int threads, void* dst, const void* src, size_t nvec, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) probe_store<0><<<blocks, threads, 0, st>>>((pf4*)dst, (const pf4*)src, nvec);
    else if (mode == 1) probe_store<1><<<blocks, threads, 0, st>>>((pf4*)dst, (const pf4*)src, nvec);
    else if (mode == 2) probe_store<2><<<blocks, threads, 0, st>>>((pf4*)dst, (const pf4*)src, nvec);
    else probe_store<3><<<blocks, threads, 0, st>>>((pf4*)dst, (const pf4*)src, nvec);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Sustained matrix-pipe rate: every wave issues `iters` x 8 independent v_mfma_f32_32x32x16_bf16 on operands it
// loaded once (no memory or LDS traffic inside the loop).  `seed` words are the operands (random bf16 data or zeros:
// the clock the chip sustains depends on the data).  out gets one value per wave so nothing is optimised away.
__global__ void __launch_bounds__(512) probe_mfma_rate(const unsigned* seed, float* out, int iters) {
    const int l = threadIdx.x & 63;
    u16x8 a[2], b[4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) a[i][j] = (unsigned short)seed[(l * 16 + i * 8 + j) & 4095];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) b[i][j] = (unsigned short)(seed[(l * 32 + i * 8 + j + 1024) & 4095] >> 16);
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                    __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (l == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = s;
}

extern "C" int probe_run_mfma_rate(const void* seed, void* out, int blocks, int threads, int iters, void* stream) {
    probe_mfma_rate<<<blocks, threads, 0, (hipStream_t)stream>>>((const unsigned*)seed, (float*)out, iters);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Cost of a software grid barrier (all workgroups co-resident): every round thread 0 of a workgroup adds 1 to a
// device-scope counter and spins on it until all gridDim.x workgroups of the round have arrived; then the workgroup
// continues.  Launch with at most one small workgroup per CU.  (Sizing a conv + BatchNorm-apply fusion that would keep a
// layer's tile in LDS across the statistics' global reduction.)
__global__ void __launch_bounds__(256) probe_grid_barrier(unsigned* counter, unsigned long long* out, int rounds) {
    const unsigned n = gridDim.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(r + 1) * n;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

extern "C" int probe_run_grid_barrier(void* counter, void* out, int blocks, int rounds, void* stream) {
    probe_grid_barrier<<<blocks, 256, 0, (hipStream_t)stream>>>((unsigned*)counter, (unsigned long long*)out, rounds);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// The same barrier, hierarchical: workgroup b arrives at the counter of ITS XCD (b % 8: workgroups are dispatched round-robin
// over the XCDs), the last arriver of an XCD arrives at the global counter, the last of those publishes the round number in
// eight flag words (one 128-byte line per XCD) and every workgroup spins on its XCD's flag only -- 32 pollers per line
// instead of 256 on one, 8 + 8 serialised read-modify-writes instead of 256.
// state: unsigned [8 XCD counters x 32 words apart][global counter][8 flags x 32 words apart]
__global__ void __launch_bounds__(256) probe_grid_barrier_hier(unsigned* state, unsigned long long* out, int rounds) {
    const unsigned n = gridDim.x, xcd = blockIdx.x & 7;
    const unsigned n_xcd = n / 8 + ((blockIdx.x & 7) < (n & 7) ? 1u : 0u);      // workgroups of this XCD
    unsigned* cnt_x = state + xcd * 32;
    unsigned* cnt_g = state + 8 * 32;
    unsigned* flags = state + 9 * 32;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) {
            const unsigned round = (unsigned)(r + 1);
            if (__hip_atomic_fetch_add(cnt_x, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == round * n_xcd - 1) {
                const unsigned groups = n < 8 ? n : 8;
                if (__hip_atomic_fetch_add(cnt_g, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == round * groups - 1) {
                    for (unsigned x = 0; x < groups; ++x)
                        __hip_atomic_store(flags + x * 32, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            while (__hip_atomic_load(flags + xcd * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

extern "C" int probe_run_grid_barrier_hier(void* state, void* out, int blocks, int rounds, void* stream) {
    probe_grid_barrier_hier<<<blocks, 256, 0, (hipStream_t)stream>>>((unsigned*)state, (unsigned long long*)out, rounds);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
