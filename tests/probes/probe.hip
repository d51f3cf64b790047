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
