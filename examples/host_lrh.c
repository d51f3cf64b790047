/* A host WITHOUT Python or torch on the drop-in boundary: plain C against include/rgda_hip.h + the HIP runtime.
 * Homogenizer.forward (LRH; /root/reference/regda/utils/local_region_homog.py:125-152) on two 4 x 4 label maps whose
 * answers can be worked out by hand -- the second one holds the 2 : 2 tie whose ratio 2 / (4 + 1e-5) falls just below
 * percent = 0.5 in fp32 (the region keeps its labels).
 *   build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ -o host_lrh examples/host_lrh.c -Iinclude -I/opt/rocm/include
 *                -Lregda_amd/csrc -lrgda_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/regda_amd/csrc -Wl,-rpath,/opt/rocm/lib
 *   (tests/test_c_host.py compiles it on the build box and runs it on the GPU box) */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <string.h>

#include "rgda_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_RGDA(x) do { int s_ = (x); if (s_ != RGDA_OK) { fprintf(stderr, "%s: %s\n", #x, rgda_strerror(s_)); return 3; } } while (0)

int main(void) {
    enum { B = 2, HW = 16, C = 3, R = 4, IGNORE = -1 };
    const int64_t regions[B * HW] = {0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3,
                                     0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3};
    const int64_t labels[B * HW] = {0, 1, 2, -1, 1, 1, 1, 0, 0, 1, 2, -1, -1, -1, 2, 2,
                                    0, 1, 2, -1, 1, 1, 0, 0, 0, 1, 2, -1, -1, -1, 2, 2};
    /* region 0 is background (left alone); region 1: 3 of 4 -> class 1 (image 0), 2 : 2 -> 0.49999.. < 0.5 -> kept (image 1);
     * region 2: 1 : 1 : 1 -> kept; region 3: both labelled pixels are class 2 -> the ignored pixels become 2 as well */
    const int64_t expect[B * HW] = {0, 1, 2, -1, 1, 1, 1, 1, 0, 1, 2, -1, 2, 2, 2, 2,
                                    0, 1, 2, -1, 1, 1, 0, 0, 0, 1, 2, -1, 2, 2, 2, 2};
    if (rgda_abi_version() != RGDA_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    int64_t *d_lab, *d_reg, *d_out;
    void* ws;
    const size_t ws_bytes = rgda_lrh_workspace(B, R, C);
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_HIP(hipMalloc((void**)&d_lab, sizeof labels));
    CHECK_HIP(hipMalloc((void**)&d_reg, sizeof regions));
    CHECK_HIP(hipMalloc((void**)&d_out, sizeof labels));
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    CHECK_HIP(hipMemcpyAsync(d_lab, labels, sizeof labels, hipMemcpyHostToDevice, st));
    CHECK_HIP(hipMemcpyAsync(d_reg, regions, sizeof regions, hipMemcpyHostToDevice, st));
    CHECK_RGDA(rgda_lrh(d_lab, d_reg, d_out, B, HW, C, IGNORE, 0.5f, R, ws, ws_bytes, (rgda_stream_t)st));
    int64_t out[B * HW];
    int flag = -1;
    CHECK_HIP(hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, st));
    /* the flag word behind hist[B][R][C] and ids[B][R] (rgda_hip.h): bit0 region id out of range, bit1 label out of range */
    CHECK_HIP(hipMemcpyAsync(&flag, (char*)ws + ((size_t)B * R * C + (size_t)B * R) * 4, 4, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    /* a NULL pointer is an argument error, reported before anything is enqueued */
    if (rgda_lrh(NULL, d_reg, d_out, B, HW, C, IGNORE, 0.5f, R, ws, ws_bytes, (rgda_stream_t)st) != RGDA_ERR_ARG) return 4;
    if (rgda_lrh(d_lab, d_reg, d_out, B, HW, C, IGNORE, 0.5f, R, ws, ws_bytes - 1, (rgda_stream_t)st) != RGDA_ERR_WORKSPACE) return 5;
    int bad = memcmp(out, expect, sizeof out) != 0 || flag != 0;
    for (int i = 0; i < B * HW; ++i) printf("%lld%c", (long long)out[i], (i % HW == HW - 1) ? '\n' : ' ');
    printf("%s (flag %d)\n", bad ? "MISMATCH" : "host_lrh ok", flag);
    hipFree(d_lab); hipFree(d_reg); hipFree(d_out); hipFree(ws); hipStreamDestroy(st);
    return bad ? 6 : 0;
}
