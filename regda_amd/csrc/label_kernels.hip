// Label path of the RegDA SSL step on gfx950: pseudo_selection, LRH (Homogenizer),
// label_refine, update_prototype, fused bilinear-upsample + cross-entropy, teacher probs.
// All fp32 / integer, HBM-bound: coalesced wide loads, LDS-resident small state,
// wavefront-level pre-reduction in front of every atomic.  No fast-math: the
// integer decisions must match the reference bit for bit (DESIGN.md "Parity").
#include "common.h"

// --------------------------------------------------------------------------------------
// pseudo_selection   (regda/gast/pseudo_generation.py:59-93)
// --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pseudo_max_kernel(const float* __restrict__ soft, float* classmax,
                                                         int* flag, int hw, int chunk) {
    const int plane = blockIdx.y;  // b*c + c
    const float* p = soft + (size_t)plane * hw;
    int beg = blockIdx.x * chunk, end = min(hw, beg + chunk);
    float mx = 0.f, mn = 0.f;
    bool bad = false;
    if ((hw & 3) == 0 && (chunk & 3) == 0) {
        const float4* p4 = (const float4*)p;
        for (int i = beg / 4 + threadIdx.x; i < end / 4; i += 256) {
            float4 v = p4[i];
            mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            mn = fminf(mn, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
        }
    } else {
        for (int i = beg + threadIdx.x; i < end; i += 256) {
            float v = p[i];
            mx = fmaxf(mx, v);
            mn = fminf(mn, v);
        }
    }
    bad = (mx > 1.f) || (mn < 0.f);
    mx = wave_max(mx);
    __shared__ float smx[4];
    __shared__ int sbad;
    if (threadIdx.x == 0) sbad = 0;
    __syncthreads();
    if (bad) sbad = 1;
    if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        // values are >= 0 on the valid path, so the uint order equals the float order
        atomicMax((unsigned*)(classmax + plane), __float_as_uint(fmaxf(m, 0.f)));
        if (sbad) atomicOr(flag, 1);
    }
}

template <int C>
__global__ void __launch_bounds__(256) pseudo_pick_kernel(const float* __restrict__ soft,
                                                          const float* __restrict__ classmax,
                                                          int64_t* __restrict__ out, int hw, float top,
                                                          float low, int ignore_label, int c_rt) {
    const int b = blockIdx.y;
    const int c = C > 0 ? C : c_rt;
    float thr[C > 0 ? C : 16];
    for (int k = 0; k < c; ++k) thr[k] = fmaxf(__fmul_rn(classmax[b * c + k], top), low);
    const float* base = soft + (size_t)b * c * hw;
    int64_t* o = out + (size_t)b * hw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        int cnt = 0, first = 0;
#pragma unroll
        for (int k = 0; k < c; ++k) {
            float v = base[(size_t)k * hw + i];
            bool pass = v > thr[k];
            if (pass && cnt == 0) first = k;
            cnt += pass ? 1 : 0;
        }
        o[i] = (cnt == 1) ? (int64_t)first : (int64_t)ignore_label;
    }
}

extern "C" size_t rgda_pseudo_select_workspace(int b, int c) { return (size_t)b * c * 4 + 16; }

extern "C" int rgda_pseudo_select(const float* soft, int64_t* out, int b, int c, int hw, float cutoff_top,
                                  float cutoff_low, int ignore_label, int classmax_ready, void* ws,
                                  size_t ws_bytes, rgda_stream_t stream) {
    if (!soft || !out || !ws || b <= 0 || c <= 0 || c > 16 || hw < 0) return RGDA_ERR_ARG;
    if (ws_bytes < rgda_pseudo_select_workspace(b, c)) return RGDA_ERR_WORKSPACE;
    if (hw == 0) return RGDA_OK;
    hipStream_t st = to_stream(stream);
    float* classmax = (float*)ws;
    int* flag = (int*)(classmax + (size_t)b * c);
    if (!classmax_ready) {
        if (zero_bytes(ws, (size_t)b * c * 4 + 4, stream) != RGDA_OK) return RGDA_ERR_LAUNCH;
        int chunk = 16384;
        dim3 grid(cdiv(hw, chunk), b * c);
        pseudo_max_kernel<<<grid, 256, 0, st>>>(soft, classmax, flag, hw, chunk);
        RGDA_CHECK_LAUNCH();
    }
    dim3 grid(min(cdiv(hw, 256), 1024), b);
    if (c == 6)
        pseudo_pick_kernel<6><<<grid, 256, 0, st>>>(soft, classmax, out, hw, cutoff_top, cutoff_low, ignore_label, c);
    else
        pseudo_pick_kernel<0><<<grid, 256, 0, st>>>(soft, classmax, out, hw, cutoff_top, cutoff_low, ignore_label, c);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// --------------------------------------------------------------------------------------
// LRH / Homogenizer   (regda/utils/local_region_homog.py:125-152)
//   pass 1: per (image, region) class histogram.  Each workgroup owns a contiguous pixel
//           chunk of ONE image, keeps an LDS histogram for region ids < lds_regions and
//           merges equal (region,class) keys of neighbouring lanes before touching it
//           (wavefront run-length reduction); the LDS bins are flushed with one global
//           atomic per non-empty bin.
//   pass 2: per region: n, m, first argmax, fp32 ratio test -> id table.
//   pass 3: gather.
// --------------------------------------------------------------------------------------
// key of the previous lane (-1 for lane 0): a DPP wavefront shift, one VALU instruction -- `__shfl_up` is a ds_bpermute
// through the LDS crossbar, and the histogram kernels are bound by exactly these merges.  Callers pass key = -1 for lanes
// that do not count, so "previous lane valid and equal" is one comparison.
static __device__ __forceinline__ int prev_lane_key(int key) {
    return __builtin_amdgcn_update_dpp(-1, key, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

static __device__ __forceinline__ void lrh_add(int key, bool valid, int* lds_hist, int lds_bins, int* ghist) {
    // merge runs of equal keys across the 64 lanes: only run heads issue an atomic
    const int lane = threadIdx.x & 63;
    if (!valid) key = -1;
    const int prev = prev_lane_key(key);
    bool head = valid && prev != key;
    unsigned long long heads = __ballot(head);
    unsigned long long valids = __ballot(valid);
    if (head) {
        // run ends at the next head or at the next invalid lane
        unsigned long long stop = (heads | ~valids) >> lane >> 1;
        int len = stop ? (__builtin_ctzll(stop) + 1) : (64 - lane);
        if (key < lds_bins)
            atomicAdd(&lds_hist[key], len);
        else
            atomicAdd(&ghist[key], len);
    }
}

// the same merge with `w` pixels per lane (all of one key): a run of lanes adds the sum of its lanes' weights
static __device__ __forceinline__ void lrh_add_w(int key, int w, bool valid, int* lds_hist, int lds_bins, int* ghist) {
    const int lane = threadIdx.x & 63;
    if (!valid) key = -1;
    const int prev = prev_lane_key(key);
    bool head = valid && prev != key;
    unsigned long long heads = __ballot(head);
    unsigned long long valids = __ballot(valid);
    unsigned long long w4 = __ballot(valid && w == 4);         // every valid lane carries 4 pixels or fewer: count the 4s
    if (head) {
        unsigned long long stop = (heads | ~valids) >> lane >> 1;
        int len = stop ? (__builtin_ctzll(stop) + 1) : (64 - lane);
        const unsigned long long run = (len == 64 ? ~0ull : ((1ull << len) - 1)) << lane;
        // lanes of the run with weight 4 add 4, the others (weight 1 .. 3) are handled by the caller's slow path
        int total = 4 * __builtin_popcountll(run & w4);
        if (key < lds_bins)
            atomicAdd(&lds_hist[key], total);
        else
            atomicAdd(&ghist[key], total);
    }
}

__global__ void __launch_bounds__(256) lrh_hist_kernel(const int64_t* __restrict__ labels,
                                                       const int64_t* __restrict__ regions, int* hist,
                                                       int* flag, int hw, int chunk, int C, int ignore_label,
                                                       int R, int lds_regions) {
    extern __shared__ int lds_hist[];
    const int b = blockIdx.y;
    const int lds_bins = lds_regions * C;
    for (int i = threadIdx.x; i < lds_bins; i += 256) lds_hist[i] = 0;
    __syncthreads();
    const int64_t* lab = labels + (size_t)b * hw;
    const int64_t* reg = regions + (size_t)b * hw;
    int* gh = hist + (size_t)b * R * C;
    int beg = blockIdx.x * chunk, end = min(hw, beg + chunk);
    int bad = 0;
    // the loop bound is wave-uniform so the shuffles/ballots in lrh_add see full waves
    constexpr int LB = 4;                                          // 256-pixel slabs whose loads are issued together
    for (int i0 = beg; i0 < end; i0 += 256 * LB) {
        long long lv[LB], rv[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int i = i0 + u * 256 + threadIdx.x;
            const bool in = i < end;
            lv[u] = in ? lab[i] : (long long)ignore_label;
            rv[u] = in ? reg[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            if (i0 + u * 256 >= end) break;                         // wave-uniform
            const int i = i0 + u * 256 + threadIdx.x;
            const bool in = i < end;
            const long long l = lv[u], r = rv[u];
            bool rok = (r >= 0) && (r < R);
            bool lok = (l >= 0) && (l < C);
            if (in && !rok) bad |= 1;
            if (in && !lok && l != ignore_label) bad |= 2;
            bool valid = in && rok && lok;
            int key = valid ? ((int)r * C + (int)l) : -1;
            lrh_add(key, valid, lds_hist, lds_bins, gh);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < lds_bins; i += 256) {
        int v = lds_hist[i];
        if (v) atomicAdd(&gh[i], v);
    }
    if (bad) atomicOr(flag, bad);
}

__global__ void __launch_bounds__(256) lrh_decide_kernel(const int* __restrict__ hist, int* ids, int total, int C,
                                                         int ignore_label, float percent) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int* h = hist + (size_t)i * C;
    int n = 0, m = h[0], arg = 0;
    for (int c = 0; c < C; ++c) {
        int v = h[c];
        n += v;
        if (v > m) { m = v; arg = c; }   // strict > keeps the FIRST maximum (torch.max)
    }
    // class_num_max / (pixel_num_4sup + 1e-5) in fp32, IEEE divide (local_region_homog.py:143)
    float ratio = __fdiv_rn((float)m, __fadd_rn((float)n, 1e-5f));
    ids[i] = (ratio < percent) ? ignore_label : arg;
}

__global__ void __launch_bounds__(256) lrh_gather_kernel(const int64_t* __restrict__ labels,
                                                         const int64_t* __restrict__ regions,
                                                         const int* __restrict__ ids, int64_t* __restrict__ out,
                                                         int hw, int R, int ignore_label) {
    const int b = blockIdx.y;
    const int64_t* lab = labels + (size_t)b * hw;
    const int64_t* reg = regions + (size_t)b * hw;
    const int* id = ids + (size_t)b * R;
    int64_t* o = out + (size_t)b * hw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        long long r = reg[i];
        long long l = lab[i];
        long long v = ignore_label;
        if (r > 0 && r < R) v = id[r];
        o[i] = (v == ignore_label) ? l : v;
    }
}

extern "C" size_t rgda_lrh_workspace(int b, int max_regions, int class_num) {
    return ((size_t)b * max_regions * class_num + (size_t)b * max_regions) * 4 + 16;
}

extern "C" int rgda_lrh(const int64_t* labels, const int64_t* regions, int64_t* out, int b, int hw,
                        int class_num, int ignore_label, float percent, int max_regions, void* ws,
                        size_t ws_bytes, rgda_stream_t stream) {
    if (!labels || !regions || !out || !ws || b <= 0 || hw < 0 || class_num <= 0 || max_regions <= 0)
        return RGDA_ERR_ARG;
    if (ws_bytes < rgda_lrh_workspace(b, max_regions, class_num)) return RGDA_ERR_WORKSPACE;
    if (hw == 0) return RGDA_OK;
    hipStream_t st = to_stream(stream);
    const int R = max_regions, C = class_num;
    int* hist = (int*)ws;
    int* ids = hist + (size_t)b * R * C;
    int* flag = ids + (size_t)b * R;
    // one clear for the histogram and the flag word (the id table between them is rewritten anyway)
    if (zero_bytes(hist, ((size_t)b * R * C + (size_t)b * R + 1) * 4, stream) != RGDA_OK) return RGDA_ERR_LAUNCH;
    int lds_regions = min(R, (48 * 1024) / (C * 4));
    // pixels per workgroup: enough workgroups to fill the chip (a 16 K chunk left half of the CUs idle and made every
    // workgroup a chain of 64 dependent load round trips), few enough that the LDS flush stays small
    int chunk = 16384;
    while (chunk > 2048 && (long long)cdiv(hw, chunk) * b < 512) chunk >>= 1;
    dim3 g1(cdiv(hw, chunk), b);
    lrh_hist_kernel<<<g1, 256, (size_t)lds_regions * C * 4, st>>>(labels, regions, hist, flag, hw, chunk, C,
                                                                    ignore_label, R, lds_regions);
    RGDA_CHECK_LAUNCH();
    lrh_decide_kernel<<<cdiv((long long)b * R, 256), 256, 0, st>>>(hist, ids, b * R, C, ignore_label, percent);
    RGDA_CHECK_LAUNCH();
    dim3 g3(min(cdiv(hw, 256), 1024), b);
    lrh_gather_kernel<<<g3, 256, 0, st>>>(labels, regions, ids, out, hw, R, ignore_label);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// --------------------------------------------------------------------------------------
// pseudo_selection + LRH in ONE pass over the soft labels (the SSL step's chain, tools/train_ssl_reg.py:224-228 ->
// regda/gast/pseudo_generation.py:76-88 -> regda/utils/local_region_homog.py:125-152): the selected hard label is never
// written as an int64 tensor and read back.
//   pick_hist : per pixel the thresholded argmax (registers) -> the (region, class) histogram as in lrh_hist_kernel, plus
//               the label as ONE byte and the region id as 16 bits for the gather (3 B per pixel instead of 16);
//               the LAST workgroup of an image to finish (a counter per image; the histogram is read back with
//               device-scope loads) decides that image's region -> label table: no decide launch.
//   gather    : 3 B per pixel in, the int64 label out.
// Four pixels per lane (16-byte loads of every soft plane, 2 x 16 bytes of region ids); lane j-th pixels of neighbouring
// lanes are 4 pixels apart, inside one region almost always, so the wavefront run-length merge still removes most atomics.
// --------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned u32x4l;
#define RGDA_LBL_SC1 16     /* sc1: device-scope (coherent across the XCDs' L2s) buffer access */

template <int C>
__global__ void __launch_bounds__(256) pick_hist_kernel(const float* __restrict__ soft, const float* __restrict__ classmax,
                                                        const int64_t* __restrict__ regions, unsigned char* __restrict__ lab8,
                                                        unsigned short* __restrict__ reg16, int* hist, int* ids, int* flag,
                                                        int* counters, int hw, int chunk, float top, float low,
                                                        int ignore_label, float percent, int R, int lds_regions) {
    extern __shared__ int lds_hist[];
    __shared__ int s_last;
    const int b = blockIdx.y;
    const int lds_bins = lds_regions * C;
    for (int i = threadIdx.x; i < lds_bins; i += 256) lds_hist[i] = 0;
    float thr[C];
#pragma unroll
    for (int k = 0; k < C; ++k) thr[k] = fmaxf(__fmul_rn(classmax[b * C + k], top), low);
    __syncthreads();
    const float* base = soft + (size_t)b * C * hw;
    const int64_t* reg = regions + (size_t)b * hw;
    int* gh = hist + (size_t)b * R * C;
    const int beg = blockIdx.x * chunk, end = min(hw, beg + chunk);      // chunk % 1024 == 0, hw % 4 == 0
    int bad = 0;
    constexpr int LB = 2;                       // 1024-pixel slabs whose loads are issued together (the kernel is latency-bound:
                                                // 512 workgroups of four slabs each)
    for (int i00 = beg; i00 < end; i00 += 1024 * LB) {                    // wave-uniform bounds: lrh_add sees full waves
        float4 vv[LB][C];
        long long rr[LB][4];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int i = i00 + u * 1024 + threadIdx.x * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[u][j] = 0;
            if (i < end) {
#pragma unroll
                for (int k = 0; k < C; ++k) vv[u][k] = *(const float4*)(base + (size_t)k * hw + i);
                const longlong2 ra = *(const longlong2*)(reg + i), rb = *(const longlong2*)(reg + i + 2);
                rr[u][0] = ra.x; rr[u][1] = ra.y; rr[u][2] = rb.x; rr[u][3] = rb.y;
            }
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
        const int i0 = i00 + u * 1024;
        if (i0 >= end) break;                                             // wave-uniform
        const int i = i0 + threadIdx.x * 4;
        const bool in = i < end;
        const float4 (&v)[C] = vv[u];
        const long long (&r4)[4] = rr[u];
        unsigned lpack = 0;
        unsigned short rs[4];
        int key[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int cnt = 0, first = 0;
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const float x = j == 0 ? v[k].x : j == 1 ? v[k].y : j == 2 ? v[k].z : v[k].w;
                const bool pass = in && x > thr[k];
                if (pass && cnt == 0) first = k;
                cnt += pass ? 1 : 0;
            }
            const bool labelled = cnt == 1;
            const long long r = r4[j];
            const bool rok = (r >= 0) && (r < R);
            if (in && !rok) bad |= 1;
            key[j] = (in && rok && labelled) ? ((int)r * C + first) : -1;
            lpack |= (labelled ? (unsigned)first : 0xffu) << (8 * j);
            rs[j] = rok ? (unsigned short)r : (unsigned short)0xffff;
        }
        // Labels and regions are piecewise constant: in most waves every lane's four pixels share one key (or none is
        // counted at all).  Those waves make ONE merged add per lane instead of four (the kernel is bound by these
        // shuffles / ballots / LDS atomics, not by its 35 B per pixel); any other wave takes the pixel-by-pixel path.
        const bool same = key[0] == key[1] && key[1] == key[2] && key[2] == key[3];
        if (__all(same)) {
            if (__any(key[0] >= 0)) lrh_add_w(key[0], 4, key[0] >= 0, lds_hist, lds_bins, gh);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) lrh_add(key[j], key[j] >= 0, lds_hist, lds_bins, gh);
        }
        if (in) {
            *(unsigned*)(lab8 + (size_t)b * hw + i) = lpack;
            *(uint2*)(reg16 + (size_t)b * hw + i) = uint2{(unsigned)rs[0] | ((unsigned)rs[1] << 16), (unsigned)rs[2] | ((unsigned)rs[3] << 16)};
        }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < lds_bins; i += 256) {
        const int x = lds_hist[i];
        if (x) atomicAdd(&gh[i], x);
    }
    if (bad) atomicOr(flag, bad);
    // ---- the last workgroup of this image decides the image's table
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's atomics have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                    // agent-scope release: the workgroup's histogram adds (ordered before this
                                                            // thread by the barrier) are visible before the arrival count moves
        s_last = (atomicAdd(&counters[b], 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                        // agent-scope acquire in front of the table's reads (one workgroup per image)
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gh, 0, R * C * 4, 0x00020000);
    // eight regions per thread and pass, all 48 loads in flight together (region after region this tail was 16 dependent
    // memory round trips: 15 of the kernel's 27 us)
    constexpr int RB = 8;
    for (int r0 = threadIdx.x; r0 < R; r0 += 256 * RB) {
        int h[RB][C];
#pragma unroll
        for (int u = 0; u < RB; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c)       // (out-of-range offsets read as zero)
                h[u][c] = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, ((r0 + u * 256) * C + c) * 4, 0, RGDA_LBL_SC1);
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = r0 + u * 256;
            if (r >= R) break;
            int n = 0, m = h[u][0], arg = 0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                n += h[u][c];
                if (h[u][c] > m) { m = h[u][c]; arg = c; }   // strict > keeps the FIRST maximum (torch.max)
            }
            const float ratio = __fdiv_rn((float)m, __fadd_rn((float)n, 1e-5f));     // local_region_homog.py:143, fp32, IEEE divide
            ids[(size_t)b * R + r] = (ratio < percent) ? ignore_label : arg;
        }
    }
    if (threadIdx.x == 0) counters[b] = 0;                  // (the workspace is cleared per call anyway)
#endif
}

__global__ void __launch_bounds__(256) lrh_gather8_kernel(const unsigned char* __restrict__ lab8,
                                                          const unsigned short* __restrict__ reg16,
                                                          const int* __restrict__ ids, int64_t* __restrict__ out, int hw,
                                                          int R, int ignore_label) {
    const int b = blockIdx.y;
    const int* id = ids + (size_t)b * R;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < hw; i += gridDim.x * 1024) {
        const unsigned lp = *(const unsigned*)(lab8 + (size_t)b * hw + i);
        const uint2 rp = *(const uint2*)(reg16 + (size_t)b * hw + i);
        const unsigned rr[4] = {rp.x & 0xffffu, rp.x >> 16, rp.y & 0xffffu, rp.y >> 16};
        long long o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned l8 = (lp >> (8 * j)) & 0xffu;
            const long long l = (l8 == 0xffu) ? (long long)ignore_label : (long long)l8;
            long long v = ignore_label;
            if (rr[j] > 0 && rr[j] < (unsigned)R) v = id[rr[j]];         // region 0 = background: left alone
            o[j] = (v == ignore_label) ? l : v;
        }
        *(longlong2*)(out + (size_t)b * hw + i) = longlong2{o[0], o[1]};
        *(longlong2*)(out + (size_t)b * hw + i + 2) = longlong2{o[2], o[3]};
    }
}

extern "C" size_t rgda_pseudo_lrh_workspace(int b, int hw, int max_regions, int class_num) {
    // int32 hist[b][R][C], int32 ids[b][R], int32 flag, int32 counters[b] (+ pad), uint16 reg16[b][hw], uint8 lab8[b][hw]
    size_t head = ((size_t)b * max_regions * class_num + (size_t)b * max_regions + 1 + (size_t)b) * 4;
    head = (head + 15) & ~(size_t)15;
    return head + (size_t)b * hw * 3 + 16;
}

extern "C" int rgda_pseudo_lrh(const float* soft, const float* classmax, const int64_t* regions, int64_t* out, int b, int hw,
                               int class_num, float cutoff_top, float cutoff_low, int ignore_label, float percent,
                               int max_regions, void* ws, size_t ws_bytes, rgda_stream_t stream) {
    if (!soft || !classmax || !regions || !out || !ws || b <= 0 || hw < 0 || max_regions <= 0 || max_regions > 65535)
        return RGDA_ERR_ARG;
    if (class_num != 6 || (hw & 3)) return RGDA_ERR_UNSUPPORTED;       // (the two-call route serves everything else)
    if (((uintptr_t)ws & 15) || ws_bytes < rgda_pseudo_lrh_workspace(b, hw, max_regions, class_num)) return RGDA_ERR_WORKSPACE;
    if (hw == 0) return RGDA_OK;
    hipStream_t st = to_stream(stream);
    const int R = max_regions, C = class_num;
    int* hist = (int*)ws;
    int* ids = hist + (size_t)b * R * C;
    int* flag = ids + (size_t)b * R;
    int* counters = flag + 1;
    size_t head = ((size_t)b * R * C + (size_t)b * R + 1 + (size_t)b) * 4;
    if (zero_bytes(ws, head, stream) != RGDA_OK) return RGDA_ERR_LAUNCH;            // ONE clear: histogram, flag, counters
    head = (head + 15) & ~(size_t)15;
    unsigned short* reg16 = (unsigned short*)((char*)ws + head);
    unsigned char* lab8 = (unsigned char*)(reg16 + (size_t)b * hw);
    int lds_regions = min(R, (48 * 1024) / (C * 4));
    int chunk = 16384, min_wg = 512;
    if (const char* e = TUNE_ENV("RGDA_LRH_LDS")) lds_regions = min(R, atoi(e));         // tuning experiments only
    if (const char* e = TUNE_ENV("RGDA_LRH_WG")) min_wg = atoi(e);                        // tuning experiments only
    while (chunk > 1024 && (long long)cdiv(hw, chunk) * b < min_wg) chunk >>= 1;
    dim3 g1(cdiv(hw, chunk), b);
    pick_hist_kernel<6><<<g1, 256, (size_t)lds_regions * C * 4, st>>>(soft, classmax, regions, lab8, reg16, hist, ids, flag, counters,
                                                                       hw, chunk, cutoff_top, cutoff_low, ignore_label, percent, R,
                                                                       lds_regions);
    RGDA_CHECK_LAUNCH();
    dim3 g3(min(cdiv(hw, 1024), 512), b);
    lrh_gather8_kernel<<<g3, 256, 0, st>>>(lab8, reg16, ids, out, hw, R, ignore_label);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// --------------------------------------------------------------------------------------
// bilinear align_corners=True helpers (torch upsample_bilinear2d semantics)
// --------------------------------------------------------------------------------------
struct Lerp {
    int i0, i1;
    float l0, l1;
};
static __device__ __forceinline__ Lerp lerp_ac(int dst, int in, int out) {
    float scale = (out > 1) ? __fdiv_rn((float)(in - 1), (float)(out - 1)) : 0.f;
    float src = __fmul_rn(scale, (float)dst);
    Lerp r;
    r.i0 = (int)src;
    r.i1 = r.i0 + ((r.i0 < in - 1) ? 1 : 0);
    r.l1 = __fsub_rn(src, (float)r.i0);
    r.l0 = __fsub_rn(1.f, r.l1);
    return r;
}
static __device__ __forceinline__ float bilerp(const float* p, int w, const Lerp& ly, const Lerp& lx) {
    float v00 = p[ly.i0 * w + lx.i0], v01 = p[ly.i0 * w + lx.i1];
    float v10 = p[ly.i1 * w + lx.i0], v11 = p[ly.i1 * w + lx.i1];
    float top = __fadd_rn(__fmul_rn(lx.l0, v00), __fmul_rn(lx.l1, v01));
    float bot = __fadd_rn(__fmul_rn(lx.l0, v10), __fmul_rn(lx.l1, v11));
    return __fadd_rn(__fmul_rn(ly.l0, top), __fmul_rn(ly.l1, bot));
}

// --------------------------------------------------------------------------------------
// label_refine   (regda/gast/alignment.py:194-265, 396-423)
// --------------------------------------------------------------------------------------
// centred prototypes pc[c][k] and their unbiased std
__global__ void __launch_bounds__(256) proto_center_kernel(const float* __restrict__ protos, float* pc, float* pstd,
                                                           int K) {
    const int c = blockIdx.x;
    const float* p = protos + (size_t)c * K;
    __shared__ float red[4];
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) s += p[k];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float mean = (red[0] + red[1] + red[2] + red[3]) / (float)K;
    __syncthreads();
    float q = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        float d = p[k] - mean;
        pc[(size_t)c * K + k] = d;
        q += d * d;
    }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) pstd[c] = sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)(K - 1));
}

// sim[b][c][p] = 1 / pearson_dist(feat[b,:,p], protos[c]).  Workgroup = PX pixels x SL k-slices (PX * SL threads):
// 32 x 16 gives 256 workgroups of 8 waves for the 8 x 32 x 32 target map, every CU streams its share of the
// 67 MB feature map (the second pass over it is served by L2).
template <int C, int PX, int SL>
__global__ void __launch_bounds__(PX * SL) pearson_sim_kernel(const float* __restrict__ feat, const float* __restrict__ pc,
                                                              const float* __restrict__ pstd, float* __restrict__ sim,
                                                              int K, int hw) {
    extern __shared__ float lds[];      // pc[C][K] then red[SL][PX][C+1]
    float* lpc = lds;
    float* red = lds + (size_t)C * K;
    const int b = blockIdx.y;
    const int lane = threadIdx.x % PX, slice = threadIdx.x / PX;
    const int p = blockIdx.x * PX + lane;
    const bool ok = p < hw;
    for (int i = threadIdx.x; i < C * K; i += PX * SL) lpc[i] = pc[i];
    const float* f = feat + (size_t)b * K * hw + (ok ? p : 0);
    const int kper = (K + SL - 1) / SL;
    const int k0 = min(slice * kper, K), k1 = min(k0 + kper, K);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
        s0 += f[(size_t)k * hw]; s1 += f[(size_t)(k + 1) * hw]; s2 += f[(size_t)(k + 2) * hw]; s3 += f[(size_t)(k + 3) * hw];
    }
    for (; k < k1; ++k) s0 += f[(size_t)k * hw];
    red[(slice * PX + lane) * (C + 1)] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float tot = 0.f;
    for (int q = 0; q < SL; ++q) tot += red[(q * PX + lane) * (C + 1)];
    const float mean = tot / (float)K;
    __syncthreads();
    float q = 0.f, cov[C];
#pragma unroll
    for (int c = 0; c < C; ++c) cov[c] = 0.f;
#pragma unroll 4
    for (k = k0; k < k1; ++k) {
        float d = f[(size_t)k * hw] - mean;
        q += d * d;
#pragma unroll
        for (int c = 0; c < C; ++c) cov[c] += d * lpc[c * K + k];
    }
    float* r = red + (slice * PX + lane) * (C + 1);
    r[0] = q;
#pragma unroll
    for (int c = 0; c < C; ++c) r[1 + c] = cov[c];
    __syncthreads();
    if (slice == 0 && ok) {
        float qq = 0.f;
        for (int s4 = 0; s4 < SL; ++s4) qq += red[(s4 * PX + lane) * (C + 1)];
        float fstd = sqrtf(qq / (float)(K - 1));
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float cv = 0.f;
            for (int s4 = 0; s4 < SL; ++s4) cv += red[(s4 * PX + lane) * (C + 1) + 1 + c];
            float bcov = cv / ((float)(K - 1) + 1e-7f);
            float dist = (-1.0f * bcov / (fstd * pstd[c] + 1e-7f) + 1.0f) * 0.5f;
            sim[((size_t)b * C + c) * hw + p] = 1.0f / dist;
        }
    }
}

// ---- superpixel view (alignment.py:238-258): per (image, superpixel, class) the MAXIMUM of the soft labels over the
// superpixel's pixels (torch_scatter.scatter(reduce='max')), gathered back per pixel, softmax_T(., temp) over the classes,
// divided by its per-pixel maximum + 1e-7; pixels of the superpixel with the largest id OF THE BATCH are `ignored`.
// The maxima are kept as order-preserving unsigned keys so that one integer atomicMax serves any sign.
__device__ __forceinline__ unsigned fkey(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
struct SupView {
    const long long* sup;      // (b, H*W) superpixel ids
    unsigned* tbl;             // (b, max_regions, C) keys of the maxima
    int* cnt;                  // [0] the largest id of the batch, [1] flag: an id outside [0, max_regions)
    int max_regions;
};

template <int C>
__global__ void __launch_bounds__(256) sup_max_kernel(const float* __restrict__ soft, SupView sv, size_t HW) {
    const int b = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = p < HW;
    long long id = in ? sv.sup[(size_t)b * HW + p] : -1;
    const bool ok = in && id >= 0 && id < sv.max_regions;
    if (in && !ok) atomicOr(sv.cnt + 1, 1);
    const int idm = wave_max_i32(ok ? (int)id : -1);
    if ((threadIdx.x & 63) == 0 && idm >= 0) atomicMax(sv.cnt, idm);
    unsigned key[C];
#pragma unroll
    for (int c = 0; c < C; ++c) key[c] = ok ? fkey(soft[((size_t)b * C + c) * HW + p]) : 0u;
    // superpixels are compact: most waves lie inside one, and then one lane carries the wave's maxima
    const int id0 = __builtin_amdgcn_readfirstlane((int)id);
    if (__all(ok && (int)id == id0)) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const unsigned m = wave_max_u32(key[c]);
            if ((threadIdx.x & 63) == 0) atomicMax(sv.tbl + ((size_t)b * sv.max_regions + id0) * C + c, m);
        }
    } else if (ok) {
#pragma unroll
        for (int c = 0; c < C; ++c) atomicMax(sv.tbl + ((size_t)b * sv.max_regions + id) * C + c, key[c]);
    }
}

constexpr int REFINE_ROWS = 8;
template <int C, bool SUP = false>
__global__ void __launch_bounds__(256) refine_apply_kernel(const float* __restrict__ sim, const float* __restrict__ p1,
                                                           const float* __restrict__ p2, const float* __restrict__ soft,
                                                           float* __restrict__ out, float* classmax, int h, int w,
                                                           int H, int W, float temp, int views, SupView spx = {}) {
    // a workgroup walks REFINE_ROWS output rows: the per-class maxima leave as one atomic per class and workgroup
    // (one per row-workgroup was 49 K same-address memory-side atomics, a large part of this kernel's time)
    const int b = blockIdx.z;
    const int X = blockIdx.x * 256 + threadIdx.x;
    const int hw = h * w;
    const size_t HW = (size_t)H * W;
    float omax[C];
#pragma unroll
    for (int c = 0; c < C; ++c) omax[c] = 0.f;
    // the horizontal half of the bilinear interpolation (bilerp's `top` / `bot`) depends only on the pair of low-res
    // rows: it is kept across the output rows that share the pair (16 of them at scale 16) -- same operations, 8x
    // fewer gathers
    const Lerp lx = lerp_ac(min(X, W - 1), w, W);
    float top[3][C], bot[3][C];
    int have_i0 = -1;
    const int Y0 = blockIdx.y * REFINE_ROWS, Y1 = min(H, (int)(blockIdx.y + 1) * REFINE_ROWS);
    float sv[C], sn[C];                                              // this row's soft labels, and the next row's in flight
#pragma unroll
    for (int c = 0; c < C; ++c) sn[c] = (X < W) ? soft[((size_t)b * C + c) * HW + (size_t)Y0 * W + X] : 0.f;
  for (int Y = Y0; Y < Y1; ++Y) {
    float o[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { o[c] = 0.f; sv[c] = sn[c]; }
    if (X < W && Y + 1 < Y1) {
#pragma unroll
        for (int c = 0; c < C; ++c) sn[c] = soft[((size_t)b * C + c) * HW + (size_t)(Y + 1) * W + X];
    }
    if (X < W) {
        const Lerp ly = lerp_ac(Y, h, H);
        if (ly.i0 != have_i0) {
            have_i0 = ly.i0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const size_t off = ((size_t)b * C + c) * hw;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (!(views & (q == 0 ? 1 : 2))) { top[q][c] = bot[q][c] = 0.f; continue; }   // view not asked for
                    const float* p = (q == 0 ? sim : (q == 1 ? p1 : p2)) + off;
                    top[q][c] = __fadd_rn(__fmul_rn(lx.l0, p[ly.i0 * w + lx.i0]), __fmul_rn(lx.l1, p[ly.i0 * w + lx.i1]));
                    bot[q][c] = __fadd_rn(__fmul_rn(lx.l0, p[ly.i1 * w + lx.i0]), __fmul_rn(lx.l1, p[ly.i1 * w + lx.i1]));
                }
            }
        }
        float a[C], z1[C], z2[C];
        float ma = -INFINITY, m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            a[c] = __fadd_rn(__fmul_rn(ly.l0, top[0][c]), __fmul_rn(ly.l1, bot[0][c]));
            z1[c] = __fdiv_rn(__fadd_rn(__fmul_rn(ly.l0, top[1][c]), __fmul_rn(ly.l1, bot[1][c])), temp);
            z2[c] = __fdiv_rn(__fadd_rn(__fmul_rn(ly.l0, top[2][c]), __fmul_rn(ly.l1, bot[2][c])), temp);
            ma = fmaxf(ma, a[c]); m1 = fmaxf(m1, z1[c]); m2 = fmaxf(m2, z2[c]);
        }
        float sa = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            a[c] = expf(a[c] - ma); z1[c] = expf(z1[c] - m1); z2[c] = expf(z2[c] - m2);
            sa += a[c]; s1 += z1[c]; s2 += z2[c];
        }
        float pmax = 0.f, lmax = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            a[c] = a[c] / sa;                                        // softmax_T(simi, 1)
            z1[c] = (z1[c] / s1 + z2[c] / s2) * 0.5f;               // (softmax + softmax) * 0.5
            pmax = fmaxf(pmax, a[c]); lmax = fmaxf(lmax, z1[c]);
        }
        float supw[C];
        bool sup_on = false;
        if constexpr (SUP) {
            const long long id = spx.sup[(size_t)b * HW + (size_t)Y * W + X];
            // `ignored`: the superpixel with the batch's largest id (alignment.py:241-243); an id outside the table was
            // flagged by sup_max_kernel and is left alone
            sup_on = id >= 0 && id < spx.max_regions && id != (long long)spx.cnt[0];
            if (sup_on) {
                float m = -INFINITY, ssum = 0.f, smax = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    supw[c] = __fdiv_rn(fkey_inv(spx.tbl[((size_t)b * spx.max_regions + id) * C + c]), temp);
                    m = fmaxf(m, supw[c]);
                }
#pragma unroll
                for (int c = 0; c < C; ++c) { supw[c] = expf(supw[c] - m); ssum += supw[c]; }
#pragma unroll
                for (int c = 0; c < C; ++c) { supw[c] = supw[c] / ssum; smax = fmaxf(smax, supw[c]); }
#pragma unroll
                for (int c = 0; c < C; ++c) supw[c] = supw[c] / (smax + 1e-7f);
            }
        }
        float tot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            // mode 'all': both views; 'p' / 'l': `weight = 0 + view` (alignment.py:212,223,236); no view (mode 's' with
            // superpixels): ones (:257)
            float wgt = (views == 3) ? a[c] / (pmax + 1e-7f) + z1[c] / (lmax + 1e-7f)
                                     : ((views & 1) ? a[c] / (pmax + 1e-7f) : ((views & 2) ? z1[c] / (lmax + 1e-7f) : 1.f));
            if constexpr (SUP) wgt = sup_on ? wgt * supw[c] : wgt;      // :254-258
            float v = wgt * sv[c];
            o[c] = v;
            tot += v;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            o[c] = o[c] / (tot + 1e-7f);
            out[((size_t)b * C + c) * HW + (size_t)Y * W + X] = o[c];
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) omax[c] = fmaxf(omax[c], o[c]);
  }
    if (classmax) {
        __shared__ float red[4][C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float m = wave_max(omax[c]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = m;
        }
        __syncthreads();
        if (threadIdx.x < C) {
            float m = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]),
                            fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
            atomicMax((unsigned*)(classmax + b * C + threadIdx.x), __float_as_uint(fmaxf(m, 0.f)));
        }
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t rgda_label_refine_classmax_offset(int b, int c, int h, int w) {
    // sim | pc (c*k is not known here: pc lives after classmax) -> sim first, classmax second
    return align256((size_t)b * c * h * w * 4);
}
extern "C" size_t rgda_label_refine_workspace(int b, int c, int h, int w) {
    // sim[b][c][hw] | classmax[b][c] + flag | pstd[c] | pc[c][K<=4096]
    return rgda_label_refine_classmax_offset(b, c, h, w) + align256((size_t)b * c * 4 + 16) + align256((size_t)c * 4) +
           (size_t)c * 4096 * 4;
}

extern "C" int rgda_label_refine(const float* feat, const float* protos, const float* p1, const float* p2,
                                 const float* soft, float* out, int b, int k, int c, int h, int w, int H, int W,
                                 float temp, void* ws, size_t ws_bytes, rgda_stream_t stream) {
    return rgda_label_refine_views(feat, protos, p1, p2, soft, out, b, k, c, h, w, H, W, temp, 3, ws, ws_bytes, stream);
}

static int refine_run(const float* feat, const float* protos, const float* p1, const float* p2, const float* soft,
                      const long long* sup, int max_regions, float* out, int b, int k, int c, int h, int w, int H, int W,
                      float temp, int views, void* ws, size_t ws_bytes, rgda_stream_t stream);

extern "C" int rgda_label_refine_views(const float* feat, const float* protos, const float* p1, const float* p2,
                                       const float* soft, float* out, int b, int k, int c, int h, int w, int H, int W,
                                       float temp, int views, void* ws, size_t ws_bytes, rgda_stream_t stream) {
    if (views < 1 || views > 3) return RGDA_ERR_ARG;
    return refine_run(feat, protos, p1, p2, soft, nullptr, 0, out, b, k, c, h, w, H, W, temp, views, ws, ws_bytes, stream);
}

extern "C" size_t rgda_label_refine_sup_workspace(int b, int c, int h, int w, int max_regions) {
    // the workspace of rgda_label_refine | keys[b][max_regions][c] | largest id, flag
    return rgda_label_refine_workspace(b, c, h, w) + align256((size_t)b * max_regions * c * 4) + 256;
}
extern "C" size_t rgda_label_refine_sup_flag_offset(int b, int c, int h, int w, int max_regions) {
    return rgda_label_refine_workspace(b, c, h, w) + align256((size_t)b * max_regions * c * 4);
}

extern "C" int rgda_label_refine_sup(const float* feat, const float* protos, const float* p1, const float* p2,
                                     const float* soft, const int64_t* label_t_sup, float* out, int b, int k, int c,
                                     int h, int w, int H, int W, float temp, int views, int max_regions, void* ws,
                                     size_t ws_bytes, rgda_stream_t stream) {
    if (views < 0 || views > 3 || !label_t_sup || max_regions <= 0) return RGDA_ERR_ARG;
    return refine_run(feat, protos, p1, p2, soft, (const long long*)label_t_sup, max_regions, out, b, k, c, h, w, H, W, temp, views, ws,
                      ws_bytes, stream);
}

static int refine_run(const float* feat, const float* protos, const float* p1, const float* p2, const float* soft,
                      const long long* sup, int max_regions, float* out, int b, int k, int c, int h, int w, int H, int W,
                      float temp, int views, void* ws, size_t ws_bytes, rgda_stream_t stream) {
    if (!soft || !out || !ws) return RGDA_ERR_ARG;
    const bool pview = views & 1, lview = views & 2;
    if ((pview && (!feat || !protos)) || (lview && (!p1 || !p2))) return RGDA_ERR_ARG;
    if (c != 6) return RGDA_ERR_UNSUPPORTED;   // ISPRS: 6 classes (regda/datasets/isprsda.py:18-26)
    if (b <= 0 || (pview && (k < 2 || k > 4096 || (k & 3))) || h <= 0 || w <= 0 || H <= 0 || W <= 0 || !(temp > 0.f))
        return RGDA_ERR_ARG;
    if (ws_bytes < (sup ? rgda_label_refine_sup_workspace(b, c, h, w, max_regions) : rgda_label_refine_workspace(b, c, h, w)))
        return RGDA_ERR_WORKSPACE;
    hipStream_t st = to_stream(stream);
    char* base = (char*)ws;
    float* sim = (float*)base;
    size_t off = rgda_label_refine_classmax_offset(b, c, h, w);
    float* classmax = (float*)(base + off);
    off += align256((size_t)b * c * 4 + 16);
    float* pstd = (float*)(base + off);
    off += align256((size_t)c * 4);
    float* pc = (float*)(base + off);
    if (zero_bytes(classmax, (size_t)b * c * 4 + 16, stream) != RGDA_OK) return RGDA_ERR_LAUNCH;
    const int hw = h * w;
    if (pview) {
    proto_center_kernel<<<c, 256, 0, st>>>(protos, pc, pstd, k);
    RGDA_CHECK_LAUNCH();
    constexpr int PX = 32, SL = 16;
    size_t lds = ((size_t)6 * k + SL * PX * 7) * 4;
    dim3 g1(cdiv(hw, PX), b);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)pearson_sim_kernel<6, PX, SL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return RGDA_ERR_LAUNCH;
    pearson_sim_kernel<6, PX, SL><<<g1, PX * SL, lds, st>>>(feat, pc, pstd, sim, k, hw);
    RGDA_CHECK_LAUNCH();
    }
    dim3 g2(cdiv(W, 256), cdiv(H, REFINE_ROWS), b);
    if (!sup) {
        refine_apply_kernel<6><<<g2, 256, 0, st>>>(sim, p1, p2, soft, out, classmax, h, w, H, W, temp, views);
        RGDA_CHECK_LAUNCH();
        return RGDA_OK;
    }
    SupView sv;
    sv.sup = sup;
    sv.max_regions = max_regions;
    const size_t tbl_bytes = align256((size_t)b * max_regions * c * 4);
    sv.tbl = (unsigned*)(base + rgda_label_refine_workspace(b, c, h, w));
    sv.cnt = (int*)((char*)sv.tbl + tbl_bytes);
    if (zero_bytes(sv.tbl, tbl_bytes + 256, stream) != RGDA_OK) return RGDA_ERR_LAUNCH;     // key 0 = below every float
    const size_t HW = (size_t)H * W;
    sup_max_kernel<6><<<dim3(cdiv(HW, (size_t)256), b), 256, 0, st>>>(soft, sv, HW);
    RGDA_CHECK_LAUNCH();
    refine_apply_kernel<6, true><<<g2, 256, 0, st>>>(sim, p1, p2, soft, out, classmax, h, w, H, W, temp, views, sv);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// --------------------------------------------------------------------------------------
// SAM region map assembly   (regda/utils/local_region_homog.py:51-56, inside SAM.get_local_regions)
// --------------------------------------------------------------------------------------
// region[p] = 1 + the LAST mask index k (generator order) with areas[k] >= threshold and masks[k][p] != 0, else 0:
// what painting the kept masks one over the other leaves.  A thread owns 16 consecutive pixels (one 16-byte load per
// mask) and walks the masks from the last to the first until all 16 are decided.
__global__ void __launch_bounds__(256) masks_to_regions_kernel(const uint8_t* __restrict__ masks,
                                                               const long long* __restrict__ areas,
                                                               int* __restrict__ regions, int K, long long HW,
                                                               long long thr) {
    const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 16;
    if (p0 >= HW) return;
    int out[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) out[e] = 0;
    const bool full = p0 + 16 <= HW && (HW & 15) == 0;
    unsigned open_mask = 0xffffu;
    for (int k = K - 1; k >= 0 && open_mask; --k) {
        if (areas[k] < thr) continue;
        const uint8_t* m = masks + (size_t)k * HW + p0;
        __attribute__((aligned(16))) uint8_t v[16];
        if (full) {
            *(uint4*)v = *(const uint4*)m;
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = (p0 + e < HW) ? m[e] : 0;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (((open_mask >> e) & 1u) && v[e]) { out[e] = k + 1; open_mask &= ~(1u << e); }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (p0 + e < HW) regions[p0 + e] = out[e];
}

extern "C" int rgda_masks_to_regions(const uint8_t* masks, const int64_t* areas, int32_t* regions, int K, int64_t HW,
                                     int64_t area_threshold, rgda_stream_t stream) {
    if (!regions || K < 0 || HW <= 0 || (K > 0 && (!masks || !areas))) return RGDA_ERR_ARG;
    const long long threads = (HW + 15) / 16;
    masks_to_regions_kernel<<<(unsigned)cdiv(threads, 256), 256, 0, to_stream(stream)>>>(
        masks, (const long long*)areas, regions, K, HW, area_threshold);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// --------------------------------------------------------------------------------------
// update_prototype   (regda/gast/alignment.py:86-90, 300-327, 456-481)
// --------------------------------------------------------------------------------------
// one workgroup per low-res cell: histogram of the s x s block over C+1 classes
__global__ void __launch_bounds__(256) downscale_label_kernel(const int64_t* __restrict__ label, int64_t* label_ds,
                                                              float* cnt, int* flag, int h, int w, int s, int C,
                                                              int ignore_label, float min_ratio) {
    __shared__ int hist[17];
    const int cell = blockIdx.x;             // b*h*w + y*w + x
    const int b = cell / (h * w), yx = cell % (h * w), y = yx / w, x = yx % w;
    const int H = h * s, W = w * s;
    if (threadIdx.x < 17) hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t* src = label + ((size_t)b * H + (size_t)y * s) * W + (size_t)x * s;
    int bad = 0;
    for (int i = threadIdx.x; i < s * s; i += 256) {
        long long l = src[(size_t)(i / s) * W + (i % s)];
        if (l == ignore_label) l = C;
        if (l < 0 || l > C) { bad = 1; continue; }
        atomicAdd(&hist[(int)l], 1);
    }
    if (bad) atomicOr(flag, 2);
    __syncthreads();
    if (threadIdx.x == 0) {
        // avg_pool2d of the one-hot: count / s^2 in fp32; torch.max keeps the first maximum
        float inv = (float)(s * s);
        float best = __fdiv_rn((float)hist[0], inv);
        int arg = 0;
        for (int c = 1; c <= C; ++c) {
            float r = __fdiv_rn((float)hist[c], inv);
            if (r > best) { best = r; arg = c; }
        }
        long long o = arg;
        if (arg == C) o = ignore_label;
        if (best < min_ratio) o = ignore_label;
        label_ds[cell] = o;
        if (o != ignore_label) atomicAdd(&cnt[(int)o], 1.0f);
    }
}

// Fast form for scale 16 and C <= 6 (the RegDA configuration: 512 -> 32, six classes + ignore): one workgroup per ROW of
// low-res cells.  A thread owns two label columns of the 16-row band (16-byte loads, fully coalesced 4 KB rows) and
// counts its 32 labels in seven 9-bit fields of one 64-bit word; the eight threads of a cell add their words with DPP
// row shifts (no LDS, no atomics), the first of them picks the class.  (The one-workgroup-per-cell kernel above spends
// 94 us on 8192 tiny workgroups with LDS atomics and a single-thread tail; this is ~8 us for the same 17 MB.)
__global__ void __launch_bounds__(256) downscale_label16_kernel(const int64_t* __restrict__ label, int64_t* label_ds,
                                                                float* cnt, int* flag, int h, int w, int C,
                                                                int ignore_label, float min_ratio) {
    const int W = w * 16;
    const int b = blockIdx.y / h, y = blockIdx.y % h;
    const int col = (blockIdx.x * 256 + threadIdx.x) * 2;           // first of this thread's two columns
    unsigned long long packed = 0;
    int bad = 0;
    if (col < W) {
        const int64_t* src = label + ((size_t)b * h * 16 + (size_t)y * 16) * W + col;
        longlong2 rows[16];                                          // the whole band in flight at once
#pragma unroll
        for (int r = 0; r < 16; ++r) rows[r] = *(const longlong2*)(src + (size_t)r * W);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const longlong2 v = rows[r];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                long long l = k ? v.y : v.x;
                if (l == ignore_label) l = C;
                if (l < 0 || l > C) bad = 1;
                else packed += 1ull << (9 * (int)l);
            }
        }
    }
    if (bad) atomicOr(flag, 2);
    // sum over the 8 threads (16 columns) of a cell: lanes 8k .. 8k+7 of a wave
    unsigned lo = (unsigned)packed, hi = (unsigned)(packed >> 32);
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        unsigned long long other = ((unsigned long long)(unsigned)__shfl_xor((int)hi, o, 64) << 32) | (unsigned)__shfl_xor((int)lo, o, 64);
        packed += other;
        lo = (unsigned)packed; hi = (unsigned)(packed >> 32);
    }
    __shared__ int wg_cnt[8];                                        // cells of this workgroup per class
    if (threadIdx.x < 8) wg_cnt[threadIdx.x] = 0;
    __syncthreads();
    if ((threadIdx.x & 7) == 0 && col < W) {
        const int x = col / 16;
        // avg_pool2d of the one-hot: count / 256 in fp32; torch.max keeps the first maximum
        float best = __fdiv_rn((float)(packed & 511u), 256.f);
        int arg = 0;
        for (int c = 1; c <= C; ++c) {
            float r = __fdiv_rn((float)((packed >> (9 * c)) & 511u), 256.f);
            if (r > best) { best = r; arg = c; }
        }
        long long o = arg;
        if (arg == C) o = ignore_label;
        if (best < min_ratio) o = ignore_label;
        label_ds[((size_t)b * h + y) * w + x] = o;
        if (o != ignore_label) atomicAdd(&wg_cnt[(int)o], 1);
    }
    // one global add per class and workgroup (whole numbers: exact in fp32 in any order); per-cell adds on six words
    // were 8192 serialised memory-side atomics, most of this kernel's time
    __syncthreads();
    if (threadIdx.x < C && wg_cnt[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (float)wg_cnt[threadIdx.x]);
}

// one workgroup per feature channel k: its 256 threads walk that channel's hw values of every image (image after
// image), the six per-class sums are reduced wave -> LDS -> thread in a fixed order and leave with plain stores -- no
// atomics, bit-reproducible prototypes
template <int C>
__global__ void __launch_bounds__(256) proto_accum_kernel(const float* __restrict__ feat,
                                                          const int64_t* __restrict__ label_ds, float* sums, int K,
                                                          int hw, int B) {
    __shared__ float red[4][C];
    const int k = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* f = feat + ((size_t)b * K + k) * hw;
        const int64_t* l = label_ds + (size_t)b * hw;
        for (int p = t; p < hw; p += 256) {
            float v = f[p];
            int c0 = (int)l[p];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (c0 == c) ? v : 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float s = wave_sum(acc[c]);
        if (lane == 0) red[wave][c] = s;
    }
    __syncthreads();
    if (t < C) sums[(size_t)t * K + k] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
}

__global__ void __launch_bounds__(256) proto_finalize_kernel(float* protos, const float* __restrict__ sums,
                                                             const float* __restrict__ cnt, int K, int total,
                                                             float one_minus_decay, float decay) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float n = cnt[i / K];
    float old = protos[i];
    float local = __fdiv_rn(sums[i], __fadd_rn(n, 1e-7f));
    if (n < 1.f) local = old;
    protos[i] = __fadd_rn(__fmul_rn(one_minus_decay, local), __fmul_rn(decay, old));
}

extern "C" size_t rgda_proto_update_workspace(int c, int k) { return ((size_t)c * k + c) * 4 + 16; }

// The sufficient statistics of _compute_local_prototypes (alignment.py:300-327): f32 sums[c][k] = sum of the features of
// the pixels whose downscaled label is class c, f32 cnt[c] = their number, then one flag word.  They ADD over batches, so
// data-parallel ranks all-reduce (sum) the first c * k + c floats and every rank applies the same totals: the prototypes
// of the concatenated global batch, not an average of per-rank prototypes.
extern "C" int rgda_proto_stats(const float* feat, const int64_t* label, int64_t* label_ds, int b, int k, int c, int h, int w,
                                int scale, int ignore_label, float min_ratio, void* stats, size_t stats_bytes,
                                rgda_stream_t stream) {
    if (!feat || !label || !label_ds || !stats) return RGDA_ERR_ARG;
    if (c != 6) return RGDA_ERR_UNSUPPORTED;
    if (b <= 0 || k <= 0 || h <= 0 || w <= 0 || scale <= 1) return RGDA_ERR_ARG;
    if (stats_bytes < rgda_proto_update_workspace(c, k)) return RGDA_ERR_WORKSPACE;
    hipStream_t st = to_stream(stream);
    float* sums = (float*)stats;
    float* cnt = sums + (size_t)c * k;
    int* flag = (int*)(cnt + c);
    if (zero_bytes(stats, rgda_proto_update_workspace(c, k), stream) != RGDA_OK) return RGDA_ERR_LAUNCH;
    if (scale == 16 && c <= 6 && !(w & 1))
        downscale_label16_kernel<<<dim3(cdiv(w * 8, 256), b * h), 256, 0, st>>>(label, label_ds, cnt, flag, h, w, c, ignore_label, min_ratio);
    else
        downscale_label_kernel<<<b * h * w, 256, 0, st>>>(label, label_ds, cnt, flag, h, w, scale, c, ignore_label, min_ratio);
    RGDA_CHECK_LAUNCH();
    proto_accum_kernel<6><<<k, 256, 0, st>>>(feat, label_ds, sums, k, h * w, b);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// local = sums / (cnt + 1e-7), the old prototype where cnt < 1 (alignment.py:318-321), then the EMA (:435-438)
extern "C" int rgda_proto_apply(float* protos, const void* stats, int c, int k, float decay, rgda_stream_t stream) {
    if (!protos || !stats) return RGDA_ERR_ARG;
    if (c <= 0 || k <= 0 || !(decay > 0.f && decay < 1.f)) return RGDA_ERR_ARG;
    const float* sums = (const float*)stats;
    const float* cnt = sums + (size_t)c * k;
    float omd = (float)(1.0 - (double)decay);
    proto_finalize_kernel<<<cdiv((long long)c * k, 256), 256, 0, to_stream(stream)>>>(protos, sums, cnt, k, c * k, omd, decay);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

extern "C" int rgda_proto_update(const float* feat, const int64_t* label, float* protos, int64_t* label_ds, int b,
                                 int k, int c, int h, int w, int scale, int ignore_label, float min_ratio,
                                 float decay, void* ws, size_t ws_bytes, rgda_stream_t stream) {
    if (!protos || !(decay > 0.f && decay < 1.f)) return RGDA_ERR_ARG;
    const int rc = rgda_proto_stats(feat, label, label_ds, b, k, c, h, w, scale, ignore_label, min_ratio, ws, ws_bytes, stream);
    if (rc != RGDA_OK) return rc;
    return rgda_proto_apply(protos, ws, c, k, decay, stream);
}

// --------------------------------------------------------------------------------------
// loss_calc(multi=True) + CrossEntropy  (regda/utils/tools.py:240-254; regda/gast/balance.py:88-101)
// forward and d(loss)/d(low-res logits) in one pass over the full-resolution pixels:
//   stage 1 (one workgroup per output row): bilinear(ac=True) logits -> log-softmax -> loss
//           partial; per-pixel gradient kept in LDS and contracted with the horizontal
//           interpolation weights -> T[b][Y][head][c][x]
//   stage 2: vertical contraction T -> g[head][b][c][y][x]; deterministic loss reduction.
// --------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) upce_row_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                       const int64_t* __restrict__ label,
                                                       const float* __restrict__ class_weight, float* partial,
                                                       float* T, int h, int w, int H, int W, int ignore_label,
                                                       float gscale, int want_grad) {
    extern __shared__ float lds[];
    // rows[2 heads][C][2][w] | G[2][C][W]
    float* rows = lds;
    float* G = lds + 2 * C * 2 * w;
    // the horizontal interpolation of every output column, kept for the contraction below: recomputing lerp_ac there (an
    // IEEE divide each) cost 13 000 calls per workgroup -- more than the per-pixel loss arithmetic
    int* lxi = (int*)(G + (want_grad ? 2 * C * W : 0));
    float* lxl = (float*)(lxi + W);
    const int b = blockIdx.y, Y = blockIdx.x;
    const int hw = h * w;
    Lerp ly = lerp_ac(Y, h, H);
    for (int i = threadIdx.x; i < 2 * C * 2 * w; i += 256) {
        int x = i % w, r = (i / w) & 1, c = (i / (2 * w)) % C, hd = i / (2 * w * C);
        const float* p = hd ? p2 : p1;
        rows[i] = p[((size_t)b * C + c) * hw + (r ? ly.i1 : ly.i0) * w + x];
    }
    __syncthreads();
    float lsum0 = 0.f, lsum1 = 0.f;
    for (int X = threadIdx.x; X < W; X += 256) {
        Lerp lx = lerp_ac(X, w, W);
        lxi[X] = lx.i0;
        lxl[X] = lx.l1;
        long long lab = label[((size_t)b * H + Y) * W + X];
        bool valid = lab != ignore_label;
        int li = valid ? (int)lab : 0;
#pragma unroll
        for (int hd = 0; hd < 2; ++hd) {
            float z[C], m = -INFINITY;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float* r = rows + ((hd * C + c) * 2) * w;
                float top = __fadd_rn(__fmul_rn(lx.l0, r[lx.i0]), __fmul_rn(lx.l1, r[lx.i1]));
                float bot = __fadd_rn(__fmul_rn(lx.l0, r[w + lx.i0]), __fmul_rn(lx.l1, r[w + lx.i1]));
                z[c] = __fadd_rn(__fmul_rn(ly.l0, top), __fmul_rn(ly.l1, bot));
                m = fmaxf(m, z[c]);
            }
            float se = 0.f, e[C];
#pragma unroll
            for (int c = 0; c < C; ++c) { e[c] = expf(z[c] - m); se += e[c]; }
            float lse = m + logf(se);
            float wgt = valid ? (class_weight ? class_weight[hd * C + li] : 1.f) : 0.f;
            float zl = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) zl = (c == li) ? z[c] : zl;
            float lp = valid ? (lse - zl) * wgt : 0.f;
            if (hd == 0) lsum0 += lp; else lsum1 += lp;
            if (want_grad) {
                float gs = wgt * gscale;
#pragma unroll
                for (int c = 0; c < C; ++c) G[(hd * C + c) * W + X] = (e[c] / se - ((c == li) ? 1.f : 0.f)) * gs;
            }
        }
    }
    __shared__ float red[2][4];
    lsum0 = wave_sum(lsum0); lsum1 = wave_sum(lsum1);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lsum0; red[1][threadIdx.x >> 6] = lsum1; }
    __syncthreads();
    if (threadIdx.x < 2)
        partial[((size_t)b * H + Y) * 2 + threadIdx.x] =
            red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (!want_grad) return;
    // horizontal contraction: T[hd][c][x] = sum_X G[hd][c][X] * Rx[X][x]
    const float inv_scale = (w > 1) ? (float)(W - 1) / (float)(w - 1) : 0.f;
    for (int o = threadIdx.x; o < 2 * C * w; o += 256) {
        int x = o % w, hc = o / w;
        int lo = (w > 1) ? max(0, (int)floorf((float)(x - 1) * inv_scale) - 1) : 0;
        int hi = (w > 1) ? min(W - 1, (int)ceilf((float)(x + 1) * inv_scale) + 1) : W - 1;
        float acc = 0.f;
        for (int X = lo; X <= hi; ++X) {
            const int i0 = lxi[X], i1 = i0 + ((i0 < w - 1) ? 1 : 0);
            const float l1 = lxl[X], l0 = __fsub_rn(1.f, l1);
            float wt = ((i0 == x) ? l0 : 0.f) + ((i1 == x) ? l1 : 0.f);
            acc += wt * G[hc * W + X];
        }
        T[(((size_t)b * H + Y) * 2 * C + hc) * w + x] = acc;
    }
}

template <int C>
__global__ void __launch_bounds__(256) upce_col_kernel(const float* __restrict__ T, float* g1, float* g2, int b_n,
                                                       int h, int w, int H) {
    int i = blockIdx.x * 256 + threadIdx.x;           // over [hd][b][c][y][x]
    int total = 2 * b_n * C * h * w;
    if (i >= total) return;
    int x = i % w, y = (i / w) % h, c = (i / (w * h)) % C, b = (i / (w * h * C)) % b_n, hd = i / (w * h * C * b_n);
    const float inv_scale = (h > 1) ? (float)(H - 1) / (float)(h - 1) : 0.f;
    int lo = (h > 1) ? max(0, (int)floorf((float)(y - 1) * inv_scale) - 1) : 0;
    int hi = (h > 1) ? min(H - 1, (int)ceilf((float)(y + 1) * inv_scale) + 1) : H - 1;
    float acc = 0.f;
    for (int Y = lo; Y <= hi; ++Y) {
        Lerp ly = lerp_ac(Y, h, H);
        float wt = ((ly.i0 == y) ? ly.l0 : 0.f) + ((ly.i1 == y) ? ly.l1 : 0.f);
        acc += wt * T[(((size_t)b * H + Y) * 2 * C + hd * C + c) * w + x];
    }
    float* g = hd ? g2 : g1;
    g[(((size_t)b * C + c) * h + y) * w + x] = acc;
}

__global__ void __launch_bounds__(256) upce_loss_kernel(const float* __restrict__ partial, float* loss, int n,
                                                        double inv_npix) {
    // deterministic: fixed strided order, double accumulation
    __shared__ double red[2][256];
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { s0 += partial[2 * i]; s1 += partial[2 * i + 1]; }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float l0 = (float)(red[0][0] * inv_npix), l1 = (float)(red[1][0] * inv_npix);   // torch.mean per head
        loss[0] = (l0 + l1) / 2.f;                                                       // loss / num (tools.py:252)
    }
}

extern "C" size_t rgda_upsample_ce_workspace(int b, int c, int h, int w, int H, int W) {
    (void)W;
    return align256((size_t)b * H * 2 * 4) + (size_t)b * H * 2 * c * w * 4;
}

extern "C" int rgda_upsample_ce(const float* p1, const float* p2, const int64_t* label, const float* class_weight,
                                float* loss, float* g1, float* g2, int b, int c, int h, int w, int H, int W,
                                int ignore_label, void* ws, size_t ws_bytes, rgda_stream_t stream) {
    if (!p1 || !p2 || !label || !loss || !ws || ((g1 == nullptr) != (g2 == nullptr))) return RGDA_ERR_ARG;
    if (c != 6) return RGDA_ERR_UNSUPPORTED;
    if (b <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return RGDA_ERR_ARG;
    if (ws_bytes < rgda_upsample_ce_workspace(b, c, h, w, H, W)) return RGDA_ERR_WORKSPACE;
    hipStream_t st = to_stream(stream);
    float* partial = (float*)ws;
    float* T = (float*)((char*)ws + align256((size_t)b * H * 2 * 4));
    const int want = g1 != nullptr;
    double npix = (double)b * H * W;
    float gscale = (float)(0.5 / npix);
    size_t lds = ((size_t)2 * 6 * 2 * w + (want ? (size_t)2 * 6 * W : 0) + (size_t)2 * W) * 4;
    if (lds > 150 * 1024) return RGDA_ERR_UNSUPPORTED;
    dim3 g(H, b);
    upce_row_kernel<6><<<g, 256, lds, st>>>(p1, p2, label, class_weight, partial, T, h, w, H, W, ignore_label, gscale, want);
    RGDA_CHECK_LAUNCH();
    if (want) {
        upce_col_kernel<6><<<cdiv((long long)2 * b * 6 * h * w, 256), 256, 0, st>>>(T, g1, g2, b, h, w, H);
        RGDA_CHECK_LAUNCH();
    }
    upce_loss_kernel<<<1, 256, 0, st>>>(partial, loss, b * H, 1.0 / npix);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// --------------------------------------------------------------------------------------
// teacher probabilities   (regda/models/Encoder.py:152-155)
// --------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) teacher_probs_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                            float* __restrict__ probs, int h, int w, int H, int W) {
    const int b = blockIdx.z, Y = blockIdx.y, X = blockIdx.x * 256 + threadIdx.x;
    if (X >= W) return;
    const int hw = h * w;
    const size_t HW = (size_t)H * W;
    Lerp ly = lerp_ac(Y, h, H), lx = lerp_ac(X, w, W);
    float z1[C], z2[C], m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        size_t off = ((size_t)b * C + c) * hw;
        z1[c] = bilerp(p1 + off, w, ly, lx);
        z2[c] = bilerp(p2 + off, w, ly, lx);
        m1 = fmaxf(m1, z1[c]); m2 = fmaxf(m2, z2[c]);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { z1[c] = expf(z1[c] - m1); z2[c] = expf(z2[c] - m2); s1 += z1[c]; s2 += z2[c]; }
#pragma unroll
    for (int c = 0; c < C; ++c)
        probs[((size_t)b * C + c) * HW + (size_t)Y * W + X] = (z1[c] / s1 + z2[c] / s2) / 2.f;
}

extern "C" int rgda_teacher_probs(const float* p1, const float* p2, float* probs, int b, int c, int h, int w, int H,
                                  int W, rgda_stream_t stream) {
    if (!p1 || !p2 || !probs || b <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return RGDA_ERR_ARG;
    if (c != 6) return RGDA_ERR_UNSUPPORTED;
    dim3 g(cdiv(W, 256), H, b);
    teacher_probs_kernel<6><<<g, 256, 0, to_stream(stream)>>>(p1, p2, probs, h, w, H, W);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// --------------------------------------------------------------------------------------
// ClassBalance._local_freq counts  (regda/gast/balance.py:45-53)
// --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) class_count_kernel(const int64_t* __restrict__ label, int* cnt, long long n,
                                                          int C) {
    __shared__ int h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        long long l = label[i];
        if (l >= 0 && l < C) atomicAdd(&h[(int)l], 1);
    }
    __syncthreads();
    if (threadIdx.x < C && h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], h[threadIdx.x]);
}

extern "C" int rgda_class_count(const int64_t* label, int32_t* cnt, int64_t n, int c, rgda_stream_t stream) {
    if (!label || !cnt || n < 0 || c <= 0 || c > 16) return RGDA_ERR_ARG;
    if (n == 0) return RGDA_OK;
    class_count_kernel<<<min(cdiv(n, 256 * 8), 2048), 256, 0, to_stream(stream)>>>(label, cnt, n, c);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}
