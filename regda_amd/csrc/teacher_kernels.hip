// Teacher / pseudo-label harness kernels: test-time-augmentation views, sliding-window
// accumulation and the align_corners=True resize of regda/utils/tools.py:61-97,132-152 and
// regda/gast/pseudo_generation.py:96-141.  All fp32 NCHW, HBM-bound permutations / stencils.
#include "common.h"

// One dihedral view.  With rot90 as torch.rot90(x, 1, (2, 3)): R(x)[i][j] = x[j][w-1-i] (input h x w -> output
// w x h) and F(x)[y][x] = x[y][w-1-x]:
//   flip_first = 1: dst = R^k(F^f(src))     (ttach Compose order of augment_image: HorizontalFlip, then Rotate90)
//   flip_first = 0: dst = F^f(R^k(src))     (deaugment_mask: inverse rotation first, then the flip)
// dst (+)= scale * view; output is Hs x Ws for even k, Ws x Hs for odd k.
__global__ void __launch_bounds__(256) dihedral_kernel(const float* __restrict__ src, float* __restrict__ dst, long long planes,
                                                       int Hs, int Ws, int hflip, int k, int flip_first, float scale,
                                                       int accumulate) {
    const int Ho = (k & 1) ? Ws : Hs, Wo = (k & 1) ? Hs : Ws;
    const long long total = planes * Ho * Wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
        const long long p = i / ((long long)Wo * Ho);
        int y = yo, x = xo, h = Ho, w = Wo;
        if (!flip_first && hflip) x = w - 1 - x;          // dst = F(t): undo the flip first
        for (int s = 0; s < k; ++s) {                      // cur = R(prev), prev is w x h: prev[x][hp_w-1-y]
            const int wp = h;
            const int py = x, px = wp - 1 - y;
            y = py; x = px;
            const int t = h; h = w; w = t;
        }
        if (flip_first && hflip) x = w - 1 - x;
        const float v = scale * src[(p * Hs + y) * Ws + x];
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

extern "C" int rgda_dihedral_nchw(const float* src, float* dst, int N, int C, int Hs, int Ws, int hflip, int rot_k,
                                  int flip_first, float scale, int accumulate, rgda_stream_t stream) {
    if (!src || !dst || N <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || rot_k < 0 || rot_k > 3) return RGDA_ERR_ARG;
    const long long total = (long long)N * C * Hs * Ws;
    int grid = (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
    dihedral_kernel<<<grid, 256, 0, to_stream(stream)>>>(src, dst, (long long)N * C, Hs, Ws, hflip ? 1 : 0, rot_k,
                                                         flip_first ? 1 : 0, scale, accumulate ? 1 : 0);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// tile[n][c][y][x] = full[n][c][y1 + y][x1 + x] inside the h x w window, 0 in the padding up to Th x Tw
// (tools.py:79-80: the crop image[:, :, y1:y2, x1:x2] + pad_image)
__global__ void __launch_bounds__(256) window_crop_kernel(const float* __restrict__ full, float* __restrict__ tile,
                                                          long long planes, int Hf, int Wf, int y1, int x1, int h, int w,
                                                          int Th, int Tw) {
    const long long total = planes * Th * Tw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % Tw), y = (int)((i / Tw) % Th);
        const long long p = i / ((long long)Tw * Th);
        tile[i] = (y < h && x < w) ? full[(p * Hf + y1 + y) * Wf + x1 + x] : 0.f;
    }
}

// full[:, :, y1:y1+h, x1:x1+w] += tile[:, :, :h, :w]; count[:, 0, window] += 1   (tools.py:91-93)
__global__ void __launch_bounds__(256) window_acc_kernel(const float* __restrict__ tile, float* __restrict__ full,
                                                         float* __restrict__ count, int N, int C, int Hf, int Wf, int y1,
                                                         int x1, int h, int w, int Th, int Tw) {
    const long long total = (long long)N * (C + 1) * h * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long long pc = i / ((long long)w * h);
        const int c = (int)(pc % (C + 1)), n = (int)(pc / (C + 1));
        if (c == C) count[((long long)n * Hf + y1 + y) * Wf + x1 + x] += 1.f;
        else full[(((long long)n * C + c) * Hf + y1 + y) * Wf + x1 + x] += tile[(((long long)n * C + c) * Th + y) * Tw + x];
    }
}

// full[n][c][p] /= count[n][0][p]   (tools.py:95)
__global__ void __launch_bounds__(256) window_norm_kernel(float* __restrict__ full, const float* __restrict__ count, int N,
                                                          int C, long long HW) {
    const long long total = (long long)N * C * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % HW, n = i / (HW * C);
        full[i] = __fdiv_rn(full[i], count[n * HW + p]);
    }
}

static inline int grid_for(long long total) { long long g = (total + 255) / 256; return (int)(g > 65535 ? 65535 : (g < 1 ? 1 : g)); }

extern "C" int rgda_window_crop(const float* full, float* tile, int N, int C, int Hf, int Wf, int y1, int x1, int h, int w,
                                int Th, int Tw, rgda_stream_t stream) {
    if (!full || !tile || N <= 0 || C <= 0 || Hf <= 0 || Wf <= 0 || y1 < 0 || x1 < 0 || h <= 0 || w <= 0 || h > Th || w > Tw ||
        y1 + h > Hf || x1 + w > Wf)
        return RGDA_ERR_ARG;
    window_crop_kernel<<<grid_for((long long)N * C * Th * Tw), 256, 0, to_stream(stream)>>>(full, tile, (long long)N * C, Hf, Wf,
                                                                                              y1, x1, h, w, Th, Tw);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

extern "C" int rgda_window_accumulate(const float* tile, float* full, float* count, int N, int C, int Hf, int Wf, int y1,
                                      int x1, int h, int w, int Th, int Tw, rgda_stream_t stream) {
    if (!full || !tile || !count || N <= 0 || C <= 0 || Hf <= 0 || Wf <= 0 || y1 < 0 || x1 < 0 || h <= 0 || w <= 0 || h > Th ||
        w > Tw || y1 + h > Hf || x1 + w > Wf)
        return RGDA_ERR_ARG;
    window_acc_kernel<<<grid_for((long long)N * (C + 1) * h * w), 256, 0, to_stream(stream)>>>(tile, full, count, N, C, Hf, Wf, y1,
                                                                                                 x1, h, w, Th, Tw);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

extern "C" int rgda_window_normalise(float* full, const float* count, int N, int C, int Hf, int Wf, rgda_stream_t stream) {
    if (!full || !count || N <= 0 || C <= 0 || Hf <= 0 || Wf <= 0) return RGDA_ERR_ARG;
    window_norm_kernel<<<grid_for((long long)N * C * Hf * Wf), 256, 0, to_stream(stream)>>>(full, count, N, C, (long long)Hf * Wf);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// F.interpolate(mode='bilinear', align_corners=True): src = dst * (in - 1) / (out - 1)   (pseudo_generation.py:135)
// (ATen's area_pixel_compute_source_index in fp32: scale = (in-1)/(out-1) computed in fp32, lambda = src - floor)
__global__ void __launch_bounds__(256) resize_ac_kernel(const float* __restrict__ src, float* __restrict__ dst, long long planes,
                                                        int h, int w, int H, int W, float sy, float sx) {
    // hipcc contracts a*b - c into one fma by default, and HIP's __fmul_rn / __fsub_rn are inline functions whose
    // bodies carry that default: fma(sy, Y, -y0) is MORE exact than ATen's fl(fl(sy*Y) - y0) and moves the
    // interpolation weight by up to an ulp of the source coordinate (1.6e-6 measured).  Plain operators, contraction off.
#pragma clang fp contract(off)
    const long long total = planes * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int X = (int)(i % W), Y = (int)((i / W) % H);
        const long long p = i / ((long long)W * H);
        const float fy = sy * (float)Y, fx = sx * (float)X;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + ((y0 < h - 1) ? 1 : 0), x1 = x0 + ((x0 < w - 1) ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float my = 1.f - ly, mx = 1.f - lx;
        const float* s = src + p * h * w;
        const float top = mx * s[y0 * w + x0] + lx * s[y0 * w + x1];       // ATen's association
        const float bot = mx * s[y1 * w + x0] + lx * s[y1 * w + x1];
        dst[i] = my * top + ly * bot;
    }
}

extern "C" int rgda_resize_bilinear_ac(const float* src, float* dst, int N, int C, int h, int w, int H, int W,
                                       rgda_stream_t stream) {
    if (!src || !dst || N <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return RGDA_ERR_ARG;
    // the scale exactly as ATen computes it (area_pixel_compute_scale, align_corners): a float division on the host
    const float sy = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
    resize_ac_kernel<<<grid_for((long long)N * C * H * W), 256, 0, to_stream(stream)>>>(src, dst, (long long)N * C, h, w, H, W,
                                                                                         sy, sx);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// tnf.pad(img, (0, 0, top, bottom)): rows only; dst has h + top + bottom rows, negative values crop (tools.py:57)
__global__ void __launch_bounds__(256) pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long long planes,
                                                       int h, int w, int top, int Ho) {
    const long long total = planes * Ho * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % Ho);
        const long long p = i / ((long long)w * Ho);
        const int ys = y - top;
        dst[i] = (ys >= 0 && ys < h) ? src[(p * h + ys) * w + x] : 0.f;
    }
}

extern "C" int rgda_pad_rows_nchw(const float* src, float* dst, int N, int C, int h, int w, int top, int bottom,
                                  rgda_stream_t stream) {
    if (!src || !dst || N <= 0 || C <= 0 || h <= 0 || w <= 0 || h + top + bottom <= 0) return RGDA_ERR_ARG;
    const int Ho = h + top + bottom;
    pad_rows_kernel<<<grid_for((long long)N * C * Ho * w), 256, 0, to_stream(stream)>>>(src, dst, (long long)N * C, h, w, top, Ho);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// ------------------------------------------------------------------ evaluation path
// cls.argmax(dim=1) (regda/utils/eval.py:43): first maximum wins, NaN-free inputs (probabilities)
__global__ void __launch_bounds__(256) argmax_nchw_kernel(const float* __restrict__ probs, int64_t* __restrict__ out, int N,
                                                          int C, long long HW) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / HW, p = i % HW;
        const float* s = probs + n * C * HW + p;
        float best = s[0];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = s[(long long)c * HW];
            if (v > best) { best = v; arg = c; }
        }
        out[i] = arg;
    }
}

extern "C" int rgda_argmax_nchw(const float* probs, int64_t* out, int N, int C, int64_t HW, rgda_stream_t stream) {
    if (!probs || !out || N <= 0 || C <= 0 || HW <= 0) return RGDA_ERR_ARG;
    argmax_nchw_kernel<<<grid_for((long long)N * HW), 256, 0, to_stream(stream)>>>(probs, out, N, C, HW);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}

// cm[t][p] += #{i : y_true[i] == t >= 0, y_pred[i] == p}  (eval.py:45-50: mask = cls_gt >= 0, then the confusion matrix
// PixelMetric accumulates).  LDS histogram per workgroup, one 64-bit atomic per nonzero cell per workgroup.
__global__ void __launch_bounds__(256) confusion_kernel(const int64_t* __restrict__ yt, const int64_t* __restrict__ yp,
                                                        unsigned long long* __restrict__ cm, long long n, int C, int* flag) {
    extern __shared__ unsigned int hist[];       // [C*C]
    for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0;
    __syncthreads();
    int bad = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long t = yt[i], p = yp[i];
        if (t < 0) continue;
        if (t >= C || p < 0 || p >= C) { bad = 1; continue; }
        atomicAdd(&hist[(int)t * C + (int)p], 1u);
    }
    if (bad) atomicOr(flag, 1);
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256)
        if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
}

extern "C" int rgda_confusion_accumulate(const int64_t* y_true, const int64_t* y_pred, int64_t* cm, int* flag, int64_t n,
                                         int C, rgda_stream_t stream) {
    if (!y_true || !y_pred || !cm || !flag || n < 0 || C <= 0 || C > 64) return RGDA_ERR_ARG;
    if (n == 0) return RGDA_OK;
    int grid = grid_for(n);
    if (grid > 1024) grid = 1024;               // each workgroup counts < 2^32 elements into its 32-bit LDS cells
    confusion_kernel<<<grid, 256, (size_t)C * C * 4, to_stream(stream)>>>(y_true, y_pred, (unsigned long long*)cm, n, C, flag);
    RGDA_CHECK_LAUNCH();
    return RGDA_OK;
}
