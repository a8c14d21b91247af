// Spatial-transformer warp (bilinear grid_sample), loss windows (SSIM / LNCC /
// smoothness), root-sum-of-squares and sensitivity normalisation for gfx950.
// All HBM/latency-bound: loss windows are staged in LDS per tile and reduced
// with wavefront shuffles; every scalar loss is reduced in two deterministic
// stages (per-workgroup partials in the caller's workspace, then one wave in
// double precision), never with floating-point atomics.
#include "san_common.h"

namespace {

constexpr int kThreads = 256;

// ------------------------------------------------------------------ sampler
__device__ __forceinline__ float reflect_coord(float x, float lo2, float hi2) {
    // reflect x into [lo2/2, hi2/2] (ATen reflect_coordinates with doubled bounds)
    if (lo2 == hi2) return 0.f;
    const float mn = lo2 * 0.5f;
    const float span = (hi2 - lo2) * 0.5f;
    x = fabsf(x - mn);
    const float extra = fmodf(x, span);
    const int flips = (int)floorf(x / span);
    return (flips & 1) ? (span - extra + mn) : (extra + mn);
}

__device__ __forceinline__ float2 operator*(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
__device__ __forceinline__ float2& operator+=(float2& a, float2 b) {
    a.x += b.x;
    a.y += b.y;
    return a;
}
template <typename T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ float zero_of<float>() { return 0.f; }
template <> __device__ __forceinline__ float2 zero_of<float2>() { return make_float2(0.f, 0.f); }

template <typename T>
__device__ __forceinline__ T sample_bilinear(const T* __restrict__ img, int H, int W, float gx, float gy, int padding) {
    float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    if (padding == 1) {
        ix = reflect_coord(ix, -1.f, 2.f * (float)W - 1.f);
        ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
        iy = reflect_coord(iy, -1.f, 2.f * (float)H - 1.f);
        iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy;
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x1 >= 0 && x1 < W;
    const bool yin0 = y0 >= 0 && y0 < H, yin1 = y1 >= 0 && y1 < H;
    T v = zero_of<T>();
    if (yin0 && xin0) v += img[y0 * W + x0] * (wx0 * wy0);
    if (yin0 && xin1) v += img[y0 * W + x1] * (wx1 * wy0);
    if (yin1 && xin0) v += img[y1 * W + x0] * (wx0 * wy1);
    if (yin1 && xin1) v += img[y1 * W + x1] * (wx1 * wy1);
    return v;
}

// grid.x over pixels, grid.y = n
__global__ void __launch_bounds__(kThreads)
warp_kernel(const float* __restrict__ img, const float* __restrict__ offset, float* __restrict__ out,
            float* __restrict__ grid_out, int C, int H, int W, int padding) {
    const int n = blockIdx.y;
    const int HW = H * W;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HW; i += gridDim.x * kThreads) {
        const int yi = i / W, xj = i - yi * W;
        const float bx = (float)(2 * xj + 1) / (float)W - 1.f;
        const float by = (float)(2 * yi + 1) / (float)H - 1.f;
        const float gx = bx + offset[((size_t)n * 2 + 0) * HW + i];
        const float gy = by + offset[((size_t)n * 2 + 1) * HW + i];
        if (grid_out) *reinterpret_cast<float2*>(grid_out + ((size_t)n * HW + i) * 2) = make_float2(gx, gy);
        for (int c = 0; c < C; ++c)
            out[((size_t)n * C + c) * HW + i] = sample_bilinear(img + ((size_t)n * C + c) * HW, H, W, gx, gy, padding);
    }
}

__global__ void __launch_bounds__(kThreads)
grid_sample_kernel(const float* __restrict__ img, const float* __restrict__ grid, float* __restrict__ out, int C,
                   int H, int W, int HO, int WO, int padding) {
    const int n = blockIdx.y;
    const int HWo = HO * WO;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HWo; i += gridDim.x * kThreads) {
        const float2 g = *reinterpret_cast<const float2*>(grid + ((size_t)n * HWo + i) * 2);
        for (int c = 0; c < C; ++c)
            out[((size_t)n * C + c) * HWo + i] =
                sample_bilinear(img + ((size_t)n * C + c) * H * W, H, W, g.x, g.y, padding);
    }
}

// interleaved complex planes (the reference samples real and imaginary parts with the same grid, augment.py:62-63)
__global__ void __launch_bounds__(kThreads)
grid_sample_c_kernel(const float2* __restrict__ img, const float* __restrict__ grid, float2* __restrict__ out, int C,
                     int H, int W, int HO, int WO, int padding) {
    const int n = blockIdx.y;
    const int HWo = HO * WO;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HWo; i += gridDim.x * kThreads) {
        const float2 g = *reinterpret_cast<const float2*>(grid + ((size_t)n * HWo + i) * 2);
        for (int c = 0; c < C; ++c)
            out[((size_t)n * C + c) * HWo + i] =
                sample_bilinear(img + ((size_t)n * C + c) * H * W, H, W, g.x, g.y, padding);
    }
}

// ------------------------------------------------------------- augmentation grid
// grid[n,i,j] = M_n @ (x_j, y_i, 1) [+ bicubic upsample of the CG x CG control offsets]  (augment.py:7-48).
// Bicubic = ATen's upsample_bicubic2d, align_corners=False: src = (dst + 0.5) * in/out - 0.5 (not
// clamped), Keys kernel A = -0.75, taps floor(src)-1 .. +2 clamped to the border.
__device__ __forceinline__ void cubic_w(float t, float (&w)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, u = 1.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * u - (A + 3.f)) * u * u + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ void __launch_bounds__(kThreads)
augment_grid_kernel(const float* __restrict__ affine, const float* __restrict__ ctrl, float* __restrict__ grid, int H,
                    int W, int CG) {
    extern __shared__ float cg[];            // [2][CG][CG] of this sample
    const int n = blockIdx.y;
    if (ctrl) {
        for (int i = threadIdx.x; i < 2 * CG * CG; i += kThreads) cg[i] = ctrl[(size_t)n * 2 * CG * CG + i];
        __syncthreads();
    }
    const float* m = affine + (size_t)n * 6;
    const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5];
    const float sy = (float)CG / (float)H, sx = (float)CG / (float)W;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < H * W; i += gridDim.x * kThreads) {
        const int yi = i / W, xj = i - yi * W;
        const float bx = (float)(2 * xj + 1) / (float)W - 1.f;
        const float by = (float)(2 * yi + 1) / (float)H - 1.f;
        float gx = m00 * bx + m01 * by + m02;
        float gy = m10 * bx + m11 * by + m12;
        if (ctrl) {
            const float srcy = ((float)yi + 0.5f) * sy - 0.5f, srcx = ((float)xj + 0.5f) * sx - 0.5f;
            const float fy = floorf(srcy), fx = floorf(srcx);
            float wy[4], wx[4];
            cubic_w(srcy - fy, wy);
            cubic_w(srcx - fx, wx);
            float ox = 0.f, oy = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int r = min(max((int)fy - 1 + a, 0), CG - 1);
                float rx = 0.f, ry = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int c = min(max((int)fx - 1 + b, 0), CG - 1);
                    rx += cg[r * CG + c] * wx[b];
                    ry += cg[CG * CG + r * CG + c] * wx[b];
                }
                ox += rx * wy[a];
                oy += ry * wy[a];
            }
            gx += ox;
            gy += oy;
        }
        *reinterpret_cast<float2*>(grid + ((size_t)n * H * W + i) * 2) = make_float2(gx, gy);
    }
}

// ------------------------------------------------------------- validation metrics
// One workgroup per image: sum of squared / absolute errors, sum of gt^2 (double) and the bins x bins
// joint histogram over [0,1]^2 in LDS (integer atomics: deterministic), from which the mutual
// information sum xlogy(Pxy, Pxy) - xlogy(Pxy, Px*Py) is evaluated in double (metrics.py:23-35,55-69).
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = san_wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(kThreads)
image_metrics_kernel(const float* __restrict__ gt, const float* __restrict__ pred, double* __restrict__ out, int hw,
                     int bins) {
    extern __shared__ int hist[];                    // [bins*bins] counts, then [2*bins] doubles behind them
    __shared__ double red[4];
    const int n = blockIdx.x;
    const float* g = gt + (size_t)n * hw;
    const float* p = pred + (size_t)n * hw;
    const int nb = bins * bins;
    for (int i = threadIdx.x; i < nb; i += kThreads) hist[i] = 0;
    __syncthreads();
    double sse = 0.0, sae = 0.0, sgg = 0.0;
    for (int i = threadIdx.x; i < hw; i += kThreads) {
        const float a = g[i], b = p[i];
        const double d = (double)a - (double)b;
        sse += d * d;
        sae += fabs(d);
        sgg += (double)a * (double)a;
        if (a >= 0.f && a <= 1.f && b >= 0.f && b <= 1.f) {           // np.histogram2d drops samples outside the range
            const int ba = min((int)floorf(a * (float)bins), bins - 1); // the right edge belongs to the last bin
            const int bb = min((int)floorf(b * (float)bins), bins - 1);
            atomicAdd(&hist[ba * bins + bb], 1);
        }
    }
    sse = block_sum_d(sse, red);
    sae = block_sum_d(sae, red);
    sgg = block_sum_d(sgg, red);
    double* marg = reinterpret_cast<double*>(hist + ((nb + 1) & ~1));  // px[bins], py[bins]
    double tot = 0.0;
    for (int i = threadIdx.x; i < nb; i += kThreads) tot += (double)hist[i];
    tot = block_sum_d(tot, red) + 1e-10;
    for (int i = threadIdx.x; i < 2 * bins; i += kThreads) {
        double s = 0.0;
        if (i < bins) {
            for (int j = 0; j < bins; ++j) s += (double)hist[i * bins + j];
        } else {
            for (int j = 0; j < bins; ++j) s += (double)hist[j * bins + (i - bins)];
        }
        marg[i] = s / tot;
    }
    __syncthreads();
    double mi = 0.0;
    for (int i = threadIdx.x; i < nb; i += kThreads) {
        const int c = hist[i];
        if (c > 0) {
            const double pxy = (double)c / tot;
            mi += pxy * log(pxy) - pxy * log(marg[i / bins] * marg[bins + i % bins]);
        }
    }
    mi = block_sum_d(mi, red);
    if (threadIdx.x == 0) {
        out[4 * n + 0] = sse;
        out[4 * n + 1] = sae;
        out[4 * n + 2] = sgg;
        out[4 * n + 3] = mi;
    }
}

// ------------------------------------------------------------- loss windows
// Tile of 32 x 8 outputs per workgroup; the (32+K-1) x (8+K-1) input windows of
// both images sit in LDS; each lane slides its K x K window over LDS.
template <int K, bool LNCC>
__global__ void __launch_bounds__(kThreads)
window_loss_kernel(const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ partial, int H,
                   int W, int OH, int OW, int pad) {
    constexpr int TW = 32, TH = 8;
    constexpr int IW = TW + K - 1, IH = TH + K - 1;
    __shared__ float sx[IH][IW + 1];
    __shared__ float sy[IH][IW + 1];
    __shared__ float red[4];
    const int n = blockIdx.z;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    const float* xp = X + (size_t)n * H * W;
    const float* yp = Y + (size_t)n * H * W;
    for (int e = threadIdx.x; e < IH * IW; e += kThreads) {
        const int r = e / IW, c = e - r * IW;
        const int gy = oy0 + r - pad, gx = ox0 + c - pad;
        float a = 0.f, b = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            a = xp[(size_t)gy * W + gx];
            b = yp[(size_t)gy * W + gx];
        }
        sx[r][c] = a;
        sy[r][c] = b;
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int ox = ox0 + lx, oy = oy0 + ly;
    float val = 0.f;
    if (ox < OW && oy < OH) {
        float s_x = 0.f, s_y = 0.f, s_xx = 0.f, s_yy = 0.f, s_xy = 0.f;
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const float a = sx[ly + r][lx + c], b = sy[ly + r][lx + c];
                s_x += a;
                s_y += b;
                s_xx = fmaf(a, a, s_xx);
                s_yy = fmaf(b, b, s_yy);
                s_xy = fmaf(a, b, s_xy);
            }
        if (LNCC) {
            const float nn = (float)(K * K);
            const float ui = s_x / nn, uj = s_y / nn;
            const float cross = s_xy - uj * s_x - ui * s_y + ui * uj * nn;
            const float ivar = s_xx - 2.f * ui * s_x + ui * ui * nn;
            const float jvar = s_yy - 2.f * uj * s_y + uj * uj * nn;
            val = cross * cross / (ivar * jvar + 1e-5f);
        } else {
            const float inv = 1.f / (float)(K * K);
            const float cov = (float)(K * K) / (float)(K * K - 1);
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float ux = s_x * inv, uy = s_y * inv;
            const float uxx = s_xx * inv, uyy = s_yy * inv, uxy = s_xy * inv;
            const float vx = cov * (uxx - ux * ux), vy = cov * (uyy - uy * uy), vxy = cov * (uxy - ux * uy);
            const float A1 = 2.f * ux * uy + C1, A2 = 2.f * vxy + C2;
            const float B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
            val = (A1 * A2) / (B1 * B2);
        }
    }
    val = san_wave_sum(val);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = val;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[((size_t)n * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// y = avg_pool2(conv2d(x, K x K kernel, zero pad K/2)): the multi-scale LNCC down-sampler
// (lnccloss.py:59-60 -> miloss.py:20-24 + avg_pool2d).  16x16 pooled outputs per workgroup;
// the 32+K-1 square input window and the kernel sit in LDS; each lane owns one pooled pixel.
template <int K>
__global__ void __launch_bounds__(kThreads)
smooth_pool_kernel(const float* __restrict__ x, const float* __restrict__ kern, float* __restrict__ y, int H, int W) {
    constexpr int T = 16, IW = 2 * T + K - 1;
    __shared__ float sx[IW][IW + 1];
    __shared__ float sk[K * K];
    const int plane = blockIdx.z;
    const int OH = H >> 1, OW = W >> 1;
    const int ox0 = blockIdx.x * T, oy0 = blockIdx.y * T;
    const float* xp = x + (size_t)plane * H * W;
    for (int e = threadIdx.x; e < K * K; e += kThreads) sk[e] = kern[e];
    for (int e = threadIdx.x; e < IW * IW; e += kThreads) {
        const int r = e / IW, c = e - r * IW;
        const int gy = 2 * oy0 + r - K / 2, gx = 2 * ox0 + c - K / 2;
        sx[r][c] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? xp[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int ox = ox0 + lx, oy = oy0 + ly;
    if (ox >= OW || oy >= OH) return;
    float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
#pragma unroll 1
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const float w = sk[r * K + c];
            s00 = fmaf(w, sx[2 * ly + r][2 * lx + c], s00);
            s01 = fmaf(w, sx[2 * ly + r][2 * lx + c + 1], s01);
            s10 = fmaf(w, sx[2 * ly + r + 1][2 * lx + c], s10);
            s11 = fmaf(w, sx[2 * ly + r + 1][2 * lx + c + 1], s11);
        }
    y[(size_t)plane * OH * OW + (size_t)oy * OW + ox] = ((s00 + s01) + (s10 + s11)) * 0.25f;
}

// squared forward differences of an NCHW [n,2,h,w] offset field
__global__ void __launch_bounds__(kThreads)
gradient_partial_kernel(const float* __restrict__ off, float* __restrict__ partial, int H, int W) {
    __shared__ float red[8];
    const int plane = blockIdx.y;   // n*2 + comp
    const float* p = off + (size_t)plane * H * W;
    float sdx = 0.f, sdy = 0.f;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < H * W; i += gridDim.x * kThreads) {
        const int y = i / W, x = i - y * W;
        const float v = p[i];
        if (x + 1 < W) {
            const float d = p[i + 1] - v;
            sdx = fmaf(d, d, sdx);
        }
        if (y + 1 < H) {
            const float d = p[i + W] - v;
            sdy = fmaf(d, d, sdy);
        }
    }
    sdx = san_wave_sum(sdx);
    sdy = san_wave_sum(sdy);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = sdx;
        red[4 + (threadIdx.x >> 6)] = sdy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = partial + ((size_t)plane * gridDim.x + blockIdx.x) * 2;
        o[0] = (red[0] + red[1]) + (red[2] + red[3]);
        o[1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

// final stage: one wave; mode 0: loss = a + b*sum/count ; mode 1 (gradient):
// partial holds (dx, dy) pairs, loss = (sum_dx/cnt_x + sum_dy/cnt_y)/2
__global__ void __launch_bounds__(64)
loss_final_kernel(const float* __restrict__ partial, int count, float* __restrict__ loss, int mode, double a, double b,
                  double denom_x, double denom_y) {
    const int lane = threadIdx.x;
    if (mode == 0) {
        double s = 0.0;
        for (int i = lane; i < count; i += 64) s += (double)partial[i];
        s = san_wave_sum_d(s);
        if (lane == 0) loss[0] = (float)(a + b * s / denom_x);
    } else {
        double sx = 0.0, sy = 0.0;
        for (int i = lane; i < count; i += 64) {
            sx += (double)partial[2 * i];
            sy += (double)partial[2 * i + 1];
        }
        sx = san_wave_sum_d(sx);
        sy = san_wave_sum_d(sy);
        if (lane == 0) loss[0] = (float)((sx / denom_x + sy / denom_y) * 0.5);
    }
}

// ------------------------------------------------------------- rss & sens
__global__ void __launch_bounds__(kThreads)
rss_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int HW, int is_complex) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HW; i += gridDim.x * kThreads) {
        float s = 0.f;
        if (is_complex) {
            for (int c = 0; c < C; ++c) {
                const float2 v = *reinterpret_cast<const float2*>(x + (((size_t)n * C + c) * HW + i) * 2);
                s += v.x * v.x + v.y * v.y;
            }
        } else {
            for (int c = 0; c < C; ++c) {
                const float v = x[((size_t)n * C + c) * HW + i];
                s = fmaf(v, v, s);
            }
        }
        out[(size_t)n * HW + i] = sqrtf(s);
    }
}

__global__ void __launch_bounds__(kThreads)
sens_normalize_kernel(const float* __restrict__ est, float* __restrict__ sens, int C, int HW) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HW; i += gridDim.x * kThreads) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            const size_t b = ((size_t)(n * C + c) * 2) * HW + i;
            const float re = est[b], im = est[b + HW];
            s += re * re + im * im;
        }
        const float d = sqrtf(s) + 1e-6f;
        for (int c = 0; c < C; ++c) {
            const size_t b = ((size_t)(n * C + c) * 2) * HW + i;
            *reinterpret_cast<float2*>(sens + (((size_t)n * C + c) * HW + i) * 2) = make_float2(est[b] / d, est[b + HW] / d);
        }
    }
}

int stream_blocks(int elems) {
    int b = san_cdiv(elems, kThreads);
    return b > 512 ? 512 : (b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

int san_warp_fwd(const float* img, const float* offset, float* out, float* grid_out, int n, int c, int h, int w,
                 int padding, void* stream) {
    SAN_CHECK_ARG(offset != nullptr, "null offset");
    SAN_CHECK_ARG((c > 0 && img && out) || (c == 0 && grid_out), "need img/out, or c == 0 with grid_out only");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0, "bad dims");
    SAN_CHECK_ARG(padding == 0 || padding == 1, "padding must be 0 (zeros) or 1 (reflection)");
    hipLaunchKernelGGL(warp_kernel, dim3(stream_blocks(h * w), n), dim3(kThreads), 0, (hipStream_t)stream, img, offset,
                       out, grid_out, c, h, w, padding);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_grid_sample_fwd(const float* img, const float* grid, float* out, int n, int c, int h, int w, int ho, int wo,
                        int padding, void* stream) {
    SAN_CHECK_ARG(img && grid && out, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "bad dims");
    SAN_CHECK_ARG(padding == 0 || padding == 1, "padding must be 0 (zeros) or 1 (reflection)");
    hipLaunchKernelGGL(grid_sample_kernel, dim3(stream_blocks(ho * wo), n), dim3(kThreads), 0, (hipStream_t)stream, img,
                       grid, out, c, h, w, ho, wo, padding);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_grid_sample_complex_fwd(const float* img, const float* grid, float* out, int n, int c, int h, int w, int ho,
                                int wo, int padding, void* stream) {
    SAN_CHECK_ARG(img && grid && out, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "bad dims");
    SAN_CHECK_ARG(padding == 0 || padding == 1, "padding must be 0 (zeros) or 1 (reflection)");
    hipLaunchKernelGGL(grid_sample_c_kernel, dim3(stream_blocks(ho * wo), n), dim3(kThreads), 0, (hipStream_t)stream,
                       (const float2*)img, grid, (float2*)out, c, h, w, ho, wo, padding);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_augment_grid(const float* affine, const float* ctrl, float* grid, int n, int h, int w, int cg, void* stream) {
    SAN_CHECK_ARG(affine && grid, "null pointer");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0, "bad dims");
    SAN_CHECK_ARG(ctrl == nullptr || (cg >= 2 && cg <= 64), "control grid size must be in [2, 64]");
    const size_t lds = ctrl ? (size_t)2 * cg * cg * sizeof(float) : 0;
    hipLaunchKernelGGL(augment_grid_kernel, dim3(stream_blocks(h * w), n), dim3(kThreads), lds, (hipStream_t)stream,
                       affine, ctrl, grid, h, w, cg);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_image_metrics(const float* gt, const float* pred, double* out, int n, int hw, int bins, void* stream) {
    SAN_CHECK_ARG(gt && pred && out, "null pointer");
    SAN_CHECK_ARG(n > 0 && hw > 0, "bad dims");
    SAN_CHECK_ARG(bins >= 2 && bins <= 128, "bins must be in [2, 128]");
    const size_t lds = (size_t)((bins * bins + 1) & ~1) * sizeof(int) + (size_t)2 * bins * sizeof(double);
    hipLaunchKernelGGL(image_metrics_kernel, dim3(n), dim3(kThreads), lds, (hipStream_t)stream, gt, pred, out, hw, bins);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

size_t san_loss_workspace_floats(int n, int h, int w) {
    size_t tiles = (size_t)san_cdiv(w, 32) * san_cdiv(h, 8) * (size_t)n;
    size_t grad = (size_t)n * 2 * 64 * 2;
    return tiles > grad ? tiles : grad;
}

int san_ssim_loss_fwd(const float* x, const float* y, float* loss, int n, int h, int w, float* ws, void* stream) {
    SAN_CHECK_ARG(x && y && loss && ws, "null pointer");
    SAN_CHECK_ARG(n > 0 && h >= 7 && w >= 7, "image smaller than the 7x7 window");
    const int oh = h - 6, ow = w - 6;
    dim3 grid(san_cdiv(ow, 32), san_cdiv(oh, 8), n);
    hipLaunchKernelGGL((window_loss_kernel<7, false>), grid, dim3(kThreads), 0, (hipStream_t)stream, x, y, ws, h, w, oh,
                       ow, 0);
    SAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, (int)(grid.x * grid.y * grid.z),
                       loss, 0, 1.0, -1.0, (double)n * oh * ow, 1.0);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_lncc_loss_fwd(const float* i, const float* j, float* loss, int n, int h, int w, int win, float* ws,
                      void* stream) {
    SAN_CHECK_ARG(i && j && loss && ws, "null pointer");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0, "bad dims");
    if (win != 9) {
        san_set_error("lncc window %d unsupported (only 9)", win);
        return SAN_E_UNSUPPORTED;
    }
    dim3 grid(san_cdiv(w, 32), san_cdiv(h, 8), n);
    hipLaunchKernelGGL((window_loss_kernel<9, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, i, j, ws, h, w, h, w,
                       4);
    SAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, (int)(grid.x * grid.y * grid.z),
                       loss, 0, 0.0, -1.0, (double)n * h * w, 1.0);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_smooth_pool_fwd(const float* x, const float* kern, float* y, int planes, int h, int w, int ksize, void* stream) {
    SAN_CHECK_ARG(x && kern && y, "null pointer");
    SAN_CHECK_ARG(planes > 0 && h >= 2 && w >= 2 && (h % 2 == 0) && (w % 2 == 0), "h, w must be even");
    if (ksize != 13) {
        san_set_error("smoothing kernel size %d unsupported (only 13 = sigma 3)", ksize);
        return SAN_E_UNSUPPORTED;
    }
    dim3 grid(san_cdiv(w / 2, 16), san_cdiv(h / 2, 16), planes);
    hipLaunchKernelGGL((smooth_pool_kernel<13>), grid, dim3(kThreads), 0, (hipStream_t)stream, x, kern, y, h, w);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_gradient_loss_fwd(const float* offset, float* loss, int n, int h, int w, float* ws, void* stream) {
    SAN_CHECK_ARG(offset && loss && ws, "null pointer");
    SAN_CHECK_ARG(n > 0 && h > 1 && w > 1, "bad dims");
    int bx = stream_blocks(h * w);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(gradient_partial_kernel, dim3(bx, n * 2), dim3(kThreads), 0, (hipStream_t)stream, offset, ws, h,
                       w);
    SAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, bx * n * 2, loss, 1, 0.0, 0.0,
                       (double)n * h * (w - 1) * 2, (double)n * (h - 1) * w * 2);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_rss(const float* x, float* out, int n, int c, int hw, int is_complex, void* stream) {
    SAN_CHECK_ARG(x && out, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && hw > 0, "bad dims");
    hipLaunchKernelGGL(rss_kernel, dim3(stream_blocks(hw), n), dim3(kThreads), 0, (hipStream_t)stream, x, out, c, hw,
                       is_complex);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_sens_normalize(const float* est_planar, float* sens, int n, int c, int h, int w, void* stream) {
    SAN_CHECK_ARG(est_planar && sens, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && h > 0 && w > 0, "bad dims");
    hipLaunchKernelGGL(sens_normalize_kernel, dim3(stream_blocks(h * w), n), dim3(kThreads), 0, (hipStream_t)stream,
                       est_planar, sens, c, h * w);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

}  // extern "C"
