// Backward of the local normalised cross-correlation loss and of the multi-scale variant's down-sampler
// (reference: lnccloss.py:7-65, miloss.py:6-24).  gfx950 / CDNA4.
//
//   cc[p] = cross^2 / (Ivar * Jvar + 1e-5),  loss = -mean_p cc[p],
//   with the five 9x9 zero-padded box sums sI, sJ, sII, sJJ, sIJ of window p and
//   uI = sI/81, uJ = sJ/81, cross = sIJ - uJ sI - uI sJ + uI uJ 81, Ivar = sII - 2 uI sI + uI^2 81 (Jvar alike).
//
// Two launches, both HBM-bound, no float atomics (bit-reproducible):
//   stage 1 (lncc_bwd_coef_kernel, the forward kernel's LDS window skeleton): per window position the five partial
//            derivatives of cc wrt its box sums;
//   stage 2 (lncc_bwd_gather_kernel): gI[q] = scale * sum_{p : |p - q|_inf <= 4} (aI[p] + 2 I[q] aII[p] + J[q] aIJ[p])
//            (gJ alike), the five coefficient planes of a 32 x 8 output tile's 40 x 16 neighbourhood staged in LDS.
// The down-sampler y = avg_pool2(conv2d(x, k13x13, zero pad 6)) gets its adjoint in one launch.
#include "san_common.h"

namespace {

constexpr int kThreads = 256;

__global__ void __launch_bounds__(kThreads)
lncc_bwd_coef_kernel(const float* __restrict__ I, const float* __restrict__ J, float* __restrict__ coef, int H, int W) {
    constexpr int K = 9, PAD = 4, TW = 32, TH = 8, IW = TW + K - 1, IH = TH + K - 1;
    __shared__ float si[IH][IW + 1];
    __shared__ float sj[IH][IW + 1];
    const int n = blockIdx.z;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    const float* ip = I + (size_t)n * H * W;
    const float* jp = J + (size_t)n * H * W;
    for (int e = threadIdx.x; e < IH * IW; e += kThreads) {
        const int r = e / IW, c = e - r * IW;
        const int gy = oy0 + r - PAD, gx = ox0 + c - PAD;
        const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
        si[r][c] = ok ? ip[(size_t)gy * W + gx] : 0.f;
        sj[r][c] = ok ? jp[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int ox = ox0 + lx, oy = oy0 + ly;
    if (ox >= W || oy >= H) return;
    float s_i = 0.f, s_j = 0.f, s_ii = 0.f, s_jj = 0.f, s_ij = 0.f;
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const float a = si[ly + r][lx + c], b = sj[ly + r][lx + c];
            s_i += a;
            s_j += b;
            s_ii = fmaf(a, a, s_ii);
            s_jj = fmaf(b, b, s_jj);
            s_ij = fmaf(a, b, s_ij);
        }
    const float nn = (float)(K * K);
    const float ui = s_i / nn, uj = s_j / nn;
    const float cross = s_ij - uj * s_i - ui * s_j + ui * uj * nn;
    const float ivar = s_ii - 2.f * ui * s_i + ui * ui * nn;
    const float jvar = s_jj - 2.f * uj * s_j + uj * uj * nn;
    const float D = ivar * jvar + 1e-5f;
    const float cc = cross * cross / D;
    const float d_cross = 2.f * cross / D, d_ivar = -cc * jvar / D, d_jvar = -cc * ivar / D;
    // uI = sI/81 and uJ = sJ/81 are functions of the sums too; collecting their terms leaves d cross / d sI = -uJ,
    // d cross / d sJ = -uI, d Ivar / d sI = -2 uI, d Jvar / d sJ = -2 uJ (what autograd of lnccloss.py:37-56 sums up to)
    const size_t o = ((size_t)n * H + oy) * W + ox;
    const size_t plane = (size_t)gridDim.z * H * W;
    coef[o] = -d_cross * uj - 2.f * d_ivar * ui;               // wrt sI
    coef[plane + o] = -d_cross * ui - 2.f * d_jvar * uj;       // wrt sJ
    coef[2 * plane + o] = d_ivar;                              // wrt sII
    coef[3 * plane + o] = d_jvar;                              // wrt sJJ
    coef[4 * plane + o] = d_cross;                             // wrt sIJ
}

__global__ void __launch_bounds__(kThreads)
lncc_bwd_gather_kernel(const float* __restrict__ I, const float* __restrict__ J, const float* __restrict__ coef,
                       const float* __restrict__ gscale_dev, float gscale, float denom, float* __restrict__ gI,
                       float* __restrict__ gJ, int accumulate, int H, int W) {
    constexpr int R = 4, TW = 32, TH = 8, CW = TW + 2 * R, CH = TH + 2 * R;
    __shared__ float sc[5][CH][CW + 1];
    const int n = blockIdx.z;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    const size_t plane = (size_t)gridDim.z * H * W;
    for (int e = threadIdx.x; e < 5 * CH * CW; e += kThreads) {
        const int k = e / (CH * CW), rem = e - k * (CH * CW);
        const int r = rem / CW, c = rem - r * CW;
        const int py = oy0 + r - R, px = ox0 + c - R;
        const bool ok = py >= 0 && py < H && px >= 0 && px < W;     // windows centred outside the image do not exist
        sc[k][r][c] = ok ? coef[(size_t)k * plane + ((size_t)n * H + py) * W + px] : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int qx = ox0 + lx, qy = oy0 + ly;
    if (qx >= W || qy >= H) return;
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2 * R + 1; ++r)
#pragma unroll
        for (int c = 0; c < 2 * R + 1; ++c)
#pragma unroll
            for (int k = 0; k < 5; ++k) a[k] += sc[k][ly + r][lx + c];
    const size_t e = ((size_t)n * H + qy) * W + qx;
    const float iv = I[e], jv = J[e];
    const float s = -(gscale * (gscale_dev ? gscale_dev[0] : 1.f)) / denom;      // loss = -mean(cc)
    if (gI) {
        const float v = s * (a[0] + 2.f * iv * a[2] + jv * a[4]);
        gI[e] = accumulate ? gI[e] + v : v;
    }
    if (gJ) {
        const float v = s * (a[1] + 2.f * jv * a[3] + iv * a[4]);
        gJ[e] = accumulate ? gJ[e] + v : v;
    }
}

// adjoint of y = avg_pool2(conv2d(x, k[K x K], zero pad K/2)):
//   gx[q] (+)= 0.25 * sum_{r,c} k[r][c] * gy[(q + K/2 - (r,c)) / 2]   (terms whose full-resolution index leaves the image drop)
template <int K>
__global__ void __launch_bounds__(kThreads)
smooth_pool_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ kern, float* __restrict__ gx, int accumulate,
                       int H, int W) {
    constexpr int TW = 32, TH = 8, P = K / 2;
    // full-resolution neighbourhood [q - P, q + P] -> pooled indices [(q0 - P) >> 1, (q0 + T - 1 + P) >> 1]
    constexpr int GW = (TW + 2 * P) / 2 + 2, GH = (TH + 2 * P) / 2 + 2;
    __shared__ float sg[GH][GW + 1];
    __shared__ float sk[K * K];
    const int plane = blockIdx.z;
    const int OH = H >> 1, OW = W >> 1;
    const int qx0 = blockIdx.x * TW, qy0 = blockIdx.y * TH;
    // floor division by two of possibly negative starts
    const int gx0 = (qx0 - P) >> 1, gy0 = (qy0 - P) >> 1;
    const float* gp = gy + (size_t)plane * OH * OW;
    for (int e = threadIdx.x; e < K * K; e += kThreads) sk[e] = kern[e];
    for (int e = threadIdx.x; e < GH * GW; e += kThreads) {
        const int r = e / GW, c = e - r * GW;
        const int py = gy0 + r, px = gx0 + c;
        sg[r][c] = (py >= 0 && py < OH && px >= 0 && px < OW) ? gp[(size_t)py * OW + px] : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int qx = qx0 + lx, qy = qy0 + ly;
    if (qx >= W || qy >= H) return;
    float acc = 0.f;
#pragma unroll 1
    for (int r = 0; r < K; ++r) {
        const int ty = qy + P - r;                       // full-resolution row of the smoothed image this tap feeds
        if (ty < 0 || ty >= (OH << 1)) continue;
        const int sr = (ty >> 1) - gy0;
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const int tx = qx + P - c;
            const float g = (tx >= 0 && tx < (OW << 1)) ? sg[sr][(tx >> 1) - gx0] : 0.f;
            acc = fmaf(sk[r * K + c], g, acc);
        }
    }
    const size_t e = (size_t)plane * H * W + (size_t)qy * W + qx;
    const float v = 0.25f * acc;
    gx[e] = accumulate ? gx[e] + v : v;
}

// bilinear grid_sample (zeros padding, align_corners = False) backward wrt the IMAGE: every output pixel scatters its
// gradient to the four texels it read.  The only float atomics of the library (the scatter has no gather form for an
// arbitrary grid): used when a caller asks autograd for d/d img of SpatialTransformer.warp; the training step never does.
__global__ void __launch_bounds__(kThreads)
grid_sample_bwd_img_kernel(const float* __restrict__ grid, const float* __restrict__ g, float* __restrict__ gimg, int C, int H,
                           int W, int HO, int WO) {
    const int n = blockIdx.y;
    const int HWo = HO * WO;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HWo; i += gridDim.x * kThreads) {
        const float2 gg = *reinterpret_cast<const float2*>(grid + ((size_t)n * HWo + i) * 2);
        const float ix = ((gg.x + 1.f) * (float)W - 1.f) * 0.5f;
        const float iy = ((gg.y + 1.f) * (float)H - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const bool xin0 = x0 >= 0 && x0 < W, xin1 = x1 >= 0 && x1 < W;
        const bool yin0 = y0 >= 0 && y0 < H, yin1 = y1 >= 0 && y1 < H;
        for (int c = 0; c < C; ++c) {
            const float go = g[((size_t)n * C + c) * HWo + i];
            float* p = gimg + ((size_t)n * C + c) * H * W;
            if (yin0 && xin0) atomicAdd(p + y0 * W + x0, go * (wx0 * wy0));
            if (yin0 && xin1) atomicAdd(p + y0 * W + x1, go * (wx1 * wy0));
            if (yin1 && xin0) atomicAdd(p + y1 * W + x0, go * (wx0 * wy1));
            if (yin1 && xin1) atomicAdd(p + y1 * W + x1, go * (wx1 * wy1));
        }
    }
}

// The same scatter WITHOUT float atomics (round 5): contributions are rounded to 64-bit fixed point (2^-40 of the largest |g|,
// i.e. finer than an fp32 sum of them could resolve) and added with INTEGER atomics -- integer addition is associative, so the
// result does not depend on the order the workgroups arrive in; a last pass converts back.  work: n*c*h*w + 1 words (the last
// one collects the bits of max |g|).
__global__ void __launch_bounds__(kThreads) gsb_amax_kernel(const float* __restrict__ g, size_t count, unsigned long long* __restrict__ amax) {
    // a non-finite gradient (Inf or NaN: fmaxf would drop the NaN) is recorded as the NaN bit pattern, which orders above every
    // finite value and above Inf in the integer maximum: the convert pass then writes NaN instead of a silently zero gradient
    // (round 6, ADVICE r5 -- F.grid_sample's autograd propagates them; a GradScaler-style overflow check must see them)
    float mx = 0.f;
    int bad = 0;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += (size_t)gridDim.x * kThreads) {
        const float a = fabsf(g[i]);
        bad |= !(a <= 3.402823466e+38f);
        mx = fmaxf(mx, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        bad |= __shfl_xor(bad, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad) atomicMax(amax, 0x7FC00000ull);
        else if (mx > 0.f) atomicMax(amax, (unsigned long long)__builtin_bit_cast(unsigned, mx));
    }
}

__device__ __forceinline__ float gsb_scale(const unsigned long long* amax) {
    const float mx = __builtin_bit_cast(float, (unsigned)*amax);
    if (!(mx > 0.f) || !(mx < INFINITY)) return 0.f;
    int e;
    frexpf(mx, &e);                                      // mx = m * 2^e, 0.5 <= m < 1
    return ldexpf(1.f, 40 - e);                          // |g| * scale < 2^40: 2^23 coinciding contributions fit in 63 bits
}

__global__ void __launch_bounds__(kThreads)
grid_sample_bwd_img_fixed_kernel(const float* __restrict__ grid, const float* __restrict__ g, unsigned long long* __restrict__ work,
                                 const unsigned long long* __restrict__ amax, int C, int H, int W, int HO, int WO) {
    const int n = blockIdx.y;
    const int HWo = HO * WO;
    const float scale = gsb_scale(amax);
    if (scale == 0.f) return;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HWo; i += gridDim.x * kThreads) {
        const float2 gg = *reinterpret_cast<const float2*>(grid + ((size_t)n * HWo + i) * 2);
        const float ix = ((gg.x + 1.f) * (float)W - 1.f) * 0.5f;
        const float iy = ((gg.y + 1.f) * (float)H - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        if (!(fx > -2.f && fx < (float)W + 1.f && fy > -2.f && fy < (float)H + 1.f)) continue;      // (also NaN / far outside)
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const bool xin0 = x0 >= 0 && x0 < W, xin1 = x1 >= 0 && x1 < W;
        const bool yin0 = y0 >= 0 && y0 < H, yin1 = y1 >= 0 && y1 < H;
        for (int c = 0; c < C; ++c) {
            const float go = g[((size_t)n * C + c) * HWo + i] * scale;           // (a power of two: exact)
            unsigned long long* p = work + ((size_t)n * C + c) * H * W;
            if (yin0 && xin0) atomicAdd(p + y0 * W + x0, (unsigned long long)__float2ll_rn(go * (wx0 * wy0)));
            if (yin0 && xin1) atomicAdd(p + y0 * W + x1, (unsigned long long)__float2ll_rn(go * (wx1 * wy0)));
            if (yin1 && xin0) atomicAdd(p + y1 * W + x0, (unsigned long long)__float2ll_rn(go * (wx0 * wy1)));
            if (yin1 && xin1) atomicAdd(p + y1 * W + x1, (unsigned long long)__float2ll_rn(go * (wx1 * wy1)));
        }
    }
}

__global__ void __launch_bounds__(kThreads) gsb_convert_kernel(const unsigned long long* __restrict__ work, const unsigned long long* __restrict__ amax,
                                                               float* __restrict__ gimg, size_t count) {
    const float scale = gsb_scale(amax);
    const double inv = scale > 0.f ? 1.0 / (double)scale : 0.0;
    const bool nonfinite = (unsigned)*amax > 0x7F7FFFFFu;      // some |g| was Inf or NaN: the whole gradient is NaN, loudly
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < count; i += (size_t)gridDim.x * kThreads)
        gimg[i] = nonfinite ? __builtin_nanf("") : (float)((double)(long long)work[i] * inv);
}

}  // namespace

extern "C" {

size_t san_grid_sample_bwd_img_work_bytes(int n, int c, int h, int w) { return ((size_t)n * c * h * w + 1) * sizeof(long long); }

int san_grid_sample_bwd_img_det(const float* grid, const float* g, float* gimg, long long* work, int n, int c, int h, int w, int ho,
                                int wo, void* stream) {
    SAN_CHECK_ARG(grid && g && gimg && work, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "bad dims");
    const size_t count = (size_t)n * c * h * w, gcount = (size_t)n * c * ho * wo;
    hipError_t e = hipMemsetAsync(work, 0, (count + 1) * sizeof(long long), (hipStream_t)stream);
    if (e != hipSuccess) {
        san_set_error("san_grid_sample_bwd_img_det: memset failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    unsigned long long* wk = reinterpret_cast<unsigned long long*>(work);
    int ba = (int)((gcount + kThreads - 1) / kThreads);
    if (ba > 1024) ba = 1024;
    hipLaunchKernelGGL(gsb_amax_kernel, dim3(ba), dim3(kThreads), 0, (hipStream_t)stream, g, gcount, wk + count);
    int bx = san_cdiv(ho * wo, kThreads);
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(grid_sample_bwd_img_fixed_kernel, dim3(bx, n), dim3(kThreads), 0, (hipStream_t)stream, grid, g, wk, wk + count, c, h,
                       w, ho, wo);
    int bc = (int)((count + kThreads - 1) / kThreads);
    if (bc > 2048) bc = 2048;
    hipLaunchKernelGGL(gsb_convert_kernel, dim3(bc), dim3(kThreads), 0, (hipStream_t)stream, wk, wk + count, gimg, count);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_grid_sample_bwd_img(const float* grid, const float* g, float* gimg, int n, int c, int h, int w, int ho, int wo,
                            void* stream) {
    SAN_CHECK_ARG(grid && g && gimg, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "bad dims");
    hipError_t e = hipMemsetAsync(gimg, 0, (size_t)n * c * h * w * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) {
        san_set_error("san_grid_sample_bwd_img: memset failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    int bx = san_cdiv(ho * wo, kThreads);
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(grid_sample_bwd_img_kernel, dim3(bx, n), dim3(kThreads), 0, (hipStream_t)stream, grid, g, gimg, c, h, w,
                       ho, wo);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

size_t san_lncc_bwd_workspace_floats(int n, int h, int w) { return (size_t)5 * n * h * w; }

int san_lncc_loss_bwd(const float* i, const float* j, float* gi, float* gj, float gscale, const float* gscale_dev,
                      int accumulate, int n, int h, int w, int win, float* ws, void* stream) {
    SAN_CHECK_ARG(i && j && ws, "null pointer");
    SAN_CHECK_ARG(gi || gj, "at least one of gi / gj must be given");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0, "bad dims");
    if (win != 9) {
        san_set_error("lncc window %d unsupported (only 9)", win);
        return SAN_E_UNSUPPORTED;
    }
    const dim3 grid(san_cdiv(w, 32), san_cdiv(h, 8), n);
    hipLaunchKernelGGL(lncc_bwd_coef_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, i, j, ws, h, w);
    SAN_LAUNCH_CHECK();
    // loss = -mean(cc): d loss / d cc[p] = -1 / (n h w), times gscale and (if given) the device scalar
    hipLaunchKernelGGL(lncc_bwd_gather_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, i, j, ws, gscale_dev, gscale,
                       (float)n * (float)h * (float)w, gi, gj, accumulate, h, w);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_smooth_pool_bwd(const float* gy, const float* kern, float* gx, int accumulate, int planes, int h, int w, int ksize,
                        void* stream) {
    SAN_CHECK_ARG(gy && kern && gx, "null pointer");
    // (h, w) = the INPUT's size; odd sizes are fine: avg_pool2d drops the last row / column, whose smoothed values get no
    // gradient while the input pixels there still feed their neighbours' smoothed values
    SAN_CHECK_ARG(planes > 0 && h >= 2 && w >= 2, "h, w must be at least 2");
    if (ksize != 13) {
        san_set_error("smoothing kernel size %d unsupported (only 13 = sigma 3)", ksize);
        return SAN_E_UNSUPPORTED;
    }
    const dim3 grid(san_cdiv(w, 32), san_cdiv(h, 8), planes);
    hipLaunchKernelGGL((smooth_pool_bwd_kernel<13>), grid, dim3(kThreads), 0, (hipStream_t)stream, gy, kern, gx, accumulate, h,
                       w);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

}  // extern "C"
