// Normalisation statistics (merge of per-tile Welford partials) and the
// element-wise materialisers of the lazy-normalisation scheme.  All of these are
// HBM-bound streaming kernels: 16-byte accesses where the layout allows,
// grid-stride loops capped at 2048 workgroups.
#include "san_common.h"

namespace {

constexpr int kThreads = 256;

// -------------------------------------------------------------------------
// one workgroup of 256 threads per output statistic; partials merged in double, fixed order.  The (count, mean, M2)
// records are read ONCE (up to 8 per thread stay in registers for the second, centred pass).
// grid: (c, n_groups) ; n_groups = n (instance/group) or 1 (batch)
__global__ void __launch_bounds__(256)
norm_finalize_kernel(const float* __restrict__ part, int n, int c, int tiles, int mode, float eps,
                     const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ scale,
                     float* __restrict__ shift, int sc_ctot, int sc_coff, float* __restrict__ aux_a,
                     float* __restrict__ aux_b, float* __restrict__ rmean, float* __restrict__ rvar,
                     long long* __restrict__ nbt, float momentum, float var_factor) {
    __shared__ double red[3][4];
    const int ch = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n_lo = (mode == SAN_NORM_BATCH) ? 0 : blockIdx.y;
    const int n_hi = (mode == SAN_NORM_BATCH) ? n : blockIdx.y + 1;
    const int total = (n_hi - n_lo) * tiles;            // records of this statistic, contiguous per sample
    constexpr int kKeep = 8;
    float kc[kKeep], km[kKeep], k2[kKeep];
    double cnt = 0.0, s = 0.0;
    auto rec = [&](int e) -> const float* {
        const int b = n_lo + e / tiles, t = e - (e / tiles) * tiles;
        return part + ((size_t)(b * c + ch) * tiles + t) * 3;
    };
#pragma unroll
    for (int i = 0; i < kKeep; ++i) {
        const int e = tid + 256 * i;
        kc[i] = km[i] = k2[i] = 0.f;
        if (e < total) {
            const float* p = rec(e);
            kc[i] = p[0];
            km[i] = p[1];
            k2[i] = p[2];
            cnt += (double)kc[i];
            s += (double)kc[i] * (double)km[i];
        }
    }
    for (int e = tid + 256 * kKeep; e < total; e += 256) {
        const float* p = rec(e);
        cnt += (double)p[0];
        s += (double)p[0] * (double)p[1];
    }
    cnt = san_wave_sum_d(cnt);
    s = san_wave_sum_d(s);
    if (lane == 0) {
        red[0][wv] = cnt;
        red[1][wv] = s;
    }
    __syncthreads();
    cnt = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    s = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double mean = cnt > 0.0 ? s / cnt : 0.0;
    double m2 = 0.0;
#pragma unroll
    for (int i = 0; i < kKeep; ++i) {
        const double d = (double)km[i] - mean;
        m2 += (double)k2[i] + (double)kc[i] * d * d;
    }
    for (int e = tid + 256 * kKeep; e < total; e += 256) {
        const float* p = rec(e);
        const double d = (double)p[1] - mean;
        m2 += (double)p[2] + (double)p[0] * d * d;
    }
    m2 = san_wave_sum_d(m2);
    if (lane == 0) red[2][wv] = m2;
    __syncthreads();
    m2 = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    if (tid != 0) return;
    const double var_b = cnt > 0.0 ? m2 / cnt : 0.0;
    const double var_u = cnt > 1.0 ? m2 / (cnt - 1.0) : 0.0;
    if (mode == SAN_NORM_INSTANCE) {
        const float sc = (float)(1.0 / sqrt(var_b + (double)eps));
        scale[blockIdx.y * sc_ctot + sc_coff + ch] = sc;
        shift[blockIdx.y * sc_ctot + sc_coff + ch] = (float)(-mean) * sc;
    } else if (mode == SAN_NORM_GROUP || mode == SAN_NORM_GROUP_BWD) {
        const float sd = (float)sqrt(var_u);
        const float sc = 1.f / (sd + eps);
        scale[blockIdx.y * sc_ctot + sc_coff + ch] = sc;
        shift[blockIdx.y * sc_ctot + sc_coff + ch] = (float)(-mean) * sc;
        if (aux_a) aux_a[blockIdx.y * c + ch] = sd;
        if (aux_b) aux_b[blockIdx.y * c + ch] = (float)mean;
        if (mode == SAN_NORM_GROUP_BWD) {
            // a constant plane (std == 0) gets no d sigma term (torch's std backward masks it to 0)
            const float isd = sd > 0.f ? 1.f / sd : 0.f;
            aux_a[(size_t)n * c + blockIdx.y * c + ch] = isd;
            aux_b[(size_t)n * c + blockIdx.y * c + ch] = -(float)mean * isd;
        }
    } else {
        const float g = gamma ? gamma[ch] : 1.f;
        const float bt = beta ? beta[ch] : 0.f;
        const float sc = g * (float)(1.0 / sqrt(var_b + (double)eps));
        const float sh = bt - (float)mean * sc;
        for (int b = 0; b < n; ++b) {
            scale[b * sc_ctot + sc_coff + ch] = sc;
            shift[b * sc_ctot + sc_coff + ch] = sh;
        }
        if (aux_a) aux_a[ch] = (float)mean;
        if (aux_b) aux_b[ch] = (float)var_u;
        // BatchNorm2d's running statistics in the same launch (san_norm_finalize_bn): the arithmetic of bn_update_running_kernel
        if (rmean) {
            rmean[ch] = rmean[ch] * (1.f - momentum) + momentum * (float)mean;
            rvar[ch] = rvar[ch] * (1.f - momentum) + momentum * ((float)var_u * var_factor);
            if (ch == 0 && nbt) *nbt += 1;
        }
    }
}

// `tiles` workgroups per (c, n) plane, each an exact two-pass (mean, then centred M2)
// over its contiguous chunk; norm_finalize merges the chunks.  grid: (tiles, c, n)
__global__ void __launch_bounds__(kThreads)
plane_stats_kernel(const float* __restrict__ x, int x_ctot, int x_coff, int c, int hw, int tiles,
                   float* __restrict__ part) {
    __shared__ float red[8];
    const int t = blockIdx.x, ch = blockIdx.y, n = blockIdx.z;
    const int chunk = (hw + tiles - 1) / tiles;
    const int lo = t * chunk;
    const int cnt = max(0, min(hw, lo + chunk) - lo);
    const float* p = x + ((size_t)(n * x_ctot + x_coff + ch)) * hw + lo;
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < cnt; i += kThreads) s += p[i];
    s = san_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = cnt > 0 ? ((red[0] + red[1]) + (red[2] + red[3])) / (float)cnt : 0.f;
    float d = 0.f;
    for (int i = tid; i < cnt; i += kThreads) {
        const float e = p[i] - mean;
        d = fmaf(e, e, d);
    }
    d = san_wave_sum(d);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = d;
    __syncthreads();
    if (tid == 0) {
        float* o = part + ((size_t)(n * c + ch) * tiles + t) * 3;
        o[0] = (float)cnt;
        o[1] = mean;
        o[2] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift, int sc_ctot, int sc_coff,
                                      int n, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * c) return;
    const int b = i / c, ch = i - b * c;
    const float sc = gamma[ch] / sqrtf(rvar[ch] + eps);
    scale[b * sc_ctot + sc_coff + ch] = sc;
    shift[b * sc_ctot + sc_coff + ch] = beta[ch] - rmean[ch] * sc;
}

// ------------------------------------------------------------ materialisers
struct EwArgs {
    const float* x;
    const float* sc;
    const float* sh;
    float slope;
    int x_ctot, x_coff;
    const float* b;
    const float* b_sc;
    const float* b_sh;
    float b_slope;
    int b_ctot, b_coff;
    float* y;
    int y_ctot, y_coff;
    int n, c, h, w;
};

__device__ __forceinline__ void load_affine(const float* sc, const float* sh, int idx, float& s, float& t) {
    s = 1.f;
    t = 0.f;
    if (sc) {
        s = sc[idx];
        t = sh[idx];
    }
}

// grid: (blocks over the plane, c, n)
__global__ void __launch_bounds__(kThreads) avgpool2_kernel(const EwArgs a) {
    const int ch = blockIdx.y, n = blockIdx.z;
    const int oh = a.h >> 1, ow = a.w >> 1;
    float s, t;
    load_affine(a.sc, a.sh, n * a.x_ctot + a.x_coff + ch, s, t);
    const float* xp = a.x + ((size_t)(n * a.x_ctot + a.x_coff + ch)) * a.h * a.w;
    float* yp = a.y + ((size_t)(n * a.y_ctot + a.y_coff + ch)) * oh * ow;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < oh * ow; i += gridDim.x * kThreads) {
        const int oy = i / ow, ox = i - oy * ow;
        const float2 r0 = *reinterpret_cast<const float2*>(xp + (size_t)(2 * oy) * a.w + 2 * ox);
        const float2 r1 = *reinterpret_cast<const float2*>(xp + (size_t)(2 * oy + 1) * a.w + 2 * ox);
        const float v = (san_act(r0.x, s, t, a.slope) + san_act(r0.y, s, t, a.slope)) +
                        (san_act(r1.x, s, t, a.slope) + san_act(r1.y, s, t, a.slope));
        yp[i] = v * 0.25f;
    }
}

// InstanceNorm finalisation + 2 x 2 average pooling of the activated plane in ONE launch (round 6): the encoder levels of the
// U-Net run conv -> IN -> LeakyReLU -> avg_pool2d (varnet.py:95-99); norm_finalize_kernel (one workgroup per plane, ~5 us of
// launch for 2-5 KB of records) and avgpool2_kernel were two launches per level.  Here EVERY workgroup of a plane merges the
// plane's records itself -- norm_finalize_kernel's order and arithmetic exactly, so the affine is the same bits -- the first one
// writes (scale, shift), and each pools its share of the plane with avgpool2_kernel's arithmetic.  The redundant merges read
// K x 19 KB per plane from L2; a workgroup spends ~2 us on them.
// grid: (K, c, n)
__global__ void __launch_bounds__(256)
norm_finalize_pool_kernel(const float* __restrict__ part, int c, int tiles, float eps, float* __restrict__ scale,
                          float* __restrict__ shift, int sc_ctot, int sc_coff, const float* __restrict__ x, int x_ctot, int x_coff,
                          float slope, float* __restrict__ y, int y_ctot, int y_coff, int h, int w) {
    __shared__ double red[3][4];
    __shared__ float aff[2];
    const int ch = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int total = tiles;
    constexpr int kKeep = 8;
    float kc[kKeep], km[kKeep], k2[kKeep];
    double cnt = 0.0, s = 0.0;
    auto rec = [&](int e) -> const float* { return part + ((size_t)(n * c + ch) * tiles + e) * 3; };
#pragma unroll
    for (int i = 0; i < kKeep; ++i) {
        const int e = tid + 256 * i;
        kc[i] = km[i] = k2[i] = 0.f;
        if (e < total) {
            const float* p = rec(e);
            kc[i] = p[0];
            km[i] = p[1];
            k2[i] = p[2];
            cnt += (double)kc[i];
            s += (double)kc[i] * (double)km[i];
        }
    }
    for (int e = tid + 256 * kKeep; e < total; e += 256) {
        const float* p = rec(e);
        cnt += (double)p[0];
        s += (double)p[0] * (double)p[1];
    }
    cnt = san_wave_sum_d(cnt);
    s = san_wave_sum_d(s);
    if (lane == 0) {
        red[0][wv] = cnt;
        red[1][wv] = s;
    }
    __syncthreads();
    cnt = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    s = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double mean = cnt > 0.0 ? s / cnt : 0.0;
    double m2 = 0.0;
#pragma unroll
    for (int i = 0; i < kKeep; ++i) {
        const double d = (double)km[i] - mean;
        m2 += (double)k2[i] + (double)kc[i] * d * d;
    }
    for (int e = tid + 256 * kKeep; e < total; e += 256) {
        const float* p = rec(e);
        const double d = (double)p[1] - mean;
        m2 += (double)p[2] + (double)p[0] * d * d;
    }
    m2 = san_wave_sum_d(m2);
    if (lane == 0) red[2][wv] = m2;
    __syncthreads();
    if (tid == 0) {
        m2 = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        const double var_b = cnt > 0.0 ? m2 / cnt : 0.0;
        const float sc = (float)(1.0 / sqrt(var_b + (double)eps));
        const float sh = (float)(-mean) * sc;
        aff[0] = sc;
        aff[1] = sh;
        if (blockIdx.x == 0) {
            scale[n * sc_ctot + sc_coff + ch] = sc;
            shift[n * sc_ctot + sc_coff + ch] = sh;
        }
    }
    __syncthreads();
    const float sc = aff[0], sh = aff[1];
    const int oh = h >> 1, ow = w >> 1;
    const float* xp = x + ((size_t)(n * x_ctot + x_coff + ch)) * h * w;
    float* yp = y + ((size_t)(n * y_ctot + y_coff + ch)) * oh * ow;
    for (int i = blockIdx.x * 256 + tid; i < oh * ow; i += gridDim.x * 256) {
        const int oy = i / ow, ox = i - oy * ow;
        const float2 r0 = *reinterpret_cast<const float2*>(xp + (size_t)(2 * oy) * w + 2 * ox);
        const float2 r1 = *reinterpret_cast<const float2*>(xp + (size_t)(2 * oy + 1) * w + 2 * ox);
        const float v = (san_act(r0.x, sc, sh, slope) + san_act(r0.y, sc, sh, slope)) +
                        (san_act(r1.x, sc, sh, slope) + san_act(r1.y, sc, sh, slope));
        yp[i] = v * 0.25f;
    }
}

// One lazily normalised channel -- its raw plane and its (scale, shift) entries -- copied from one tensor into up to 16 others of
// the same shape (round 6).  The 12 cascades of a training step keep their own U-Net input buffers, and every one of them holds the
// SAME InstanceNorm-ed reference image as channel 2 (varnet.py:315-319): set_ref ran apply + plane_stats + norm_finalize per cascade,
// 36 launches of 4-10 us at the head of a step; now once, plus this launch.
struct ReplArgs {
    const float* src;
    const float* ssc;
    const float* ssh;
    float* dst[16];
    float* dsc[16];
    float* dsh[16];
    int ctot, ch, n, hw, count;
};
__global__ void __launch_bounds__(kThreads) replicate_channel_kernel(const ReplArgs a) {
    const int n = blockIdx.y, k = blockIdx.z;
    const size_t off = ((size_t)(n * a.ctot + a.ch)) * a.hw;
    const float* sp = a.src + off;
    float* dp = a.dst[k] + off;
    if ((a.hw & 3) == 0 && ((reinterpret_cast<uintptr_t>(sp) | reinterpret_cast<uintptr_t>(dp)) & 15) == 0) {
        for (int i = blockIdx.x * kThreads + threadIdx.x; i < a.hw / 4; i += gridDim.x * kThreads)
            reinterpret_cast<float4*>(dp)[i] = reinterpret_cast<const float4*>(sp)[i];
    } else {
        for (int i = blockIdx.x * kThreads + threadIdx.x; i < a.hw; i += gridDim.x * kThreads) dp[i] = sp[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.dsc[k][n * a.ctot + a.ch] = a.ssc[n * a.ctot + a.ch];
        a.dsh[k][n * a.ctot + a.ch] = a.ssh[n * a.ctot + a.ch];
    }
}

__global__ void __launch_bounds__(kThreads) upsample2_kernel(const EwArgs a) {
    const int ch = blockIdx.y, n = blockIdx.z;
    const int ow = a.w * 2;
    float s, t;
    load_affine(a.sc, a.sh, n * a.x_ctot + a.x_coff + ch, s, t);
    const float* xp = a.x + ((size_t)(n * a.x_ctot + a.x_coff + ch)) * a.h * a.w;
    float* yp = a.y + ((size_t)(n * a.y_ctot + a.y_coff + ch)) * (size_t)(4 * a.h * a.w);
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < a.h * a.w; i += gridDim.x * kThreads) {
        const int iy = i / a.w, ix = i - iy * a.w;
        const float v = san_act(xp[i], s, t, a.slope);
        const float2 vv = make_float2(v, v);
        *reinterpret_cast<float2*>(yp + (size_t)(2 * iy) * ow + 2 * ix) = vv;
        *reinterpret_cast<float2*>(yp + (size_t)(2 * iy + 1) * ow + 2 * ix) = vv;
    }
}

// inverse pixel shuffle of the transposed convolution: y[n, 4c + 2dy + dx, i, j] = x[n, c, 2i+dy, 2j+dx]
// (a.h, a.w are the OUTPUT plane dims; grid.y runs over the c input channels)
__global__ void __launch_bounds__(kThreads) unshuffle2_kernel(const EwArgs a) {
    const int ch = blockIdx.y, n = blockIdx.z;
    const int h = a.h, w = a.w, iw = 2 * a.w;
    const float* xp = a.x + ((size_t)(n * a.x_ctot + a.x_coff + ch)) * (size_t)(4 * h * w);
    float* yp = a.y + ((size_t)(n * a.y_ctot + a.y_coff + 4 * ch)) * (size_t)(h * w);
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < h * w; i += gridDim.x * kThreads) {
        const int iy = i / w, ix = i - iy * w;
        const float2 r0 = *reinterpret_cast<const float2*>(xp + (size_t)(2 * iy) * iw + 2 * ix);
        const float2 r1 = *reinterpret_cast<const float2*>(xp + (size_t)(2 * iy + 1) * iw + 2 * ix);
        yp[i] = r0.x;
        yp[(size_t)h * w + i] = r0.y;
        yp[(size_t)2 * h * w + i] = r1.x;
        yp[(size_t)3 * h * w + i] = r1.y;
    }
}

// h*w passed as a.h (a.w == 1)
__global__ void __launch_bounds__(kThreads) add_kernel(const EwArgs a) {
    const int ch = blockIdx.y, n = blockIdx.z;
    const int hw = a.h;
    float s0, t0, s1, t1;
    load_affine(a.sc, a.sh, n * a.x_ctot + a.x_coff + ch, s0, t0);
    load_affine(a.b_sc, a.b_sh, n * a.b_ctot + a.b_coff + ch, s1, t1);
    const float* xp = a.x + ((size_t)(n * a.x_ctot + a.x_coff + ch)) * hw;
    const float* bp = a.b + ((size_t)(n * a.b_ctot + a.b_coff + ch)) * hw;
    float* yp = a.y + ((size_t)(n * a.y_ctot + a.y_coff + ch)) * hw;
    if ((hw & 3) == 0) {
        for (int i = blockIdx.x * kThreads + threadIdx.x; i < hw / 4; i += gridDim.x * kThreads) {
            const float4 u = reinterpret_cast<const float4*>(xp)[i];
            const float4 v = reinterpret_cast<const float4*>(bp)[i];
            float4 o;
            o.x = san_act(u.x, s0, t0, a.slope) + san_act(v.x, s1, t1, a.b_slope);
            o.y = san_act(u.y, s0, t0, a.slope) + san_act(v.y, s1, t1, a.b_slope);
            o.z = san_act(u.z, s0, t0, a.slope) + san_act(v.z, s1, t1, a.b_slope);
            o.w = san_act(u.w, s0, t0, a.slope) + san_act(v.w, s1, t1, a.b_slope);
            reinterpret_cast<float4*>(yp)[i] = o;
        }
    } else {
        for (int i = blockIdx.x * kThreads + threadIdx.x; i < hw; i += gridDim.x * kThreads)
            yp[i] = san_act(xp[i], s0, t0, a.slope) + san_act(bp[i], s1, t1, a.b_slope);
    }
}

__global__ void __launch_bounds__(kThreads) apply_kernel(const EwArgs a) {
    const int ch = blockIdx.y, n = blockIdx.z;
    const int hw = a.h;
    float s0, t0;
    load_affine(a.sc, a.sh, n * a.x_ctot + a.x_coff + ch, s0, t0);
    const float* xp = a.x + ((size_t)(n * a.x_ctot + a.x_coff + ch)) * hw;
    float* yp = a.y + ((size_t)(n * a.y_ctot + a.y_coff + ch)) * hw;
    if ((hw & 3) == 0) {
        for (int i = blockIdx.x * kThreads + threadIdx.x; i < hw / 4; i += gridDim.x * kThreads) {
            const float4 u = reinterpret_cast<const float4*>(xp)[i];
            float4 o;
            o.x = san_act(u.x, s0, t0, a.slope);
            o.y = san_act(u.y, s0, t0, a.slope);
            o.z = san_act(u.z, s0, t0, a.slope);
            o.w = san_act(u.w, s0, t0, a.slope);
            reinterpret_cast<float4*>(yp)[i] = o;
        }
    } else {
        for (int i = blockIdx.x * kThreads + threadIdx.x; i < hw; i += gridDim.x * kThreads)
            yp[i] = san_act(xp[i], s0, t0, a.slope);
    }
}


// window copy between planes of different sizes (NormUnet.pad / unpad, the U-Net's reflect pad and its adjoint):
// y[oy, ox] = T(x)[oy - off_y, ox - off_x], T = the lazy read of x.  mode 0: zero outside x; 1: one reflected row /
// column at the bottom / right (F.pad 'reflect': the edge itself is not repeated); 2: adjoint of mode 1 (x is the
// gradient of the padded plane: its extra row hy / column wy folds back onto row hy-2 / column wy-2 of y).
struct WinArgs {
    const float* x;
    const float* sc;
    const float* sh;
    float slope;
    int x_ctot, x_coff, hx, wx;
    float* y;
    int y_ctot, y_coff, hy, wy;
    int off_y, off_x, mode;
};

__global__ void __launch_bounds__(kThreads) window_copy_kernel(const WinArgs a) {
    const int ch = blockIdx.y, n = blockIdx.z;
    float s, t;
    load_affine(a.sc, a.sh, n * a.x_ctot + a.x_coff + ch, s, t);
    const float* xp = a.x + ((size_t)(n * a.x_ctot + a.x_coff + ch)) * a.hx * a.wx;
    float* yp = a.y + ((size_t)(n * a.y_ctot + a.y_coff + ch)) * a.hy * a.wy;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < a.hy * a.wy; i += gridDim.x * kThreads) {
        const int oy = i / a.wy, ox = i - oy * a.wy;
        int sy = oy - a.off_y, sx = ox - a.off_x;
        float v = 0.f;
        if (a.mode == 1) {
            if (sy >= a.hx) sy = 2 * (a.hx - 1) - sy;
            if (sx >= a.wx) sx = 2 * (a.wx - 1) - sx;
            v = san_act(xp[(size_t)sy * a.wx + sx], s, t, a.slope);
        } else if (a.mode == 2) {
            // x = gradient of the reflect-padded plane (hx >= hy, wx >= wy); rows / columns beyond y fold back
            const bool fy = a.hx > a.hy && oy == 2 * (a.hy - 1) - a.hy, fx = a.wx > a.wy && ox == 2 * (a.wy - 1) - a.wy;
            v = san_act(xp[(size_t)oy * a.wx + ox], s, t, a.slope);
            if (fy) v += san_act(xp[(size_t)a.hy * a.wx + ox], s, t, a.slope);
            if (fx) v += san_act(xp[(size_t)oy * a.wx + a.wy], s, t, a.slope);
            if (fy && fx) v += san_act(xp[(size_t)a.hy * a.wx + a.wy], s, t, a.slope);
        } else if (sy >= 0 && sy < a.hx && sx >= 0 && sx < a.wx) {
            v = san_act(xp[(size_t)sy * a.wx + sx], s, t, a.slope);
        }
        yp[i] = v;
    }
}

dim3 ew_grid(int elems_per_plane, int c, int n) {
    int bx = san_cdiv(elems_per_plane, kThreads * 4);
    if (bx < 1) bx = 1;
    long cap = 4096 / ((long)c * n > 0 ? (long)c * n : 1);
    if (cap < 1) cap = 1;
    if (bx > cap) bx = (int)cap;
    return dim3(bx, c, n);
}

int check_view(int ctot, int coff, int c) { return coff >= 0 && c > 0 && coff + c <= ctot; }

}  // namespace

extern "C" {

int san_norm_finalize(const float* part, int n, int c, int tiles, int mode, float eps, const float* gamma,
                      const float* beta, float* scale, float* shift, int sc_ctot, int sc_coff, float* aux_a,
                      float* aux_b, void* stream) {
    SAN_CHECK_ARG(part && scale && shift, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && tiles > 0, "bad dims");
    SAN_CHECK_ARG(mode >= 0 && mode <= 3, "bad mode");
    SAN_CHECK_ARG(mode != SAN_NORM_GROUP_BWD || (aux_a && aux_b), "GROUP_BWD needs both aux arrays ([2][n][c])");
    SAN_CHECK_ARG(check_view(sc_ctot, sc_coff, c), "bad scale/shift view");
    dim3 grid(c, mode == SAN_NORM_BATCH ? 1 : n);
    hipLaunchKernelGGL(norm_finalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, part, n, c, tiles, mode, eps,
                       gamma, beta, scale, shift, sc_ctot, sc_coff, aux_a, aux_b, (float*)nullptr, (float*)nullptr,
                       (long long*)nullptr, 0.f, 1.f);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_norm_finalize_bn(const float* part, int n, int c, int tiles, float eps, const float* gamma, const float* beta,
                         float* scale, float* shift, int sc_ctot, int sc_coff, float* aux_a, float* aux_b, float* rmean,
                         float* rvar, long long* num_batches_tracked, float momentum, float var_factor, void* stream) {
    SAN_CHECK_ARG(part && scale && shift && rmean && rvar, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && tiles > 0, "bad dims");
    SAN_CHECK_ARG(check_view(sc_ctot, sc_coff, c), "bad scale/shift view");
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(c, 1), dim3(256), 0, (hipStream_t)stream, part, n, c, tiles, SAN_NORM_BATCH,
                       eps, gamma, beta, scale, shift, sc_ctot, sc_coff, aux_a, aux_b, rmean, rvar, num_batches_tracked,
                       momentum, var_factor);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_plane_stat_tiles(int hw) {
    int t = san_cdiv(hw, 4096);
    return t < 1 ? 1 : (t > 32 ? 32 : t);
}

int san_plane_stats(const float* x, int x_ctot, int x_coff, int c, int n, int hw, float* part, void* stream) {
    SAN_CHECK_ARG(x && part, "null pointer");
    SAN_CHECK_ARG(n > 0 && hw > 0 && check_view(x_ctot, x_coff, c), "bad dims");
    const int tiles = san_plane_stat_tiles(hw);
    hipLaunchKernelGGL(plane_stats_kernel, dim3(tiles, c, n), dim3(kThreads), 0, (hipStream_t)stream, x, x_ctot,
                       x_coff, c, hw, tiles, part);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_bn_eval_affine(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                       float* scale, float* shift, int sc_ctot, int sc_coff, int n, int c, void* stream) {
    SAN_CHECK_ARG(gamma && beta && rmean && rvar && scale && shift, "null pointer");
    SAN_CHECK_ARG(n > 0 && check_view(sc_ctot, sc_coff, c), "bad dims");
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(san_cdiv(n * c, 256)), dim3(256), 0, (hipStream_t)stream, gamma,
                       beta, rmean, rvar, eps, scale, shift, sc_ctot, sc_coff, n, c);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_avgpool2_fwd(const float* x, int x_ctot, int x_coff, const float* sc, const float* sh, float slope, float* y,
                     int y_ctot, int y_coff, int n, int c, int h, int w, void* stream) {
    SAN_CHECK_ARG(x && y, "null pointer");
    SAN_CHECK_ARG(n > 0 && h >= 2 && w >= 2 && (h % 2 == 0) && (w % 2 == 0), "h, w must be even");
    SAN_CHECK_ARG(check_view(x_ctot, x_coff, c) && check_view(y_ctot, y_coff, c), "bad channel view");
    SAN_CHECK_ARG((sc == nullptr) == (sh == nullptr), "scale/shift must come together");
    EwArgs a{};
    a.x = x; a.sc = sc; a.sh = sh; a.slope = slope; a.x_ctot = x_ctot; a.x_coff = x_coff;
    a.y = y; a.y_ctot = y_ctot; a.y_coff = y_coff; a.n = n; a.c = c; a.h = h; a.w = w;
    hipLaunchKernelGGL(avgpool2_kernel, ew_grid((h / 2) * (w / 2) * 4, c, n), dim3(kThreads), 0, (hipStream_t)stream, a);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

// san_norm_finalize(SAN_NORM_INSTANCE) of the records `part` [n, c, tiles, 3] of x's channels [x_coff, x_coff + c) AND
// san_avgpool2_fwd of x read through the affine just computed, in one launch: y [n, y_ctot, h/2, w/2] channels from y_coff =
// avg_pool2d(lrelu(IN(x), slope)).  scale / shift: [n, sc_ctot] views at sc_coff (the same channels of x).  h, w even.
int san_norm_finalize_pool(const float* part, int n, int c, int tiles, float eps, float* scale, float* shift, int sc_ctot, int sc_coff,
                           const float* x, int x_ctot, int x_coff, float slope, float* y, int y_ctot, int y_coff, int h, int w,
                           void* stream) {
    SAN_CHECK_ARG(part && scale && shift && x && y, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && tiles > 0 && h >= 2 && w >= 2 && (h % 2 == 0) && (w % 2 == 0), "bad dims (h, w even)");
    SAN_CHECK_ARG(check_view(sc_ctot, sc_coff, c) && check_view(x_ctot, x_coff, c) && check_view(y_ctot, y_coff, c), "bad channel view");
    int K = san_cdiv((h / 2) * (w / 2), 256 * 12);      // ~12 output pixels per thread
    if (K < 1) K = 1;
    if (K > 8) K = 8;
    hipLaunchKernelGGL(norm_finalize_pool_kernel, dim3(K, c, n), dim3(256), 0, (hipStream_t)stream, part, c, tiles, eps, scale, shift,
                       sc_ctot, sc_coff, x, x_ctot, x_coff, slope, y, y_ctot, y_coff, h, w);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

// Channel `ch` of src [n, ctot, hw] (raw values) and its lazy-affine entries src_scale / src_shift [n, ctot] copied into `count`
// (<= 16) tensors of the same shape: dst / dst_scale / dst_shift are HOST arrays of `count` device pointers.  The cascades' shared
// reference channel (NormUnet.set_ref, varnet.py:315-319) in one launch instead of three per cascade.
int san_replicate_channel(const float* src, const float* src_scale, const float* src_shift, const void* dst, const void* dst_scale,
                          const void* dst_shift, int count, int n, int ctot, int ch, int hw, void* stream) {
    SAN_CHECK_ARG(src && src_scale && src_shift && dst && dst_scale && dst_shift, "null pointer");
    SAN_CHECK_ARG(count >= 1 && count <= 16 && n > 0 && hw > 0 && ch >= 0 && ch < ctot, "bad dims (count <= 16)");
    ReplArgs a{};
    a.src = src;
    a.ssc = src_scale;
    a.ssh = src_shift;
    for (int k = 0; k < count; ++k) {
        a.dst[k] = static_cast<float* const*>(dst)[k];
        a.dsc[k] = static_cast<float* const*>(dst_scale)[k];
        a.dsh[k] = static_cast<float* const*>(dst_shift)[k];
        SAN_CHECK_ARG(a.dst[k] && a.dsc[k] && a.dsh[k], "null destination");
    }
    a.ctot = ctot;
    a.ch = ch;
    a.n = n;
    a.hw = hw;
    a.count = count;
    int bx = san_cdiv(hw, kThreads * 8);
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(replicate_channel_kernel, dim3(bx, n, count), dim3(kThreads), 0, (hipStream_t)stream, a);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_upsample2_fwd(const float* x, int x_ctot, int x_coff, const float* sc, const float* sh, float slope, float* y,
                      int y_ctot, int y_coff, int n, int c, int h, int w, void* stream) {
    SAN_CHECK_ARG(x && y, "null pointer");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0, "bad dims");
    SAN_CHECK_ARG(check_view(x_ctot, x_coff, c) && check_view(y_ctot, y_coff, c), "bad channel view");
    SAN_CHECK_ARG((sc == nullptr) == (sh == nullptr), "scale/shift must come together");
    EwArgs a{};
    a.x = x; a.sc = sc; a.sh = sh; a.slope = slope; a.x_ctot = x_ctot; a.x_coff = x_coff;
    a.y = y; a.y_ctot = y_ctot; a.y_coff = y_coff; a.n = n; a.c = c; a.h = h; a.w = w;
    hipLaunchKernelGGL(upsample2_kernel, ew_grid(h * w * 4, c, n), dim3(kThreads), 0, (hipStream_t)stream, a);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_unshuffle2_fwd(const float* x, int x_ctot, int x_coff, float* y, int y_ctot, int y_coff, int n, int c, int h,
                       int w, void* stream) {
    SAN_CHECK_ARG(x && y, "null pointer");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0, "bad dims");
    SAN_CHECK_ARG(check_view(x_ctot, x_coff, c) && check_view(y_ctot, y_coff, 4 * c), "bad channel view");
    EwArgs a{};
    a.x = x; a.x_ctot = x_ctot; a.x_coff = x_coff; a.y = y; a.y_ctot = y_ctot; a.y_coff = y_coff;
    a.n = n; a.c = c; a.h = h; a.w = w;
    hipLaunchKernelGGL(unshuffle2_kernel, ew_grid(h * w * 4, c, n), dim3(kThreads), 0, (hipStream_t)stream, a);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_add_fwd(const float* a_, int a_ctot, int a_coff, const float* a_sc, const float* a_sh, float a_slope,
                const float* b, int b_ctot, int b_coff, const float* b_sc, const float* b_sh, float b_slope, float* y,
                int y_ctot, int y_coff, int n, int c, int hw, void* stream) {
    SAN_CHECK_ARG(a_ && b && y, "null pointer");
    SAN_CHECK_ARG(n > 0 && hw > 0, "bad dims");
    SAN_CHECK_ARG(check_view(a_ctot, a_coff, c) && check_view(b_ctot, b_coff, c) && check_view(y_ctot, y_coff, c),
                  "bad channel view");
    SAN_CHECK_ARG(((a_sc == nullptr) == (a_sh == nullptr)) && ((b_sc == nullptr) == (b_sh == nullptr)),
                  "scale/shift must come together");
    EwArgs a{};
    a.x = a_; a.sc = a_sc; a.sh = a_sh; a.slope = a_slope; a.x_ctot = a_ctot; a.x_coff = a_coff;
    a.b = b; a.b_sc = b_sc; a.b_sh = b_sh; a.b_slope = b_slope; a.b_ctot = b_ctot; a.b_coff = b_coff;
    a.y = y; a.y_ctot = y_ctot; a.y_coff = y_coff; a.n = n; a.c = c; a.h = hw; a.w = 1;
    hipLaunchKernelGGL(add_kernel, ew_grid(hw, c, n), dim3(kThreads), 0, (hipStream_t)stream, a);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_apply_fwd(const float* x, int x_ctot, int x_coff, const float* sc, const float* sh, float slope, float* y,
                  int y_ctot, int y_coff, int n, int c, int hw, void* stream) {
    SAN_CHECK_ARG(x && y, "null pointer");
    SAN_CHECK_ARG(n > 0 && hw > 0, "bad dims");
    SAN_CHECK_ARG(check_view(x_ctot, x_coff, c) && check_view(y_ctot, y_coff, c), "bad channel view");
    SAN_CHECK_ARG((sc == nullptr) == (sh == nullptr), "scale/shift must come together");
    EwArgs a{};
    a.x = x; a.sc = sc; a.sh = sh; a.slope = slope; a.x_ctot = x_ctot; a.x_coff = x_coff;
    a.y = y; a.y_ctot = y_ctot; a.y_coff = y_coff; a.n = n; a.c = c; a.h = hw; a.w = 1;
    hipLaunchKernelGGL(apply_kernel, ew_grid(hw, c, n), dim3(kThreads), 0, (hipStream_t)stream, a);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_window_copy_fwd(const float* x, int x_ctot, int x_coff, const float* sc, const float* sh, float slope, int hx,
                        int wx, float* y, int y_ctot, int y_coff, int hy, int wy, int off_y, int off_x, int mode, int n,
                        int c, void* stream) {
    SAN_CHECK_ARG(x && y, "null pointer");
    SAN_CHECK_ARG(n > 0 && hx > 0 && wx > 0 && hy > 0 && wy > 0, "bad dims");
    SAN_CHECK_ARG(check_view(x_ctot, x_coff, c) && check_view(y_ctot, y_coff, c), "bad channel view");
    SAN_CHECK_ARG((sc == nullptr) == (sh == nullptr), "scale/shift must come together");
    SAN_CHECK_ARG(mode >= 0 && mode <= 2, "bad mode");
    if (mode == 1) SAN_CHECK_ARG(off_y == 0 && off_x == 0 && hy - hx >= 0 && hy - hx <= 1 && wy - wx >= 0 && wy - wx <= 1 &&
                                 hx >= 2 && wx >= 2, "reflect: pads one row / column at the bottom / right");
    if (mode == 2) SAN_CHECK_ARG(off_y == 0 && off_x == 0 && hx - hy >= 0 && hx - hy <= 1 && wx - wy >= 0 && wx - wy <= 1 &&
                                 hy >= 2 && wy >= 2, "reflect adjoint: x is the padded plane");
    WinArgs a{};
    a.x = x; a.sc = sc; a.sh = sh; a.slope = slope; a.x_ctot = x_ctot; a.x_coff = x_coff; a.hx = hx; a.wx = wx;
    a.y = y; a.y_ctot = y_ctot; a.y_coff = y_coff; a.hy = hy; a.wy = wy; a.off_y = off_y; a.off_x = off_x; a.mode = mode;
    hipLaunchKernelGGL(window_copy_kernel, ew_grid(hy * wy, c, n), dim3(kThreads), 0, (hipStream_t)stream, a);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

}  // extern "C"
