// ConvTranspose2d 2x2 stride 2 and the weight repacking kernel for gfx950.
#include "san_conv_common.h"

namespace {

constexpr int kThreads = 256;

// ------------------------------------------------------------------------
// ConvTranspose2d 2x2 stride 2: every input pixel feeds a private 2x2 output
// block, so it is four independent 1x1 convolutions.  Lane = one input pixel,
// CO_T channels x 4 outputs in registers; inputs come straight from global
// (no reuse between lanes), weights through SGPRs.
struct TconvArgs {
    const float* in_scale;
    const float* in_shift;
    float* part;
    float in_slope;
    int x_ctot, x_coff, cin;
    int y_ctot, y_coff, cout;
    int N, H, W;
    int co_t, groups, WY, WC, tiles;
};

template <int CO_T>
__global__ void __launch_bounds__(kThreads)
tconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                 const TconvArgs a) {
    __shared__ float red[4 * CO_T * 4 * 2 + 4];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wc = wave % a.WC;
    const int wy = wave / a.WC;
    const int n = blockIdx.z;
    const int grp = blockIdx.y * a.WC + wc;
    const bool g_ok = grp < a.groups;
    const int HW = a.H * a.W;
    const int pix = (blockIdx.x * a.WY + wy) * 64 + lane;
    const bool ok = g_ok && pix < HW;
    const int py = pix / a.W;
    const int px = pix - py * a.W;

    float acc[CO_T][4];
#pragma unroll
    for (int c = 0; c < CO_T; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[c][t] = 0.f;

    if (g_ok) {
        const float* xb = x + ((size_t)(n * a.x_ctot + a.x_coff)) * HW + (ok ? pix : 0);
        const float* wb = wp + (size_t)grp * a.cin * (4 * CO_T);
        // 8 input channels per round: the 8 loads are issued back to back (memory-level
        // parallelism) and the next round's loads are in flight while this round's FMAs run
        constexpr int U = 8;
        const int aff = n * a.x_ctot + a.x_coff;
        float nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = (u < a.cin) ? xb[(size_t)u * HW] : 0.f;
        for (int c0 = 0; c0 < a.cin; c0 += U) {
            float cur[U];
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            if (c0 + U < a.cin) {
#pragma unroll
                for (int u = 0; u < U; ++u) nxt[u] = (c0 + U + u < a.cin) ? xb[(size_t)(c0 + U + u) * HW] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c0 + u < a.cin) {
                    float sc = 1.f, sh = 0.f;
                    if (a.in_scale) {
                        sc = a.in_scale[aff + c0 + u];
                        sh = a.in_shift[aff + c0 + u];
                    }
                    const float v = san_act(cur[u], sc, sh, a.in_slope);
                    const float* wci = wb + (size_t)(c0 + u) * (4 * CO_T);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int c = 0; c < CO_T; ++c) acc[c][t] = fmaf(wci[t * CO_T + c], v, acc[c][t]);
                }
            }
        }
    }

    if (a.part) {
        float* r1 = red;                      // [4][CO_T]
        float* rc = red + 4 * CO_T;           // [4]
        float* r2 = rc + 4;                   // [4][CO_T]
        float cnt = san_wave_sum(ok ? 4.f : 0.f);
        if (lane == 0) rc[wave] = cnt;
#pragma unroll
        for (int c = 0; c < CO_T; ++c) {
            float s = ok ? (acc[c][0] + acc[c][1]) + (acc[c][2] + acc[c][3]) : 0.f;
            s = san_wave_sum(s);
            if (lane == 0) r1[wave * CO_T + c] = s;
        }
        __syncthreads();
        float tcnt = 0.f;
        for (int v = 0; v < a.WY; ++v) tcnt += rc[v * a.WC + wc];
        const float inv = tcnt > 0.f ? 1.f / tcnt : 0.f;
#pragma unroll
        for (int c = 0; c < CO_T; ++c) {
            float tot = 0.f;
            for (int v = 0; v < a.WY; ++v) tot += r1[(v * a.WC + wc) * CO_T + c];
            const float mean = tot * inv;
            float d = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float e = acc[c][t] - mean;
                d += e * e;
            }
            d = san_wave_sum(ok ? d : 0.f);
            if (lane == 0) r2[wave * CO_T + c] = d;
        }
        __syncthreads();
        if (wy == 0 && lane < CO_T && g_ok) {
            const int co = grp * CO_T + lane;
            if (co < a.cout) {
                float tot = 0.f, m2 = 0.f;
                for (int v = 0; v < a.WY; ++v) {
                    tot += r1[(v * a.WC + wc) * CO_T + lane];
                    m2 += r2[(v * a.WC + wc) * CO_T + lane];
                }
                float* o = a.part + ((size_t)(n * a.cout + co) * a.tiles + blockIdx.x) * 3;
                o[0] = tcnt;
                o[1] = tot * inv;
                o[2] = m2;
            }
        }
    }
    if (!ok) return;
    const int OW = 2 * a.W;
#pragma unroll
    for (int c = 0; c < CO_T; ++c) {
        const int co = grp * CO_T + c;
        if (co >= a.cout) break;
        float* dst = y + ((size_t)(n * a.y_ctot + a.y_coff + co) * (2 * a.H) + 2 * py) * OW + 2 * px;
        *reinterpret_cast<float2*>(dst) = make_float2(acc[c][0], acc[c][1]);
        *reinterpret_cast<float2*>(dst + OW) = make_float2(acc[c][2], acc[c][3]);
    }
}

// w [cout, cin, ks, ks] (or [cin, cout, ks, ks] transposed) -> [groups][cin][taps][co_t]
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ packed, int cout, int cin,
                                    int taps, int co_t, int groups, int transposed) {
    const int total = groups * cin * taps * co_t;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int c = i % co_t;
        int t = (i / co_t) % taps;
        int ci = (i / (co_t * taps)) % cin;
        int g = i / (co_t * taps * cin);
        int co = g * co_t + c;
        float v = 0.f;
        if (co < cout) v = transposed ? w[((size_t)ci * cout + co) * taps + t] : w[((size_t)co * cin + ci) * taps + t];
        packed[i] = v;
    }
}

void tconv_geom(int H, int W, int cout, TconvArgs& a) {
    a.co_t = san_pick_co_t(cout);
    a.groups = san_cdiv(cout, a.co_t);
    a.WC = a.groups >= 4 ? 4 : (a.groups >= 2 ? 2 : 1);
    a.WY = 4 / a.WC;
    a.tiles = san_cdiv(H * W, 64 * a.WY);
}

}  // namespace

extern "C" {

int san_conv_pack_weights(const float* w, float* packed, int cout, int cin, int ks, int transposed, void* stream) {
    SAN_CHECK_ARG(w && packed, "null pointer");
    SAN_CHECK_ARG(cout > 0 && cin > 0 && ks >= 1 && ks <= 3, "bad dims");
    int ct = san_pick_co_t(cout);
    int groups = san_cdiv(cout, ct);
    int total = groups * cin * ks * ks * ct;
    int blocks = san_cdiv(total, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, cout, cin,
                       ks * ks, ct, groups, transposed);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_tconv_stat_tiles(int h, int w, int cout) {
    TconvArgs a{};
    tconv_geom(h, w, cout, a);
    return a.tiles;
}

int san_tconv2x2_fwd(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                     float in_slope, const float* w_packed, float* y, int y_ctot, int y_coff, int cout,
                     float* part_stats, int n, int h, int w, void* stream) {
    SAN_CHECK_ARG(x && w_packed && y, "null pointer");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "bad dims");
    SAN_CHECK_ARG(x_coff >= 0 && x_coff + cin <= x_ctot && y_coff >= 0 && y_coff + cout <= y_ctot, "bad channel view");
    SAN_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "in_scale/in_shift must come together");
    TconvArgs a{};
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_slope = in_slope;
    a.part = part_stats;
    a.x_ctot = x_ctot;
    a.x_coff = x_coff;
    a.cin = cin;
    a.y_ctot = y_ctot;
    a.y_coff = y_coff;
    a.cout = cout;
    a.N = n;
    a.H = h;
    a.W = w;
    tconv_geom(h, w, cout, a);
    dim3 grid(a.tiles, san_cdiv(a.groups, a.WC), n);
    hipStream_t s = (hipStream_t)stream;
    switch (a.co_t) {
        case 2: hipLaunchKernelGGL((tconv_fwd_kernel<2>), grid, dim3(kThreads), 0, s, x, w_packed, y, a); break;
        case 4: hipLaunchKernelGGL((tconv_fwd_kernel<4>), grid, dim3(kThreads), 0, s, x, w_packed, y, a); break;
        case 8: hipLaunchKernelGGL((tconv_fwd_kernel<8>), grid, dim3(kThreads), 0, s, x, w_packed, y, a); break;
        case 16: hipLaunchKernelGGL((tconv_fwd_kernel<16>), grid, dim3(kThreads), 0, s, x, w_packed, y, a); break;
        case 18: hipLaunchKernelGGL((tconv_fwd_kernel<18>), grid, dim3(kThreads), 0, s, x, w_packed, y, a); break;
        default: san_set_error("bad co_t %d", a.co_t); return SAN_E_UNSUPPORTED;
    }
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

}  // extern "C"
