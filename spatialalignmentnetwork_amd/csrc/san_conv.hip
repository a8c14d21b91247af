// Direct fp32 convolution for gfx950 with wave-uniform (scalar-register) weights.
//
// Why not an MFMA implicit GEMM here: the cascade U-Net's channel counts are
// 3/18/36/72/144/288.  fp32-input MFMA runs at exactly the fp32 VALU rate on
// CDNA4 (157 TF both), so for fp32 parity the matrix core buys no throughput,
// while its 16/32-wide tiles would waste 44 % (Cout 18 -> 32) and 25 %
// (36 -> 48) of the work on the layers that hold half the MACs.  Instead every
// lane owns a 1x4 strip of output pixels and CO_T output channels in registers
// (CO_T in {2,4,8,16,18}: zero padding waste for all of the above), the input
// halo tile is staged once in LDS with the producer's lazy normalisation
// (scale/shift/LeakyReLU) applied on the way in, and the weights are read
// through the scalar cache into SGPRs: one v_fmac_f32 per MAC with an SGPR
// operand, 36*CO_T FMAs per 18 LDS dwords.
//
// The epilogue fuses bias, the per-tile (count, mean, M2) statistics that
// InstanceNorm / BatchNorm need (so normalisation never re-reads the tensor),
// and an optional per-(n, c) output affine (NormUnet.unnorm).
#include "san_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPX = 4;

struct ConvGeom {
    int co_t;     // output channels per lane
    int groups;   // ceil(cout / co_t)
    int LX, LY;   // lanes per wave along x / y (LX*LY <= 64)
    int WY, WC;   // waves along y / along cout groups (WY*WC == 4)
    int TW, TH;   // workgroup pixel tile
    int tiles_x, tiles_y;
    int ck;       // input channels staged per LDS round
    int pitch;    // LDS row pitch (floats)
    int rows_t;   // staged rows per channel
};

struct ConvArgs {
    const float* in_scale;
    const float* in_shift;
    const float* bias;
    const float* out_scale;
    const float* out_shift;
    float* part;
    float in_slope;
    int x_ctot, x_coff, cin;
    int y_ctot, y_coff, cout;
    int N, H, W;
    ConvGeom g;
};

int pick_co_t(int cout) {
    const int cand[5] = {18, 16, 8, 4, 2};
    int best = 2, best_waste = 1 << 30;
    for (int i = 0; i < 5; ++i) {
        int ct = cand[i];
        int waste = san_cdiv(cout, ct) * ct - cout;
        if (waste < best_waste) {
            best_waste = waste;
            best = ct;
        }
    }
    return best;
}

ConvGeom conv_geom(int H, int W, int cin, int cout, int ks) {
    ConvGeom g{};
    g.co_t = pick_co_t(cout);
    g.groups = san_cdiv(cout, g.co_t);
    g.WC = g.groups >= 4 ? 4 : (g.groups >= 2 ? 2 : 1);
    g.WY = 4 / g.WC;
    const int lxc[5] = {8, 10, 5, 4, 16};
    double best = -1.0;
    for (int i = 0; i < 5; ++i) {
        int LX = lxc[i], LY = 64 / LX;
        int TW = LX * kPX, TH = LY * g.WY;
        double ex = (double)W / (san_cdiv(W, TW) * TW);
        double ey = (double)H / (san_cdiv(H, TH) * TH);
        double el = (double)(LX * LY) / 64.0;
        double e = ex * ey * el;
        if (e > best + 1e-9) {
            best = e;
            g.LX = LX;
            g.LY = LY;
        }
    }
    g.TW = g.LX * kPX;
    g.TH = g.LY * g.WY;
    g.tiles_x = san_cdiv(W, g.TW);
    g.tiles_y = san_cdiv(H, g.TH);
    const int pad = ks / 2;
    g.pitch = g.TW + (pad ? 4 : 0);
    g.rows_t = g.TH + 2 * pad;
    int per_ch = g.rows_t * g.pitch * (int)sizeof(float);
    g.ck = (32 * 1024) / per_ch;
    if (g.ck < 1) g.ck = 1;
    if (g.ck > cin) g.ck = cin;
    if (g.ck > 16) g.ck = 16;
    return g;
}

extern __shared__ __attribute__((aligned(16))) float conv_lds[];

template <int CO_T, int KS>
__global__ void __launch_bounds__(kThreads)
conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                const ConvArgs a) {
    constexpr int PAD = KS / 2;
    constexpr int TAPS = KS * KS;
    const ConvGeom& G = a.g;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wc = wave % G.WC;
    const int wy = wave / G.WC;
    const int ly = lane / G.LX;
    const int lx = lane - ly * G.LX;
    const bool lane_ok = ly < G.LY;
    const int ty = blockIdx.x / G.tiles_x;
    const int tx = blockIdx.x - ty * G.tiles_x;
    const int x0 = tx * G.TW, y0 = ty * G.TH;
    const int n = blockIdx.z;
    const int grp = blockIdx.y * G.WC + wc;       // wave-uniform
    const bool g_ok = grp < G.groups;
    const int H = a.H, W = a.W;
    const int pitch = G.pitch, rows_t = G.rows_t;
    const int cols_t = G.TW + 2 * PAD;

    float acc[CO_T][kPX];
#pragma unroll
    for (int c = 0; c < CO_T; ++c)
#pragma unroll
        for (int p = 0; p < kPX; ++p) acc[c][p] = 0.f;

    const int my_row = wy * G.LY + ly;   // row of this lane's outputs inside the tile

    for (int c0 = 0; c0 < a.cin; c0 += G.ck) {
        const int ckk = min(G.ck, a.cin - c0);
        __syncthreads();
        // ---- stage ckk channels x rows_t rows x cols_t cols, lazily normalised
        {
            int r = wave, ci = 0;
            while (r >= rows_t) {
                r -= rows_t;
                ++ci;
            }
            const int total = ckk * rows_t;
            for (int rr = wave; rr < total; rr += 4) {
                const int gy = y0 - PAD + r;
                const int chan = a.x_coff + c0 + ci;
                float sc = 1.f, sh = 0.f;
                if (a.in_scale) {
                    sc = a.in_scale[n * a.x_ctot + chan];
                    sh = a.in_shift[n * a.x_ctot + chan];
                }
                const bool row_ok = (gy >= 0) && (gy < H);
                const float* src = x + ((size_t)(n * a.x_ctot + chan) * H + (row_ok ? gy : 0)) * W;
                float* dst = conv_lds + (ci * rows_t + r) * pitch;
                for (int col = lane; col < cols_t; col += 64) {
                    const int gx = x0 - PAD + col;
                    float v = 0.f;
                    if (row_ok && gx >= 0 && gx < W) v = san_act(src[gx], sc, sh, a.in_slope);
                    dst[col] = v;
                }
                r += 4;
                while (r >= rows_t) {
                    r -= rows_t;
                    ++ci;
                }
            }
        }
        __syncthreads();
        if (lane_ok && g_ok) {
            const float* wbase = wp + ((size_t)grp * a.cin + c0) * (TAPS * CO_T);
            for (int ci = 0; ci < ckk; ++ci) {
                const float* wci = wbase + ci * (TAPS * CO_T);
                const float* t = conv_lds + (ci * rows_t + my_row) * pitch + kPX * lx;
                if (KS == 3) {
                    float in[3][6];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const float4 v4 = *reinterpret_cast<const float4*>(t + ky * pitch);
                        const float2 v2 = *reinterpret_cast<const float2*>(t + ky * pitch + 4);
                        in[ky][0] = v4.x;
                        in[ky][1] = v4.y;
                        in[ky][2] = v4.z;
                        in[ky][3] = v4.w;
                        in[ky][4] = v2.x;
                        in[ky][5] = v2.y;
                    }
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int c = 0; c < CO_T; ++c) {
                                const float wv = wci[(ky * 3 + kx) * CO_T + c];
#pragma unroll
                                for (int p = 0; p < kPX; ++p) acc[c][p] = fmaf(wv, in[ky][kx + p], acc[c][p]);
                            }
                } else {
                    const float4 v4 = *reinterpret_cast<const float4*>(t);
                    const float in[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int c = 0; c < CO_T; ++c) {
                        const float wv = wci[c];
#pragma unroll
                        for (int p = 0; p < kPX; ++p) acc[c][p] = fmaf(wv, in[p], acc[c][p]);
                    }
                }
            }
        }
    }

    // ------------------------------------------------------------ epilogue
    const int oy = y0 + my_row;
    const int ox = x0 + kPX * lx;
    bool valid[kPX];
#pragma unroll
    for (int p = 0; p < kPX; ++p) valid[p] = lane_ok && g_ok && oy < H && (ox + p) < W;

    if (a.bias) {
#pragma unroll
        for (int c = 0; c < CO_T; ++c) {
            const int co = grp * CO_T + c;
            const float b = (g_ok && co < a.cout) ? a.bias[co] : 0.f;
#pragma unroll
            for (int p = 0; p < kPX; ++p) acc[c][p] += b;
        }
    }

    if (a.part) {
        __syncthreads();
        float* red = conv_lds;               // [4][CO_T]
        float* redc = conv_lds + 4 * CO_T;   // [4]
        float* red2 = redc + 4;              // [4][CO_T]
        float cnt = 0.f;
#pragma unroll
        for (int p = 0; p < kPX; ++p) cnt += valid[p] ? 1.f : 0.f;
        cnt = san_wave_sum(cnt);
        if (lane == 0) redc[wave] = cnt;
#pragma unroll
        for (int c = 0; c < CO_T; ++c) {
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < kPX; ++p) s += valid[p] ? acc[c][p] : 0.f;
            s = san_wave_sum(s);
            if (lane == 0) red[wave * CO_T + c] = s;
        }
        __syncthreads();
        float tcnt = 0.f;
        for (int v = 0; v < G.WY; ++v) tcnt += redc[v * G.WC + wc];
        const float inv = tcnt > 0.f ? 1.f / tcnt : 0.f;
#pragma unroll
        for (int c = 0; c < CO_T; ++c) {
            float tot = 0.f;
            for (int v = 0; v < G.WY; ++v) tot += red[(v * G.WC + wc) * CO_T + c];
            const float mean = tot * inv;
            float d = 0.f;
#pragma unroll
            for (int p = 0; p < kPX; ++p) {
                const float e = acc[c][p] - mean;
                d += valid[p] ? e * e : 0.f;
            }
            d = san_wave_sum(d);
            if (lane == 0) red2[wave * CO_T + c] = d;
        }
        __syncthreads();
        if (wy == 0 && lane < CO_T && g_ok) {
            const int co = grp * CO_T + lane;
            if (co < a.cout) {
                float tot = 0.f, m2 = 0.f;
                for (int v = 0; v < G.WY; ++v) {
                    tot += red[(v * G.WC + wc) * CO_T + lane];
                    m2 += red2[(v * G.WC + wc) * CO_T + lane];
                }
                const int tiles = G.tiles_x * G.tiles_y;
                float* o = a.part + ((size_t)(n * a.cout + co) * tiles + blockIdx.x) * 3;
                o[0] = tcnt;
                o[1] = tot * inv;
                o[2] = m2;
            }
        }
    }

    if (!(lane_ok && g_ok) || oy >= H) return;
    const bool vec = ((W & 3) == 0) && (ox + 3 < W);
#pragma unroll
    for (int c = 0; c < CO_T; ++c) {
        const int co = grp * CO_T + c;
        if (co >= a.cout) break;
        float os = 1.f, ob = 0.f;
        if (a.out_scale) {
            os = a.out_scale[n * a.cout + co];
            ob = a.out_shift[n * a.cout + co];
        }
        float* dst = y + ((size_t)(n * a.y_ctot + a.y_coff + co) * H + oy) * W + ox;
        if (vec) {
            float4 o;
            o.x = fmaf(acc[c][0], os, ob);
            o.y = fmaf(acc[c][1], os, ob);
            o.z = fmaf(acc[c][2], os, ob);
            o.w = fmaf(acc[c][3], os, ob);
            *reinterpret_cast<float4*>(dst) = o;
        } else {
#pragma unroll
            for (int p = 0; p < kPX; ++p)
                if (ox + p < W) dst[p] = fmaf(acc[c][p], os, ob);
        }
    }
}

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
        for (int ci = 0; ci < a.cin; ++ci) {
            float sc = 1.f, sh = 0.f;
            if (a.in_scale) {
                sc = a.in_scale[n * a.x_ctot + a.x_coff + ci];
                sh = a.in_shift[n * a.x_ctot + a.x_coff + ci];
            }
            const float v = san_act(xb[(size_t)ci * HW], sc, sh, a.in_slope);
            const float* wci = wb + ci * (4 * CO_T);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int c = 0; c < CO_T; ++c) acc[c][t] = fmaf(wci[t * CO_T + c], v, acc[c][t]);
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
    a.co_t = pick_co_t(cout);
    a.groups = san_cdiv(cout, a.co_t);
    a.WC = a.groups >= 4 ? 4 : (a.groups >= 2 ? 2 : 1);
    a.WY = 4 / a.WC;
    a.tiles = san_cdiv(H * W, 64 * a.WY);
}

template <int KS>
int launch_conv(const float* x, const float* wp, float* y, const ConvArgs& a, hipStream_t s) {
    const ConvGeom& g = a.g;
    dim3 grid(g.tiles_x * g.tiles_y, san_cdiv(g.groups, g.WC), a.N);
    size_t lds = (size_t)g.ck * g.rows_t * g.pitch * sizeof(float);
    size_t need = (size_t)(4 * g.co_t * 2 + 4) * sizeof(float);
    if (lds < need) lds = need;
    switch (g.co_t) {
        case 2: hipLaunchKernelGGL((conv_fwd_kernel<2, KS>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
        case 4: hipLaunchKernelGGL((conv_fwd_kernel<4, KS>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
        case 8: hipLaunchKernelGGL((conv_fwd_kernel<8, KS>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
        case 16: hipLaunchKernelGGL((conv_fwd_kernel<16, KS>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
        case 18: hipLaunchKernelGGL((conv_fwd_kernel<18, KS>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
        default: san_set_error("bad co_t %d", g.co_t); return SAN_E_UNSUPPORTED;
    }
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

}  // namespace

extern "C" {

size_t san_conv_packed_floats(int cout, int cin, int ks) {
    int ct = pick_co_t(cout);
    return (size_t)san_cdiv(cout, ct) * ct * (size_t)cin * ks * ks;
}

int san_conv_pack_weights(const float* w, float* packed, int cout, int cin, int ks, int transposed, void* stream) {
    SAN_CHECK_ARG(w && packed, "null pointer");
    SAN_CHECK_ARG(cout > 0 && cin > 0 && ks >= 1 && ks <= 3, "bad dims");
    int ct = pick_co_t(cout);
    int groups = san_cdiv(cout, ct);
    int total = groups * cin * ks * ks * ct;
    int blocks = san_cdiv(total, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, cout, cin,
                       ks * ks, ct, groups, transposed);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_conv_stat_tiles(int h, int w, int cin, int cout, int ks) {
    ConvGeom g = conv_geom(h, w, cin, cout, ks);
    return g.tiles_x * g.tiles_y;
}

int san_conv2d_fwd(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                   float in_slope, const float* w_packed, const float* bias, float* y, int y_ctot, int y_coff,
                   int cout, const float* out_scale, const float* out_shift, float* part_stats, int n, int h, int w,
                   int ks, void* stream) {
    SAN_CHECK_ARG(x && w_packed && y, "null pointer");
    SAN_CHECK_ARG(ks == 1 || ks == 3, "ks must be 1 or 3");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "bad dims");
    SAN_CHECK_ARG(x_coff >= 0 && x_coff + cin <= x_ctot && y_coff >= 0 && y_coff + cout <= y_ctot, "bad channel view");
    SAN_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "in_scale/in_shift must come together");
    SAN_CHECK_ARG((out_scale == nullptr) == (out_shift == nullptr), "out_scale/out_shift must come together");
    ConvArgs a{};
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_slope = in_slope;
    a.bias = bias;
    a.out_scale = out_scale;
    a.out_shift = out_shift;
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
    a.g = conv_geom(h, w, cin, cout, ks);
    if (ks == 3) return launch_conv<3>(x, w_packed, y, a, (hipStream_t)stream);
    return launch_conv<1>(x, w_packed, y, a, (hipStream_t)stream);
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
