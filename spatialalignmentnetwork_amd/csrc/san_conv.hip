// Direct fp32 convolution (3x3 / 1x1) for gfx950 with wave-uniform (scalar-register) weights.
//
// Why not an MFMA implicit GEMM here: the cascade U-Net's channel counts are
// 3/18/36/72/144/288.  fp32-input MFMA runs at exactly the fp32 VALU rate on
// CDNA4 (157 TF both), so for fp32 parity the matrix core buys no throughput,
// while its 16/32-wide tiles would waste 44 % (Cout 18 -> 32) and 25 %
// (36 -> 48) of the work on the layers that hold half the MACs.  Instead every
// lane owns a 1 x PX strip of output pixels and CO_T output channels in
// registers (CO_T in {2,4,8,16,18}: zero padding waste for all of the above),
// the input halo tile is staged in LDS with the producer's lazy normalisation
// (scale/shift/LeakyReLU) applied on the way in, and the weights are read
// through the scalar cache into SGPRs (v_pk_fma_f32 with an SGPR-pair operand).
//
// Pipeline per workgroup (4 waves): global -> registers for channel chunk k+1 is
// issued before the FMAs of chunk k, so HBM/L2 latency hides under compute; one
// LDS buffer, two barriers per chunk.
//
// Geometry is chosen per layer on the host (conv_geom): PX in {4,2,1} shrinks
// the per-lane tile at the deep, low-resolution levels so that every layer
// still launches >= ~2 waves per SIMD; LX x LY lanes per wave follow the image
// width (32/40/20/16/64-wide tiles); WC waves share one staged input tile across
// output-channel groups, WY waves stack in y.
//
// The epilogue fuses bias, the per-tile (count, mean, M2) statistics that
// InstanceNorm / BatchNorm need (so normalisation never re-reads the tensor),
// and an optional per-(n, c) output affine (NormUnet.unnorm).
#include "san_conv_common.h"

namespace {

constexpr int kThreads = 256;

struct ConvGeom {
    int co_t;     // output channels per lane
    int px;       // output pixels per lane (along x)
    int groups;   // ceil(cout / co_t)
    int LX, LY;   // lanes per wave along x / y (LX*LY <= 64)
    int WY, WC;   // waves along y / along cout groups (WY*WC == 4)
    int TW, TH;   // workgroup pixel tile
    int tiles_x, tiles_y;
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

// per-PX staging constants: SLOTS = ceil(max staged tile elements / 256), CK = channels per round
template <int PX> struct StageCfg;
template <> struct StageCfg<4> { static constexpr int SLOTS = 5, CK = 4; };
template <> struct StageCfg<2> { static constexpr int SLOTS = 3, CK = 8; };
template <> struct StageCfg<1> { static constexpr int SLOTS = 2, CK = 8; };

int stage_slots(int px) { return px == 4 ? 5 : (px == 2 ? 3 : 2); }
int stage_ck(int px) { return px == 4 ? 4 : 8; }

ConvGeom conv_geom(int N, int H, int W, int cin, int cout, int ks) {
    (void)cin;
    ConvGeom g{};
    g.co_t = san_pick_co_t(cout);
    g.groups = san_cdiv(cout, g.co_t);
    g.WC = g.groups >= 4 ? 4 : (g.groups >= 2 ? 2 : 1);
    g.WY = 4 / g.WC;
    const int pad = ks / 2;
    const int pxc[3] = {4, 2, 1};
    const double pxw[3] = {1.0, 0.92, 0.80};           // smaller strips pay more LDS traffic per FMA
    const int lxc[8] = {8, 10, 5, 4, 16, 20, 32, 64};
    double best = -1.0;
    for (int ip = 0; ip < 3; ++ip) {
        const int PX = pxc[ip];
        for (int il = 0; il < 8; ++il) {
            const int LX = lxc[il], LY = 64 / LX;
            if (LY < 1) continue;
            const int TW = LX * PX, TH = LY * g.WY;
            if ((TW + 2 * pad) * (TH + 2 * pad) > stage_slots(PX) * kThreads) continue;
            const int tx = san_cdiv(W, TW), ty = san_cdiv(H, TH);
            const double ex = (double)W / (tx * TW), ey = (double)H / (ty * TH);
            const double el = (double)(LX * LY) / 64.0;
            const double waves = (double)tx * ty * N * 4.0 * san_cdiv(g.groups, g.WC);
            const double fill = waves >= 2048.0 ? 1.0 : (0.25 + 0.75 * waves / 2048.0);   // want >= 2 waves per SIMD
            const double halo = (double)(TW * TH) / ((TW + 2 * pad) * (TH + 2 * pad));     // staged bytes that are payload
            const double e = ex * ey * el * fill * pxw[ip] * (0.8 + 0.2 * halo);
            if (e > best + 1e-9) {
                best = e;
                g.px = PX;
                g.LX = LX;
                g.LY = LY;
            }
        }
    }
    g.TW = g.LX * g.px;
    g.TH = g.LY * g.WY;
    g.tiles_x = san_cdiv(W, g.TW);
    g.tiles_y = san_cdiv(H, g.TH);
    const int cols = g.TW + 2 * pad;
    g.pitch = (cols + 3) & ~3;
    g.rows_t = g.TH + 2 * pad;
    return g;
}

extern __shared__ __attribute__((aligned(16))) float conv_lds[];

template <int PX, int N>
__device__ __forceinline__ void load_row(const float* t, float (&in)[N]) {
    // N = PX + 2 (3x3) or PX (1x1) consecutive floats starting at t (aligned to PX floats)
    if (PX == 4) {
        const float4 v4 = *reinterpret_cast<const float4*>(t);
        in[0] = v4.x;
        in[1] = v4.y;
        in[2] = v4.z;
        in[3] = v4.w;
        if (N > 4) {
            const float2 v2 = *reinterpret_cast<const float2*>(t + 4);
            in[4] = v2.x;
            in[N - 1] = v2.y;
        }
    } else if (PX == 2) {
        const float2 a = *reinterpret_cast<const float2*>(t);
        in[0] = a.x;
        in[1] = a.y;
        if (N > 2) {
            const float2 b = *reinterpret_cast<const float2*>(t + 2);
            in[2] = b.x;
            in[N - 1] = b.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) in[i] = t[i];
    }
}

template <int CO_T, int KS, int PX>
__global__ void __launch_bounds__(kThreads)
conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                const ConvArgs a) {
    constexpr int PAD = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int SLOTS = StageCfg<PX>::SLOTS;
    constexpr int CK = StageCfg<PX>::CK;
    constexpr int NIN = PX + 2 * PAD;
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
    const int tile_elems = rows_t * cols_t;
    const int tile_stride = rows_t * pitch;       // floats per staged channel
    const size_t HW = (size_t)H * W;

    // Per-thread staging slots: element e = tid + s*256 of one channel's (rows_t x cols_t)
    // halo tile.  The geometry is the same for every channel, so the global offset inside a
    // plane and the LDS offset inside a channel tile are computed once.
    int goff[SLOTS], loff[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int e = tid + s * kThreads;
        goff[s] = -1;
        loff[s] = -1;
        if (e < tile_elems) {
            const int r = e / cols_t, col = e - r * cols_t;
            const int gy = y0 - PAD + r, gx = x0 - PAD + col;
            loff[s] = r * pitch + col;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) goff[s] = gy * W + gx;
        }
    }

    float acc[CO_T][PX];
#pragma unroll
    for (int c = 0; c < CO_T; ++c)
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[c][p] = 0.f;

    const int my_row = wy * G.LY + ly;   // row of this lane's outputs inside the tile
    const float* xn = x + (size_t)(n * a.x_ctot + a.x_coff) * HW;
    const int aff = n * a.x_ctot + a.x_coff;

    // ---- software pipeline: global -> registers for chunk k+1 is in flight while chunk k computes
    float stage[CK][SLOTS];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int ci = 0; ci < CK; ++ci) {
            const bool ch_ok = (c0 + ci) < a.cin;
            const float* src = xn + (size_t)(ch_ok ? (c0 + ci) : 0) * HW;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) stage[ci][s] = (ch_ok && goff[s] >= 0) ? src[goff[s]] : 0.f;
        }
    };
    prefetch(0);

    for (int c0 = 0; c0 < a.cin; c0 += CK) {
        const int ckk = min(CK, a.cin - c0);
        __syncthreads();   // every wave is done reading the previous chunk
#pragma unroll
        for (int ci = 0; ci < CK; ++ci) {
            float sc = 1.f, sh = 0.f;
            if (a.in_scale && ci < ckk) {
                sc = a.in_scale[aff + c0 + ci];
                sh = a.in_shift[aff + c0 + ci];
            }
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (loff[s] >= 0)
                    conv_lds[ci * tile_stride + loff[s]] = goff[s] >= 0 ? san_act(stage[ci][s], sc, sh, a.in_slope) : 0.f;
        }
        __syncthreads();
        if (c0 + CK < a.cin) prefetch(c0 + CK);
        if (lane_ok && g_ok) {
            const float* wbase = wp + ((size_t)grp * a.cin + c0) * (TAPS * CO_T);
            for (int ci = 0; ci < ckk; ++ci) {
                const float* wci = wbase + ci * (TAPS * CO_T);
                const float* t = conv_lds + ci * tile_stride + my_row * pitch + PX * lx;
                float in[KS][NIN];
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) load_row<PX, NIN>(t + ky * pitch, in[ky]);
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                        for (int c = 0; c < CO_T; ++c) {
                            const float wv = wci[(ky * KS + kx) * CO_T + c];
#pragma unroll
                            for (int p = 0; p < PX; ++p) acc[c][p] = fmaf(wv, in[ky][kx + p], acc[c][p]);
                        }
            }
        }
    }

    // ------------------------------------------------------------ epilogue
    const int oy = y0 + my_row;
    const int ox = x0 + PX * lx;
    bool valid[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) valid[p] = lane_ok && g_ok && oy < H && (ox + p) < W;

    if (a.bias) {
#pragma unroll
        for (int c = 0; c < CO_T; ++c) {
            const int co = grp * CO_T + c;
            const float b = (g_ok && co < a.cout) ? a.bias[co] : 0.f;
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[c][p] += b;
        }
    }

    if (a.part) {
        __syncthreads();
        float* red = conv_lds;               // [4][CO_T]
        float* redc = conv_lds + 4 * CO_T;   // [4]
        float* red2 = redc + 4;              // [4][CO_T]
        float cnt = 0.f;
#pragma unroll
        for (int p = 0; p < PX; ++p) cnt += valid[p] ? 1.f : 0.f;
        cnt = san_wave_sum(cnt);
        if (lane == 0) redc[wave] = cnt;
#pragma unroll
        for (int c = 0; c < CO_T; ++c) {
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < PX; ++p) s += valid[p] ? acc[c][p] : 0.f;
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
            for (int p = 0; p < PX; ++p) {
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
    const bool vec = (PX > 1) && ((W % PX) == 0) && (ox + PX - 1 < W);
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
        float o[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) o[p] = fmaf(acc[c][p], os, ob);
        if (vec && PX == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[PX - 1]);
        } else if (vec && PX == 2) {
            *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[PX - 1]);
        } else {
#pragma unroll
            for (int p = 0; p < PX; ++p)
                if (ox + p < W) dst[p] = o[p];
        }
    }
}

template <int CO_T, int KS>
void launch_px(const float* x, const float* wp, float* y, const ConvArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    switch (a.g.px) {
        case 4: hipLaunchKernelGGL((conv_fwd_kernel<CO_T, KS, 4>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
        case 2: hipLaunchKernelGGL((conv_fwd_kernel<CO_T, KS, 2>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
        default: hipLaunchKernelGGL((conv_fwd_kernel<CO_T, KS, 1>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
    }
}

template <int KS>
int launch_conv(const float* x, const float* wp, float* y, const ConvArgs& a, hipStream_t s) {
    const ConvGeom& g = a.g;
    dim3 grid(g.tiles_x * g.tiles_y, san_cdiv(g.groups, g.WC), a.N);
    size_t lds = (size_t)stage_ck(g.px) * g.rows_t * g.pitch * sizeof(float);
    size_t need = (size_t)(4 * g.co_t * 2 + 4) * sizeof(float);
    if (lds < need) lds = need;
    switch (g.co_t) {
        case 2: launch_px<2, KS>(x, wp, y, a, grid, lds, s); break;
        case 4: launch_px<4, KS>(x, wp, y, a, grid, lds, s); break;
        case 8: launch_px<8, KS>(x, wp, y, a, grid, lds, s); break;
        case 16: launch_px<16, KS>(x, wp, y, a, grid, lds, s); break;
        case 18: launch_px<18, KS>(x, wp, y, a, grid, lds, s); break;
        default: san_set_error("bad co_t %d", g.co_t); return SAN_E_UNSUPPORTED;
    }
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

}  // namespace

extern "C" {

int san_conv_stat_tiles(int n, int h, int w, int cin, int cout, int ks) {
    ConvGeom g = conv_geom(n, h, w, cin, cout, ks);
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
    a.g = conv_geom(n, h, w, cin, cout, ks);
    if (ks == 3) return launch_conv<3>(x, w_packed, y, a, (hipStream_t)stream);
    return launch_conv<1>(x, w_packed, y, a, (hipStream_t)stream);
}

}  // extern "C"
