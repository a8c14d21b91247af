// fp32 convolution (3x3 / 1x1) on the CDNA4 matrix cores, built around
// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 outer products per
// instruction.  Lane l = 4b + j supplies one A value (row j of block b) and one
// B value (column j of block b) and receives D[i][lane] = A[4b+i] * B[4b+j]
// (i = result register), verified on gfx950.  Mapping used here:
//
//      B operand : one activation per lane  -> 64 output pixels per instruction
//      A operand : 4 weights (one output-channel quad) replicated in every block
//      D         : lane l accumulates 4 output channels of ITS pixel
//
// i.e. the matrix core is used as a register-blocked outer-product engine with
// 4-channel granularity: the U-Net's awkward channel counts cost at most
// 18 -> 20 padding (10 %), nothing for 36/72/144/288/32/64, where 16- or 32-wide
// MFMA tiles would waste 44 % / 25 % on the layers that hold half of the MACs,
// and -- unlike the plain-VALU direct convolution this replaces -- the inner loop
// has no VALU work at all: per input channel and tap a wave issues CQ*G MFMAs
// (8 cycles each, exact fp32 FMA chains, bit-identical to fmaf) fed by G + 2
// LDS reads.  fp32-input MFMA peaks at the fp32 vector rate (157 TF), so this is
// about issue efficiency, not a higher roof.
//
// Workgroup = 4 waves.  WC waves share one staged input tile and take different
// output-channel groups of CW = 4*CQ channels; WY waves stack in y.  A wave owns
// G pixel groups of GW x GH = 64 lanes stacked in y.  Input tile: global ->
// registers (prefetched one channel chunk ahead) -> lazy normalisation -> LDS.
// Weights: packed [group][cin][tap][j][cq] so that a lane's CQ weights are
// contiguous; staged per chunk to LDS by the same pipeline.
//
// Epilogue: bias, per-tile (count, mean, M2) statistics for Instance/BatchNorm,
// optional per-(n, c) output affine, coalesced stores (a wave writes GW
// consecutive pixels of one channel).
#include <stdlib.h>

#include "san_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kCK = 4;        // input channels per staged chunk
constexpr int kSlots = 5;     // most staged input elements per thread per channel (<= 1280 per tile); template SL = 2, 3 or 5
constexpr int kMaxWSlots = 5; // staged weight float4s per thread per chunk (template WS in {1,2,3,5})

typedef float f4 __attribute__((ext_vector_type(4)));

struct MGeom {
    int cw;       // output channels per wave (4*CQ)
    int cq, cqp;  // quads per wave, padded quad stride in the packed layout (multiple of 4)
    int G;        // pixel groups per wave
    int GW, GH;   // lanes per group along x / y (GW*GH <= 64)
    int groups;   // ceil(cout / cw)
    int WY, WC;
    int TW, TH;
    int tiles_x, tiles_y;
    int pitch, rows_t;
};

struct MArgs {
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
    int shuffle;   // 1: ConvTranspose2d 2x2 s2 evaluated as a 1x1 conv to 4*cout channels + 2x2 pixel shuffle
    MGeom g;
};

// Output channels per wave: maximise (useful / padded channels) x (a preference for wide
// waves: CQ quads share every activation read, narrow waves re-read the tile more often).
int pick_cw(int cout) {
    const int cand[4] = {20, 16, 8, 4};
    const double pref[4] = {1.0, 1.0, 0.8, 0.6};
    int best = 4;
    double best_e = -1.0;
    for (int i = 0; i < 4; ++i) {
        const int cw = cand[i];
        const double e = pref[i] * (double)cout / (double)(san_cdiv(cout, cw) * cw);
        if (e > best_e + 1e-9) {
            best_e = e;
            best = cw;
        }
    }
    return best;
}

MGeom mfma_geom(int N, int H, int W, int cout, int ks) {
    MGeom g{};
    g.cw = pick_cw(cout);
    g.cq = g.cw / 4;
    g.cqp = (g.cq + 3) & ~3;
    g.groups = san_cdiv(cout, g.cw);
    // WC = 1: every workgroup stages only its own output-channel group's weights.  Sharing one
    // input tile between 2-4 groups (WC > 1) measured 0-50 % SLOWER on MI355X (more staging
    // registers -> 2 waves/SIMD, more LDS write traffic per barrier), so it is not used.
    g.WC = 1;
    g.WY = 4 / g.WC;
    const int pad = ks / 2;
    const int gwc[7] = {64, 32, 16, 8, 20, 40, 10};
    const int Gc[3] = {4, 2, 1};
    const double Gw[3] = {1.0, 0.9, 0.75};      // fewer groups per wave = fewer MFMAs per weight read
    double best = -1.0;
    for (int ig = 0; ig < 3; ++ig) {
        const int G = Gc[ig];
        for (int iw = 0; iw < 7; ++iw) {
            const int GW = gwc[iw], GH = 64 / GW;
            const int TW = GW, TH = GH * G * g.WY;
            if ((TW + 2 * pad) * (TH + 2 * pad) > kSlots * kThreads) continue;
            const int tx = san_cdiv(W, TW), ty = san_cdiv(H, TH);
            const double ex = (double)W / (tx * TW), ey = (double)H / (ty * TH);
            const double el = (double)(GW * GH) / 64.0;
            const double waves = (double)tx * ty * N * 4.0 * san_cdiv(g.groups, g.WC);
            const double fill = waves >= 1024.0 ? 1.0 : (0.5 + 0.5 * waves / 1024.0);   // >= 1 wave per SIMD
            const double halo = (double)(TW * TH) / ((TW + 2 * pad) * (TH + 2 * pad));
            const double e = ex * ey * el * fill * Gw[ig] * (0.8 + 0.2 * halo);
            if (e > best + 1e-9) {
                best = e;
                g.G = G;
                g.GW = GW;
                g.GH = GH;
            }
        }
    }
    g.TW = g.GW;
    g.TH = g.GH * g.G * g.WY;
    g.tiles_x = san_cdiv(W, g.TW);
    g.tiles_y = san_cdiv(H, g.TH);
    // LDS row pitch: the 32 lanes of a ds_read_b32 group sit in 32/GW lane rows that are G tile rows
    // apart; pick the smallest pitch whose lane rows fall on disjoint banks
    g.pitch = g.TW + 2 * pad;
    if (g.GW < 32) {
        int best_p = g.pitch, best_c = 1 << 30;
        for (int p = g.TW + 2 * pad; p < g.TW + 2 * pad + 32; ++p) {
            int cnt[32] = {0}, worst = 0;
            for (int l = 0; l < 32 && l < g.GW * g.GH; ++l) {
                const int b = ((l / g.GW) * g.G * p + l % g.GW) & 31;
                if (++cnt[b] > worst) worst = cnt[b];
            }
            if (worst < best_c) {
                best_c = worst;
                best_p = p;
            }
        }
        g.pitch = best_p;
    }
    g.rows_t = g.TH + 2 * pad;
    return g;
}

extern __shared__ __attribute__((aligned(16))) float mf_lds[];

template <int CQ, int KS, int G, int WS, int SL>
__global__ void __launch_bounds__(kThreads)
conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y, const MArgs a) {
    constexpr int PAD = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int CQP = (CQ + 3) & ~3;
    constexpr int CW = 4 * CQ;
    constexpr int WROW = 4 * CQP;                 // packed floats per (channel, tap) per group
    const MGeom& M = a.g;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wc = wave % M.WC;
    const int wy = wave / M.WC;
    const bool lane_ok = lane < M.GW * M.GH;
    const int lrow = lane_ok ? lane / M.GW : 0;
    const int lcol = lane_ok ? lane - lrow * M.GW : 0;
    // XCD-aware order: the 1-D grid's ids are dealt round-robin to the 8 XCDs (id % 8); each XCD gets a
    // contiguous run of the logical order (n, tile row-major, channel group fastest), so the channel
    // groups that read the same input tile, and neighbouring tiles that share halo rows / columns,
    // are served by ONE L2 instead of refilling up to eight of them from the fabric.
    const int ngrp = (M.groups + M.WC - 1) / M.WC;
    const int ntile = M.tiles_x * M.tiles_y;
    int lin;
    {
        const int total = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3;
        lin = xcd * (total >> 3) + min(xcd, total & 7) + slot;
    }
    const int bgrp = lin % ngrp;
    const int btile = (lin / ngrp) % ntile;
    const int ty = btile / M.tiles_x;
    const int tx = btile - ty * M.tiles_x;
    const int x0 = tx * M.TW, y0 = ty * M.TH;
    const int n = lin / (ngrp * ntile);
    const int grp0 = bgrp * M.WC;                 // first output-channel group of this workgroup
    const int grp = grp0 + wc;                    // wave-uniform
    const bool g_ok = grp < M.groups;
    const int H = a.H, W = a.W;
    const int pitch = M.pitch, rows_t = M.rows_t;
    const int cols_t = M.TW + 2 * PAD;
    const int tile_elems = rows_t * cols_t;
    const int tile_stride = rows_t * pitch;
    const size_t HW = (size_t)H * W;

    float* lds_in = mf_lds;                                   // [kCK][rows_t][pitch]
    const int w_chunk = kCK * TAPS * WROW;                    // weight floats per group per chunk
    float* lds_w = mf_lds + ((kCK * tile_stride + kThreads + 3) & ~3);   // [WC][kCK][TAPS][4][CQP], 16-byte aligned
    // 256 spare floats between the regions: slots that fall outside the tile store to their own
    // dummy word (same-address LDS stores from many lanes serialise, one lane per cycle)
    const int trash = kCK * tile_stride + tid;

    // ---- staging maps (computed once): input halo tile and the weight chunk
    int goff[SL], loff[SL];
    bool inb[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int e = tid + s * kThreads;
        const int r = e / cols_t, col = e - r * cols_t;
        const int gy = y0 - PAD + r, gx = x0 - PAD + col;
        const bool in_tile = e < tile_elems;
        inb[s] = in_tile && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[s] = inb[s] ? gy * W + gx : 0;
        loff[s] = in_tile ? r * pitch + col : -1 - 0 * r;
    }
    // weights: the chunk of group g is w_chunk contiguous floats at wp[(g*cin + c0) * TAPS * WROW];
    // float4 slot q of this thread covers element (tid + q*256)*4 of the WC concatenated group chunks
    const int w_total4 = M.WC * w_chunk / 4;

    float stage[kCK][SL];
    f4 wstage[WS];
    float psc[kCK], psh[kCK];          // the chunk's lazy affine, fetched with the tile (not at first use)
    const int aff = n * a.x_ctot + a.x_coff;
    auto prefetch = [&](int c0) {
        const int cin_last = a.cin - 1;
#pragma unroll
        for (int ci = 0; ci < kCK; ++ci) {
            psc[ci] = 1.f;
            psh[ci] = 0.f;
            if (a.in_scale) {
                psc[ci] = a.in_scale[aff + min(c0 + ci, cin_last)];
                psh[ci] = a.in_shift[aff + min(c0 + ci, cin_last)];
            }
        }
#pragma unroll
        for (int ci = 0; ci < kCK; ++ci) {
            const float* src = x + (size_t)(n * a.x_ctot + a.x_coff + min(c0 + ci, cin_last)) * HW;
#pragma unroll
            for (int s = 0; s < SL; ++s) stage[ci][s] = src[goff[s]];
        }
#pragma unroll
        for (int q = 0; q < WS; ++q) {
            const int e4 = tid + q * kThreads;
            const int gsel = min((e4 * 4) / w_chunk, M.WC - 1);
            const int within = e4 * 4 - gsel * w_chunk;       // float offset inside the group's chunk
            const int gg = min(grp0 + gsel, M.groups - 1);
            // the last chunk may run past cin: the packed buffer is padded by one chunk (zeros)
            const float* src = wp + ((size_t)gg * a.cin + c0) * (TAPS * WROW) + within;
            wstage[q] = (e4 < w_total4) ? *reinterpret_cast<const f4*>(src) : f4{0.f, 0.f, 0.f, 0.f};
        }
    };

    f4 acc[G][CQ];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int c = 0; c < CQ; ++c) acc[g][c] = f4{0.f, 0.f, 0.f, 0.f};

    const int row0 = wy * (M.GH * G) + lrow * G;   // this lane's G pixels are rows row0 .. row0 + G - 1 of the tile
    const int in_base = row0 * pitch + lcol;
    const float* wl = lds_w + wc * w_chunk + (lane & 3) * CQP;

    prefetch(0);
    for (int c0 = 0; c0 < a.cin; c0 += kCK) {
        const int ckk = min(kCK, a.cin - c0);
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < kCK; ++ci) {
            const float sc = psc[ci], sh = psh[ci];
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const float v = inb[s] ? san_act(stage[ci][s], sc, sh, a.in_slope) : 0.f;
                lds_in[loff[s] >= 0 ? ci * tile_stride + loff[s] : trash] = v;
            }
        }
#pragma unroll
        for (int q = 0; q < WS; ++q) {
            const int e4 = tid + q * kThreads;
            if (e4 < w_total4) *reinterpret_cast<f4*>(lds_w + e4 * 4) = wstage[q];
        }
        __syncthreads();
        if (c0 + kCK < a.cin) prefetch(c0 + kCK);
        if (g_ok) {
            // A lane's G pixels are vertically adjacent (row0 + g), so for one input channel the G
            // 3x3 windows overlap: (G + KS - 1) x KS activations per lane serve all G*TAPS products
            // (18 LDS reads instead of 36 at G = 4), all at immediate offsets.  Per tap the CQ weight
            // quads arrive as CQP/4 16-byte reads.  Both are fetched one step ahead of the CQ*G MFMAs
            // that consume them; the MFMAs of a tap form one solid block (switching between matrix
            // and vector/LDS issue costs cycles on gfx950, scratch/probe/mfma_il.hip).
            constexpr int WR = G + KS - 1;
            float win[2][WR][KS];
            float wq[2][CQP];
            auto load_win = [&](int ci, float (&w)[WR][KS]) {
                const float* tin = lds_in + ci * tile_stride + in_base;
#pragma unroll
                for (int r = 0; r < WR; ++r)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) w[r][kx] = tin[r * pitch + kx];
            };
            auto load_w = [&](int ci, int tap, float (&q)[CQP]) {
                const float* tw = wl + (ci * TAPS + tap) * WROW;
#pragma unroll
                for (int c4 = 0; c4 < CQP; c4 += 4) {
                    const f4 t4 = *reinterpret_cast<const f4*>(tw + c4);
                    q[c4] = t4[0];
                    q[c4 + 1] = t4[1];
                    q[c4 + 2] = t4[2];
                    q[c4 + 3] = t4[3];
                }
            };
            load_win(0, win[0]);
            load_w(0, 0, wq[0]);
#pragma unroll
            for (int ci = 0; ci < kCK; ++ci) {
                if (ci < ckk) {
#pragma unroll
                    for (int tap = 0; tap < TAPS; ++tap) {
                        constexpr int dummy = 0;
                        (void)dummy;
                        const int step = ci * TAPS + tap;
                        if (tap + 1 < TAPS) {
                            load_w(ci, tap + 1, wq[(step + 1) & 1]);
                        } else {
                            load_w(min(ci + 1, kCK - 1), 0, wq[(step + 1) & 1]);   // clamped: unused after the last channel
                        }
                        if (tap == 0) load_win(min(ci + 1, kCK - 1), win[(ci + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int c = 0; c < CQ; ++c)
                                acc[g][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[step & 1][c], win[ci & 1][g + tap / KS][tap % KS],
                                                                               acc[g][c], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }

    // ------------------------------------------------------------ epilogue
    const int ox = x0 + lcol;
    bool valid[G];
    int oyg[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        oyg[g] = y0 + row0 + g;
        valid[g] = lane_ok && g_ok && oyg[g] < H && ox < W;
    }
    if (a.bias) {
#pragma unroll
        for (int c = 0; c < CQ; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = grp * CW + 4 * c + i;
                const float bv = (g_ok && co < a.cout) ? a.bias[co] : 0.f;
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g][c][i] += bv;
            }
    }

    if (a.part) {
        // Per-WAVE statistics tile (no LDS, no barrier): single pass over the wave's G*64 outputs
        // with every value shifted by a pilot sample c (one valid output of the same channel), so
        //   mean = c + S1/n,  M2 = S2 - S1^2/n   with S1 = sum(x-c), S2 = sum((x-c)^2)
        // has no catastrophic cancellation; wave totals by DPP butterflies (wave_total), partials
        // merged later by san_norm_finalize (Chan).  Tile index = tile * WY + wy.
        const unsigned long long vm = __ballot(valid[0]);
        const int src_lane = vm ? (int)__ffsll((long long)vm) - 1 : 0;
        float cnt = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) cnt += valid[g] ? 1.f : 0.f;
        cnt = san_wave_total(cnt);
        const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
        float my_mean = 0.f, my_m2 = 0.f;
#pragma unroll
        for (int c = 0; c < CQ; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pilot = __builtin_bit_cast(
                    float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[0][c][i]), src_lane));
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float e = valid[g] ? acc[g][c][i] - pilot : 0.f;
                    s1 += e;
                    s2 = fmaf(e, e, s2);
                }
                const float S1 = san_wave_total(s1), S2 = san_wave_total(s2);
                if (lane == 4 * c + i) {
                    my_mean = pilot + S1 * inv;
                    my_m2 = fmaxf(S2 - S1 * S1 * inv, 0.f);
                }
            }
        if (lane < CW && g_ok) {
            const int co = grp * CW + lane;
            if (co < a.cout) {
                const int tiles = M.tiles_x * M.tiles_y * M.WY;
                float* o = a.part + ((size_t)(n * a.cout + co) * tiles + btile * M.WY + wy) * 3;
                o[0] = cnt;
                o[1] = cnt > 0.f ? my_mean : 0.f;
                o[2] = cnt > 0.f ? my_m2 : 0.f;
            }
        }
    }

    if (!(lane_ok && g_ok) || ox >= W) return;
    if (a.shuffle) {
        // channel c' = 4*co + t holds tap t = 2*dy + dx of output channel co: a lane's f4 accumulator
        // is exactly the 2x2 output block of its input pixel -> two 8-byte stores per channel,
        // contiguous across lanes
        const int OW = 2 * W;
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
            const int cp = grp * CW + 4 * c;
            if (cp < a.cout) {
                float* dst = y + (size_t)(n * a.y_ctot + a.y_coff + (cp >> 2)) * (4 * HW) + 2 * ox;
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (oyg[g] < H) {
                        float* d = dst + (size_t)(2 * oyg[g]) * OW;
                        *reinterpret_cast<float2*>(d) = make_float2(acc[g][c][0], acc[g][c][1]);
                        *reinterpret_cast<float2*>(d + OW) = make_float2(acc[g][c][2], acc[g][c][3]);
                    }
            }
        }
        return;
    }
    // Output affine of every channel first, THEN the stores: on gfx9 loads and stores share vmcnt, so a
    // load between two stores makes each store wait for the previous one to complete.
    float osv[CQ][4], obv[CQ][4];
#pragma unroll
    for (int c = 0; c < CQ; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            osv[c][i] = 1.f;
            obv[c][i] = 0.f;
            if (a.out_scale) {
                const int co = min(grp * CW + 4 * c + i, a.cout - 1);
                osv[c][i] = a.out_scale[n * a.cout + co];
                obv[c][i] = a.out_shift[n * a.cout + co];
            }
        }
    int ooff[G];
#pragma unroll
    for (int g = 0; g < G; ++g) ooff[g] = oyg[g] * W + ox;
#pragma unroll
    for (int c = 0; c < CQ; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = grp * CW + 4 * c + i;
            if (co < a.cout) {
                float* dst = y + (size_t)(n * a.y_ctot + a.y_coff + co) * HW;
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (oyg[g] < H) dst[ooff[g]] = fmaf(acc[g][c][i], osv[c][i], obv[c][i]);
            }
        }
}

// w [cout, cin, ks, ks] -> packed [groups][cin (+ kCK zero rows)][taps][4][CQP]; cout = g*CW + 4*cq + j
__global__ void pack_mfma_kernel(const float* __restrict__ w, float* __restrict__ packed, int cout, int cin, int taps,
                                 int cw, int cqp, int groups, int transposed) {
    const int wrow = 4 * cqp;
    const size_t total = (size_t)groups * cin * taps * wrow;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total + (size_t)kCK * taps * wrow;
         i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < total) {
            const int cq = (int)(i % cqp);
            const int j = (int)((i / cqp) % 4);
            const int t = (int)((i / wrow) % taps);
            const int ci = (int)((i / ((size_t)wrow * taps)) % cin);
            const int g = (int)(i / ((size_t)wrow * taps * cin));
            const int co = g * cw + 4 * cq + j;
            // transposed: cout = 4*Cout' virtual channels c' = 4*co + tap of a ConvTranspose2d weight [cin, Cout', 2, 2]
            // transposed == 2: data-gradient weights of a Conv2d [cin_fwd = cout here ... ]: the forward
            // weight is [ci][co][taps] in THIS kernel's naming (its cout = forward cin), taps flipped
            if (4 * cq < cw && co < cout) {
                if (transposed == 1) v = w[(size_t)ci * cout + co];
                else if (transposed == 2) v = w[((size_t)ci * cout + co) * taps + (taps - 1 - t)];
                else v = w[((size_t)co * cin + ci) * taps + t];
            }
        }
        packed[i] = v;
    }
}

// One pack job per blockIdx.y; the job table lives in device memory: 8 x int64 per job =
// {w, packed, cout, cin, taps, cw, cqp, groups | transposed << 32} (built by san_conv_pack_job).
__global__ void pack_mfma_batch_kernel(const long long* __restrict__ jobs) {
    const long long* j = jobs + 8 * (size_t)blockIdx.y;
    const float* w = reinterpret_cast<const float*>(j[0]);
    float* packed = reinterpret_cast<float*>(j[1]);
    const int cout = (int)j[2], cin = (int)j[3], taps = (int)j[4], cw = (int)j[5], cqp = (int)j[6];
    const int groups = (int)(j[7] & 0xffffffffll), transposed = (int)(j[7] >> 32);
    const int wrow = 4 * cqp;
    const size_t total = (size_t)groups * cin * taps * wrow;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total + (size_t)kCK * taps * wrow;
         i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < total) {
            const int cq = (int)(i % cqp);
            const int jj = (int)((i / cqp) % 4);
            const int t = (int)((i / wrow) % taps);
            const int ci = (int)((i / ((size_t)wrow * taps)) % cin);
            const int g = (int)(i / ((size_t)wrow * taps * cin));
            const int co = g * cw + 4 * cq + jj;
            if (4 * cq < cw && co < cout) {
                if (transposed == 1) v = w[(size_t)ci * cout + co];
                else if (transposed == 2) v = w[((size_t)ci * cout + co) * taps + (taps - 1 - t)];
                else v = w[((size_t)co * cin + ci) * taps + t];
            }
        }
        packed[i] = v;
    }
}

// SL: staged elements per thread per channel.  Small images (40^2, 20^2 tiles) need 2; issuing the other
// three as masked loads + stores made those layers staging-bound.
template <int CQ, int KS, int G, int SL>
void launch_ws(const float* x, const float* wp, float* y, const MArgs& a, dim3 grid, size_t lds, int ws, hipStream_t s) {
    if (KS == 1 || ws <= 1) {
        hipLaunchKernelGGL((conv_mfma_kernel<CQ, KS, G, 1, SL>), grid, dim3(kThreads), lds, s, x, wp, y, a);
    } else if (KS == 3 && ws == 2) {
        hipLaunchKernelGGL((conv_mfma_kernel<CQ, 3, G, 2, SL>), grid, dim3(kThreads), lds, s, x, wp, y, a);
    } else if (KS == 3 && ws == 3) {
        hipLaunchKernelGGL((conv_mfma_kernel<CQ, 3, G, 3, SL>), grid, dim3(kThreads), lds, s, x, wp, y, a);
    } else {
        hipLaunchKernelGGL((conv_mfma_kernel<CQ, 3, G, 5, SL>), grid, dim3(kThreads), lds, s, x, wp, y, a);
    }
}

template <int CQ, int KS, int G>
void launch_sl(const float* x, const float* wp, float* y, const MArgs& a, dim3 grid, size_t lds, int ws, hipStream_t s) {
    const int pad = KS / 2;
    const int tile_elems = (a.g.TW + 2 * pad) * (a.g.TH + 2 * pad);
    if (tile_elems <= 2 * kThreads)
        launch_ws<CQ, KS, G, 2>(x, wp, y, a, grid, lds, ws, s);
    else if (tile_elems <= 3 * kThreads)
        launch_ws<CQ, KS, G, 3>(x, wp, y, a, grid, lds, ws, s);
    else
        launch_ws<CQ, KS, G, kSlots>(x, wp, y, a, grid, lds, ws, s);
}

template <int CQ, int KS>
void launch_g(const float* x, const float* wp, float* y, const MArgs& a, dim3 grid, size_t lds, int ws, hipStream_t s) {
    switch (a.g.G) {
        case 4: launch_sl<CQ, KS, 4>(x, wp, y, a, grid, lds, ws, s); break;
        case 2: launch_sl<CQ, KS, 2>(x, wp, y, a, grid, lds, ws, s); break;
        default: launch_sl<CQ, KS, 1>(x, wp, y, a, grid, lds, ws, s); break;
    }
}

template <int KS>
int launch_mfma(const float* x, const float* wp, float* y, const MArgs& a, hipStream_t s) {
    const MGeom& g = a.g;
    dim3 grid(g.tiles_x * g.tiles_y * san_cdiv(g.groups, g.WC) * a.N, 1, 1);
    const int taps = KS * KS;
    size_t in_fl = ((size_t)kCK * g.rows_t * g.pitch + kThreads + 3) & ~(size_t)3;
    size_t w_fl = (size_t)g.WC * kCK * taps * 4 * g.cqp;
    int ws = (int)((w_fl / 4 + kThreads - 1) / kThreads);
    if (ws > kMaxWSlots) {
        san_set_error("weight chunk too large for the staging slots");
        return SAN_E_UNSUPPORTED;
    }
    ws = ws <= 1 ? 1 : (ws == 2 ? 2 : (ws == 3 ? 3 : 5));
    size_t lds = (in_fl + w_fl) * sizeof(float);
    size_t need = (size_t)(4 * g.cw * 2 + 4) * sizeof(float);
    if (lds < need) lds = need;
    switch (g.cq) {
        case 1: launch_g<1, KS>(x, wp, y, a, grid, lds, ws, s); break;
        case 2: launch_g<2, KS>(x, wp, y, a, grid, lds, ws, s); break;
        case 4: launch_g<4, KS>(x, wp, y, a, grid, lds, ws, s); break;
        case 5: launch_g<5, KS>(x, wp, y, a, grid, lds, ws, s); break;
        default: san_set_error("bad cq %d", g.cq); return SAN_E_UNSUPPORTED;
    }
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Direct fp32 convolution for layers with a handful of channels on ONE side (round 5): the cascade's first convolution
// (4 -> 18, 3x3), its output convolution (18 -> 2, 1x1) and their data gradients (2 -> 18 1x1, 18 -> 4 3x3).  These move 13 + 59 MB
// per launch at N = 8 (12 us of HBM time) and need 648 FMAs per pixel (7 us of the vector pipe), but took 45 / 18 / 47 / 44 us on
// the outer-product kernel above (its per-(channel, tap) MFMA blocks are tiny there, the launch is latency-bound).  Here:
// tile 64 x 16 pixels, thread = 4 consecutive pixels of one row x ALL CW output channels of its group (4 CW accumulators);
// the halo tile of CK input channels (lazy affine + LeakyReLU applied once, while staging) and the group's whole weight set
// sit in LDS; per (channel, row) a thread reads its 6 (3x3) or 4 (1x1) inputs with one or two wide LDS reads, the CW weights of
// a tap are broadcast reads (same address in every lane).  Pure v_fma_f32: exact fp32 chains like the kernel above.
// Same packed weights ([group][cin][tap][4][CQP], forward or data-gradient packing), same epilogue contract (bias, per-wave
// (count, mean, M2) statistics tiles, per-(n, c) output affine, channel views) -- san_conv2d_fwd routes here by shape alone.
constexpr int kDTW = 64, kDTH = 16, kDPX = 4;

__host__ __device__ inline bool direct_ok(int cin, int cout, int ks) {
    if (ks != 1 && ks != 3) return false;
    return cin <= 4 || (cout <= 4 && cin <= 48);
}

// CK = 4: cin <= 4, ONE staged chunk; the output channels are produced four at a time (CW / 4 passes over the staged tile), each
// quad's bias / statistics / stores issued as soon as it is complete, so that the stores of one quad drain while the next is
// computed (with all CW x 4 accumulators finished at once, every resident workgroup stored at the same moment and the launch
// took compute + store time).  Measured at N = 8, 320^2 (rocprofv3, us): 4 -> 18 3x3 47.5 -> 35.2 (ablation: loads + stores alone 15,
// the FMAs alone 23 -- a plain v_fma_f32 issues at half the packed / MFMA fp32 rate -- statistics 6), 2 -> 18 1x1 28.0 -> 14.5,
// 18 -> 4 3x3 37.2 -> ~22, 18 -> 2 1x1 18.2 -> 17.2; 2 -> 8 3x3 on 15 planes of 640 x 368: 60 -> 41.  CK = 6: cout <= 4 (CW = 4), the input channels arrive in chunks of six
// whose global loads are issued a whole compute phase ahead.
template <int CK, int KS, int CW>
__global__ void __launch_bounds__(kThreads)
conv_direct_kernel(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y, const MArgs a) {
    static_assert(CK == 4 || CW == 4, "many input channels only with one output quad");
    constexpr int PAD = KS / 2, TAPS = KS * KS;
    constexpr int TWH = kDTW + 2 * PAD, THH = kDTH + 2 * PAD;
    constexpr int PITCH = (TWH + 3) & ~3;                  // 68 (3x3) / 64 (1x1): rows start 16-byte aligned
    constexpr int TILE = THH * PITCH;
    constexpr int SL = (THH * TWH + kThreads - 1) / kThreads;
    constexpr int NX = kDPX + 2 * PAD;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int tx = tid & 15, ty = tid >> 4;                // thread = pixels (4 tx .. 4 tx + 3, ty) of the tile
    const int H = a.H, W = a.W, cin = a.cin;
    const size_t HW = (size_t)H * W;
    const int tiles_x = (W + kDTW - 1) / kDTW, tiles_y = (H + kDTH - 1) / kDTH, ntile = tiles_x * tiles_y;
    const int groups = a.g.groups, cqp = a.g.cqp;
    int lin;
    {
        const int total = gridDim.x, id = blockIdx.x;      // XCD-contiguous logical order, as in the kernel above
        const int xcd = id & 7, slot = id >> 3;
        lin = xcd * (total >> 3) + min(xcd, total & 7) + slot;
    }
    const int grp = lin % groups;
    const int tile = (lin / groups) % ntile;
    const int n = lin / (groups * ntile);
    const int tyy = tile / tiles_x, txx = tile - tyy * tiles_x;
    const int x0 = txx * kDTW, y0 = tyy * kDTH;

    float* lds_x = mf_lds;                                  // [CK][THH][PITCH]
    float* lds_w = mf_lds + CK * TILE;                      // [cin][TAPS][CW], channel co = 4 cq + j of the group at [co]
    int goff[SL], loff[SL];
    bool inb[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int e = tid + s * kThreads;
        const int r = e / TWH, c = e - r * TWH;
        const int gy = y0 - PAD + r, gx = x0 - PAD + c;
        const bool in_tile = e < THH * TWH;
        inb[s] = in_tile && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[s] = inb[s] ? gy * W + gx : 0;
        loff[s] = in_tile ? r * PITCH + c : -1;
    }
    const float* xn = x + (size_t)(n * a.x_ctot + a.x_coff) * HW;
    float v[CK][SL];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int ch = 0; ch < CK; ++ch) {
            const float* xc = xn + (size_t)min(c0 + ch, cin - 1) * HW;
#pragma unroll
            for (int s = 0; s < SL; ++s) v[ch][s] = inb[s] ? xc[goff[s]] : 0.f;
        }
    };
    auto stage = [&](int c0) {
#pragma unroll
        for (int ch = 0; ch < CK; ++ch) {
            const int ci = min(c0 + ch, cin - 1);           // (channels past cin are staged but never read)
            float sc = 1.f, sh = 0.f;
            if (a.in_scale) {
                sc = a.in_scale[n * a.x_ctot + a.x_coff + ci];
                sh = a.in_shift[n * a.x_ctot + a.x_coff + ci];
            }
#pragma unroll
            for (int s = 0; s < SL; ++s)
                if (loff[s] >= 0) lds_x[ch * TILE + loff[s]] = inb[s] ? san_act(v[ch][s], sc, sh, a.in_slope) : 0.f;
        }
    };
    fetch(0);
    {
        const float* src = wp + (size_t)grp * cin * TAPS * 4 * cqp;
        const int total = cin * TAPS * CW;
        for (int i = tid; i < total; i += kThreads) {
            const int row = i / CW, co = i - row * CW;
            lds_w[i] = src[(size_t)row * 4 * cqp + (co & 3) * cqp + (co >> 2)];
        }
    }
    const int oy = y0 + ty, ox = x0 + kDPX * tx;
    bool valid[kDPX];
#pragma unroll
    for (int p = 0; p < kDPX; ++p) valid[p] = oy < H && ox + p < W;
    const int cbase = grp * CW;
    const bool quad = valid[kDPX - 1] && (W & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    float cnt = 0.f, inv = 0.f;
    int src_lane = 0;
    if (a.part) {
        const unsigned long long vm = __ballot(valid[0]);
        src_lane = vm ? (int)__ffsll((long long)vm) - 1 : 0;
#pragma unroll
        for (int p = 0; p < kDPX; ++p) cnt += valid[p] ? 1.f : 0.f;
        cnt = san_wave_total(cnt);
        inv = cnt > 0.f ? 1.f / cnt : 0.f;
    }
    float acc[kDPX][4];
    auto clear = [&]() {
#pragma unroll
        for (int p = 0; p < kDPX; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[p][i] = 0.f;
    };
    // the staged chunk c0.. x the weight columns 4 cq .. 4 cq + 3
    auto compute = [&](int c0, int cq) {
#pragma unroll
        for (int ch = 0; ch < CK; ++ch) {
            if (c0 + ch < cin) {                            // (workgroup-uniform)
                const float* wrow = lds_w + (size_t)(c0 + ch) * TAPS * CW + 4 * cq;
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
                    const float* xr = lds_x + ch * TILE + (ty + ky) * PITCH + kDPX * tx;
                    float xv[NX];
                    const f4 q = *reinterpret_cast<const f4*>(xr);
                    xv[0] = q[0]; xv[1] = q[1]; xv[2] = q[2]; xv[3] = q[3];
                    if constexpr (KS == 3) {
                        typedef float f2v __attribute__((ext_vector_type(2)));
                        const f2v r2 = *reinterpret_cast<const f2v*>(xr + 4);
                        xv[4] = r2[0]; xv[5] = r2[1];
                    }
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const f4 wv = *reinterpret_cast<const f4*>(wrow + (ky * KS + kx) * CW);      // broadcast read
#pragma unroll
                        for (int p = 0; p < kDPX; ++p)
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[p][i] = fmaf(xv[p + kx], wv[i], acc[p][i]);
                    }
                }
            }
        }
    };
    // bias, per-WAVE (count, mean, M2) statistics (pilot-shifted single pass: the format san_norm_finalize merges), output affine,
    // stores of output channels cbase + 4 cq ..
    auto finish = [&](int cq) {
        float my_mean = 0.f, my_m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = cbase + 4 * cq + i;
            if (a.bias) {
                const float bv = co < a.cout ? a.bias[co] : 0.f;
#pragma unroll
                for (int p = 0; p < kDPX; ++p) acc[p][i] += bv;
            }
            if (a.part) {
                const float pilot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[0][i]), src_lane));
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int p = 0; p < kDPX; ++p) {
                    const float e = valid[p] ? acc[p][i] - pilot : 0.f;
                    s1 += e;
                    s2 = fmaf(e, e, s2);
                }
                const float S1 = san_wave_total(s1), S2 = san_wave_total(s2);
                if (lane == i) {
                    my_mean = pilot + S1 * inv;
                    my_m2 = fmaxf(S2 - S1 * S1 * inv, 0.f);
                }
            }
        }
        if (a.part && lane < 4 && cbase + 4 * cq + lane < a.cout) {
            const int tiles = ntile * 4;
            float* o = a.part + ((size_t)(n * a.cout + cbase + 4 * cq + lane) * tiles + tile * 4 + wave) * 3;
            o[0] = cnt;
            o[1] = cnt > 0.f ? my_mean : 0.f;
            o[2] = cnt > 0.f ? my_m2 : 0.f;
        }
        if (valid[0]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = cbase + 4 * cq + i;
                if (co < a.cout) {
                    float os = 1.f, ob = 0.f;
                    if (a.out_scale) {
                        os = a.out_scale[n * a.cout + co];
                        ob = a.out_shift[n * a.cout + co];
                    }
                    float* dst = y + (size_t)(n * a.y_ctot + a.y_coff + co) * HW + (size_t)oy * W + ox;
                    if (quad) {
                        *reinterpret_cast<f4*>(dst) = f4{fmaf(acc[0][i], os, ob), fmaf(acc[1][i], os, ob), fmaf(acc[2][i], os, ob), fmaf(acc[3][i], os, ob)};
                    } else {
#pragma unroll
                        for (int p = 0; p < kDPX; ++p)
                            if (valid[p]) dst[p] = fmaf(acc[p][i], os, ob);
                    }
                }
            }
        }
    };

    if constexpr (CK == 4) {
        stage(0);
        __syncthreads();
        for (int cq = 0; cq < CW / 4; ++cq) {
            if (cbase + 4 * cq >= a.cout) break;            // (padding quads of the last group)
            clear();
            compute(0, cq);
            finish(cq);
        }
    } else {
        clear();
        for (int c0 = 0; c0 < cin; c0 += CK) {
            if (c0) __syncthreads();                        // everyone is done with the previous chunk's tile
            stage(c0);
            __syncthreads();
            if (c0 + CK < cin) fetch(c0 + CK);              // in flight while this chunk is computed
            compute(c0, 0);
        }
        finish(0);
    }
}

template <int CK, int KS>
int launch_direct_cw(const float* x, const float* wp, float* y, const MArgs& a, hipStream_t s) {
    constexpr int PAD = KS / 2;
    constexpr int TILE = (kDTH + 2 * PAD) * ((kDTW + 2 * PAD + 3) & ~3);
    const int tiles = san_cdiv(a.W, kDTW) * san_cdiv(a.H, kDTH);
    dim3 grid(tiles * a.g.groups * a.N, 1, 1);
    const size_t lds = ((size_t)CK * TILE + (size_t)a.cin * KS * KS * a.g.cw) * sizeof(float);
    if (lds > 64 * 1024) {
        san_set_error("direct convolution: %d input channels x %d do not fit the weight table", a.cin, a.g.cw);
        return SAN_E_UNSUPPORTED;
    }
    if constexpr (CK == 4) {
        switch (a.g.cw) {
            case 4: hipLaunchKernelGGL((conv_direct_kernel<4, KS, 4>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
            case 8: hipLaunchKernelGGL((conv_direct_kernel<4, KS, 8>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
            case 16: hipLaunchKernelGGL((conv_direct_kernel<4, KS, 16>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
            case 20: hipLaunchKernelGGL((conv_direct_kernel<4, KS, 20>), grid, dim3(kThreads), lds, s, x, wp, y, a); break;
            default: san_set_error("bad cw %d", a.g.cw); return SAN_E_UNSUPPORTED;
        }
    } else {
        if (a.g.cw != 4) {
            san_set_error("direct convolution: cw %d with %d input channels", a.g.cw, a.cin);
            return SAN_E_UNSUPPORTED;
        }
        hipLaunchKernelGGL((conv_direct_kernel<CK, KS, 4>), grid, dim3(kThreads), lds, s, x, wp, y, a);
    }
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

template <int KS>
int launch_direct(const float* x, const float* wp, float* y, const MArgs& a, hipStream_t s) {
    if (a.cin <= 4) return launch_direct_cw<4, KS>(x, wp, y, a, s);
    return launch_direct_cw<6, KS>(x, wp, y, a, s);
}

}  // namespace

extern "C" {

size_t san_conv_packed_floats(int cout, int cin, int ks) {
    if (ks == 2) {  // ConvTranspose2d 2x2 s2 == 1x1 conv to 4*cout virtual channels
        cout *= 4;
        ks = 1;
    }
    const int cw = pick_cw(cout);
    const int cqp = ((cw / 4) + 3) & ~3;
    return (size_t)san_cdiv(cout, cw) * cin * ks * ks * 4 * cqp + (size_t)kCK * ks * ks * 4 * cqp;
}

int san_conv_pack_weights_fwd(const float* w, float* packed, int cout, int cin, int ks, void* stream) {
    SAN_CHECK_ARG(w && packed, "null pointer");
    SAN_CHECK_ARG(cout > 0 && cin > 0 && (ks == 1 || ks == 3), "bad dims");
    const int cw = pick_cw(cout);
    const int cqp = ((cw / 4) + 3) & ~3;
    const int groups = san_cdiv(cout, cw);
    size_t total = (size_t)groups * cin * ks * ks * 4 * cqp;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, cout, cin, ks * ks,
                       cw, cqp, groups, 0);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_conv_pack_weights_dgrad(const float* w, float* packed, int cout, int cin, int ks, void* stream) {
    // w is the FORWARD weight [cout, cin, ks, ks]; the packed result drives san_conv2d_fwd with
    // cin' = cout, cout' = cin (buffer size: san_conv_packed_floats(cin, cout, ks))
    SAN_CHECK_ARG(w && packed, "null pointer");
    SAN_CHECK_ARG(cout > 0 && cin > 0 && (ks == 1 || ks == 3), "bad dims");
    const int cw = pick_cw(cin);
    const int cqp = ((cw / 4) + 3) & ~3;
    const int groups = san_cdiv(cin, cw);
    size_t total = (size_t)groups * cout * ks * ks * 4 * cqp;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, cin, cout, ks * ks,
                       cw, cqp, groups, 2);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_conv_pack_job(long long* job8, const float* w, float* packed, int cout, int cin, int ks, int mode) {
    // mode 0: Conv2d forward [cout,cin,ks,ks]; 1: ConvTranspose2d 2x2 [cin,cout,2,2] (ks = 2);
    // 2: Conv2d data gradient (w is the forward weight; the packed result has cout' = cin, cin' = cout)
    SAN_CHECK_ARG(job8 && w && packed, "null pointer");
    SAN_CHECK_ARG(cout > 0 && cin > 0, "bad dims");
    int pc = cout, pcin = cin, taps = ks * ks;
    if (mode == 1) {
        SAN_CHECK_ARG(ks == 2, "transposed packing is for ConvTranspose2d 2x2 only");
        pc = 4 * cout;
        taps = 1;
    } else {
        SAN_CHECK_ARG(ks == 1 || ks == 3, "ks must be 1 or 3");
        SAN_CHECK_ARG(mode == 0 || mode == 2, "mode must be 0, 1 or 2");
        if (mode == 2) {
            pc = cin;
            pcin = cout;
        }
    }
    const int cw = pick_cw(pc);
    const int cqp = ((cw / 4) + 3) & ~3;
    const int groups = san_cdiv(pc, cw);
    job8[0] = (long long)(uintptr_t)w;
    job8[1] = (long long)(uintptr_t)packed;
    job8[2] = pc;
    job8[3] = pcin;
    job8[4] = taps;
    job8[5] = cw;
    job8[6] = cqp;
    job8[7] = (long long)groups | ((long long)mode << 32);
    return SAN_OK;
}

int san_conv_pack_batch(const long long* jobs_dev, int njobs, void* stream) {
    SAN_CHECK_ARG(jobs_dev && njobs > 0, "empty job table");
    hipLaunchKernelGGL(pack_mfma_batch_kernel, dim3(16, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

// san_conv_direct_enable(0) or SAN_CONV_DIRECT=0 in the environment: every layer on the outer-product kernel (A/B, tests)
static bool g_direct = !(getenv("SAN_CONV_DIRECT") && atoi(getenv("SAN_CONV_DIRECT")) == 0);

int san_conv_direct_enable(int on) {
    const int prev = g_direct ? 1 : 0;
    if (on >= 0) g_direct = on != 0;
    return prev;
}

int san_conv_stat_tiles(int n, int h, int w, int cin, int cout, int ks) {
    if (g_direct && direct_ok(cin, cout, ks)) return san_cdiv(w, kDTW) * san_cdiv(h, kDTH) * 4;      // one tile per wave
    MGeom g = mfma_geom(n, h, w, cout, ks);
    return g.tiles_x * g.tiles_y * g.WY;   // one statistics tile per wave row
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
    MArgs a{};
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
    a.g = mfma_geom(n, h, w, cout, ks);
    if (g_direct && direct_ok(cin, cout, ks)) {
        if (ks == 3) return launch_direct<3>(x, w_packed, y, a, (hipStream_t)stream);
        return launch_direct<1>(x, w_packed, y, a, (hipStream_t)stream);
    }
    if (ks == 3) return launch_mfma<3>(x, w_packed, y, a, (hipStream_t)stream);
    return launch_mfma<1>(x, w_packed, y, a, (hipStream_t)stream);
}

int san_conv_pack_weights(const float* w, float* packed, int cout, int cin, int ks, int transposed, void* stream) {
    SAN_CHECK_ARG(w && packed, "null pointer");
    if (!transposed) return san_conv_pack_weights_fwd(w, packed, cout, cin, ks, stream);
    SAN_CHECK_ARG(cout > 0 && cin > 0 && ks == 2, "transposed packing is for ConvTranspose2d 2x2 only");
    const int cv = 4 * cout;
    const int cw = pick_cw(cv);
    const int cqp = ((cw / 4) + 3) & ~3;
    const int groups = san_cdiv(cv, cw);
    size_t total = (size_t)groups * cin * 4 * cqp;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, cv, cin, 1, cw, cqp,
                       groups, 1);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_tconv_stat_tiles(int n, int h, int w, int cout) {
    MGeom g = mfma_geom(n, h, w, 4 * cout, 1);
    return 4 * g.tiles_x * g.tiles_y * g.WY;   // 4 virtual channels (taps) per output channel
}

int san_tconv2x2_fwd(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                     float in_slope, const float* w_packed, float* y, int y_ctot, int y_coff, int cout,
                     float* part_stats, int n, int h, int w, void* stream) {
    SAN_CHECK_ARG(x && w_packed && y, "null pointer");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "bad dims");
    SAN_CHECK_ARG(x_coff >= 0 && x_coff + cin <= x_ctot && y_coff >= 0 && y_coff + cout <= y_ctot, "bad channel view");
    SAN_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "in_scale/in_shift must come together");
    MArgs a{};
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_slope = in_slope;
    a.part = part_stats;
    a.x_ctot = x_ctot;
    a.x_coff = x_coff;
    a.cin = cin;
    a.y_ctot = y_ctot;
    a.y_coff = y_coff;
    a.cout = 4 * cout;
    a.N = n;
    a.H = h;
    a.W = w;
    a.shuffle = 1;
    a.g = mfma_geom(n, h, w, 4 * cout, 1);
    return launch_mfma<1>(x, w_packed, y, a, (hipStream_t)stream);
}

}  // extern "C"
