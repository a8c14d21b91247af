// 3x3 weight gradient on the bf16 matrix cores with fp32-grade accuracy ("bf16x3", as san_conv_bf16.hip):
//   dw[co][ci][ky][kx] (+)= sum_{n,y,x} dy[n][co][y][x] * T(x)[n][ci][y+ky-1][x+kx-1]
// (the backward of F.conv2d at varnet.py:140,143 and unet.py:119-140 w.r.t. the weights).
//
// As a GEMM per tap the contraction index is the PIXEL, and v_mfma_f32_16x16x32_bf16 wants 8 consecutive
// contraction elements per lane -- i.e. 8 neighbouring pixels of one channel row: in NCHW that is already
// contiguous memory.  So there is no transposition and no LDS in the main loop:
//
//   pass 1 (split_planes_kernel, HBM-bound): T(x) and dy are split ONCE into three bf16 planes each
//     (a = a1 + a2 + a3), stored [part][n][c / 16][Hp][Wp / 8][c % 16][8 pixels] with a zero frame (one zero
//     row on top, zero rows below, one zero 8-pixel piece left and right, channels padded to 16) so that
//     every read of pass 2 is unconditional -- and so that the 64 lanes of a wave (16 channels x 4
//     neighbouring pieces) read 1 KB of CONTIGUOUS memory per operand load.
//   pass 2 (wgrad_bf16x3_kernel): a wave owns one block of 16 input channels x NB blocks of 16 output
//     channels x all 9 taps (9 NB accumulator tiles) and four "piece columns": an 8-pixel-wide column of L
//     image rows per group of 16 lanes.  Per row step a lane loads its own 16-byte operand pieces straight from
//     L2 into the MFMA operand registers (3 parts of one new x row, 3 parts x NB of one dy row; both are
//     fetched one step ahead), derives the kx = 0 / 2 operands of the x row with 5 v_alignbit per part from
//     the aligned piece and its two neighbour dwords, keeps the three x rows of the window in registers and
//     issues 54 NB MFMAs (9 taps x NB x the six part products a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1).
//     The four waves of a workgroup work on the same channel tile over different pixels and add their
//     tiles through LDS, so one partial tile per workgroup goes to HBM.
//   pass 3 (wgrad_bf16x3_reduce_kernel): fixed-order sum over the workgroups' partial tiles -> dw.
//
// Deterministic (no atomics).  Accuracy is that of the forward bf16x3 kernel (dropped terms O(2^-24)).
#include "san_common.h"

#include <cstdint>
#include <mutex>
#include <vector>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float fl2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(2))) _Float16 hf2;
union Frag {
    u32x4 u;
    bf8 v;
    h8 h;
};

constexpr int kWaves = 4;
constexpr int kWT = kWaves * 64;

__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
    fl2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
}
// (f0, f1) -> three packed bf16 pairs with p1 + p2 + p3 == f to ~2^-24 relative (low half = f0)
__device__ __forceinline__ void split3_pair(float f0, float f1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = cvt_pk(f0, f1);
    const float r0 = f0 - __builtin_bit_cast(float, p1 << 16), r1 = f1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = cvt_pk(r0, r1);
    p3 = cvt_pk(r0 - __builtin_bit_cast(float, p2 << 16), r1 - __builtin_bit_cast(float, p2 & 0xffff0000u));
}

// Two fp16 parts instead of three bf16 parts (F16 forms of the direct and the 1x1 kernel: three products instead of six, see
// san_conv_bf16.hip "f16x2").  fp16 has the mantissa (2 x 11 bits) but not the range for gradients, so dy is multiplied by a
// power of two S that brings the tensor's largest magnitude (a device scalar the producing kernel maintained with an
// integer atomic max: order-independent, deterministic) to ~2^13, and the accumulators by 1 / S at the end: both exact.
// Elements more than 2^27 below the maximum lose relative precision gradually (absolute floor 2^-25 / S).
__device__ __forceinline__ uint32_t cvt_pk_h(float a, float b) {
    fl2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hf2));
}
template <bool F16>
__device__ __forceinline__ void split_pair(float f0, float f1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    if constexpr (F16) {
        p1 = cvt_pk_h(f0, f1);
        const hf2 h = __builtin_bit_cast(hf2, p1);
        p2 = cvt_pk_h(f0 - (float)h[0], f1 - (float)h[1]);
        p3 = 0u;
    } else {
        split3_pair(f0, f1, p1, p2, p3);
    }
}
template <bool F16>
__device__ __forceinline__ f4 mma(const Frag& x, const Frag& y, f4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(x.h, y.h, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x.v, y.v, c, 0, 0, 0);
}
// S = 2^(13 - floor(log2 max)) and 1 / S from the tensor's amax record (san_common.h; all zero -> 1, 1); whole waves call this
__device__ __forceinline__ void amax_scale(const uint32_t* amax, float& S, float& invS) {
    S = 1.f;
    invS = 1.f;
    if (amax) {
        const uint32_t b = san_amax_read(amax);
        int e = (int)((b >> 23) & 255u);
        if (b != 0u) {
            e = e < 14 ? 14 : (e > 250 ? 250 : e);
            S = __builtin_bit_cast(float, (uint32_t)(267 - e) << 23);
            invS = __builtin_bit_cast(float, (uint32_t)(e - 13) << 23);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// pass 1: fp32 NCHW view (+ lazy affine / LeakyReLU) -> three framed bf16 planes.  One thread per 8-pixel piece.
struct SplitOne {
    const float* src;
    const float* scale;
    const float* shift;
    float slope;
    int ctot, coff, C, CB;     // channel view; CB = blocks of 16 channels written (>= cdiv(C, 16), the rest zero)
    uint16_t* out;
    long long plane;           // elements per part plane
    long long total;           // pieces = N * CB * Hp * npw * 16
    int vec;                   // rows are 16-byte aligned and W % 4 == 0
};
struct SplitArgs {
    SplitOne t[2];             // x, dy: one launch
    unsigned blocks0;          // workgroups of t[0]
    int H, W, Hp, npw;         // framed rows, framed pieces per row (= Wp / 8)
};

__global__ void __launch_bounds__(256) split_planes_kernel(const SplitArgs a) {
    const int which = blockIdx.x >= a.blocks0;
    const SplitOne& o = a.t[which];
    const long long idx = (long long)(blockIdx.x - (which ? a.blocks0 : 0u)) * 256 + threadIdx.x;
    if (idx >= o.total) return;
    // [n][cb][row][piece][c16]: a wave = 16 channels x 4 pieces reads 16 full 128-byte lines, writes 1 KB contiguous
    const int c16 = (int)(idx & 15);
    long long t = idx >> 4;
    const int pc = (int)(t % a.npw);
    t /= a.npw;
    const int row = (int)(t % a.Hp);
    const int ncb = (int)(t / a.Hp);
    const int n = ncb / o.CB, c = (ncb - n * o.CB) * 16 + c16;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
    const int x0 = 8 * (pc - 1), y = row - 1;
    if (c < o.C && y >= 0 && y < a.H && x0 >= 0 && x0 < a.W) {
        const float* p = o.src + ((size_t)(n * o.ctot + o.coff + c) * a.H + y) * a.W + x0;
        float sc = 1.f, sh = 0.f;
        if (o.scale) {
            sc = o.scale[n * o.ctot + o.coff + c];
            sh = o.shift[n * o.ctot + o.coff + c];
        }
        if (o.vec) {
            const f4 lo = *reinterpret_cast<const f4*>(p);
            f4 hi = {0.f, 0.f, 0.f, 0.f};
            const bool has_hi = x0 + 4 < a.W;
            if (has_hi) hi = *reinterpret_cast<const f4*>(p + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = san_act(lo[i], sc, sh, o.slope);
                v[4 + i] = has_hi ? san_act(hi[i], sc, sh, o.slope) : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (x0 + i < a.W) v[i] = san_act(p[i], sc, sh, o.slope);
        }
    }
    u32x4 q1, q2, q3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t p1, p2, p3;
        split3_pair(v[2 * i], v[2 * i + 1], p1, p2, p3);
        q1[i] = p1;
        q2[i] = p2;
        q3[i] = p3;
    }
    uint16_t* w = o.out + idx * 8;
    *reinterpret_cast<u32x4*>(w) = q1;
    *reinterpret_cast<u32x4*>(w + o.plane) = q2;
    *reinterpret_cast<u32x4*>(w + 2 * o.plane) = q3;
}

// ---------------------------------------------------------------------------------------------------------
// pass 2
struct WBArgs {
    const uint16_t* xs;        // framed planes of T(x): [3][N][ci_b][Hp][npw][16][8]
    const uint16_t* dys;       // framed planes of dy:   [3][N][ncog * NB][Hp][npw][16][8]
    long long xplane, dyplane; // elements per part plane
    float* partial;            // [P][9][cin_pad][cout_pad]
    int N, cin, cout;
    int Hp, npw, npc, nbands, L;
    int Q;                     // piece columns = N * nbands * npc
    int P;                     // workgroup partitions of the piece columns (16 columns each)
    int ci_b, ncog, cin_pad, cout_pad;
    int rowstride;             // elements per framed row of one channel block (npw * 128)
};

struct XRow {
    Frag f[3][3];              // [part][kx]: the 8 pixels x0 + kx - 1 .. of one channel row
};

template <int NB>
__global__ void __launch_bounds__(kWT) wgrad_bf16x3_kernel(const WBArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;

    // XCD-aware order: consecutive logical ids (channel tile fastest, then pixel partition) share an L2
    int lin;
    {
        const int total = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3;
        lin = xcd * (total >> 3) + min(xcd, total & 7) + slot;
    }
    const int tiles = a.ci_b * a.ncog;
    const int tile = lin % tiles, pp = lin / tiles;
    const int cib = tile % a.ci_b, cog = tile / a.ci_b;

    // this lane group's piece column
    const int q = (pp * kWaves + wave) * 4 + g;
    const bool valid = q < a.Q;
    const int qc = min(q, a.Q - 1);
    const int xp = qc % a.npc;
    const int band = (qc / a.npc) % a.nbands;
    const int n = qc / (a.npc * a.nbands);
    const int r0 = band * a.L;                          // first image row; framed row index = image row + 1

    const int rowstride = a.rowstride;
    int xcur = (((n * a.ci_b + cib) * a.Hp + r0) * a.npw + xp + 1) * 128 + c * 8;      // framed row r0 = image row r0 - 1
    int dybase[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)   // a group without a column reads the zero top row of the frame, with row stride 0
        dybase[nb] = (((n * a.ncog * NB + cog * NB + nb) * a.Hp + (valid ? r0 + 1 : 0)) * a.npw + xp + 1) * 128 + c * 8;
    const int dystep = valid ? rowstride : 0;
    int dyrow = 0;

    const uint16_t* xpl[3] = {a.xs, a.xs + a.xplane, a.xs + 2 * a.xplane};
    const uint16_t* dpl[3] = {a.dys, a.dys + a.dyplane, a.dys + 2 * a.dyplane};

    u32x4 rd[3];
    uint32_t rm[3], rp[3];
    auto load_raw = [&]() {                             // the x row at xcur: aligned piece + the dword on each side
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint16_t* s = xpl[p] + xcur;
            rd[p] = *reinterpret_cast<const u32x4*>(s);
            rm[p] = *reinterpret_cast<const uint32_t*>(s - 122);      // pixels 6, 7 of the piece to the left
            rp[p] = *reinterpret_cast<const uint32_t*>(s + 128);      // pixels 0, 1 of the piece to the right
        }
        xcur += rowstride;
    };
    auto make_row = [&](XRow& r) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const u32x4 d = rd[p];
            const uint32_t s01 = __builtin_amdgcn_alignbit(d[1], d[0], 16), s12 = __builtin_amdgcn_alignbit(d[2], d[1], 16),
                           s23 = __builtin_amdgcn_alignbit(d[3], d[2], 16);
            r.f[p][1].u = d;
            r.f[p][0].u = u32x4{__builtin_amdgcn_alignbit(d[0], rm[p], 16), s01, s12, s23};
            r.f[p][2].u = u32x4{s01, s12, s23, __builtin_amdgcn_alignbit(rp[p], d[3], 16)};
        }
    };
    Frag dyf[NB][3];
    auto load_dy = [&](int nb) {
#pragma unroll
        for (int p = 0; p < 3; ++p) dyf[nb][p].u = *reinterpret_cast<const u32x4*>(dpl[p] + dybase[nb] + dyrow);
    };

    f4 acc[9][NB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f4{0.f, 0.f, 0.f, 0.f};

    XRow ra, rb, rc;
    load_raw();
    make_row(ra);
    load_raw();
    make_row(rb);
    load_raw();
    make_row(rc);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) load_dy(nb);
    dyrow += dystep;

    // one image row: r0 = rows above / at / below the dy row; the x row two below is fetched during the step
    // and replaces r0's registers at its end; dy block nb of the next row is fetched once block nb is done.
    // (the scheduling barriers pin the prefetches where they are written: left alone, the scheduler sinks every load
    // to just before its first use -- fewer live registers, fully exposed latency)
    auto step = [&](XRow& w0, XRow& w1, XRow& w2) {
        load_raw();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int pa = 0; pa < 3; ++pa)
#pragma unroll
                for (int pb = 0; pb < 3 - pa; ++pb) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        acc[kx][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0.f[pa][kx].v, dyf[nb][pb].v, acc[kx][nb], 0, 0, 0);
                        acc[3 + kx][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1.f[pa][kx].v, dyf[nb][pb].v, acc[3 + kx][nb], 0, 0, 0);
                        acc[6 + kx][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2.f[pa][kx].v, dyf[nb][pb].v, acc[6 + kx][nb], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            load_dy(nb);
            __builtin_amdgcn_sched_barrier(0);
        }
        dyrow += dystep;
        make_row(w0);
    };
    for (int t = 0; t < a.L; t += 3) {
        step(ra, rb, rc);
        step(rb, rc, ra);
        step(rc, ra, rb);
    }

    // ---- the four waves' tiles -> one: waves 1..3 park theirs in LDS, wave 0 adds them in a fixed order
    f4* red = reinterpret_cast<f4*>(smem);
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) red[((wave - 1) * 9 * NB + t * NB + nb) * 64 + lane] = acc[t][nb];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = a.partial + (size_t)pp * 9 * a.cin_pad * a.cout_pad;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f4 s = acc[t][nb];
#pragma unroll
                for (int w = 0; w < kWaves - 1; ++w) s += red[(w * 9 * NB + t * NB + nb) * 64 + lane];
                // D[m = 4 g + r][n = c]: rows = input channel of the block, columns = output channel
                float* o = out + ((size_t)t * a.cin_pad + cib * 16 + 4 * g) * a.cout_pad + (cog * NB + nb) * 16 + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(size_t)r * a.cout_pad] = s[r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// pass 2, direct form: the same tiling, but the wave reads the fp32 tensors themselves (no pass 1, no split planes)
// and splits its operand pieces in registers: ~52 VALU per dy piece and ~120 per x row (lazy affine, LeakyReLU,
// zero frame by predication, three-way split, the two shifted copies) against 54 NB MFMAs per row step.  That
// costs the matrix pipe 25-50 % more issue time per step than pass 2 proper but saves the HBM round trip of the
// split planes (10 bytes per element written and read back), which is the larger cost wherever the channel
// counts are low or the images large.  Needs 16-byte aligned rows (W % 4 == 0).
struct WDArgs {
    const float* x;
    const float* in_scale;
    const float* in_shift;
    float in_slope;
    const float* dy;
    float* partial;            // [P][9][cin_pad][cout_pad]
    int x_ctot, x_coff, cin, dy_ctot, dy_coff, cout;
    int H, W, npc, nbands, L;
    int Q, P, ci_b, ncog, cin_pad, cout_pad;
    const uint32_t* amax;      // F16 form: bits of max |dy| (device scalar)
};

// NP: operand parts used (3 = fp32-equivalent, six products; 2 = three products; 1 = plain bf16), san_set_conv_precision
template <int NB, int NP, bool F16>
__global__ void __launch_bounds__(kWT) wgrad_bf16x3_direct_kernel(const WDArgs a) {
    static_assert(!F16 || NP == 2, "the fp16 form has two parts");
    float dyS, dyInvS;
    amax_scale(F16 ? a.amax : nullptr, dyS, dyInvS);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    int lin;
    {
        const int total = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3;
        lin = xcd * (total >> 3) + min(xcd, total & 7) + slot;
    }
    const int tiles = a.ci_b * a.ncog;
    const int tile = lin % tiles, pp = lin / tiles;
    const int cib = tile % a.ci_b, cog = tile / a.ci_b;

    const int q = (pp * kWaves + wave) * 4 + g;
    const bool valid = q < a.Q;
    const int qc = min(q, a.Q - 1);
    const int xp = qc % a.npc;
    const int band = (qc / a.npc) % a.nbands;
    const int n = qc / (a.npc * a.nbands);
    const int r0 = band * a.L;
    const int rend = valid ? min(r0 + a.L, a.H) : 0;   // dy rows [r0, rend) are this column's
    const int H = a.H, W = a.W;
    const int x0 = 8 * xp;
    const bool hi_ok = x0 + 4 < W, left_ok = x0 > 0, right_ok = x0 + 8 < W;
    const int d_hi = hi_ok ? 4 : 0, d_l = left_ok ? -1 : 0, d_r = right_ok ? 8 : 0;    // masked reads stay inside the row

    const int ci = min(cib * 16 + c, a.cin - 1);        // lanes past the last channel redo it; their tile rows are never read
    const int xbase = ((n * a.x_ctot + a.x_coff + ci) * H) * W + x0;
    float sc = 1.f, sh = 0.f;
    if (a.in_scale) {
        sc = a.in_scale[n * a.x_ctot + a.x_coff + ci];
        sh = a.in_shift[n * a.x_ctot + a.x_coff + ci];
    }
    const float slope = a.in_slope;
    int dybase[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int co = min((cog * NB + nb) * 16 + c, a.cout - 1);
        dybase[nb] = ((n * a.dy_ctot + a.dy_coff + co) * H) * W + x0;
    }

    // ---- x rows: raw fetch (aligned halves + the pixel on each side), then activation, frame, split, shifts
    f4 xlo, xhi;
    float xl, xr_;
    int xrow = r0 - 1;                                  // image row of the raw registers
    auto load_x = [&]() {
        const float* p = a.x + xbase + min(max(xrow, 0), H - 1) * W;
        xlo = *reinterpret_cast<const f4*>(p);
        xhi = *reinterpret_cast<const f4*>(p + d_hi);
        xl = p[d_l];
        xr_ = p[d_r];
    };
    auto make_row = [&](XRow& r) {
        const bool ok = (unsigned)xrow < (unsigned)H;
        const bool okh = ok && hi_ok;
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = ok ? san_act(xlo[i], sc, sh, slope) : 0.f;
            v[4 + i] = okh ? san_act(xhi[i], sc, sh, slope) : 0.f;
        }
        const float vl = (ok && left_ok) ? san_act(xl, sc, sh, slope) : 0.f;
        const float vr = (ok && right_ok) ? san_act(xr_, sc, sh, slope) : 0.f;
        uint32_t d[3][4], dm[3], dp[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) split_pair<F16>(v[2 * i], v[2 * i + 1], d[0][i], d[1][i], d[2][i]);
        split_pair<F16>(0.f, vl, dm[0], dm[1], dm[2]);       // high half = the pixel to the left
        split_pair<F16>(vr, 0.f, dp[0], dp[1], dp[2]);       // low half = the pixel to the right
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint32_t s01 = __builtin_amdgcn_alignbit(d[p][1], d[p][0], 16), s12 = __builtin_amdgcn_alignbit(d[p][2], d[p][1], 16),
                           s23 = __builtin_amdgcn_alignbit(d[p][3], d[p][2], 16);
            r.f[p][1].u = u32x4{d[p][0], d[p][1], d[p][2], d[p][3]};
            r.f[p][0].u = u32x4{__builtin_amdgcn_alignbit(d[p][0], dm[p], 16), s01, s12, s23};
            r.f[p][2].u = u32x4{s01, s12, s23, __builtin_amdgcn_alignbit(dp[p], d[p][3], 16)};
        }
        ++xrow;
    };

    // ---- dy rows
    f4 dlo[NB], dhi[NB];
    int dyrow = r0;                                     // image row of the raw registers
    auto load_dy = [&]() {
        const int ro = min(dyrow, H - 1) * W;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float* p = a.dy + dybase[nb] + ro;
            dlo[nb] = *reinterpret_cast<const f4*>(p);
            dhi[nb] = *reinterpret_cast<const f4*>(p + d_hi);
        }
    };
    Frag dyf[NB][3];
    auto make_dy = [&]() {
        const bool ok = dyrow < rend;
        const bool okh = ok && hi_ok;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            uint32_t d[3][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                split_pair<F16>(ok ? dlo[nb][2 * i] * dyS : 0.f, ok ? dlo[nb][2 * i + 1] * dyS : 0.f, d[0][i], d[1][i], d[2][i]);
                split_pair<F16>(okh ? dhi[nb][2 * i] * dyS : 0.f, okh ? dhi[nb][2 * i + 1] * dyS : 0.f, d[0][2 + i], d[1][2 + i], d[2][2 + i]);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) dyf[nb][p].u = u32x4{d[p][0], d[p][1], d[p][2], d[p][3]};
        }
        ++dyrow;
    };

    f4 acc[9][NB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f4{0.f, 0.f, 0.f, 0.f};

    XRow ra, rb, rc;
    load_x();
    make_row(ra);
    load_x();
    make_row(rb);
    load_x();
    make_row(rc);
    load_dy();
    load_x();                                           // row r0 + 2, converted at the end of the first step
    __builtin_amdgcn_sched_barrier(0);

    auto step = [&](XRow& w0, XRow& w1, XRow& w2) {
        make_dy();                                      // the dy row fetched during the previous step
        __builtin_amdgcn_sched_barrier(0);
        load_dy();                                      // next dy row: a whole step ahead of its use
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int pa = 0; pa < NP; ++pa)
#pragma unroll
                for (int pb = 0; pb < NP - pa; ++pb)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        acc[kx][nb] = mma<F16>(w0.f[pa][kx], dyf[nb][pb], acc[kx][nb]);
                        acc[3 + kx][nb] = mma<F16>(w1.f[pa][kx], dyf[nb][pb], acc[3 + kx][nb]);
                        acc[6 + kx][nb] = mma<F16>(w2.f[pa][kx], dyf[nb][pb], acc[6 + kx][nb]);
                    }
        __builtin_amdgcn_sched_barrier(0);
        make_row(w0);                                   // the x row fetched during the previous step replaces the oldest
        load_x();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < a.L; t += 3) {
        step(ra, rb, rc);
        step(rb, rc, ra);
        step(rc, ra, rb);
    }

    if constexpr (F16) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[t][nb] *= dyInvS;
    }
    f4* red = reinterpret_cast<f4*>(smem);
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) red[((wave - 1) * 9 * NB + t * NB + nb) * 64 + lane] = acc[t][nb];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = a.partial + (size_t)pp * 9 * a.cin_pad * a.cout_pad;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f4 s = acc[t][nb];
#pragma unroll
                for (int w = 0; w < kWaves - 1; ++w) s += red[(w * 9 * NB + t * NB + nb) * 64 + lane];
                float* o = out + ((size_t)t * a.cin_pad + cib * 16 + 4 * g) * a.cout_pad + (cog * NB + nb) * 16 + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(size_t)r * a.cout_pad] = s[r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 1x1 weight gradient (the 2x2 transposed convolutions run as 1x1 convolutions to 4 cout + pixel shuffle, and the
// alignment net's 1x1 layers): no spatial coupling, so a channel plane is one flat run of H*W pixels.  A wave owns
// one block of 16 input channels x NB blocks of 16 output channels and a span of 32 L consecutive pixels; per step
// its four lane groups take four neighbouring 8-pixel pieces (128 contiguous bytes per channel), split them in
// registers and issue 6 NB MFMAs.  VALU-bound (~52 operations per piece against 6 MFMAs), which is still several
// times faster than the fp32 kernel on these short-K, wide-channel shapes.  Needs H*W % 4 == 0 and 16-byte bases.
struct W1Args {
    const float* x;
    const float* in_scale;
    const float* in_shift;
    float in_slope;
    const float* dy;
    float* partial;            // [P][1][cin_pad][cout_pad]
    int x_ctot, x_coff, cin, dy_ctot, dy_coff, cout;
    int HW, L, S;              // pixels per plane, steps per span, spans per image
    int Q, P, ci_b, ncog, cin_pad, cout_pad;
    const uint32_t* amax;      // F16 form: bits of max |dy| (device scalar)
};

template <int NB, int NP, bool F16>
__global__ void __launch_bounds__(kWT) wgrad1x1_bf16x3_kernel(const W1Args a) {
    static_assert(!F16 || NP == 2, "the fp16 form has two parts");
    float dyS, dyInvS;
    amax_scale(F16 ? a.amax : nullptr, dyS, dyInvS);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    int lin;
    {
        const int total = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3;
        lin = xcd * (total >> 3) + min(xcd, total & 7) + slot;
    }
    const int tiles = a.ci_b * a.ncog;
    const int tile = lin % tiles, pp = lin / tiles;
    const int cib = tile % a.ci_b, cog = tile / a.ci_b;

    const int q = pp * kWaves + wave;                   // this wave's span
    const bool valid = q < a.Q;
    const int qc = min(q, a.Q - 1);
    const int n = qc / a.S, sp = qc - n * a.S;
    const int HW = a.HW;
    const int pend = valid ? min((sp + 1) * 32 * a.L, HW) : 0;       // pixels [sp * 32 L, pend) are this wave's
    int px = sp * 32 * a.L + 8 * g;                     // this lane group's piece of the current step

    const int ci = min(cib * 16 + c, a.cin - 1);
    const int xbase = (n * a.x_ctot + a.x_coff + ci) * HW;
    float sc = 1.f, sh = 0.f;
    if (a.in_scale) {
        sc = a.in_scale[n * a.x_ctot + a.x_coff + ci];
        sh = a.in_shift[n * a.x_ctot + a.x_coff + ci];
    }
    const float slope = a.in_slope;
    int dybase[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) dybase[nb] = (n * a.dy_ctot + a.dy_coff + min((cog * NB + nb) * 16 + c, a.cout - 1)) * HW;

    f4 xlo, xhi, dlo[NB], dhi[NB];
    int pxl = px;                                       // pixel of the raw registers
    auto load = [&]() {
        const int p0 = min(pxl, HW - 4), p1 = min(pxl + 4, HW - 4);      // masked reads stay inside the plane
        xlo = *reinterpret_cast<const f4*>(a.x + xbase + p0);
        xhi = *reinterpret_cast<const f4*>(a.x + xbase + p1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            dlo[nb] = *reinterpret_cast<const f4*>(a.dy + dybase[nb] + p0);
            dhi[nb] = *reinterpret_cast<const f4*>(a.dy + dybase[nb] + p1);
        }
    };
    Frag xf[3], dyf[NB][3];
    auto convert = [&]() {
        const bool ok = pxl < pend, okh = pxl + 4 < pend;
        uint32_t d[3][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            split_pair<F16>(ok ? san_act(xlo[2 * i], sc, sh, slope) : 0.f, ok ? san_act(xlo[2 * i + 1], sc, sh, slope) : 0.f, d[0][i], d[1][i],
                        d[2][i]);
            split_pair<F16>(okh ? san_act(xhi[2 * i], sc, sh, slope) : 0.f, okh ? san_act(xhi[2 * i + 1], sc, sh, slope) : 0.f, d[0][2 + i],
                        d[1][2 + i], d[2][2 + i]);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) xf[p].u = u32x4{d[p][0], d[p][1], d[p][2], d[p][3]};
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                split_pair<F16>(ok ? dlo[nb][2 * i] * dyS : 0.f, ok ? dlo[nb][2 * i + 1] * dyS : 0.f, d[0][i], d[1][i], d[2][i]);
                split_pair<F16>(okh ? dhi[nb][2 * i] * dyS : 0.f, okh ? dhi[nb][2 * i + 1] * dyS : 0.f, d[0][2 + i], d[1][2 + i], d[2][2 + i]);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) dyf[nb][p].u = u32x4{d[p][0], d[p][1], d[p][2], d[p][3]};
        }
        pxl += 32;
    };

    f4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
    load();
    for (int t = 0; t < a.L; ++t) {
        convert();                                      // the pieces fetched during the previous step
        __builtin_amdgcn_sched_barrier(0);
        load();                                         // next step's pieces (clamped past the end; masked when converted)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pa = 0; pa < NP; ++pa)
#pragma unroll
            for (int pb = 0; pb < NP - pa; ++pb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[nb] = mma<F16>(xf[pa], dyf[nb][pb], acc[nb]);
        __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (F16) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] *= dyInvS;
    }
    f4* red = reinterpret_cast<f4*>(smem);
    if (wave > 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) red[((wave - 1) * NB + nb) * 64 + lane] = acc[nb];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = a.partial + (size_t)pp * a.cin_pad * a.cout_pad;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f4 s = acc[nb];
#pragma unroll
            for (int w = 0; w < kWaves - 1; ++w) s += red[(w * NB + nb) * 64 + lane];
            float* o = out + ((size_t)cib * 16 + 4 * g) * a.cout_pad + (cog * NB + nb) * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(size_t)r * a.cout_pad] = s[r];
        }
    }
}

// pass 3: dw[co][ci][tap] (+)= sum_pp partial[pp][tap][ci][co], fixed order.  64 consecutive output channels per
// thread row (coalesced partial reads), the partitions dealt to 4 thread rows whose sums meet in LDS in a fixed order.
// transposed: dw is [ci][co][tap] (the transposed convolution's weight layout).
struct RJob {
    const float* partial;
    float* dw;
    int P, cin, cout, cin_pad, cout_pad, accumulate, taps, transposed;
};

// (a thread owns FOUR consecutive output channels: 16-byte loads from the padded tile; the order of the sum per element is
// unchanged: partitions pp = row, row + 4, ... in four interleaved chains, then the four thread rows in order)
__host__ __device__ inline int wgrad_reduce_units(const RJob& j) { return ((j.cout + 3) / 4) * j.cin * j.taps; }
__device__ __forceinline__ void wgrad_reduce_body(const RJob& j, int blk) {
    __shared__ f4 red[3][64];
    const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
    const int cq = (j.cout + 3) / 4;
    const int e = blk * 64 + lane;
    const bool live = e < cq * j.cin * j.taps;
    const int ec = live ? e : 0;
    const int q4 = ec % cq;
    const int ci = (ec / cq) % j.cin;
    const int tap = ec / (cq * j.cin);
    const size_t stride = (size_t)j.taps * j.cin_pad * j.cout_pad;
    const float* p = j.partial + ((size_t)tap * j.cin_pad + ci) * j.cout_pad + 4 * q4;      // cout_pad % 16 == 0: aligned, in the tile
    const f4 z = f4{0.f, 0.f, 0.f, 0.f};
    f4 s0 = z, s1 = z, s2 = z, s3 = z;
    int pp = row;
    for (; pp + 12 < j.P; pp += 16) {
        s0 += *reinterpret_cast<const f4*>(p + (size_t)pp * stride);
        s1 += *reinterpret_cast<const f4*>(p + (size_t)(pp + 4) * stride);
        s2 += *reinterpret_cast<const f4*>(p + (size_t)(pp + 8) * stride);
        s3 += *reinterpret_cast<const f4*>(p + (size_t)(pp + 12) * stride);
    }
    for (; pp < j.P; pp += 4) s0 += *reinterpret_cast<const f4*>(p + (size_t)pp * stride);
    f4 s = (s0 + s1) + (s2 + s3);
    if (row > 0) red[row - 1][lane] = s;
    __syncthreads();
    if (row == 0 && live) {
        s = ((s + red[0][lane]) + red[1][lane]) + red[2][lane];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = 4 * q4 + k;
            if (co < j.cout) {
                float* o = j.transposed ? j.dw + ((size_t)ci * j.cout + co) * j.taps + tap : j.dw + ((size_t)co * j.cin + ci) * j.taps + tap;
                *o = j.accumulate ? *o + s[k] : s[k];
            }
        }
    }
}

__global__ void __launch_bounds__(256) wgrad_bf16x3_reduce_kernel(const RJob j) { wgrad_reduce_body(j, blockIdx.x); }

// Deferred form: up to kRBatch layers' reductions in ONE launch (san_wgrad_defer / san_wgrad_defer_flush).  The jobs ride in
// the kernel arguments (no table upload: nothing to copy, capture-safe); block b belongs to the job whose block range holds it.
constexpr int kRBatch = 48;
struct RBatch {
    RJob j[kRBatch];
    int first[kRBatch + 1];    // first block of each job; first[n] = total
    int n;
};

__global__ void __launch_bounds__(256) wgrad_reduce_batch_kernel(const RBatch b) {
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.first[k + 1]) ++k;
    wgrad_reduce_body(b.j[k], (int)blockIdx.x - b.first[k]);
}

std::mutex g_defer_mu;
std::vector<RJob> g_defer;
int g_defer_on = 0;

// the reduction of one weight gradient: now, or (deferred mode) queued for the next san_wgrad_defer_flush
int reduce_or_defer(const RJob& j, hipStream_t s) {
    {
        std::lock_guard<std::mutex> lk(g_defer_mu);
        if (g_defer_on) {
            g_defer.push_back(j);
            return SAN_OK;
        }
    }
    hipLaunchKernelGGL(wgrad_bf16x3_reduce_kernel, dim3(san_cdiv(wgrad_reduce_units(j), 64)), dim3(256), 0, s, j);
    return SAN_OK;
}

struct WBPlan {
    int ci_b, nco_b, NB, ncog, npc, nbands, L, Hp, Wp, Q, P, tiles, cin_pad, cout_pad;
    long long xplane, dyplane;
    size_t x_bytes, dy_bytes, partial_bytes;
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// max_nb: 4 for the split-plane form (NB = 5 spills there), 5 for the direct form
WBPlan wb_plan(int n, int h, int w, int cin, int cout, int max_nb = 4) {
    WBPlan p{};
    p.ci_b = san_cdiv(cin, 16);
    p.nco_b = san_cdiv(cout, 16);
    // output-channel blocks per wave: the split of nco_b into groups that wastes the fewest blocks, ties to the larger
    int best = 0;
    double best_eff = 0.0;
    for (int nb = max_nb; nb >= 1; --nb) {
        const int gcount = san_cdiv(p.nco_b, nb);
        const double eff = (double)p.nco_b / (gcount * nb);
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best = nb;
        }
    }
    p.NB = best;
    p.ncog = san_cdiv(p.nco_b, p.NB);
    p.tiles = p.ci_b * p.ncog;
    p.npc = san_cdiv(w, 8);
    // Row bands: a workgroup (one per CU: 4 waves x ~400 registers) runs L row steps plus about 4 steps' worth of
    // fixed cost (window fill, LDS reduction, launch), and the grid runs in rounds of 256 workgroups: take the
    // band count with the smallest rounds x (L + 4).
    long long best_cost = -1;
    for (int nbands = 1; nbands <= (h + 2) / 3; ++nbands) {
        const int L = 3 * san_cdiv(san_cdiv(h, nbands), 3);
        const int nb2 = san_cdiv(h, L);
        const int wgs = san_cdiv(n * nb2 * p.npc, 16) * p.tiles;
        const long long cost = (long long)san_cdiv(wgs, 256) * (L + 4);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            p.L = L;
            p.nbands = nb2;
        }
    }
    p.Hp = p.nbands * p.L + 3;
    p.Wp = 8 * (p.npc + 2);
    p.Q = n * p.nbands * p.npc;
    p.P = san_cdiv(p.Q, 16);
    p.cin_pad = p.ci_b * 16;
    p.cout_pad = p.ncog * p.NB * 16;
    p.xplane = (long long)n * p.cin_pad * p.Hp * p.Wp;
    p.dyplane = (long long)n * p.cout_pad * p.Hp * p.Wp;
    p.x_bytes = align256((size_t)p.xplane * 6);
    p.dy_bytes = align256((size_t)p.dyplane * 6);
    p.partial_bytes = align256((size_t)p.P * 9 * p.cin_pad * p.cout_pad * sizeof(float));
    return p;
}

template <int NB>
int launch_wb(const WBArgs& a, int grid, hipStream_t s) {
    constexpr size_t lds = (size_t)(kWaves - 1) * 9 * NB * 64 * sizeof(f4);
    static SanPerDevice configured;
    const int dev__ = san_current_device();
    if (!configured.has(dev__)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16x3_kernel<NB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            san_set_error("cannot reserve %d bytes of LDS for the bf16x3 weight gradient", (int)lds);
            return SAN_E_UNSUPPORTED;
        }
        configured.mark(dev__);
    }
    hipLaunchKernelGGL((wgrad_bf16x3_kernel<NB>), dim3(grid), dim3(kWT), lds, s, a);
    return SAN_OK;
}

int g_wgrad_np = 3;            // operand parts (san_set_conv_precision)

template <int NB, int NP, bool F16 = false>
int launch_wdn(const WDArgs& a, int grid, hipStream_t s) {
    constexpr size_t lds = (size_t)(kWaves - 1) * 9 * NB * 64 * sizeof(f4);
    static SanPerDevice configured;
    const int dev__ = san_current_device();
    if (!configured.has(dev__)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16x3_direct_kernel<NB, NP, F16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            san_set_error("cannot reserve %d bytes of LDS for the bf16x3 weight gradient", (int)lds);
            return SAN_E_UNSUPPORTED;
        }
        configured.mark(dev__);
    }
    hipLaunchKernelGGL((wgrad_bf16x3_direct_kernel<NB, NP, F16>), dim3(grid), dim3(kWT), lds, s, a);
    return SAN_OK;
}

template <int NB>
int launch_wd(const WDArgs& a, int grid, hipStream_t s) {
    if (a.amax && g_wgrad_np == 3) return launch_wdn<NB, 2, true>(a, grid, s);      // two fp16 parts, dy scaled by its maximum
    switch (g_wgrad_np) {
        case 1: return launch_wdn<NB, 1>(a, grid, s);
        case 2: return launch_wdn<NB, 2>(a, grid, s);
        default: return launch_wdn<NB, 3>(a, grid, s);
    }
}

SplitOne split_one(const float* src, int ctot, int coff, int C, int CB, const float* scale, const float* shift, float slope,
                   int n, int w, const WBPlan& p, uint16_t* out, long long plane) {
    SplitOne o{};
    o.src = src;
    o.scale = scale;
    o.shift = shift;
    o.slope = slope;
    o.ctot = ctot;
    o.coff = coff;
    o.C = C;
    o.CB = CB;
    o.out = out;
    o.plane = plane;
    o.total = (long long)n * CB * p.Hp * (p.Wp / 8) * 16;
    o.vec = ((uintptr_t)src & 15) == 0 && (w % 4) == 0;
    return o;
}

struct W1Plan {
    int ci_b, nco_b, NB, ncog, L, S, Q, P, cin_pad, cout_pad;
    size_t partial_bytes;
};

W1Plan w1_plan(int n, int hw, int cin, int cout) {
    W1Plan p{};
    p.ci_b = san_cdiv(cin, 16);
    p.nco_b = san_cdiv(cout, 16);
    int best = 1;
    double best_eff = 0.0;
    for (int nb = 5; nb >= 1; --nb) {
        const double eff = (double)p.nco_b / (san_cdiv(p.nco_b, nb) * nb);
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best = nb;
        }
    }
    p.NB = best;
    p.ncog = san_cdiv(p.nco_b, p.NB);
    const int tiles = p.ci_b * p.ncog;
    const int steps = san_cdiv(hw, 32);                 // per image
    long long best_cost = -1;
    for (int L = 1; L <= steps; ++L) {
        const int S = san_cdiv(steps, L);
        const int wgs = san_cdiv(n * S, kWaves) * tiles;
        const long long cost = (long long)san_cdiv(wgs, 256) * (L + 6);    // ~6 steps' worth of fixed cost per workgroup
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            p.L = L;
            p.S = S;
        }
    }
    p.Q = n * p.S;
    p.P = san_cdiv(p.Q, kWaves);
    p.cin_pad = p.ci_b * 16;
    p.cout_pad = p.ncog * p.NB * 16;
    p.partial_bytes = align256((size_t)p.P * p.cin_pad * p.cout_pad * sizeof(float));
    return p;
}

template <int NB>
int launch_w1(const W1Args& a, int grid, hipStream_t s) {
    constexpr size_t lds = (size_t)(kWaves - 1) * NB * 64 * sizeof(f4);
    if (a.amax && g_wgrad_np == 3) {
        hipLaunchKernelGGL((wgrad1x1_bf16x3_kernel<NB, 2, true>), dim3(grid), dim3(kWT), lds, s, a);
        return SAN_OK;
    }
    switch (g_wgrad_np) {
        case 1: hipLaunchKernelGGL((wgrad1x1_bf16x3_kernel<NB, 1, false>), dim3(grid), dim3(kWT), lds, s, a); break;
        case 2: hipLaunchKernelGGL((wgrad1x1_bf16x3_kernel<NB, 2, false>), dim3(grid), dim3(kWT), lds, s, a); break;
        default: hipLaunchKernelGGL((wgrad1x1_bf16x3_kernel<NB, 3, false>), dim3(grid), dim3(kWT), lds, s, a); break;
    }
    return SAN_OK;
}

// the direct form (no split planes) needs 16-byte aligned rows and 32-bit element offsets into the fp32 tensors
int g_wb_mode = -1;            // -1 automatic, 0 split planes, 1 direct (tests / tuning: san_conv_wgrad_bf16x3_set_mode)
bool wb_direct(const float* x, const float* dy, int n, int h, int w, int x_ctot, int dy_ctot) {
    const bool can = (w % 4) == 0 && ((((uintptr_t)x) | ((uintptr_t)dy)) & 15) == 0 &&
                     (double)n * (x_ctot > dy_ctot ? x_ctot : dy_ctot) * h * w < (double)(1ll << 30);
    if (!can) return false;
    if (g_wb_mode >= 0) return g_wb_mode == 1;
    return true;
}

}  // namespace

// the split-plane form (rows that are not 16-byte aligned) always runs with three parts
void san_wgrad_set_parts(int parts) { g_wgrad_np = parts; }

extern "C" {

// tuning / test hook: -1 automatic choice between the two forms of pass 2, 0 split planes, 1 direct
int san_conv_wgrad_bf16x3_set_mode(int mode) {
    g_wb_mode = mode;
    return SAN_OK;
}

// 1 when the bf16x3 weight gradient is the faster choice for this layer (what the dispatcher asks); the kernel
// itself runs any 3x3 layer whose split planes stay below 2^30 elements (san_conv_wgrad_bf16x3_supported).
int san_conv_wgrad_bf16x3_supported(int n, int h, int w, int cin, int cout, int ks) {
    if (ks != 3 || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
    const WBPlan p = wb_plan(n, h, w, cin, cout);
    if (p.xplane >= (1ll << 30) || p.dyplane >= (1ll << 30)) return 0;      // 32-bit element offsets in the kernel
    return 1;
}

int san_conv_wgrad_bf16x3_eligible(int n, int h, int w, int cin, int cout, int ks) {
    if (!san_conv_wgrad_bf16x3_supported(n, h, w, cin, cout, ks)) return 0;
    if (h < 6 || w < 8) return 0;
    const double work = (double)n * h * w * (cin > 16 ? cin : 16) * (cout > 16 ? cout : 16);
    if (work < 2.0e7) return 0;                               // too little work to pay for the extra launch
    if ((w % 4) == 0) return 1;   // direct form: measured faster than the fp32 kernel from 3 -> 18 @320^2 (71 vs 88 us) upwards
    // split-plane form (rows not 16-byte aligned): the planes' HBM round trip needs full tiles to pay off
    if (cin < 16 || cout < 16) return 0;
    if ((cin < 32 || cout < 32) && (double)n * h * w > 8.0 * 160 * 160) return 0;
    return 1;
}

// bytes of scratch san_conv2d_wgrad_bf16x3 needs (split planes of x and dy + the partial tiles)
size_t san_conv_wgrad_bf16x3_scratch_bytes(int n, int h, int w, int cin, int cout) {
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
    const WBPlan p = wb_plan(n, h, w, cin, cout), d = wb_plan(n, h, w, cin, cout, 5);
    const size_t split = p.x_bytes + p.dy_bytes + p.partial_bytes;       // either form may be taken (pointer alignment)
    return split > d.partial_bytes ? split : d.partial_bytes;
}

static int wgrad3_impl(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                       float in_slope, const float* dy, int dy_ctot, int dy_coff, int cout, float* dw, int accumulate,
                       void* scratch, int n, int h, int w, const void* dy_amax, void* stream) {
    SAN_CHECK_ARG(x && dy && dw && scratch, "null pointer");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "bad dims");
    SAN_CHECK_ARG(x_coff >= 0 && x_coff + cin <= x_ctot && dy_coff >= 0 && dy_coff + cout <= dy_ctot, "bad channel view");
    SAN_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "in_scale/in_shift must come together");
    SAN_CHECK_ARG(san_conv_wgrad_bf16x3_supported(n, h, w, cin, cout, 3), "layer too large for 32-bit plane offsets (see san_conv_wgrad_bf16x3_supported)");
    SAN_CHECK_ARG(((uintptr_t)scratch & 15) == 0, "scratch must be 16-byte aligned");
    const bool direct = wb_direct(x, dy, n, h, w, x_ctot, dy_ctot);
    const WBPlan p = wb_plan(n, h, w, cin, cout, direct ? 5 : 4);
    hipStream_t s = (hipStream_t)stream;
    unsigned char* base = static_cast<unsigned char*>(scratch);
    uint16_t* xs = reinterpret_cast<uint16_t*>(base);
    uint16_t* dys = reinterpret_cast<uint16_t*>(base + p.x_bytes);
    float* partial = direct ? reinterpret_cast<float*>(base) : reinterpret_cast<float*>(base + p.x_bytes + p.dy_bytes);

    const int grid = p.P * p.tiles;
    int rc = SAN_OK;
    if (direct) {
        WDArgs d{};
        d.x = x;
        d.in_scale = in_scale;
        d.in_shift = in_shift;
        d.in_slope = in_slope;
        d.dy = dy;
        d.partial = partial;
        d.x_ctot = x_ctot;
        d.x_coff = x_coff;
        d.cin = cin;
        d.dy_ctot = dy_ctot;
        d.dy_coff = dy_coff;
        d.cout = cout;
        d.H = h;
        d.W = w;
        d.npc = p.npc;
        d.nbands = p.nbands;
        d.L = p.L;
        d.Q = p.Q;
        d.P = p.P;
        d.ci_b = p.ci_b;
        d.ncog = p.ncog;
        d.cin_pad = p.cin_pad;
        d.cout_pad = p.cout_pad;
        d.amax = static_cast<const uint32_t*>(dy_amax);
        switch (p.NB) {
            case 1: rc = launch_wd<1>(d, grid, s); break;
            case 2: rc = launch_wd<2>(d, grid, s); break;
            case 3: rc = launch_wd<3>(d, grid, s); break;
            case 4: rc = launch_wd<4>(d, grid, s); break;
            default: rc = launch_wd<5>(d, grid, s); break;
        }
    } else {
        SplitArgs sa{};
        sa.t[0] = split_one(x, x_ctot, x_coff, cin, p.ci_b, in_scale, in_shift, in_slope, n, w, p, xs, p.xplane);
        sa.t[1] = split_one(dy, dy_ctot, dy_coff, cout, p.cout_pad / 16, nullptr, nullptr, 1.f, n, w, p, dys, p.dyplane);
        sa.blocks0 = (unsigned)((sa.t[0].total + 255) / 256);
        sa.H = h;
        sa.W = w;
        sa.Hp = p.Hp;
        sa.npw = p.Wp / 8;
        hipLaunchKernelGGL(split_planes_kernel, dim3(sa.blocks0 + (unsigned)((sa.t[1].total + 255) / 256)), dim3(256), 0, s, sa);
        SAN_LAUNCH_CHECK();

        WBArgs a{};
        a.xs = xs;
        a.dys = dys;
        a.xplane = p.xplane;
        a.dyplane = p.dyplane;
        a.partial = partial;
        a.N = n;
        a.cin = cin;
        a.cout = cout;
        a.Hp = p.Hp;
        a.npw = p.Wp / 8;
        a.npc = p.npc;
        a.nbands = p.nbands;
        a.L = p.L;
        a.Q = p.Q;
        a.P = p.P;
        a.ci_b = p.ci_b;
        a.ncog = p.ncog;
        a.cin_pad = p.cin_pad;
        a.cout_pad = p.cout_pad;
        a.rowstride = a.npw * 128;
        switch (p.NB) {
            case 1: rc = launch_wb<1>(a, grid, s); break;
            case 2: rc = launch_wb<2>(a, grid, s); break;
            case 3: rc = launch_wb<3>(a, grid, s); break;
            default: rc = launch_wb<4>(a, grid, s); break;
        }
    }
    if (rc != SAN_OK) return rc;
    SAN_LAUNCH_CHECK();
    const int count = cout * cin * 9;
    (void)count;
    reduce_or_defer(RJob{partial, dw, p.P, cin, cout, p.cin_pad, p.cout_pad, accumulate, 9, 0}, s);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

// ---- 1x1 form (see wgrad1x1_bf16x3_kernel) ----
int san_conv1x1_wgrad_bf16x3_eligible(int n, int h, int w, int cin, int cout) {
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
    if (((long long)h * w) % 4 != 0) return 0;
    if ((double)n * (cin > cout ? cin : cout) * h * w >= (double)(1ll << 30)) return 0;      // 32-bit element offsets
    if ((double)n * h * w * (cin > 16 ? cin : 16) * (cout > 16 ? cout : 16) < 2.0e7) return 0;
    return 1;
}

size_t san_conv1x1_wgrad_bf16x3_scratch_bytes(int n, int h, int w, int cin, int cout) {
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
    return w1_plan(n, h * w, cin, cout).partial_bytes;
}

static int wgrad1_impl(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                       float in_slope, const float* dy, int dy_ctot, int dy_coff, int cout, float* dw, int accumulate,
                       int transposed, void* scratch, int n, int h, int w, const void* dy_amax, void* stream) {
    SAN_CHECK_ARG(x && dy && dw && scratch, "null pointer");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "bad dims");
    SAN_CHECK_ARG(x_coff >= 0 && x_coff + cin <= x_ctot && dy_coff >= 0 && dy_coff + cout <= dy_ctot, "bad channel view");
    SAN_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "in_scale/in_shift must come together");
    SAN_CHECK_ARG(((long long)h * w) % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)dy)) & 15) == 0, "needs 16-byte aligned channel planes");
    SAN_CHECK_ARG((double)n * (x_ctot > dy_ctot ? x_ctot : dy_ctot) * h * w < (double)(1ll << 30), "tensor too large for 32-bit offsets");
    SAN_CHECK_ARG(((uintptr_t)scratch & 15) == 0, "scratch must be 16-byte aligned");
    const W1Plan p = w1_plan(n, h * w, cin, cout);
    hipStream_t s = (hipStream_t)stream;
    W1Args a{};
    a.x = x;
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_slope = in_slope;
    a.dy = dy;
    a.partial = static_cast<float*>(scratch);
    a.x_ctot = x_ctot;
    a.x_coff = x_coff;
    a.cin = cin;
    a.dy_ctot = dy_ctot;
    a.dy_coff = dy_coff;
    a.cout = cout;
    a.HW = h * w;
    a.L = p.L;
    a.S = p.S;
    a.Q = p.Q;
    a.P = p.P;
    a.ci_b = p.ci_b;
    a.ncog = p.ncog;
    a.cin_pad = p.cin_pad;
    a.cout_pad = p.cout_pad;
    a.amax = static_cast<const uint32_t*>(dy_amax);
    const int grid = p.P * p.ci_b * p.ncog;
    int rc = SAN_OK;
    switch (p.NB) {
        case 1: rc = launch_w1<1>(a, grid, s); break;
        case 2: rc = launch_w1<2>(a, grid, s); break;
        case 3: rc = launch_w1<3>(a, grid, s); break;
        case 4: rc = launch_w1<4>(a, grid, s); break;
        default: rc = launch_w1<5>(a, grid, s); break;
    }
    if (rc != SAN_OK) return rc;
    SAN_LAUNCH_CHECK();
    const int count = cout * cin;
    (void)count;
    reduce_or_defer(RJob{a.partial, dw, p.P, cin, cout, p.cin_pad, p.cout_pad, accumulate, 1, transposed}, s);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_conv2d_wgrad_bf16x3(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                            float in_slope, const float* dy, int dy_ctot, int dy_coff, int cout, float* dw, int accumulate,
                            void* scratch, int n, int h, int w, void* stream) {
    return wgrad3_impl(x, x_ctot, x_coff, cin, in_scale, in_shift, in_slope, dy, dy_ctot, dy_coff, cout, dw, accumulate, scratch, n, h,
                       w, nullptr, stream);
}

int san_conv1x1_wgrad_bf16x3(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                             float in_slope, const float* dy, int dy_ctot, int dy_coff, int cout, float* dw, int accumulate,
                             int transposed, void* scratch, int n, int h, int w, void* stream) {
    return wgrad1_impl(x, x_ctot, x_coff, cin, in_scale, in_shift, in_slope, dy, dy_ctot, dy_coff, cout, dw, accumulate, transposed,
                       scratch, n, h, w, nullptr, stream);
}

// The same with dy_amax = device pointer to the bits of max |dy| (maintained by san_act_bwd* with an integer atomic max):
// in the fp32-equivalent mode the kernels then run on two fp16 parts per operand (three products), dy scaled by a power
// of two derived from that maximum.  dy_amax == NULL: exactly the entry points above.
int san_conv2d_wgrad_bf16x3_amax(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                                 float in_slope, const float* dy, int dy_ctot, int dy_coff, int cout, float* dw, int accumulate,
                                 void* scratch, const void* dy_amax, int n, int h, int w, void* stream) {
    return wgrad3_impl(x, x_ctot, x_coff, cin, in_scale, in_shift, in_slope, dy, dy_ctot, dy_coff, cout, dw, accumulate, scratch, n, h,
                       w, dy_amax, stream);
}

int san_conv1x1_wgrad_bf16x3_amax(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                                  float in_slope, const float* dy, int dy_ctot, int dy_coff, int cout, float* dw, int accumulate,
                                  int transposed, void* scratch, const void* dy_amax, int n, int h, int w, void* stream) {
    return wgrad1_impl(x, x_ctot, x_coff, cin, in_scale, in_shift, in_slope, dy, dy_ctot, dy_coff, cout, dw, accumulate, transposed,
                       scratch, n, h, w, dy_amax, stream);
}

// Deferred reductions.  While on, the bf16x3 / fp16-part weight-gradient entry points launch only their main kernel and queue
// the fixed-order reduction of their partial tiles; san_wgrad_defer_flush launches the queued reductions, up to 48 layers per
// launch (the jobs ride in the kernel arguments).  The caller gives every queued weight gradient its own `scratch` and flushes
// before a second gradient of the same dw is queued (each dw element is written by exactly one thread of one launch, so the
// result is bit-identical to the immediate form).  ~320 launches per training step become ~8.
int san_wgrad_defer(int on) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    const int prev = g_defer_on;
    g_defer_on = on ? 1 : 0;
    return prev;
}

int san_wgrad_defer_pending(void) {
    std::lock_guard<std::mutex> lk(g_defer_mu);
    return (int)g_defer.size();
}

int san_wgrad_defer_flush(void* stream) {
    std::vector<RJob> jobs;
    {
        std::lock_guard<std::mutex> lk(g_defer_mu);
        jobs.swap(g_defer);
    }
    for (size_t at = 0; at < jobs.size(); at += kRBatch) {
        RBatch b{};
        b.n = (int)(jobs.size() - at < (size_t)kRBatch ? jobs.size() - at : (size_t)kRBatch);
        int total = 0;
        for (int k = 0; k < b.n; ++k) {
            b.j[k] = jobs[at + k];
            b.first[k] = total;
            total += san_cdiv(wgrad_reduce_units(b.j[k]), 64);
        }
        b.first[b.n] = total;
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, b);
        SAN_LAUNCH_CHECK();
    }
    return SAN_OK;
}

}  // extern "C"
