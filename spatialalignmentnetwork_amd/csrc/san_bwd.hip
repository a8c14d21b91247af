// Backward building blocks for the conv / norm stack (gfx950).
//
//   * data gradient of a convolution  = the forward MFMA kernel (san_conv_mfma.hip) run on dy
//     with flipped + transposed weights (san_conv_pack_weights_dgrad): no new conv code;
//   * weight gradient                 = conv_wgrad_kernel below: the 4x4x1 MFMA with the PIXEL
//     axis as the sixteen independent blocks (each block accumulates the 4x4 outer product
//     dy[4 co] x a[4 ci] of its own pixel; the blocks are summed once at the end), per-partition
//     partials + a deterministic second-stage reduction (no float atomics);
//   * InstanceNorm + LeakyReLU backward through the lazy-normalisation representation:
//     bwd_stats (two plane reductions) + act_bwd (element-wise), see the formulas at the kernels.
#include "san_common.h"
#include <cstdint>
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int kThreads = 256;
typedef float f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ wgrad
// dW[co][ci][tap] = sum over pixels of dy[co][p] * act(x)[ci][p + tap]: a GEMM whose reduction axis
// is the PIXEL axis (819,200 long at 320x320 x 8) and whose output is tiny (18 x 18 x 9 at level 0).
// One workgroup = 4 waves; wave w owns input-channel quad w of the block's (up to) 16 input channels
// and ALL X output-channel quads of the block, i.e. X*TAPS accumulator quads, so each staged input
// element feeds X MFMAs and each dy element feeds TAPS MFMAs.  A pixel tile of th x tw pixels
// (th*tw <= 256, geometry chosen per layer on the host so that narrow images fill the 16-pixel
// groups) is staged through LDS with the lazy activation applied; per group of 16 linearised
// pixels a wave reads X + TAPS operands and issues X*TAPS 4x4x1 MFMAs (block b = pixel b).
constexpr int kCIB = 16;       // input channels per workgroup (one quad per wave)
__device__ __forceinline__ int san_cdiv_dev(int a, int b) { return (a + b - 1) / b; }
constexpr int kVT = 512;       // threads of the pipelined kernel: 8 waves, 2 per SIMD
constexpr int kDCS = 264;      // LDS channel stride of the dy tile (== 8 mod 32: the four channels x eight pixels
                               // of a 32-lane ds_read_b32 group land on 32 different banks)

struct WgradArgs {
    const float* x;          // forward input (raw) + its lazy affine
    const float* in_scale;
    const float* in_shift;
    const float* dy;         // gradient wrt the conv output (materialised)
    float* partial;          // [P][cout][cin][taps]
    float in_slope;
    int x_ctot, x_coff, cin;
    int dy_ctot, dy_coff, cout;
    int N, H, W;
    int tw, th, aw, a_count, npx, groups;    // tile geometry: th x tw pixels, halo row pitch aw, 16-pixel groups
    int tiles_x, tiles_y, P;
    int cib_q;               // generic kernel: input-channel quads per workgroup (<= 4; balanced over the z blocks)
    int P_rem;               // pipelined kernel: pixel partitions of the remainder input-channel block (<= P)
    int co_blocks, n_full;   // pipelined kernel: 1-D grid = co_blocks * (n_full * P + P_rem) workgroups
};

struct WgradPlan {
    int vec;                 // 16-byte staged, pipelined kernel (W % 4 == 0) or the generic scalar one
    int tw, th, aw, a_count, npx, groups, tiles_x, tiles_y;
    int X, co_blocks, cib_q, ci_blocks, P, P_rem;
};

template <int KS, int X>
__global__ void __launch_bounds__(kThreads) conv_wgrad_kernel(const WgradArgs a) {
    constexpr int PAD = KS / 2;
    constexpr int TAPS = KS * KS;
    constexpr int ACS = KS == 3 ? 456 : 264;         // LDS channel stride of the input tile (== 8 mod 32)
    constexpr int ASLOTS = KS == 3 ? 2 : 1;          // staged input elements per thread per channel
    constexpr int CO = 4 * X;
    __shared__ float aT[kCIB * ACS];
    __shared__ float dT[CO * kDCS];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int b = lane >> 2, q = lane & 3;
    const int co0 = blockIdx.y * CO;
    const int ci0 = blockIdx.z * a.cib_q * 4;
    const int nstage = min(4 * a.cib_q, kCIB);       // channels this block stages (multiple of 4)
    const bool wave_on = wave < a.cib_q && ci0 + 4 * wave < a.cin;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;

    f4 acc[X][TAPS];
#pragma unroll
    for (int c = 0; c < X; ++c)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[c][t] = f4{0.f, 0.f, 0.f, 0.f};

    // per-thread staging slots (tile-relative, computed once): element e of the dense halo tile
    int ar[ASLOTS], ac[ASLOTS];
    bool a_in[ASLOTS];
#pragma unroll
    for (int s = 0; s < ASLOTS; ++s) {
        const int e = tid + s * kThreads;
        a_in[s] = e < a.a_count;
        ar[s] = a_in[s] ? e / a.aw : 0;
        ac[s] = a_in[s] ? e - ar[s] * a.aw : 0;
    }
    const bool d_in = tid < a.npx;
    const int dr = d_in ? tid / a.tw : 0, dc = d_in ? tid - dr * a.tw : 0;

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int total_tiles = tiles_per_img * a.N;
    float sa[kCIB][ASLOTS], sd[CO];
    bool a_ok[ASLOTS];
    bool d_ok = false;
    int tn = 0;
    // global -> registers for one tile (loads are unconditional on clamped addresses)
    auto prefetch = [&](int tile) {
        const int n = tile / tiles_per_img;
        const int tr = tile - n * tiles_per_img;
        const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        const int x0 = tx * a.tw, y0 = ty * a.th;
        tn = n;
        int aoff[ASLOTS];
#pragma unroll
        for (int s = 0; s < ASLOTS; ++s) {
            const int gy = y0 - PAD + ar[s], gx = x0 - PAD + ac[s];
            a_ok[s] = a_in[s] && gy >= 0 && gy < H && gx >= 0 && gx < W;
            aoff[s] = a_ok[s] ? gy * W + gx : 0;
        }
#pragma unroll
        for (int c = 0; c < kCIB; ++c)
            if (c < nstage) {
                const int ci = min(ci0 + c, a.cin - 1);
                const float* src = a.x + (size_t)(n * a.x_ctot + a.x_coff + ci) * HW;
#pragma unroll
                for (int s = 0; s < ASLOTS; ++s) sa[c][s] = src[aoff[s]];
            }
        const int gy = y0 + dr, gx = x0 + dc;
        d_ok = d_in && gy < H && gx < W;
        const int doff = d_ok ? gy * W + gx : 0;
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const int co = min(co0 + c, a.cout - 1);
            sd[c] = a.dy[(size_t)(n * a.dy_ctot + a.dy_coff + co) * HW + doff];
        }
    };

    const float* ab = aT + (4 * wave + q) * ACS;
    const float* db = dT + q * kDCS + b;
    if ((int)blockIdx.x < total_tiles) prefetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < total_tiles; tile += a.P) {
        __syncthreads();
        // ---- registers -> LDS with the lazy activation; zero outside the image / channel range
#pragma unroll
        for (int c = 0; c < kCIB; ++c)
            if (c < nstage) {
                const bool ch_ok = (ci0 + c) < a.cin;
                float sc = 1.f, sh = 0.f;
                if (a.in_scale) {
                    sc = a.in_scale[tn * a.x_ctot + a.x_coff + min(ci0 + c, a.cin - 1)];
                    sh = a.in_shift[tn * a.x_ctot + a.x_coff + min(ci0 + c, a.cin - 1)];
                }
#pragma unroll
                for (int s = 0; s < ASLOTS; ++s)
                    if (a_in[s])
                        aT[c * ACS + tid + s * kThreads] = (ch_ok && a_ok[s]) ? san_act(sa[c][s], sc, sh, a.in_slope) : 0.f;
            }
#pragma unroll
        for (int c = 0; c < CO; ++c) dT[c * kDCS + tid] = (d_ok && (co0 + c) < a.cout) ? sd[c] : 0.f;
        __syncthreads();
        if (tile + a.P < total_tiles) prefetch(tile + a.P);     // next tile's loads fly during the MFMAs
        if (wave_on) {
            // ---- groups of 16 linearised tile pixels: block b of the MFMA = pixel 16 g + b.  Operands of
            // group g+1 are read from LDS before the X*TAPS MFMAs of group g are issued.
            float dcur[X], bcur[TAPS];
            // lane's pixel of group g: LDS offset off = y*aw + x of the halo tile, advanced incrementally
            const int p0 = min(b, a.npx - 1);
            int px = p0 % a.tw;
            int off = (p0 / a.tw) * a.aw + px;
            const int off_last = (a.th - 1) * a.aw + a.tw - 1;     // pad pixels of the last group: dy is 0 there
            auto load_group = [&](int g, int o, float (&dq)[X], float (&bq)[TAPS]) {
                const float* ap = ab + o;
#pragma unroll
                for (int c = 0; c < X; ++c) dq[c] = db[4 * c * kDCS + 16 * g];
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) bq[ky * KS + kx] = ap[ky * a.aw + kx];
            };
            float dalt[X], balt[TAPS];
            constexpr int NLD = X + TAPS;
            constexpr int MPL = (X * TAPS) / NLD > 0 ? (X * TAPS) / NLD : 1;
            const int wrap = a.aw - a.tw;
            auto advance = [&]() {
                px += 16;
                off += 16;
                if (a.tw >= 16) {
                    const bool w = px >= a.tw;
                    px -= w ? a.tw : 0;
                    off += w ? wrap : 0;
                } else {
                    while (px >= a.tw) {
                        px -= a.tw;
                        off += wrap;
                    }
                }
            };
            // one half step: LDS reads of group g+1 interleaved under the X*TAPS MFMAs of group g
            auto half = [&](int g, float (&dq)[X], float (&bq)[TAPS], float (&dn)[X], float (&bn)[TAPS]) {
                advance();
                load_group(min(g + 1, a.groups - 1), min(off, off_last), dn, bn);
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int c = 0; c < X; ++c)
                        acc[c][t] = __builtin_amdgcn_mfma_f32_4x4x1f32(dq[c], bq[t], acc[c][t], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MPL, 0);
                }
            };
            load_group(0, off, dcur, bcur);
#pragma unroll 1
            for (int g = 0; g < a.groups; g += 2) {
                half(g, dcur, bcur, dalt, balt);
                if (g + 1 < a.groups) half(g + 1, dalt, balt, dcur, bcur);
            }
        }
    }
    if (!wave_on) return;
    // ---- sum the 16 blocks: lanes 4b + j, b = 0..15, hold partial D[i][j].  Rotations by 4 and 8 lanes
    // inside each row of 16 (DPP), then the four rows through the crossbar.
    const int ci = ci0 + 4 * wave + q;          // lane (b = 0, j = q) owns column ci
#pragma unroll
    for (int c = 0; c < X; ++c)
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc[c][t][i];
                v += san_dpp_get<0x124, 0xf>(v);    // row_ror:4
                v += san_dpp_get<0x128, 0xf>(v);    // row_ror:8
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                acc[c][t][i] = v;
            }
    if (b == 0 && ci < a.cin) {
#pragma unroll
        for (int c = 0; c < X; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + 4 * c + i;
                if (co < a.cout) {
                    float* o = a.partial + (((size_t)blockIdx.x * a.cout + co) * a.cin + ci) * TAPS;
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) o[t] = acc[c][t][i];
                }
            }
    }
}

// Same arithmetic for the layers that matter (W % 4 == 0: every VarNet / alignment layer), built
// around one measured fact: a wave issues MFMA, VALU, LDS and memory instructions strictly one
// after the other (scratch/probe/mfma_mix.hip: 8.7 cycles per 4x4x1 MFMA alone, 17.5 with 14 LDS
// reads + 30 VALU per 45 MFMAs), so the staging work only overlaps the matrix work when ANOTHER
// wave on the same SIMD supplies it.  Hence 8 waves = 2 per SIMD: wave w owns input-channel quad
// w & 3 and output-channel part w >> 2 (X quads), so the two waves of a SIMD share the input
// operand rows and split the output channels.  Staging uses 16-byte accesses and is software
// pipelined: LDS is double buffered; while the MFMAs of tile t run out of buffer t&1, staging
// chunk k runs next to pixel group 2k: it writes the register-held float4 of tile t+1
// (activation applied) into the other buffer and issues the global load of tile t+2 into the
// same registers.  One barrier per tile; global latency has a whole tile of matrix work to hide
// under.  The halo tile starts 4 columns left of the pixel tile so every row is 16-byte aligned.
template <int KS, int X>
__global__ void __launch_bounds__(kVT) conv_wgrad_vec_kernel(const WgradArgs a) {
    constexpr int PAD = KS / 2;
    constexpr int XO = KS == 3 ? 4 : 0;              // halo columns left of the tile (aligned)
    constexpr int TAPS = KS * KS;
    constexpr int ACS = KS == 3 ? 456 : 264;
    constexpr int NA = KS == 3 ? 4 : 2;              // float4 slots per thread, input tile (16 ch x <= 114 / 64)
    constexpr int ND = X;                            // float4 slots per thread, dy tile (8X ch x <= 64)
    constexpr int CO = 8 * X;                        // output channels per workgroup (two parts of X quads)
    constexpr int BUF = kCIB * ACS + CO * kDCS;      // floats per LDS buffer
    constexpr int MAXG = 16;
    static_assert(NA + ND <= MAXG, "one staging chunk per pixel group");
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int b = lane >> 2, q = lane & 3;
    // Input-channel blocks are 4 quads (16 channels) plus one remainder block.  A block with nci
    // quads gives every quad 8 / ncip waves (ncip = nci rounded up to 1, 2 or 4), which split the
    // block's output-channel quads between them: a remainder block of one quad (cin = 18, 36)
    // still runs MFMAs on all four SIMDs instead of one.  Waves w and w + 4 share a SIMD.
    // The remainder block is cheaper per tile, so it gets fewer pixel partitions (P_rem <= P); the
    // grid is 1-D with exactly co_blocks * (n_full * P + P_rem) workgroups (never more than the
    // 256 CUs: a second dispatch round costs 20-50 % here).
    int px, by, bz, Pb;
    {
        // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs (id % 8).  Give each XCD a
        // contiguous run of the logical order below, in which the channel block varies fastest, so the
        // workgroups that share a pixel partition's x tiles (same input block) and dy tiles (same output
        // block) run on the same XCD and hit in its L2 instead of each refilling it from the fabric.
        const int total = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3;
        const int L = xcd * (total >> 3) + min(xcd, total & 7) + slot;
        const int nblk = a.n_full * a.co_blocks, full_ids = nblk * a.P;
        if (L < full_ids) {
            px = L / nblk;
            const int blk = L - px * nblk;
            bz = blk / a.co_blocks;
            by = blk - bz * a.co_blocks;
            Pb = a.P;
        } else {
            const int r = L - full_ids;
            bz = a.n_full;
            px = r / a.co_blocks;
            by = r - px * a.co_blocks;
            Pb = a.P_rem;
        }
    }
    const int co0 = by * CO;
    const int ci0 = bz * kCIB;
    const int nci = min(4, san_cdiv_dev(a.cin, 4) - 4 * bz);
    const int ncip = nci > 2 ? 4 : nci;
    const int ncp = 8 / ncip;
    const int ciq = wave & (ncip - 1);
    // co part of this wave; the second half of the parts is taken in reverse so that the two waves of
    // a SIMD (w, w + 4) get part sizes that add up evenly
    const int cop_raw = wave / ncip;
    const int cop = (ncp >= 4 && cop_raw >= ncp / 2) ? ncp - 1 - (cop_raw - ncp / 2) : cop_raw;
    const int nstage = 4 * nci;
    const int qblk = min(2 * X, san_cdiv_dev(a.cout, 4) - 2 * X * by);
    const int qfirst = (cop * qblk) / ncp;
    const int nq = ciq < nci ? ((cop + 1) * qblk) / ncp - qfirst : 0;       // 0: this wave only stages
    const int H = a.H, W = a.W;
    const int HW = H * W;

    for (int i = tid; i < 2 * BUF + 4 * kVT; i += kVT) lds[i] = 0.f;     // pad pixels stay finite

    // ---- tile-invariant staging slots.  Every slot loads and stores unconditionally (branch-free
    // chunks keep the staging code inside the MFMA basic blocks): slots past the staged range
    // store into a per-thread trash line, channels past cin / cout store zeros in place.
    const int aw4 = a.aw >> 2, nv4 = a.a_count >> 2, tw4 = a.tw >> 2, nd4 = a.npx >> 2;
    int a_lds[NA], a_rel[NA], a_rc[NA], a_ch[NA];
    bool a_valid[NA];
#pragma unroll
    for (int s = 0; s < NA; ++s) {
        const int e = tid + s * kVT;
        const int c = e / nv4;
        const int v = e - c * nv4;
        const int row = v / aw4, col = 4 * (v - row * aw4);
        a_valid[s] = c < nstage && ci0 + c < a.cin;
        a_ch[s] = min(ci0 + c, a.cin - 1) - ci0;
        a_lds[s] = c < kCIB ? c * ACS + 4 * v : 2 * BUF + 4 * tid;
        a_rel[s] = a_ch[s] * HW + row * W + col;
        a_rc[s] = row | (col << 16);
    }
    int d_lds[ND], d_rel[ND], d_rc[ND];
    bool d_valid[ND];
#pragma unroll
    for (int s = 0; s < ND; ++s) {
        const int e = tid + s * kVT;
        const int c = e / nd4;
        const int v = e - c * nd4;
        const int row = v / tw4, col = 4 * (v - row * tw4);
        d_valid[s] = c < CO && co0 + c < a.cout;
        d_lds[s] = c < CO ? kCIB * ACS + c * kDCS + 4 * v : 2 * BUF + 4 * tid;
        d_rel[s] = (min(co0 + c, a.cout - 1) - co0) * HW + row * W + col;
        d_rc[s] = row | (col << 16);
    }

    // ---- this workgroup's contiguous run of tiles
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int total_tiles = tiles_per_img * a.N;
    const int t0 = (int)(((long long)px * total_tiles) / Pb);
    const int t1 = (int)(((long long)(px + 1) * total_tiles) / Pb);

    f4 sa[NA], sd[ND];
    float ssc[NA], ssh[NA];
    bool a_ok[NA], d_ok[ND];
    const bool has_aff = a.in_scale != nullptr;
    const float* scp = has_aff ? a.in_scale : a.x;
    const float* shp = has_aff ? a.in_shift : a.x;
    // tile coordinates of the tile being loaded (wave-uniform)
    int ln = 0, ly0 = 0, lx0 = 0;
    auto set_tile = [&](int tile) {
        const int n = tile / tiles_per_img;
        const int tr = tile - n * tiles_per_img;
        const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
        ln = n;
        ly0 = ty * a.th;
        lx0 = tx * a.tw;
    };
    auto load_a = [&](int s) {
        const int row = a_rc[s] & 0xffff, col = a_rc[s] >> 16;
        const int gy = ly0 - PAD + row, gx = lx0 - XO + col;
        a_ok[s] = a_valid[s] & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
        const float* base = a.x + (size_t)(ln * a.x_ctot + a.x_coff + ci0) * HW;
        const int off = a_ok[s] ? a_rel[s] + (ly0 - PAD) * W + lx0 - XO : 0;
        sa[s] = *reinterpret_cast<const f4*>(base + off);
        const int k = has_aff ? ln * a.x_ctot + a.x_coff + ci0 + a_ch[s] : 0;
        const float lsc = scp[k], lsh = shp[k];
        // out-of-image / out-of-range elements become act(0*x + 0) = 0: no select at store time
        ssc[s] = a_ok[s] ? (has_aff ? lsc : 1.f) : 0.f;
        ssh[s] = a_ok[s] ? (has_aff ? lsh : 0.f) : 0.f;
    };
    auto load_d = [&](int s) {
        const int row = d_rc[s] & 0xffff, col = d_rc[s] >> 16;
        const int gy = ly0 + row, gx = lx0 + col;
        d_ok[s] = d_valid[s] & (gy < H) & (gx < W);
        const float* base = a.dy + (size_t)(ln * a.dy_ctot + a.dy_coff + co0) * HW;
        const int off = d_ok[s] ? d_rel[s] + ly0 * W + lx0 : 0;
        sd[s] = *reinterpret_cast<const f4*>(base + off);
    };
    auto write_a = [&](int s, float* buf0, int bufoff) {
        f4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = fmaf(sa[s][i], ssc[s], ssh[s]);
            r[i] = fmaxf(v, v * a.in_slope);            // LeakyReLU for 0 <= slope <= 1 (host-checked)
        }
        *reinterpret_cast<f4*>(buf0 + (a_lds[s] < 2 * BUF ? a_lds[s] + bufoff : a_lds[s])) = r;
    };
    auto write_d = [&](int s, float* buf0, int bufoff) {
        const f4 r = d_ok[s] ? sd[s] : f4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f4*>(buf0 + (d_lds[s] < 2 * BUF ? d_lds[s] + bufoff : d_lds[s])) = r;
    };
    // staging chunk g: NA input slots first, then ND dy slots; store the held tile, fetch the next
    auto chunk = [&](int g, int bufoff) {
        if (g < NA) {
            write_a(g, lds, bufoff);
            load_a(g);
        } else if (g < NA + ND) {
            write_d(g - NA, lds, bufoff);
            load_d(g - NA);
        }
    };

    // ---- prologue: tile t0 -> buffer 0, tile t0+1 -> registers
    set_tile(t0);
#pragma unroll
    for (int s = 0; s < NA; ++s) load_a(s);
#pragma unroll
    for (int s = 0; s < ND; ++s) load_d(s);
    __syncthreads();                                   // zero fill done before the first writes
    set_tile(min(t0 + 1, t1 - 1));
#pragma unroll
    for (int g = 0; g < NA + ND; ++g) chunk(g, 0);
    __syncthreads();

    // ---- main loop.  The pixel tile is 16 x 16: group g = tile row g, MFMA block b = column b, so
    // consecutive groups slide the 3-row input window down by one row: only the new row (KS
    // operands) and the XE dy operands are read from LDS per group, all at immediate offsets; the
    // 16 groups are one straight-line block.  XE = the wave's own number of output-channel quads
    // (0..X, wave-uniform): a wave never issues MFMAs for quads it does not own, because they
    // would come straight out of its SIMD partner's matrix-pipe time.
    constexpr int AW = 16 + 2 * XO;
    auto run = [&](auto xe_tag) {
        constexpr int XE = decltype(xe_tag)::value;
        constexpr int XR = XE > 0 ? XE : 1;
        f4 acc[XR][TAPS];
#pragma unroll
        for (int c = 0; c < XR; ++c)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[c][t] = f4{0.f, 0.f, 0.f, 0.f};
        for (int tile = t0; tile < t1; ++tile) {
            const float* cur = lds + ((tile - t0) & 1) * BUF;
            const int nxt = (((tile - t0) & 1) ^ 1) * BUF;       // the held tile (tile+1) goes to the other buffer
            set_tile(min(tile + 2, t1 - 1));                     // past the run: a redundant reload, never consumed
            if constexpr (XE == 0) {
#pragma unroll
                for (int k = 0; k < NA + ND; ++k) chunk(k, nxt);
            } else {
                const float* ab = cur + (4 * ciq + q) * ACS + b + XO - PAD;
                const float* db = cur + kCIB * ACS + (4 * qfirst + q) * kDCS + b;
                float bw[KS][KS];              // input window row r lives in bw[r % KS]
                float dq[2][XE];
#pragma unroll
                for (int ky = 0; ky < KS - 1; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) bw[ky][kx] = ab[ky * AW + kx];
#pragma unroll
                for (int c = 0; c < XE; ++c) dq[0][c] = db[4 * c * kDCS];
#pragma unroll
                for (int g = 0; g < MAXG; ++g) {
                    if (g < a.th) {                     // tile rows (wave-uniform)
                        // operands of this group's last window row, and the next group's dy
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) bw[(g + KS - 1) % KS][kx] = ab[(g + KS - 1) * AW + kx];
                        if (g + 1 < MAXG) {
#pragma unroll
                            for (int c = 0; c < XE; ++c) dq[(g + 1) & 1][c] = db[4 * c * kDCS + 16 * (g + 1)];
                        }
                        // Solid MFMA block: switching between matrix and vector issue costs ~4 extra cycles per
                        // switch on gfx950 (scratch/probe/mfma_il.hip: 45 MFMA + 45 VALU runs at 113 TF blocked,
                        // 82 TF interleaved 1:1), so the staging VALU work below must not be woven in.
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                            for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                                for (int c = 0; c < XE; ++c)
                                    acc[c][ky * KS + kx] = __builtin_amdgcn_mfma_f32_4x4x1f32(
                                        dq[g & 1][c], bw[(g + ky) % KS][kx], acc[c][ky * KS + kx], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (g < NA + ND) chunk(g, nxt);
                }
            }
            __syncthreads();
        }
        if constexpr (XE > 0) {
            // sum the 16 blocks: lanes 4b + j hold D_b[i][j]; rotations inside each row of 16, then the four rows
            const int ci = ci0 + 4 * ciq + q;
#pragma unroll
            for (int c = 0; c < XE; ++c)
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[c][t][i];
                        v += san_dpp_get<0x124, 0xf>(v);    // row_ror:4
                        v += san_dpp_get<0x128, 0xf>(v);    // row_ror:8
                        v += __shfl_xor(v, 16, 64);
                        v += __shfl_xor(v, 32, 64);
                        acc[c][t][i] = v;
                    }
            if (b == 0 && ci < a.cin) {
#pragma unroll
                for (int c = 0; c < XE; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int co = co0 + 4 * (qfirst + c) + i;
                        if (co < a.cout) {
                            float* o = a.partial + (((size_t)px * a.cout + co) * a.cin + ci) * TAPS;
#pragma unroll
                            for (int t = 0; t < TAPS; ++t) o[t] = acc[c][t][i];
                        }
                    }
            }
        }
    };
    if (nq <= 0) {
        run(std::integral_constant<int, 0>{});
    } else if (nq == 1) {
        run(std::integral_constant<int, 1>{});
    } else if (nq == 2) {
        if constexpr (X >= 2) run(std::integral_constant<int, 2>{});
    } else if (nq == 3) {
        if constexpr (X >= 3) run(std::integral_constant<int, 3>{});
    } else {
        if constexpr (X >= 4) run(std::integral_constant<int, 4>{});
    }
}

// dW[i] (+)= sum_p partial[p][i] in a fixed order (deterministic): four lanes share one output
// element, lane k adds partitions p = k, k+4, ... with four independent loads in flight, and the
// four partial sums are combined in the order ((0+1)+(2+3)).  (A single thread walking up to 256
// partitions one dependent load at a time made this step cost as much as a small convolution.)
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int count, int P, int accumulate,
                    int cin_taps, int split, int P_rem) {
    const int k = threadIdx.x & 3;
    for (int i = blockIdx.x * 64 + (threadIdx.x >> 2); i < ((count + 63) & ~63); i += gridDim.x * 64) {
        float s = 0.f;
        if (i < count) {
            // input-channel columns >= split belong to the remainder block, which wrote P_rem partitions
            const int Pi = (i % cin_taps) >= split ? P_rem : P;
            const float* src = partial + i;
            int p = k;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (; p + 12 < Pi; p += 16) {
                s0 += src[(size_t)p * count];
                s1 += src[(size_t)(p + 4) * count];
                s2 += src[(size_t)(p + 8) * count];
                s3 += src[(size_t)(p + 12) * count];
            }
            for (; p < Pi; p += 4) s0 += src[(size_t)p * count];
            s = (s0 + s1) + (s2 + s3);
        }
        // lanes 4e .. 4e+3 hold the four strided sums of element e
        const float a = s + san_dpp_get<0xB1, 0xf>(s);      // quad_perm [1,0,3,2]: (0+1), (2+3)
        const float t = a + san_dpp_get<0x4E, 0xf>(a);      // quad_perm [2,3,0,1]: ((0+1)+(2+3))
        if (k == 0 && i < count) dw[i] = accumulate ? dw[i] + t : t;
    }
}

// Tile geometry, channel blocking and pixel partitions of one weight-gradient launch.
WgradPlan wgrad_plan(int n, int h, int w, int cin, int cout, int ks, bool allow_vec) {
    WgradPlan p{};
    const int pad = ks / 2;
    const int amax = ks == 3 ? 456 : 256;
    p.vec = allow_vec && (w % 4 == 0);
    const int xpad = p.vec ? (ks == 3 ? 4 : 0) : pad;      // the vector kernel's halo is 4 columns wide (alignment)
    if (p.vec) {                                           // 16 columns x up to 16 rows, rows balanced over the image
        p.tw = 16;
        p.th = san_cdiv(h, san_cdiv(h, 16));
    }
    // pixel tile: minimise  tiles * (groups + staging cost in group units)
    long best = -1;
    for (int k = 1; k <= w && !p.vec; ++k) {
        const int tw = san_cdiv(w, k);
        if (tw > 256) continue;
        for (int th = 1; th <= h && th * tw <= 256; ++th) {
            if ((th + 2 * pad) * (tw + 2 * xpad) > amax) break;
            const int groups = san_cdiv(th * tw, 16);
            const long cost = (long)san_cdiv(w, tw) * san_cdiv(h, th) * (groups + 4);
            if (best < 0 || cost < best) {
                best = cost;
                p.tw = tw;
                p.th = th;
            }
        }
    }
    p.aw = p.tw + 2 * xpad;
    p.a_count = (p.th + 2 * pad) * p.aw;
    p.npx = p.th * p.tw;
    p.groups = san_cdiv(p.npx, 16);
    p.tiles_x = san_cdiv(w, p.tw);
    p.tiles_y = san_cdiv(h, p.th);
    // output-channel quads per wave: fewest (blocks * (X + fixed per-group cost)).  The pipelined
    // kernel covers 2X quads per workgroup (two co parts) under a 256-register budget: X <= 4.
    const int qco = san_cdiv(cout, 4);
    static const int xs_scalar[4] = {2, 3, 5, 6}, xs_vec[4] = {1, 2, 3, 4};
    const int* xs = p.vec ? xs_vec : xs_scalar;
    const int nx = p.vec && ks == 3 ? 3 : 4;         // 3x3: X = 4 would spill (36 accumulator quads + staging)
    const int span = p.vec ? 2 : 1;
    float bx = 0.f;
    for (int i = 0; i < nx; ++i) {
        const float c = san_cdiv(qco, span * xs[i]) * (xs[i] + (p.vec ? 0.75f : 1.5f));
        if (p.X == 0 || c < bx) {
            bx = c;
            p.X = xs[i];
        }
    }
    p.co_blocks = san_cdiv(qco, span * p.X);
    const int qci = san_cdiv(cin, 4);
    p.ci_blocks = san_cdiv(qci, 4);
    p.cib_q = san_cdiv(qci, p.ci_blocks);
    const int tiles = p.tiles_x * p.tiles_y * n;
    int P;
    p.P_rem = 0;
    if (p.vec) {
        // One workgroup per CU (LDS).  Full input-channel blocks keep a SIMD busy for qblk quads per pixel
        // group, the remainder block (qci % 4 quads) for fewer (see the kernel): split the 256 workgroup
        // slots between them in proportion to that work (+1.5 quads of per-tile staging overhead).
        const int qblk = qco < span * p.X ? qco : span * p.X;
        const int rem = qci % 4;
        const int n_full = qci / 4;
        const float w_full = qblk + 1.5f;
        const float w_rem = (rem == 0 ? 0.f : rem == 1 ? (qblk + 3) / 4 : rem == 2 ? (qblk + 1) / 2 : qblk) + (rem ? 1.5f : 0.f);
        const float slots = 256.f / p.co_blocks;
        const float unit = slots / (n_full * w_full + w_rem);
        P = n_full ? (int)(unit * w_full) : (int)(unit * w_rem);
        if (P < 1) P = 1;
        if (rem) {
            int pr = n_full ? (int)(slots - (float)n_full * P) : P;
            if (pr < 1) pr = 1;
            if (pr > P) pr = P;
            p.P_rem = pr;
        }
    } else {
        P = san_cdiv(768, p.co_blocks * p.ci_blocks);
    }
    if (P > tiles) P = tiles;
    if (P > 256) P = 256;
    if (P < 1) P = 1;
    p.P = P;
    if (p.P_rem > p.P || p.P_rem < 1) p.P_rem = p.P;
    return p;
}

// ------------------------------------------------- norm + activation backward
// Forward (lazy):  yh = sc*y + sh,  a = lrelu(yh, slope).   Given G = dL/da:
//   u  = G * (yh >= 0 ? 1 : slope)
//   mode 0 (plain affine, e.g. eval BatchNorm / identity): dy = sc * u
//   mode 1 (InstanceNorm, biased var):  dy = sc * (u - mean(u) - yh * mean(u*yh))   per (n, c) plane
// bwd_stats writes per-chunk (sum u, sum u*yh) partials; act_bwd sums the chunks itself.
typedef float bf4 __attribute__((ext_vector_type(4)));

// Largest |dy| of a launch, for the fp16-format gradient kernels (they scale dy by a power of two derived from it): every
// workgroup folds its maximum into the tensor's amax record (san_common.h: san_amax_record, 64 atomic lines, no finalising launch).

// Optional second source of the incoming gradient: g(p) += scale * g2(p / 2) with g2 a [n, c2, h/2, w/2] view -- the
// encoder levels of the U-Net backward, where dL/d(block output) = skip-connection gradient + the average pool's adjoint
// (x 0.25, nearest 2 x 2) of the pooled gradient.  Read here, the up-sampled tensor and the sum are never written.
struct G2Src {
    const float* p;        // null: no second source
    int ctot, coff;
    float scale;
    int w4;                // plane width / 4 (width % 4 == 0, height even)
    float inv_w4;
};
// the four values at flattened pixels 4 i .. 4 i + 3 of the (h x w) plane whose half-resolution plane starts at q
__device__ __forceinline__ bf4 g2_add(bf4 gv, const float* __restrict__ q, int i, int w4, float inv_w4, float scale) {
    const int row = (int)(((float)i + 0.5f) * inv_w4);              // exact for i < 2^22
    const int c4 = i - row * w4;
    const float2 v = *reinterpret_cast<const float2*>(q + (size_t)(row >> 1) * (2 * w4) + 2 * c4);
    gv[0] = fmaf(scale, v.x, gv[0]);
    gv[1] = fmaf(scale, v.x, gv[1]);
    gv[2] = fmaf(scale, v.y, gv[2]);
    gv[3] = fmaf(scale, v.y, gv[3]);
    return gv;
}

// Where dy goes.  shuf: the pixel un-shuffle of the transposed convolution's backward folded into the store -- dy of the
// [n, C, h, w] activation lands as [n, 4 C, h/2, w/2] with channel 4 c + 2 (row & 1) + (col & 1) (what san_unshuffle2_fwd
// produced in a second pass over the tensor): a lane's four consecutive pixels are two pairs of neighbouring positions in
// two destination planes, i.e. two 8-byte stores.  acc: add to what dy holds (gradient accumulation across cascades).
struct DyDst {
    int shuf, acc;
    int w4;                // plane width / 4
    float inv_w4;
};
__device__ __forceinline__ void dy_store(float* __restrict__ base, int i, bf4 o, const DyDst d) {
    // base: the (n, channel) plane of dy in the plain form; the (n, 4 * channel) plane in the shuffled form
    if (!d.shuf) {
        bf4* q = reinterpret_cast<bf4*>(base) + i;
        if (d.acc) o += *q;
        *q = o;
        return;
    }
    typedef float bf2 __attribute__((ext_vector_type(2)));
    const int row = (int)(((float)i + 0.5f) * d.inv_w4);            // exact for i < 2^22
    const int c4 = i - row * d.w4;                                  // pixels 4 c4 .. 4 c4 + 3 of image row `row`
    // half-resolution plane: width 2 w4, this row's positions 2 c4, 2 c4 + 1; channel offset 2 (row & 1) + dx
    float* q = base + (size_t)(row >> 1) * (2 * d.w4) + 2 * c4;
    bf2 e = bf2{o[0], o[2]}, f = bf2{o[1], o[3]};
    float* qe = q + (size_t)(2 * (row & 1)) * d.shuf;               // d.shuf = elements per destination plane (h/2 * w/2)
    float* qf = qe + d.shuf;
    if (d.acc) {
        e += *reinterpret_cast<bf2*>(qe);
        f += *reinterpret_cast<bf2*>(qf);
    }
    *reinterpret_cast<bf2*>(qe) = e;
    *reinterpret_cast<bf2*>(qf) = f;
}

// Loads of one thread's V float4 of g and y (+ the optional half-resolution second source), ALL issued before anything waits for
// one of them: with `if (i < n4)` around each load the compiler emitted load, s_waitcnt vmcnt(0), arithmetic V times over -- two
// requests in flight per thread (round 6; the 160 x 160 plane kernel ran at half the rate the same bytes stream at).  Indices
// past the plane are clamped to its last element (a valid address) and their values dropped afterwards.
template <int V, int T>
__device__ __forceinline__ void act_bwd_load(const bf4* __restrict__ gp, const bf4* __restrict__ yp, const float* __restrict__ q2,
                                             const G2Src g2, int base, int n4, float s, float b, float slope, bf4 (&u)[V],
                                             bf4 (&yh)[V], float& s1, float& s2, int tix = -1) {
    const int tx = tix >= 0 ? tix : (int)threadIdx.x;   // index within the T threads that share the plane (chunk)
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int i = min(base + tx + T * k, n4 - 1);
        u[k] = gp[i];
        yh[k] = yp[i];
    }
    if (q2) {                                       // (uniform: a kernel argument)
        typedef float bf2 __attribute__((ext_vector_type(2)));
        bf2 v2[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int i = min(base + tx + T * k, n4 - 1);
            const int row = (int)(((float)i + 0.5f) * g2.inv_w4);           // exact for i < 2^22
            const int c4 = i - row * g2.w4;
            v2[k] = *reinterpret_cast<const bf2*>(q2 + (size_t)(row >> 1) * (2 * g2.w4) + 2 * c4);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            u[k][0] = fmaf(g2.scale, v2[k][0], u[k][0]);
            u[k][1] = fmaf(g2.scale, v2[k][0], u[k][1]);
            u[k][2] = fmaf(g2.scale, v2[k][1], u[k][2]);
            u[k][3] = fmaf(g2.scale, v2[k][1], u[k][3]);
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const bool in = base + tx + T * k < n4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = in ? fmaf(yh[k][e], s, b) : 0.f;
            const float uu = in ? u[k][e] * (t >= 0.f ? 1.f : slope) : 0.f;
            yh[k][e] = t;
            u[k][e] = uu;
            s1 += uu;
            s2 = fmaf(uu, t, s2);
        }
    }
}

// InstanceNorm + LeakyReLU backward of a whole (sample, channel) plane in ONE pass: the plane's g and y (<= V float4 per
// thread each, 512 threads) stay in registers between the two reductions and the write of dy, so g and y are read once
// instead of twice and there is one launch instead of two.  Planes of up to 512 * 4 * V values (V = 13: 160 x 160).
// (SAN_NO_PK32: san_common.h -- the kernel this was found on)
template <int V>
__global__ void __launch_bounds__(512) SAN_NO_PK32 act_bwd_plane_kernel(const float* __restrict__ g, int g_ctot, int g_coff,
                                                            const float* __restrict__ y, int y_ctot, int y_coff,
                                                            const float* __restrict__ sc, const float* __restrict__ sh, float slope,
                                                            float* __restrict__ dy, int d_ctot, int d_coff, int hw, unsigned* amax,
                                                            const G2Src g2, const DyDst dd) {
    __shared__ float red[16];
    const int ch = blockIdx.x, n = blockIdx.y;
    const float* q2 = g2.p ? g2.p + ((size_t)(n * g2.ctot + g2.coff + ch)) * (hw >> 2) : nullptr;
    const float s = sc ? sc[n * y_ctot + y_coff + ch] : 1.f;
    const float b = sh ? sh[n * y_ctot + y_coff + ch] : 0.f;
    const bf4* gp = reinterpret_cast<const bf4*>(g + ((size_t)(n * g_ctot + g_coff + ch)) * hw);
    const bf4* yp = reinterpret_cast<const bf4*>(y + ((size_t)(n * y_ctot + y_coff + ch)) * hw);
    // (shuffled form: d_ctot / d_coff count the 4 C destination channels of hw / 4 elements each: the same byte offset)
    float* dp = dy + (dd.shuf ? ((size_t)(n * d_ctot + d_coff + 4 * ch)) * (hw >> 2) : ((size_t)(n * d_ctot + d_coff + ch)) * hw);
    const int n4 = hw >> 2;
    bf4 u[V], yh[V];
    float s1 = 0.f, s2 = 0.f;
    act_bwd_load<V, 512>(gp, yp, q2, g2, 0, n4, s, b, slope, u, yh, s1, s2);
    s1 = san_wave_total(s1);
    s2 = san_wave_total(s2);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = s1;
        red[8 + (threadIdx.x >> 6)] = s2;
    }
    __syncthreads();
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        t1 += (double)red[w];
        t2 += (double)red[8 + w];
    }
    const float m1 = (float)(t1 / hw), m2 = (float)(t2 / hw);
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int i = threadIdx.x + 512 * k;
        if (i < n4) {
            bf4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = s * (u[k][e] - m1 - yh[k][e] * m2);
                mx = fmaxf(mx, fabsf(o[e]));
            }
            dy_store(dp, i, o, dd);
        }
    }
    san_amax_record<8>(amax, blockIdx.y * gridDim.x + blockIdx.x, mx);
}

// ---- one-pass norm + activation backward of planes that do not fit one workgroup (round 6) ------------------------------------
// The two-kernel form reads g and y twice (bwd_stats, then act_bwd) because the plane means must exist before the first dy can
// be written: at 320 x 320 that is 5 plane passes and two launches per layer, 2.2 ms of a training step in bwd_stats alone.
// Here a CLUSTER of K workgroups shares one reduction domain (InstanceNorm: the K chunks of one (sample, channel) plane;
// BatchNorm: the n * K chunks of one channel): every workgroup keeps its chunk of u and yh in registers, publishes its two partial
// sums, waits for the other members and then writes dy -- g and y are read once, one launch.
//   record of a cluster (int32 words, zero between launches):  [2 * members] partial sums | arrived | departed
//   publish:  agent-scope (write-through) stores of the two floats, s_waitcnt vmcnt(0), relaxed agent-scope add on `arrived`
//   wait:     one lane polls `arrived` with agent-scope loads (+ s_sleep), then the members' sums are read with agent-scope loads
//             and added in member order in double -- every member computes the same bits, run to run
//   leave:    add on `departed`; the last one out zeroes the record (sums and counters) for the next launch
// Waiting on other workgroups of the same launch needs them to be scheduled: cluster members are CONSECUTIVE workgroups of the
// grid and the dispatcher hands workgroups out in order, so the members of the oldest unfinished cluster are all resident or
// finished -- it always completes and frees its slots (no cycle among waiting clusters).  The poll is bounded all the same: a
// member that gives up writes NaN means, loudly wrong instead of a hung GPU.
constexpr int kClT = 256;                           // threads per cluster member
constexpr int kClPollMax = 1 << 21;                 // ~ seconds
__host__ __device__ __forceinline__ int cluster_rec_words(int members) { return ((2 * members + 2 + 31) / 32) * 32; }      // whole 128-byte lines

template <int V, int BN>
__global__ void __launch_bounds__(kClT) SAN_NO_PK32
act_bwd_cluster_kernel(const float* __restrict__ g, int g_ctot, int g_coff, const float* __restrict__ y, int y_ctot, int y_coff,
                       const float* __restrict__ sc, const float* __restrict__ sh, float slope, float* __restrict__ dy, int d_ctot,
                       int d_coff, int hw, unsigned* amax, const G2Src g2, const DyDst dd, unsigned* __restrict__ sync,
                       const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dgamma,
                       float* __restrict__ dbeta, double cnt) {
    __shared__ float red[8];
    __shared__ float pay[2 * 256];                  // the members' partial sums (members <= 256)
    __shared__ double tot[2];
    // InstanceNorm: grid (K, c, n), cluster = (ch, n); BatchNorm: grid (K, n, c), cluster = ch -- members consecutive either way
    const int K = gridDim.x, chunk = blockIdx.x;
    const int ch = BN ? blockIdx.z : blockIdx.y, n = BN ? blockIdx.y : blockIdx.z;
    const int members = BN ? K * (int)gridDim.y : K;
    const int me = BN ? n * K + chunk : chunk;
    const int cluster = BN ? ch : n * (int)gridDim.y + ch;
    const float* q2 = g2.p ? g2.p + ((size_t)(n * g2.ctot + g2.coff + ch)) * (hw >> 2) : nullptr;
    const float s = sc ? sc[n * y_ctot + y_coff + ch] : 1.f;
    const float b = sh ? sh[n * y_ctot + y_coff + ch] : 0.f;
    const bf4* gp = reinterpret_cast<const bf4*>(g + ((size_t)(n * g_ctot + g_coff + ch)) * hw);
    const bf4* yp = reinterpret_cast<const bf4*>(y + ((size_t)(n * y_ctot + y_coff + ch)) * hw);
    float* dp = dy + (dd.shuf ? ((size_t)(n * d_ctot + d_coff + 4 * ch)) * (hw >> 2) : ((size_t)(n * d_ctot + d_coff + ch)) * hw);
    const int n4 = hw >> 2;
    const int base = chunk * (kClT * V);
    bf4 u[V], yh[V];
    float s1 = 0.f, s2 = 0.f;
    act_bwd_load<V, kClT>(gp, yp, q2, g2, base, n4, s, b, slope, u, yh, s1, s2);
    s1 = san_wave_total(s1);
    s2 = san_wave_total(s2);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = s1;
        red[4 + (threadIdx.x >> 6)] = s2;
    }
    __syncthreads();
    unsigned* rec = sync + (size_t)cluster * cluster_rec_words(members);
    if (threadIdx.x == 0) {
        const float p1 = (red[0] + red[1]) + (red[2] + red[3]), p2 = (red[4] + red[5]) + (red[6] + red[7]);
        __hip_atomic_store(rec + 2 * me, __builtin_bit_cast(unsigned, p1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rec + 2 * me + 1, __builtin_bit_cast(unsigned, p2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the sums are in memory before the arrival can be seen
        __hip_atomic_fetch_add(rec + 2 * members, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(rec + 2 * members, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)members) {
            if (++spins > kClPollMax) break;
            __builtin_amdgcn_s_sleep(2);
        }
        red[0] = spins > kClPollMax ? 1.f : 0.f;
    }
    __syncthreads();
    const bool gave_up = red[0] != 0.f;
    for (int j = threadIdx.x; j < 2 * members; j += kClT)
        pay[j] = __builtin_bit_cast(float, __hip_atomic_load(rec + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    if (threadIdx.x < 64) {                         // member order, double: identical in every member
        double t1 = 0.0, t2 = 0.0;
        if (members <= 64) {
            if (threadIdx.x == 0)
                for (int j = 0; j < members; ++j) {
                    t1 += (double)pay[2 * j];
                    t2 += (double)pay[2 * j + 1];
                }
        } else {
            for (int j = threadIdx.x; j < members; j += 64) {
                t1 += (double)pay[2 * j];
                t2 += (double)pay[2 * j + 1];
            }
            t1 = san_wave_sum_d(t1);
            t2 = san_wave_sum_d(t2);
        }
        // everything this member needs from the record has been read: leave; the last one out zeroes the WHOLE record (sums and
        // counters), so the buffer is all zeros between launches whatever cluster size the next user of it has
        unsigned left = 0u;
        if (threadIdx.x == 0) {
            tot[0] = t1;
            tot[1] = t2;
            left = __hip_atomic_fetch_add(rec + 2 * members + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        left = (unsigned)__shfl((int)left, 0, 64);
        if (left == (unsigned)members - 1u)
            for (int j = threadIdx.x; j < 2 * members + 2; j += 64)
                __hip_atomic_store(rec + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    float m1, m2, pp = 1.f, qq = 0.f;
    if (BN) {
        // BatchNorm (unet.py:125): dbeta = S1, dgamma = (S2 - beta S1) / gamma; dy = sc (u - dbeta/cnt - yn dgamma/cnt), yn = (yh - beta) / gamma
        const double ga = (double)gamma[ch], be = (double)beta[ch];
        const double dg = (tot[1] - be * tot[0]) / ga;
        m1 = (float)(tot[0] / cnt);
        m2 = (float)(dg / cnt);
        pp = (float)(1.0 / ga);
        qq = (float)(-be / ga);
        if (me == 0 && threadIdx.x == 0) {
            dgamma[ch] += (float)dg;
            dbeta[ch] += (float)tot[0];
        }
    } else {
        m1 = (float)(tot[0] / hw);
        m2 = (float)(tot[1] / hw);
    }
    if (gave_up) m1 = m2 = __builtin_nanf("");
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const int i = base + threadIdx.x + kClT * k;
        if (i < n4) {
            bf4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = BN ? s * (u[k][e] - m1 - fmaf(pp, yh[k][e], qq) * m2) : s * (u[k][e] - m1 - yh[k][e] * m2);
                mx = fmaxf(mx, fabsf(o[e]));
            }
            dy_store(dp, i, o, dd);
        }
    }
    san_amax_record<kClT / 64>(amax, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, mx);
}

__global__ void __launch_bounds__(kThreads) SAN_NO_PK32
bwd_stats_kernel(const float* __restrict__ g, int g_ctot, int g_coff, const float* __restrict__ y, int y_ctot, int y_coff,
                 const float* __restrict__ sc, const float* __restrict__ sh, float slope, int c, int hw, int tiles,
                 float* __restrict__ part, const G2Src g2) {
    __shared__ float red[8];
    const int t = blockIdx.x, ch = blockIdx.y, n = blockIdx.z;
    const float* q2 = g2.p ? g2.p + ((size_t)(n * g2.ctot + g2.coff + ch)) * (hw >> 2) : nullptr;
    const int chunk = (((hw + tiles - 1) / tiles) + 3) & ~3;       // multiples of 4: every chunk starts 16-byte aligned
    const int lo = t * chunk;
    const int cnt = max(0, min(hw, lo + chunk) - lo);
    const float* gp = g + ((size_t)(n * g_ctot + g_coff + ch)) * hw + lo;
    const float* yp = y + ((size_t)(n * y_ctot + y_coff + ch)) * hw + lo;
    const float s = sc ? sc[n * y_ctot + y_coff + ch] : 1.f;
    const float b = sh ? sh[n * y_ctot + y_coff + ch] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if ((((uintptr_t)gp | (uintptr_t)yp) & 15) == 0) {            // 16-byte loads (chunks of a 4-divisible plane are 4-divisible)
        const int c4 = cnt >> 2;
        for (int i = threadIdx.x; i < c4; i += kThreads) {
            bf4 gv = reinterpret_cast<const bf4*>(gp)[i];
            const bf4 yv = reinterpret_cast<const bf4*>(yp)[i];
            if (q2) gv = g2_add(gv, q2, (lo >> 2) + i, g2.w4, g2.inv_w4, g2.scale);       // (host: hw % 4 == 0 with g2)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float yh = fmaf(yv[e], s, b);
                const float u = gv[e] * (yh >= 0.f ? 1.f : slope);
                s1 += u;
                s2 = fmaf(u, yh, s2);
            }
        }
        for (int i = 4 * c4 + threadIdx.x; i < cnt; i += kThreads) {
            const float yh = fmaf(yp[i], s, b);
            const float u = gp[i] * (yh >= 0.f ? 1.f : slope);
            s1 += u;
            s2 = fmaf(u, yh, s2);
        }
    } else {
        for (int i = threadIdx.x; i < cnt; i += kThreads) {
            const float yh = fmaf(yp[i], s, b);
            const float u = gp[i] * (yh >= 0.f ? 1.f : slope);
            s1 += u;
            s2 = fmaf(u, yh, s2);
        }
    }
    s1 = san_wave_total(s1);
    s2 = san_wave_total(s2);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = s1;
        red[4 + (threadIdx.x >> 6)] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)(n * c + ch) * tiles + t) * 2;
        o[0] = (red[0] + red[1]) + (red[2] + red[3]);
        o[1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

__global__ void __launch_bounds__(kThreads) SAN_NO_PK32
act_bwd_kernel(const float* __restrict__ g, int g_ctot, int g_coff, const float* __restrict__ y, int y_ctot, int y_coff,
               const float* __restrict__ sc, const float* __restrict__ sh, float slope, const float* __restrict__ part,
               int tiles, int mode, float* __restrict__ dy, int d_ctot, int d_coff, int c, int hw, unsigned* amax,
               const G2Src g2, const DyDst dd) {
    const int ch = blockIdx.y, n = blockIdx.z;
    const float* q2 = g2.p ? g2.p + ((size_t)(n * g2.ctot + g2.coff + ch)) * (hw >> 2) : nullptr;
    const float s = sc ? sc[n * y_ctot + y_coff + ch] : 1.f;
    const float b = sh ? sh[n * y_ctot + y_coff + ch] : 0.f;
    float m1 = 0.f, m2 = 0.f;
    if (mode == 1) {
        double t1 = 0.0, t2 = 0.0;
        const float* p = part + ((size_t)(n * c + ch) * tiles) * 2;
        for (int t = 0; t < tiles; ++t) {
            t1 += (double)p[2 * t];
            t2 += (double)p[2 * t + 1];
        }
        m1 = (float)(t1 / hw);
        m2 = (float)(t2 / hw);
    }
    const float* gp = g + ((size_t)(n * g_ctot + g_coff + ch)) * hw;
    const float* yp = y + ((size_t)(n * y_ctot + y_coff + ch)) * hw;
    float* dp = dy + (dd.shuf ? ((size_t)(n * d_ctot + d_coff + 4 * ch)) * (hw >> 2) : ((size_t)(n * d_ctot + d_coff + ch)) * hw);
    float mx = 0.f;
    if ((hw & 3) == 0 && ((((uintptr_t)gp | (uintptr_t)yp | (dd.shuf ? (uintptr_t)0 : (uintptr_t)dp))) & 15) == 0) {
        for (int i = blockIdx.x * kThreads + threadIdx.x; i < (hw >> 2); i += gridDim.x * kThreads) {
            bf4 gv = reinterpret_cast<const bf4*>(gp)[i];
            const bf4 yv = reinterpret_cast<const bf4*>(yp)[i];
            if (q2) gv = g2_add(gv, q2, i, g2.w4, g2.inv_w4, g2.scale);
            bf4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float yh = fmaf(yv[e], s, b);
                const float u = gv[e] * (yh >= 0.f ? 1.f : slope);
                o[e] = s * (u - m1 - yh * m2);
                mx = fmaxf(mx, fabsf(o[e]));
            }
            dy_store(dp, i, o, dd);
        }
        san_amax_record<kThreads / 64>(amax, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, mx);
        return;
    }
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < hw; i += gridDim.x * kThreads) {
        const float yh = fmaf(yp[i], s, b);
        const float u = gp[i] * (yh >= 0.f ? 1.f : slope);
        const float o = s * (u - m1 - yh * m2) + (dd.acc ? dp[i] : 0.f);      // (the shuffled form always takes the vector path: host check)
        dp[i] = o;
        mx = fmaxf(mx, fabsf(o));
    }
    san_amax_record<kThreads / 64>(amax, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, mx);
}


// ---------------------------------------------------------- k-space / loss backward pieces
// dL/d(dc_weight) partial sums: -sum M[w] * Re(conj(G) * (k - k0)) over one plane chunk
__global__ void __launch_bounds__(kThreads)
dc_weight_grad_kernel(const float2* __restrict__ G, const float2* __restrict__ k, const float2* __restrict__ k0,
                      const float* __restrict__ mask, float* __restrict__ partial, size_t total, int W) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kThreads) {
        const float m = mask[i % W];
        const float2 g = G[i], a = k[i], b = k0[i];
        s -= m * (g.x * (a.x - b.x) + g.y * (a.y - b.y));
    }
    s = san_wave_total(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// gS[n,c] += sign1 * conj(r[n]) * t1[n,c] + x[n,c] * conj(gm[n]);  r, gm planar [n,2,hw]
__global__ void __launch_bounds__(kThreads) SAN_NO_PK32
sens_grad_acc_kernel(float2* __restrict__ gS, const float* __restrict__ r, const float2* __restrict__ t1,
                     const float2* __restrict__ x, const float* __restrict__ gm, float sign1, int C, int HW) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HW; i += gridDim.x * kThreads) {
        const float rr = r[(size_t)n * 2 * HW + i], ri = r[(size_t)n * 2 * HW + HW + i];
        const float gr = gm[(size_t)n * 2 * HW + i], gi = gm[(size_t)n * 2 * HW + HW + i];
        for (int c = 0; c < C; ++c) {
            const size_t e = ((size_t)n * C + c) * HW + i;
            const float2 t = t1[e], xv = x[e];
            float2 acc = gS[e];
            // conj(r) * t = (rr - i ri)(t.x + i t.y)
            acc.x += sign1 * (rr * t.x + ri * t.y);
            acc.y += sign1 * (rr * t.y - ri * t.x);
            // x * conj(gm) = (x.x + i x.y)(gr - i gi)
            acc.x += xv.x * gr + xv.y * gi;
            acc.y += xv.y * gr - xv.x * gi;
            gS[e] = acc;
        }
    }
}

// image-domain cascade backward (see san_dc_rows): the same sensitivity-map accumulation, and in the same pass the
// regulariser-input gradient joins the state gradient:  gd[n,c] += gm[n] * S[n,c]   (m = sum_c conj(S_c) x_c).  gS may be null.
__global__ void __launch_bounds__(kThreads) SAN_NO_PK32
sens_grad_prop_kernel(float2* __restrict__ gS, const float* __restrict__ r, const float2* __restrict__ t1,
                      const float2* __restrict__ x, const float* __restrict__ gm, float sign1, float2* __restrict__ gd,
                      const float2* __restrict__ S, int C, int HW) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HW; i += gridDim.x * kThreads) {
        const float gr = gm[(size_t)n * 2 * HW + i], gi = gm[(size_t)n * 2 * HW + HW + i];
        float rr = 0.f, ri = 0.f;
        if (gS) {
            rr = r[(size_t)n * 2 * HW + i];
            ri = r[(size_t)n * 2 * HW + HW + i];
        }
        for (int c = blockIdx.z; c < C; c += gridDim.z) {       // coils are independent: one per workgroup (grid z) when there are several
            const size_t e = ((size_t)n * C + c) * HW + i;
            if (gS) {
                const float2 t = t1[e], xv = x[e];
                float2 acc = gS[e];
                acc.x += sign1 * (rr * t.x + ri * t.y);
                acc.y += sign1 * (rr * t.y - ri * t.x);
                acc.x += xv.x * gr + xv.y * gi;
                acc.y += xv.y * gr - xv.x * gi;
                gS[e] = acc;
            }
            const float2 sv = S[e];
            float2 g = gd[e];
            g.x += gr * sv.x - gi * sv.y;
            g.y += gr * sv.y + gi * sv.x;
            gd[e] = g;
        }
    }
}

// S_c = e_c / d, d = sqrt(sum |e_c|^2) + eps:  g_e_c = G_c/d - T * e_c / rss,  T = Re(sum conj(G_c) e_c) / d^2
__global__ void __launch_bounds__(kThreads)
sens_normalize_bwd_kernel(const float* __restrict__ est, const float2* __restrict__ gS, float* __restrict__ gest, int C,
                          int HW) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HW; i += gridDim.x * kThreads) {
        float ss = 0.f, dot = 0.f;
        for (int c = 0; c < C; ++c) {
            const size_t b = ((size_t)(n * C + c) * 2) * HW + i;
            const float re = est[b], im = est[b + HW];
            const float2 g = gS[((size_t)n * C + c) * HW + i];
            ss += re * re + im * im;
            dot += g.x * re + g.y * im;
        }
        const float rs = sqrtf(ss);
        const float d = rs + 1e-6f;
        const float T = dot / (d * d);
        const float inv_rs = rs > 0.f ? 1.f / rs : 0.f;
        for (int c = 0; c < C; ++c) {
            const size_t b = ((size_t)(n * C + c) * 2) * HW + i;
            const float re = est[b], im = est[b + HW];
            const float2 g = gS[((size_t)n * C + c) * HW + i];
            gest[b] = g.x / d - T * re * inv_rs;
            gest[b + HW] = g.y / d - T * im * inv_rs;
        }
    }
}

// y = sqrt(sum_c |x_c|^2): gx_c = g * x_c / y   (complex interleaved or real x)
__global__ void __launch_bounds__(kThreads)
rss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g,
               float* __restrict__ gx, int C, int HW, int is_complex) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HW; i += gridDim.x * kThreads) {
        const float yv = y[(size_t)n * HW + i];
        const float f = yv > 0.f ? g[(size_t)n * HW + i] / yv : 0.f;
        for (int c = 0; c < C; ++c) {
            const size_t e = ((size_t)n * C + c) * HW + i;
            if (is_complex) {
                const float2 v = *reinterpret_cast<const float2*>(x + 2 * e);
                *reinterpret_cast<float2*>(gx + 2 * e) = make_float2(v.x * f, v.y * f);
            } else {
                gx[e] = x[e] * f;
            }
        }
    }
}

// SSIM backward, stage 1: per window position the derivatives of S wrt (uy, uyy, uxy)
__global__ void __launch_bounds__(kThreads)
ssim_bwd_coef_kernel(const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ coef, int H, int W,
                     int OH, int OW) {
    constexpr int K = 7, TW = 32, TH = 8, IW = TW + K - 1, IH = TH + K - 1;
    __shared__ float sx[IH][IW + 1];
    __shared__ float sy[IH][IW + 1];
    const int n = blockIdx.z;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    const float* xp = X + (size_t)n * H * W;
    const float* yp = Y + (size_t)n * H * W;
    for (int e = threadIdx.x; e < IH * IW; e += kThreads) {
        const int r = e / IW, c = e - r * IW;
        const int gy = oy0 + r, gx = ox0 + c;
        const bool ok = gy < H && gx < W;
        sx[r][c] = ok ? xp[(size_t)gy * W + gx] : 0.f;
        sy[r][c] = ok ? yp[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int ox = ox0 + lx, oy = oy0 + ly;
    if (ox >= OW || oy >= OH) return;
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
    const float inv = 1.f / 49.f, cov = 49.f / 48.f, C1 = 1e-4f, C2 = 9e-4f;
    const float ux = s_x * inv, uy = s_y * inv, uxx = s_xx * inv, uyy = s_yy * inv, uxy = s_xy * inv;
    const float vx = cov * (uxx - ux * ux), vy = cov * (uyy - uy * uy), vxy = cov * (uxy - ux * uy);
    const float A1 = 2.f * ux * uy + C1, A2 = 2.f * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
    const float Sv = (A1 * A2) / (B1 * B2);
    const float dA1 = A2 / (B1 * B2), dA2 = A1 / (B1 * B2), dB1 = -Sv / B1, dB2 = -Sv / B2;
    const float d_uy = dA1 * 2.f * ux - dA2 * 2.f * cov * ux + dB1 * 2.f * uy - dB2 * 2.f * cov * uy;
    const float d_uyy = dB2 * cov;
    const float d_uxy = dA2 * 2.f * cov;
    const size_t o = ((size_t)n * OH + oy) * OW + ox;
    const size_t plane = (size_t)gridDim.z * OH * OW;
    coef[o] = d_uy;
    coef[plane + o] = d_uyy;
    coef[2 * plane + o] = d_uxy;
}

// stage 2: g_Y[q] = scale/49 * sum_{windows p containing q} (d_uy[p] + 2 Y[q] d_uyy[p] + X[q] d_uxy[p])
__global__ void __launch_bounds__(kThreads)
ssim_bwd_gather_kernel(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ coef,
                       float* __restrict__ gY, int N, int H, int W, int OH, int OW, float num, const float* __restrict__ num_dev,
                       float denom) {
    const int n = blockIdx.y;
    const float scale = -(num * (num_dev ? num_dev[0] : 1.f)) / denom;       // loss = 1 - mean(S)
    const size_t plane = (size_t)N * OH * OW;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < H * W; i += gridDim.x * kThreads) {
        const int qy = i / W, qx = i - qy * W;
        float a = 0.f, b = 0.f, c = 0.f;
        for (int py = max(qy - 6, 0); py <= min(qy, OH - 1); ++py)
            for (int px = max(qx - 6, 0); px <= min(qx, OW - 1); ++px) {
                const size_t o = ((size_t)n * OH + py) * OW + px;
                a += coef[o];
                b += coef[plane + o];
                c += coef[2 * plane + o];
            }
        const size_t e = (size_t)n * H * W + i;
        gY[e] = scale * (1.f / 49.f) * (a + 2.f * Y[e] * b + X[e] * c);
    }
}

// generic normalisation backward with host-provided per-(n,c) coefficients (m1, m2, p, q):
//   yh = sc*y + sh, u = g*(yh >= 0 ? 1 : slope),  dy = sc * (u - m1 - (p*yh + q) * m2)
// (BatchNorm training: m1 = mean(u), m2 = mean(u*yn) over N,H,W with yn = (yh - beta)/gamma)
__global__ void __launch_bounds__(kThreads)
act_bwd_coef_kernel(const float* __restrict__ g, int g_ctot, int g_coff, const float* __restrict__ y, int y_ctot,
                    int y_coff, const float* __restrict__ sc, const float* __restrict__ sh, float slope,
                    const float* __restrict__ coef, float* __restrict__ dy, int d_ctot, int d_coff, int c, int hw, unsigned* amax) {
    const int ch = blockIdx.y, n = blockIdx.z;
    const float s = sc ? sc[n * y_ctot + y_coff + ch] : 1.f;
    const float b = sh ? sh[n * y_ctot + y_coff + ch] : 0.f;
    const float* cf = coef + ((size_t)n * c + ch) * 4;
    const float m1 = cf[0], m2 = cf[1], p = cf[2], q = cf[3];
    const float* gp = g + ((size_t)(n * g_ctot + g_coff + ch)) * hw;
    const float* yp = y + ((size_t)(n * y_ctot + y_coff + ch)) * hw;
    float* dp = dy + ((size_t)(n * d_ctot + d_coff + ch)) * hw;
    float mx = 0.f;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < hw; i += gridDim.x * kThreads) {
        const float yh = fmaf(yp[i], s, b);
        const float u = gp[i] * (yh >= 0.f ? 1.f : slope);
        const float o = s * (u - m1 - fmaf(p, yh, q) * m2);
        dp[i] = o;
        mx = fmaxf(mx, fabsf(o));
    }
    san_amax_record<kThreads / 64>(amax, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, mx);
}

// bilinear warp backward wrt the sampling grid (zeros padding, align_corners = False):
// g_off[n, 0/1, i, j] = sum_c g[n,c,i,j] * d out / d (x, y)   in normalised units (ix = ((x+1)W-1)/2)
__global__ void __launch_bounds__(kThreads) SAN_NO_PK32
warp_bwd_grid_kernel(const float* __restrict__ img, const float* __restrict__ grid, const float* __restrict__ g,
                     float* __restrict__ g_off, int C, int H, int W) {
    const int n = blockIdx.y;
    const int HW = H * W;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HW; i += gridDim.x * kThreads) {
        const float2 gg = *reinterpret_cast<const float2*>(grid + ((size_t)n * HW + i) * 2);
        const float ix = ((gg.x + 1.f) * (float)W - 1.f) * 0.5f;
        const float iy = ((gg.y + 1.f) * (float)H - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const bool xin0 = x0 >= 0 && x0 < W, xin1 = x1 >= 0 && x1 < W;
        const bool yin0 = y0 >= 0 && y0 < H, yin1 = y1 >= 0 && y1 < H;
        float gx = 0.f, gy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* p = img + ((size_t)n * C + c) * HW;
            const float v00 = (yin0 && xin0) ? p[y0 * W + x0] : 0.f;
            const float v01 = (yin0 && xin1) ? p[y0 * W + x1] : 0.f;
            const float v10 = (yin1 && xin0) ? p[y1 * W + x0] : 0.f;
            const float v11 = (yin1 && xin1) ? p[y1 * W + x1] : 0.f;
            const float go = g[((size_t)n * C + c) * HW + i];
            gx += go * ((v01 - v00) * wy0 + (v11 - v10) * wy1);
            gy += go * ((v10 - v00) * wx0 + (v11 - v01) * wx1);
        }
        g_off[((size_t)n * 2 + 0) * HW + i] = gx * (0.5f * (float)W);
        g_off[((size_t)n * 2 + 1) * HW + i] = gy * (0.5f * (float)H);
    }
}

// d/ds of gscale * (mean(dW^2) + mean(dH^2))/2 for an NCHW [n,2,h,w] field; accumulates into g
__global__ void __launch_bounds__(kThreads)
gradient_loss_bwd_kernel(const float* __restrict__ off, float* __restrict__ g, int H, int W, float num,
                         const float* __restrict__ num_dev, float den_x, float den_y, int accumulate) {
    const int plane = blockIdx.y;
    const float gs = num * (num_dev ? num_dev[0] : 1.f);
    const float cx = gs / den_x, cy = gs / den_y;
    const float* p = off + (size_t)plane * H * W;
    float* gp = g + (size_t)plane * H * W;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < H * W; i += gridDim.x * kThreads) {
        const int y = i / W, x = i - y * W;
        const float v = p[i];
        float d = 0.f;
        if (x > 0) d += cx * (v - p[i - 1]);
        if (x + 1 < W) d -= cx * (p[i + 1] - v);
        if (y > 0) d += cy * (v - p[i - W]);
        if (y + 1 < H) d -= cy * (p[i + W] - v);
        gp[i] = accumulate ? gp[i] + d : d;
    }
}

template <int KS, int X>
static int wgrad_vec_launch(dim3 grid, hipStream_t s, const WgradArgs& a) {
    constexpr int ACS = KS == 3 ? 456 : 264;
    constexpr size_t bytes = (2 * (size_t)(kCIB * ACS + 8 * X * kDCS) + 4 * kVT) * sizeof(float);
    static SanPerDevice configured;
    const int dev__ = san_current_device();
    if (!configured.has(dev__)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_vec_kernel<KS, X>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        {
            san_set_error("cannot reserve %d bytes of LDS for the weight-gradient kernel", (int)bytes);
            return SAN_E_UNSUPPORTED;
        }
        configured.mark(dev__);
    }
    hipLaunchKernelGGL((conv_wgrad_vec_kernel<KS, X>), grid, dim3(kVT), bytes, s, a);
    return SAN_OK;
}


// ---- small finalisation kernels that replace chains of one-element-per-channel ATen calls in the backward tapes ----
// BatchNorm training backward (unet.py:125): from the per-chunk (sum u, sum u*yh) of san_plane_dot_stats, per channel
//   dbeta = S1, dgamma = (S2 - beta S1) / gamma (accumulated into the parameter gradients) and the coefficients
//   (m1, m2, p, q) = (dbeta/cnt, dgamma/cnt, 1/gamma, -beta/gamma) san_act_bwd_coef wants, for every sample.
// One WAVE per channel (round 5; before: one thread walked all n * tiles records in a dependent chain of double additions,
// 14-34 us per launch, 26 launches in the alignment network's backward): lanes stride over the (sample, chunk) records, the 64
// partial sums meet in a fixed-order butterfly in double -- deterministic.
__global__ void __launch_bounds__(64) bn_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ coef, int n, int c,
                                                              int tiles, double cnt) {
    const int ch = blockIdx.x, lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    const int total = n * tiles;
    for (int idx = lane; idx < total; idx += 64) {
        const int i = idx / tiles, t = idx - i * tiles;
        const float* p = part + (((size_t)i * c + ch) * tiles + t) * 2;
        s1 += (double)p[0];
        s2 += (double)p[1];
    }
    s1 = san_wave_sum_d(s1);
    s2 = san_wave_sum_d(s2);
    const double ga = (double)gamma[ch], be = (double)beta[ch];
    const double dg = (s2 - be * s1) / ga;
    if (lane == 0) {
        dgamma[ch] += (float)dg;
        dbeta[ch] += (float)s1;
    }
    const float m1 = (float)(s1 / cnt), m2 = (float)(dg / cnt), pp = (float)(1.0 / ga), qq = (float)(-be / ga);
    for (int i = lane; i < n; i += 64) {
        float* o = coef + ((size_t)i * c + ch) * 4;
        o[0] = m1;
        o[1] = m2;
        o[2] = pp;
        o[3] = qq;
    }
}

// BatchNorm2d running statistics (unet.py:125, momentum m): running = (1 - m) running + m batch, the batch variance
// scaled by var_factor (unbiased / count-scale correction); num_batches_tracked += 1
__global__ void bn_update_running_kernel(float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt,
                                         const float* __restrict__ bmean, const float* __restrict__ bvar, int c, float m,
                                         float var_factor) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch == 0 && nbt) *nbt += 1;
    if (ch >= c) return;
    rmean[ch] = rmean[ch] * (1.f - m) + m * bmean[ch];
    rvar[ch] = rvar[ch] * (1.f - m) + m * (bvar[ch] * var_factor);
}

// bias gradient from san_plane_stats chunks (count, mean, m2): db[c] += sum_{n,t} count * mean
__global__ void __launch_bounds__(64) bias_grad_kernel(const float* __restrict__ part, float* __restrict__ db, int n, int c,
                                                        int tiles) {
    // one wave per channel, lanes stride over the (sample, chunk) records; fixed-order butterfly sum in double
    const int ch = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int e = lane; e < n * tiles; e += 64) {
        const int i = e / tiles, t = e - i * tiles;
        const float* p = part + (((size_t)i * c + ch) * tiles + t) * 3;
        s += (double)p[0] * (double)p[1];
    }
    s = san_wave_sum_d(s);
    if (lane == 0) db[ch] += (float)s;
}

// NormUnet backward (varnet.py:246-332): with B = chunk sums of (g_out, g_out*U), A = chunk sums of (g_xh, g_xh*xh),
// s, t = the input's group-norm affine, sd = its std:  dmu = B1 - A1 s, dsig = B2 - A2 s, cco = dsig/(s (nel-1) sd);
// the input gradient is then  g_m = (s g_xh + dmu/nel) + (cco s m + cco t),  written as two lazy affines.
__global__ void normunet_bwd_coefs_kernel(const float* __restrict__ partB, const float* __restrict__ partA, int tiles,
                                          const float* __restrict__ scale, const float* __restrict__ shift, int x_ctot,
                                          const float* __restrict__ stdv, double nel, float* __restrict__ a_sc,
                                          float* __restrict__ a_sh, int g_ctot, float* __restrict__ m_sc,
                                          float* __restrict__ m_sh, int b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int width = g_ctot > x_ctot ? g_ctot : x_ctot;
    if (i >= b * width) return;
    const int bi = i / width, j = i - bi * width;
    float asc = 0.f, ash = 0.f, msc = 0.f, msh = 0.f;
    if (j < 2) {
        const float* pb = partB + ((size_t)bi * 2 + j) * tiles * 2;
        const float* pa = partA + ((size_t)bi * 2 + j) * tiles * 2;
        double b1 = 0.0, b2 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int t = 0; t < tiles; ++t) {
            b1 += (double)pb[2 * t];
            b2 += (double)pb[2 * t + 1];
            a1 += (double)pa[2 * t];
            a2 += (double)pa[2 * t + 1];
        }
        const double s = (double)scale[bi * x_ctot + j], t = (double)shift[bi * x_ctot + j], sd = (double)stdv[bi * 2 + j];
        const double dmu = b1 - a1 * s, dsig = b2 - a2 * s;
        // a constant plane has sd == 0: torch's std backward masks that case to 0 (and the forward divides by sd + 1e-6)
        const double cco = sd > 0.0 ? dsig / (s * (nel - 1.0) * sd) : 0.0;
        asc = (float)s;
        ash = (float)(dmu / nel);
        msc = (float)(cco * s);
        msh = (float)(cco * t);
    }
    if (j < g_ctot) {
        a_sc[bi * g_ctot + j] = asc;
        a_sh[bi * g_ctot + j] = ash;
    }
    if (j < x_ctot) {
        m_sc[bi * x_ctot + j] = msc;
        m_sh[bi * x_ctot + j] = msh;
    }
}

// ---- NormUnet backward, the two ends of a cascade's U-Net backward as ONE launch each (round 6) ----------------------------------
// HEAD (varnet.py:321-332 backwards): with U = unnorm^-1(out) = out * isd + nshift and g = dL/d out,
//     g_u = g * std  (dL/dU)           part_b[n, ch, t] = chunk sums (sum g, sum g U)  (what san_plane_dot_stats writes)
// replaces san_plane_dot_stats + san_apply_fwd + san_plane_stats + san_bias_grad_from_stats: the final 1x1 convolution's bias gradient
// sum(g_u) = std * sum(g) comes out of part_b in the tail launch.  Chunks, summation order and rounding are bwd_stats_kernel's.
__global__ void __launch_bounds__(kThreads) SAN_NO_PK32
normunet_bwd_head_kernel(const float* __restrict__ g, const float* __restrict__ out, const float* __restrict__ isd,
                         const float* __restrict__ nshift, const float* __restrict__ stdv, float* __restrict__ gu,
                         float* __restrict__ part, int hw, int tiles) {
    __shared__ float red[8];
    const int t = blockIdx.x, ch = blockIdx.y, n = blockIdx.z;
    const int chunk = (((hw + tiles - 1) / tiles) + 3) & ~3;
    const int lo = t * chunk;
    const int cnt = max(0, min(hw, lo + chunk) - lo);
    const size_t plane = ((size_t)(n * 2 + ch)) * hw + lo;
    const float* gp = g + plane;
    const float* op = out + plane;
    float* up = gu + plane;
    const float s = isd[n * 2 + ch], b = nshift[n * 2 + ch], sd = stdv[n * 2 + ch];
    float s1 = 0.f, s2 = 0.f;
    if ((((uintptr_t)gp | (uintptr_t)op | (uintptr_t)up) & 15) == 0) {
        const int c4 = cnt >> 2;
        for (int i = threadIdx.x; i < c4; i += kThreads) {
            const bf4 gv = reinterpret_cast<const bf4*>(gp)[i];
            const bf4 ov = reinterpret_cast<const bf4*>(op)[i];
            bf4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float U = fmaf(ov[e], s, b);
                s1 += gv[e];
                s2 = fmaf(gv[e], U, s2);
                o[e] = fmaf(gv[e], sd, 0.f);
            }
            reinterpret_cast<bf4*>(up)[i] = o;
        }
        for (int i = 4 * c4 + threadIdx.x; i < cnt; i += kThreads) {
            const float U = fmaf(op[i], s, b);
            s1 += gp[i];
            s2 = fmaf(gp[i], U, s2);
            up[i] = fmaf(gp[i], sd, 0.f);
        }
    } else {
        for (int i = threadIdx.x; i < cnt; i += kThreads) {
            const float U = fmaf(op[i], s, b);
            s1 += gp[i];
            s2 = fmaf(gp[i], U, s2);
            up[i] = fmaf(gp[i], sd, 0.f);
        }
    }
    s1 = san_wave_total(s1);
    s2 = san_wave_total(s2);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = s1;
        red[4 + (threadIdx.x >> 6)] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)(n * 2 + ch) * tiles + t) * 2;
        o[0] = (red[0] + red[1]) + (red[2] + red[3]);
        o[1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

// TAIL: everything after the U-Net's own backward, one launch over the plane (replaces san_normunet_bwd_coefs + san_add_fwd +
// san_sens_grad_prop + the reference channel's san_act_bwd + san_partials_add + the bias gradient):
//   coefficients of dL/dm from part_b (head) and part_x = chunk sums (sum g_xh, sum g_xh xh) of the U-Net input channels
//   (every workgroup derives them itself, in san_normunet_bwd_coefs' order and precision: identical bits everywhere);
//   g_m = (a_sc g_xh + a_sh) + (m_sc m + m_sh) per pixel, NOT stored: gd[n, c] += g_m S[n, c] and the sensitivity-map
//   accumulation gS[n, c] += sign1 conj(r) t1 + x conj(g_m) at once (san_sens_grad_prop);
//   g_ref (+)= InstanceNorm backward of the reference channel (channel 2 of g_xh / xin; plane sums from part_x);
//   one workgroup adds the scalar gradients: db[j] += sum_n std[n, j] sum_t part_b[n, j, t, 0] (bias of the last 1x1 convolution)
//   and dcw[0] += dcw_scale * sum(dcw_part) (dc_weight, varnet.py:523).
struct NuTailArgs {
    const float* part_b;       // [b, 2, tiles, 2]
    const float* part_x;       // [b, xc, tiles, 2], xc = 2 or 3 (with the reference channel)
    const float* x;            // xin buffer [b, x_ctot, hw] (channels 0, 1 = m planar, 2 = reference)
    const float* x_sc;         // [b, x_ctot]
    const float* x_sh;
    const float* stdv;         // [b, 2]
    const float* gxh;          // [b, g_ctot, hw]
    float2* gd;                // [b, C, hw] in / out
    const float2* S;           // [b, C, hw]
    float2* gS;                // [b, C, hw] or null
    const float* r;            // planar [b, 2, hw] (with gS)
    const float2* t1;          // [b, C, hw] (with gS)
    const float2* xs;          // [b, C, hw] (with gS)
    float* g_ref;              // [b, 1, hw] or null
    float* db;                 // [2] or null
    const float* dcw_part;     // or null
    float* dcw;
    double nel;
    float sign1, dcw_scale;
    int tiles, x_ctot, g_ctot, xc, C, hw, b, dcw_count, ref_acc;
};

__global__ void __launch_bounds__(kThreads) SAN_NO_PK32 normunet_bwd_tail_kernel(const NuTailArgs a) {
    __shared__ float cf[12];        // asc0 ash0 msc0 msh0 | asc1 ash1 msc1 msh1 | s2 b2 m1 m2
    __shared__ double red[kThreads / 64];
    const int n = blockIdx.y;
    if (threadIdx.x < 2) {
        const int j = threadIdx.x;
        const float* pb = a.part_b + ((size_t)n * 2 + j) * a.tiles * 2;
        const float* pa = a.part_x + ((size_t)n * a.xc + j) * a.tiles * 2;
        double b1 = 0.0, b2 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int t = 0; t < a.tiles; ++t) {
            b1 += (double)pb[2 * t];
            b2 += (double)pb[2 * t + 1];
            a1 += (double)pa[2 * t];
            a2 += (double)pa[2 * t + 1];
        }
        const double s = (double)a.x_sc[n * a.x_ctot + j], t = (double)a.x_sh[n * a.x_ctot + j], sd = (double)a.stdv[n * 2 + j];
        const double dmu = b1 - a1 * s, dsig = b2 - a2 * s;
        const double cco = sd > 0.0 ? dsig / (s * (a.nel - 1.0) * sd) : 0.0;
        cf[4 * j] = (float)s;
        cf[4 * j + 1] = (float)(dmu / a.nel);
        cf[4 * j + 2] = (float)(cco * s);
        cf[4 * j + 3] = (float)(cco * t);
    } else if (threadIdx.x == 2 && a.g_ref) {
        const float* p = a.part_x + ((size_t)n * a.xc + 2) * a.tiles * 2;
        double t1 = 0.0, t2 = 0.0;
        for (int t = 0; t < a.tiles; ++t) {
            t1 += (double)p[2 * t];
            t2 += (double)p[2 * t + 1];
        }
        cf[8] = a.x_sc[n * a.x_ctot + 2];
        cf[9] = a.x_sh[n * a.x_ctot + 2];
        cf[10] = (float)(t1 / a.hw);
        cf[11] = (float)(t2 / a.hw);
    }
    __syncthreads();
    const float asc0 = cf[0], ash0 = cf[1], msc0 = cf[2], msh0 = cf[3], asc1 = cf[4], ash1 = cf[5], msc1 = cf[6], msh1 = cf[7];
    const size_t HW = a.hw;
    const float* x0 = a.x + ((size_t)n * a.x_ctot) * HW;
    const float* g0 = a.gxh + ((size_t)n * a.g_ctot) * HW;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < a.hw; i += gridDim.x * kThreads) {
        const float gr = fmaf(g0[i], asc0, ash0) + fmaf(x0[i], msc0, msh0);
        const float gi = fmaf(g0[HW + i], asc1, ash1) + fmaf(x0[HW + i], msc1, msh1);
        float rr = 0.f, ri = 0.f;
        if (a.gS) {
            rr = a.r[(size_t)n * 2 * HW + i];
            ri = a.r[(size_t)n * 2 * HW + HW + i];
        }
        for (int c = blockIdx.z; c < a.C; c += gridDim.z) {
            const size_t e = ((size_t)n * a.C + c) * HW + i;
            if (a.gS) {
                const float2 t = a.t1[e], xv = a.xs[e];
                float2 acc = a.gS[e];
                acc.x += a.sign1 * (rr * t.x + ri * t.y);
                acc.y += a.sign1 * (rr * t.y - ri * t.x);
                acc.x += xv.x * gr + xv.y * gi;
                acc.y += xv.y * gr - xv.x * gi;
                a.gS[e] = acc;
            }
            const float2 sv = a.S[e];
            float2 gg = a.gd[e];
            gg.x += gr * sv.x - gi * sv.y;
            gg.y += gr * sv.y + gi * sv.x;
            a.gd[e] = gg;
        }
        if (a.g_ref && blockIdx.z == 0) {
            const float yh = fmaf(x0[2 * HW + i], cf[8], cf[9]);
            const float u = g0[2 * HW + i];                         // (the reference enters without an activation: slope 1)
            const float o = cf[8] * (u - cf[10] - yh * cf[11]);
            float* q = a.g_ref + (size_t)n * HW + i;
            *q = a.ref_acc ? *q + o : o;
        }
    }
    // the scalar gradients: the launch's first workgroup, fixed order (thread-strided double sums + a fixed tree)
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        if (a.db && threadIdx.x < 128) {
            // bias of the last 1x1 convolution: one wave per channel, lanes over (sample, chunk), double butterfly
            const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
            double sb = 0.0;
            for (int e = lane; e < a.b * a.tiles; e += 64) {
                const int bi = e / a.tiles, t = e - bi * a.tiles;
                sb += (double)a.stdv[bi * 2 + j] * (double)a.part_b[(((size_t)bi * 2 + j) * a.tiles + t) * 2];
            }
            sb = san_wave_sum_d(sb);
            if (lane == 0) a.db[j] += (float)sb;
        }
        if (a.dcw_part) {
            double sd = 0.0;
            for (int i = threadIdx.x; i < a.dcw_count; i += kThreads) sd += (double)a.dcw_part[i];
            sd = san_wave_sum_d(sd);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sd;
            __syncthreads();
            if (threadIdx.x == 0) {
                double tot = 0.0;
                for (int w = 0; w < kThreads / 64; ++w) tot += red[w];
                a.dcw[0] += (float)((double)a.dcw_scale * tot);
            }
        }
    }
}

// dst[0] += scale * sum(part[0 .. count)): double accumulation in a fixed order (thread-strided sums, then a fixed tree), one
// workgroup.  The scalar parameter gradients that arrive as per-workgroup partials (dc_weight) without host-side glue.
__global__ void __launch_bounds__(256) partials_add_kernel(const float* __restrict__ part, int count, float scale,
                                                           float* __restrict__ dst) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) s += (double)part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dst[0] += (float)((double)scale * red[0]);
}

template <int BN>
static int cluster_launch(int v, dim3 grid, hipStream_t s, const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff,
                          const float* sc, const float* sh, float slope, float* dy, int d_ctot, int d_coff, int hw, unsigned* amax,
                          const G2Src q, const DyDst dd, unsigned* sync, const float* gamma, const float* beta, float* dgamma,
                          float* dbeta, double cnt) {
#define SAN_ABC(V) hipLaunchKernelGGL((act_bwd_cluster_kernel<V, BN>), grid, dim3(kClT), 0, s, g, g_ctot, g_coff, y, y_ctot, y_coff, sc, sh, slope, dy, d_ctot, d_coff, hw, amax, q, dd, sync, gamma, beta, dgamma, dbeta, cnt)
    if (v <= 1) SAN_ABC(1);
    else if (v <= 2) SAN_ABC(2);
    else if (v <= 4) SAN_ABC(4);
    else SAN_ABC(7);
#undef SAN_ABC
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

}  // namespace

extern "C" {

int san_conv_wgrad_partitions(int n, int h, int w, int cin, int cout, int ks) {
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || (ks != 1 && ks != 3)) return 0;
    const int pv = wgrad_plan(n, h, w, cin, cout, ks, true).P, ps = wgrad_plan(n, h, w, cin, cout, ks, false).P;
    return pv > ps ? pv : ps;
}

int san_conv2d_wgrad(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                     float in_slope, const float* dy, int dy_ctot, int dy_coff, int cout, float* dw, int accumulate,
                     float* partial, int n, int h, int w, int ks, void* stream) {
    SAN_CHECK_ARG(x && dy && dw && partial, "null pointer");
    SAN_CHECK_ARG(ks == 1 || ks == 3, "ks must be 1 or 3");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "bad dims");
    SAN_CHECK_ARG(x_coff >= 0 && x_coff + cin <= x_ctot && dy_coff >= 0 && dy_coff + cout <= dy_ctot, "bad channel view");
    SAN_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "in_scale/in_shift must come together");
    const bool aligned = (((uintptr_t)x | (uintptr_t)dy) & 15) == 0;
    const bool lrelu01 = in_slope >= 0.f && in_slope <= 1.f;          // the pipelined kernel evaluates LeakyReLU as max(v, slope*v)
    const WgradPlan p = wgrad_plan(n, h, w, cin, cout, ks, aligned && lrelu01);
    WgradArgs a{};
    a.x = x;
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_slope = in_slope;
    a.dy = dy;
    a.partial = partial;
    a.x_ctot = x_ctot;
    a.x_coff = x_coff;
    a.cin = cin;
    a.dy_ctot = dy_ctot;
    a.dy_coff = dy_coff;
    a.cout = cout;
    a.N = n;
    a.H = h;
    a.W = w;
    a.tw = p.tw;
    a.th = p.th;
    a.aw = p.aw;
    a.a_count = p.a_count;
    a.npx = p.npx;
    a.groups = p.groups;
    a.tiles_x = p.tiles_x;
    a.tiles_y = p.tiles_y;
    a.P = p.P;
    a.P_rem = p.P_rem;
    a.co_blocks = p.co_blocks;
    a.n_full = san_cdiv(cin, 4) / 4;
    a.cib_q = p.cib_q;
    dim3 grid(p.P, p.co_blocks, p.ci_blocks);
    if (p.vec) {
        const int has_rem = (san_cdiv(cin, 4) % 4) != 0;
        grid = dim3(p.co_blocks * (a.n_full * p.P + (has_rem ? p.P_rem : 0)), 1, 1);
    }
    hipStream_t s = (hipStream_t)stream;
#define SAN_WGRAD_LAUNCH(KS, X) hipLaunchKernelGGL((conv_wgrad_kernel<KS, X>), grid, dim3(kThreads), 0, s, a)
    if (p.vec) {
#define SAN_WGRAD_VEC(KS, X)                                          \
    do {                                                              \
        const int rc = wgrad_vec_launch<KS, X>(grid, s, a);           \
        if (rc != SAN_OK) return rc;                                  \
    } while (0)
        if (ks == 3) {
            switch (p.X) {
                case 1: SAN_WGRAD_VEC(3, 1); break;
                case 2: SAN_WGRAD_VEC(3, 2); break;
                default: SAN_WGRAD_VEC(3, 3); break;
            }
        } else {
            switch (p.X) {
                case 1: SAN_WGRAD_VEC(1, 1); break;
                case 2: SAN_WGRAD_VEC(1, 2); break;
                case 3: SAN_WGRAD_VEC(1, 3); break;
                default: SAN_WGRAD_VEC(1, 4); break;
            }
        }
#undef SAN_WGRAD_VEC
    } else if (ks == 3) {
        switch (p.X) {
            case 2: SAN_WGRAD_LAUNCH(3, 2); break;
            case 3: SAN_WGRAD_LAUNCH(3, 3); break;
            case 5: SAN_WGRAD_LAUNCH(3, 5); break;
            default: SAN_WGRAD_LAUNCH(3, 6); break;
        }
    } else {
        switch (p.X) {
            case 2: SAN_WGRAD_LAUNCH(1, 2); break;
            case 3: SAN_WGRAD_LAUNCH(1, 3); break;
            case 5: SAN_WGRAD_LAUNCH(1, 5); break;
            default: SAN_WGRAD_LAUNCH(1, 6); break;
        }
    }
#undef SAN_WGRAD_LAUNCH
    SAN_LAUNCH_CHECK();
    const int count = cout * cin * ks * ks;
    int blocks = san_cdiv(count, 64);
    if (blocks > 2048) blocks = 2048;
    const int taps = ks * ks;
    const int split = p.vec ? a.n_full * kCIB * taps : cin * taps;          // first column of the remainder block
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, partial, dw, count, p.P, accumulate, cin * taps,
                       split, p.vec ? p.P_rem : p.P);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_plane_dot_stats(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff, const float* sc,
                        const float* sh, float slope, float* part, int n, int c, int hw, void* stream) {
    SAN_CHECK_ARG(g && y && part, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && hw > 0, "bad dims");
    SAN_CHECK_ARG((sc == nullptr) == (sh == nullptr), "scale/shift must come together");
    SAN_CHECK_ARG(g_coff >= 0 && g_coff + c <= g_ctot && y_coff >= 0 && y_coff + c <= y_ctot, "bad channel view");
    int tiles = san_cdiv(hw, 4096);
    tiles = tiles < 1 ? 1 : (tiles > 32 ? 32 : tiles);
    hipLaunchKernelGGL(bwd_stats_kernel, dim3(tiles, c, n), dim3(kThreads), 0, (hipStream_t)stream, g, g_ctot, g_coff, y,
                       y_ctot, y_coff, sc, sh, slope, c, hw, tiles, part, G2Src{nullptr, 0, 0, 0.f, 1, 1.f});
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

static int g_act_bwd_fused = getenv("SAN_NO_ACT_BWD_FUSED") ? 0 : 1;     // tuning hook: 0 = always the two-kernel form

int san_bwd_stat_tiles(int hw) {
    int t = san_cdiv(hw, 4096);
    return t < 1 ? 1 : (t > 32 ? 32 : t);
}

static int act_bwd_impl(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff, const float* sc,
                        const float* sh, float slope, int mode, float* part, float* dy, int d_ctot, int d_coff, int n, int c,
                        int hw, unsigned* amax, void* stream, const float* g2 = nullptr, int g2_ctot = 0,
                        int g2_coff = 0, float g2_scale = 0.f, int w = 0, int flags = 0) {
    SAN_CHECK_ARG(g && y && dy, "null pointer");
    DyDst dd{0, (flags & 2) ? 1 : 0, 1, 1.f};
    if (flags & 1) {
        // un-shuffled destination [n, d_ctot, h/2, w/2]: channels [d_coff, d_coff + 4 c)
        SAN_CHECK_ARG(w > 0 && (w & 3) == 0 && hw % w == 0 && ((hw / w) & 1) == 0 && (hw >> 2) < (1 << 22), "shuffled store: width % 4 == 0, even height");
        SAN_CHECK_ARG(((((uintptr_t)g | (uintptr_t)y)) & 15) == 0 && ((uintptr_t)dy & 7) == 0, "shuffled store: 16-byte aligned inputs, 8-byte aligned destination");
        SAN_CHECK_ARG(d_coff >= 0 && d_coff + 4 * c <= d_ctot, "bad channel view (shuffled destination)");
        dd = DyDst{hw >> 2, dd.acc, w >> 2, 1.f / (float)(w >> 2)};
    }
    G2Src q{nullptr, 0, 0, 0.f, 1, 1.f};
    if (g2) {
        SAN_CHECK_ARG(w > 0 && (w & 3) == 0 && hw % w == 0 && ((hw / w) & 1) == 0, "second gradient source: width % 4 == 0, even height");
        SAN_CHECK_ARG(g2_coff >= 0 && g2_coff + c <= g2_ctot, "bad channel view (second source)");
        SAN_CHECK_ARG(((((uintptr_t)g | (uintptr_t)y | (uintptr_t)dy)) & 15) == 0 && ((uintptr_t)g2 & 7) == 0 && (hw >> 2) < (1 << 22),
                      "second gradient source: 16-byte aligned tensors");
        q = G2Src{g2, g2_ctot, g2_coff, g2_scale, w >> 2, 1.f / (float)(w >> 2)};
    }
    SAN_CHECK_ARG(n > 0 && c > 0 && hw > 0, "bad dims");
    SAN_CHECK_ARG(mode == 0 || mode == 1, "mode must be 0 (affine) or 1 (instance norm)");
    SAN_CHECK_ARG((sc == nullptr) == (sh == nullptr), "scale/shift must come together");
    SAN_CHECK_ARG(mode == 0 || part != nullptr, "instance-norm backward needs the partial buffer");
    SAN_CHECK_ARG(g_coff >= 0 && g_coff + c <= g_ctot && y_coff >= 0 && y_coff + c <= y_ctot && d_coff >= 0 &&
                      (dd.shuf || d_coff + c <= d_ctot), "bad channel view");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = san_bwd_stat_tiles(hw);
    if (mode == 1 && (hw & 3) == 0 && hw <= 512 * 4 * 13 && g_act_bwd_fused &&
        ((((uintptr_t)g | (uintptr_t)y | (dd.shuf ? (uintptr_t)0 : (uintptr_t)dy))) & 15) == 0) {
        // the whole plane fits one workgroup's registers: statistics and gradient in one pass
        const int v = san_cdiv(hw >> 2, 512);
        const dim3 grid(c, n);
#define SAN_ABP(V) hipLaunchKernelGGL((act_bwd_plane_kernel<V>), grid, dim3(512), 0, s, g, g_ctot, g_coff, y, y_ctot, y_coff, sc, sh, slope, dy, d_ctot, d_coff, hw, amax, q, dd)
        if (v <= 1) SAN_ABP(1);
        else if (v <= 2) SAN_ABP(2);
        else if (v <= 4) SAN_ABP(4);
        else if (v <= 7) SAN_ABP(7);
        else SAN_ABP(13);
#undef SAN_ABP
        SAN_LAUNCH_CHECK();
        return SAN_OK;
    }
    if (mode == 1) {
        hipLaunchKernelGGL(bwd_stats_kernel, dim3(tiles, c, n), dim3(kThreads), 0, s, g, g_ctot, g_coff, y, y_ctot,
                           y_coff, sc, sh, slope, c, hw, tiles, part, q);
        SAN_LAUNCH_CHECK();
    }
    int bx = san_cdiv(hw, kThreads * 4);
    long cap = 4096 / ((long)c * n);
    if (cap < 1) cap = 1;
    if (bx > cap) bx = (int)cap;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(bx, c, n), dim3(kThreads), 0, s, g, g_ctot, g_coff, y, y_ctot, y_coff, sc,
                       sh, slope, part, tiles, mode, dy, d_ctot, d_coff, c, hw, amax, q, dd);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_act_bwd(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff, const float* sc,
                const float* sh, float slope, int mode, float* part, float* dy, int d_ctot, int d_coff, int n, int c,
                int hw, void* stream) {
    return act_bwd_impl(g, g_ctot, g_coff, y, y_ctot, y_coff, sc, sh, slope, mode, part, dy, d_ctot, d_coff, n, c, hw, nullptr, stream);
}

int san_act_bwd_amax(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff, const float* sc,
                     const float* sh, float slope, int mode, float* part, float* dy, int d_ctot, int d_coff, void* amax,
                     int n, int c, int hw, void* stream) {
    return act_bwd_impl(g, g_ctot, g_coff, y, y_ctot, y_coff, sc, sh, slope, mode, part, dy, d_ctot, d_coff, n, c, hw,
                        static_cast<unsigned*>(amax), stream);
}

// san_act_bwd_amax with g(p) + g2_scale * g2(p / 2) as the incoming gradient (g2 = a [n, g2_ctot, h/2, w/2] tensor, c channels
// from g2_coff): the U-Net encoder's "skip gradient + average-pool adjoint" sum without materialising it.  hw = h * w, w % 4
// == 0, h even, 16-byte aligned tensors; amax may be NULL.
int san_act_bwd_up_amax(const float* g, int g_ctot, int g_coff, const float* g2, int g2_ctot, int g2_coff, float g2_scale,
                        const float* y, int y_ctot, int y_coff, const float* sc, const float* sh, float slope, int mode,
                        float* part, float* dy, int d_ctot, int d_coff, void* amax, int n, int c, int hw, int w, void* stream) {
    SAN_CHECK_ARG(g2 != nullptr, "null second source");
    return act_bwd_impl(g, g_ctot, g_coff, y, y_ctot, y_coff, sc, sh, slope, mode, part, dy, d_ctot, d_coff, n, c, hw,
                        static_cast<unsigned*>(amax), stream, g2, g2_ctot, g2_coff, g2_scale, w);
}

// san_act_bwd_amax with a destination mode (flags): 1 = dy is stored pixel-UNSHUFFLED as [n, d_ctot, h/2, w/2], channel
// d_coff + 4 ch + 2 (row & 1) + (col & 1) (the transposed convolution's backward: san_unshuffle2_fwd folded into the store; w =
// plane width, w % 4 == 0, even height); 2 = dy += (accumulate).  amax may be NULL.
int san_act_bwd_ex_amax(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff, const float* sc,
                        const float* sh, float slope, int mode, float* part, float* dy, int d_ctot, int d_coff, void* amax,
                        int n, int c, int hw, int w, int flags, void* stream) {
    SAN_CHECK_ARG((flags & ~3) == 0, "flags: 1 = shuffled store, 2 = accumulate");
    return act_bwd_impl(g, g_ctot, g_coff, y, y_ctot, y_coff, sc, sh, slope, mode, part, dy, d_ctot, d_coff, n, c, hw,
                        static_cast<unsigned*>(amax), stream, nullptr, 0, 0, 0.f, w, flags);
}

// ---- one-pass forms on workgroup clusters (act_bwd_cluster_kernel) ----
// SAN_ACT_BWD_CLUSTER=0: off (the two-kernel form everywhere); SAN_ACT_BWD_CLUSTER_MIN: smallest plane (pixels) that takes the cluster
// form in InstanceNorm mode (default: everything the one-workgroup plane kernel cannot hold); SAN_ACT_BWD_CLUSTER_V: float4 per thread.
static int g_cluster_on = (getenv("SAN_ACT_BWD_CLUSTER") && atoi(getenv("SAN_ACT_BWD_CLUSTER")) == 0) ? 0 : 1;
// Measured (N = 8, rocprof-free event timing, scratch/r6_act_bwd_cluster.py): 18 x 320^2 two launches 47.4 us, clusters of 4 float4
// per thread 36.2 (7: 37.6, 2: 69); 36 x 160^2 one-workgroup planes 19.3, clusters 17.0 (7: 19.8); 72 x 80^2 10.8 vs 11.3 -> planes
// from 160 x 160 up take the cluster form.  BatchNorm clusters span the batch: 64 x 160^2 three launches 44.6 us, 32 members of 7
// float4 35.7 (56 members of 4: 48.7); 64 x 80^2 27.0 -> 12.0; 32 x 320^2 with 120 members 138 vs 85 (a cluster that is a tenth of
// what the chip holds leaves its early members idle): clusters of up to 64 members only.
static int g_cluster_min = getenv("SAN_ACT_BWD_CLUSTER_MIN") ? atoi(getenv("SAN_ACT_BWD_CLUSTER_MIN")) : 160 * 160;
static int g_cluster_v = getenv("SAN_ACT_BWD_CLUSTER_V") ? atoi(getenv("SAN_ACT_BWD_CLUSTER_V")) : 4;
static int g_bn_members_max = getenv("SAN_BN_BWD_CLUSTER_MEMBERS") ? atoi(getenv("SAN_BN_BWD_CLUSTER_MEMBERS")) : 64;
static int g_bn_cluster_on = (getenv("SAN_BN_BWD_CLUSTER") && atoi(getenv("SAN_BN_BWD_CLUSTER")) == 0) ? 0 : 1;

static int cluster_v_for(int n4, int want = 0) {
    // float4 per thread: the configured value, or the smallest instantiated one that covers a small plane with ONE member
    const int one = san_cdiv(n4, kClT);
    if (want <= 0) want = g_cluster_v;
    int v = want <= 1 ? 1 : want <= 2 ? 2 : want <= 4 ? 4 : 7;
    if (one <= v) v = one <= 1 ? 1 : one <= 2 ? 2 : one <= 4 ? 4 : 7;
    return v;
}

int san_act_bwd_cluster_set_tuning(int on, int min_hw, int v, int bn_on) {
    if (bn_on >= 0) g_bn_cluster_on = bn_on ? 1 : 0;
    if (on >= 0) g_cluster_on = on ? 1 : 0;
    if (min_hw > 0) g_cluster_min = min_hw;
    if (v > 0) g_cluster_v = v;
    return SAN_OK;
}

// int32 words of zero-initialised scratch san_act_bwd_in wants for [n, c] planes of hw pixels; 0: use san_act_bwd*_amax
int san_act_bwd_in_sync_words(int n, int c, int hw) {
    if (!g_cluster_on || n <= 0 || c <= 0 || hw < g_cluster_min || (hw & 3)) return 0;
    const int K = san_cdiv(hw >> 2, kClT * cluster_v_for(hw >> 2));
    if (K < 2 || K > 256) return 0;
    return n * c * cluster_rec_words(K);
}

// InstanceNorm + LeakyReLU backward in ONE pass for planes of any size (round 6): san_act_bwd_amax / _up_amax / _ex_amax in one
// entry (g2 may be NULL; flags as san_act_bwd_ex_amax; w = plane width, needed with g2 or flags & 1).  sync: the scratch of
// san_act_bwd_in_sync_words(n, c, hw) int32 words, zeroed ONCE by the caller (the kernel leaves it zero).  Needs hw % 4 == 0 and
// 16-byte aligned g, y (and dy unless flags & 1).
int san_act_bwd_in(const float* g, int g_ctot, int g_coff, const float* g2, int g2_ctot, int g2_coff, float g2_scale,
                   const float* y, int y_ctot, int y_coff, const float* sc, const float* sh, float slope, float* dy, int d_ctot,
                   int d_coff, void* amax, int n, int c, int hw, int w, int flags, void* sync, void* stream) {
    SAN_CHECK_ARG(g && y && dy && sync, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && hw > 0 && (hw & 3) == 0, "bad dims (hw % 4 == 0)");
    SAN_CHECK_ARG((flags & ~3) == 0, "flags: 1 = shuffled store, 2 = accumulate");
    SAN_CHECK_ARG((sc == nullptr) == (sh == nullptr), "scale/shift must come together");
    DyDst dd{0, (flags & 2) ? 1 : 0, 1, 1.f};
    if (flags & 1) {
        SAN_CHECK_ARG(w > 0 && (w & 3) == 0 && hw % w == 0 && ((hw / w) & 1) == 0 && (hw >> 2) < (1 << 22), "shuffled store: width % 4 == 0, even height");
        SAN_CHECK_ARG(((uintptr_t)dy & 7) == 0, "shuffled store: 8-byte aligned destination");
        SAN_CHECK_ARG(d_coff >= 0 && d_coff + 4 * c <= d_ctot, "bad channel view (shuffled destination)");
        dd = DyDst{hw >> 2, dd.acc, w >> 2, 1.f / (float)(w >> 2)};
    }
    G2Src q{nullptr, 0, 0, 0.f, 1, 1.f};
    if (g2) {
        SAN_CHECK_ARG(w > 0 && (w & 3) == 0 && hw % w == 0 && ((hw / w) & 1) == 0 && (hw >> 2) < (1 << 22), "second gradient source: width % 4 == 0, even height");
        SAN_CHECK_ARG(g2_coff >= 0 && g2_coff + c <= g2_ctot && ((uintptr_t)g2 & 7) == 0, "bad second source");
        q = G2Src{g2, g2_ctot, g2_coff, g2_scale, w >> 2, 1.f / (float)(w >> 2)};
    }
    SAN_CHECK_ARG(((((uintptr_t)g | (uintptr_t)y | (dd.shuf ? (uintptr_t)0 : (uintptr_t)dy))) & 15) == 0, "16-byte aligned tensors");
    SAN_CHECK_ARG(g_coff >= 0 && g_coff + c <= g_ctot && y_coff >= 0 && y_coff + c <= y_ctot && d_coff >= 0 &&
                      (dd.shuf || d_coff + c <= d_ctot), "bad channel view");
    const int v = cluster_v_for(hw >> 2);
    const int K = san_cdiv(hw >> 2, kClT * v);
    SAN_CHECK_ARG(K >= 1 && K <= 256, "plane too large for one cluster");
    return cluster_launch<0>(v, dim3(K, c, n), (hipStream_t)stream, g, g_ctot, g_coff, y, y_ctot, y_coff, sc, sh, slope, dy, d_ctot,
                             d_coff, hw, static_cast<unsigned*>(amax), q, dd, static_cast<unsigned*>(sync), nullptr, nullptr, nullptr,
                             nullptr, 1.0);
}

// Training-mode BatchNorm2d + LeakyReLU backward (unet.py:125) in ONE pass: san_plane_dot_stats + san_bn_bwd_finalize +
// san_act_bwd_coef_amax.  dgamma / dbeta are accumulated into; sync: san_bn_act_bwd_sync_words(n, c, hw) zeroed int32 words
// (0 words: shape not covered, use the three-launch form).
int san_bn_act_bwd_sync_words(int n, int c, int hw) {
    if (!g_bn_cluster_on || n <= 0 || c <= 0 || hw <= 0 || (hw & 3)) return 0;
    const int K = san_cdiv(hw >> 2, kClT * cluster_v_for(hw >> 2, 7));
    if ((long long)K * n > (g_bn_members_max < 256 ? g_bn_members_max : 256)) return 0;
    return c * cluster_rec_words(K * n);
}

int san_bn_act_bwd(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff, const float* sc, const float* sh,
                   float slope, const float* gamma, const float* beta, float* dgamma, float* dbeta, float* dy, int d_ctot,
                   int d_coff, void* amax, int n, int c, int hw, void* sync, void* stream) {
    SAN_CHECK_ARG(g && y && dy && sync && gamma && beta && dgamma && dbeta, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && hw > 0 && (hw & 3) == 0, "bad dims (hw % 4 == 0)");
    SAN_CHECK_ARG((sc == nullptr) == (sh == nullptr), "scale/shift must come together");
    SAN_CHECK_ARG(((((uintptr_t)g | (uintptr_t)y | (uintptr_t)dy)) & 15) == 0, "16-byte aligned tensors");
    SAN_CHECK_ARG(g_coff >= 0 && g_coff + c <= g_ctot && y_coff >= 0 && y_coff + c <= y_ctot && d_coff >= 0 && d_coff + c <= d_ctot,
                  "bad channel view");
    const int v = cluster_v_for(hw >> 2, 7);
    const int K = san_cdiv(hw >> 2, kClT * v);
    SAN_CHECK_ARG((long long)K * n <= 256, "too many cluster members");
    return cluster_launch<1>(v, dim3(K, n, c), (hipStream_t)stream, g, g_ctot, g_coff, y, y_ctot, y_coff, sc, sh, slope, dy, d_ctot,
                             d_coff, hw, static_cast<unsigned*>(amax), G2Src{nullptr, 0, 0, 0.f, 1, 1.f}, DyDst{0, 0, 1, 1.f},
                             static_cast<unsigned*>(sync), gamma, beta, dgamma, dbeta, (double)n * (double)hw);
}

// uint32 words of one amax record (see san_common.h)
int san_amax_record_words(void) { return SAN_AMAX_WORDS; }

int san_dc_weight_grad(const float* g, const float* k, const float* k0, const float* mask, float* partial, int planes,
                       int h, int w, void* stream) {
    SAN_CHECK_ARG(g && k && k0 && mask && partial, "null pointer");
    SAN_CHECK_ARG(planes > 0 && h > 0 && w > 0, "bad dims");
    hipLaunchKernelGGL(dc_weight_grad_kernel, dim3(256), dim3(kThreads), 0, (hipStream_t)stream, (const float2*)g,
                       (const float2*)k, (const float2*)k0, mask, partial, (size_t)planes * h * w, w);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_sens_grad_acc(float* gs, const float* r_planar, const float* t1, const float* x, const float* gm_planar,
                      float sign1, int n, int c, int hw, void* stream) {
    SAN_CHECK_ARG(gs && r_planar && t1 && x && gm_planar, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && hw > 0, "bad dims");
    int bx = san_cdiv(hw, kThreads);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(sens_grad_acc_kernel, dim3(bx, n), dim3(kThreads), 0, (hipStream_t)stream, (float2*)gs, r_planar,
                       (const float2*)t1, (const float2*)x, gm_planar, sign1, c, hw);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_sens_grad_prop(float* gs, const float* r_planar, const float* t1, const float* x, const float* gm_planar,
                       float sign1, float* gd, const float* sens, int n, int c, int hw, void* stream) {
    SAN_CHECK_ARG(gm_planar && gd && sens, "null pointer");
    SAN_CHECK_ARG(!gs || (r_planar && t1 && x), "the sensitivity-map accumulation needs r, t1 and x");
    SAN_CHECK_ARG(n > 0 && c > 0 && hw > 0, "bad dims");
    // (a serial coil loop per pixel was 15 dependent read-modify-write round trips at 15 x 640 x 368: 78 us for 141 MB)
    int bx = san_cdiv(hw, kThreads);
    const int cap = c > 1 ? 2048 : 256;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(sens_grad_prop_kernel, dim3(bx, n, c > 64 ? 64 : c), dim3(kThreads), 0, (hipStream_t)stream, (float2*)gs, r_planar,
                       (const float2*)t1, (const float2*)x, gm_planar, sign1, (float2*)gd, (const float2*)sens, c, hw);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_sens_normalize_bwd(const float* est_planar, const float* gs, float* gest_planar, int n, int c, int hw,
                           void* stream) {
    SAN_CHECK_ARG(est_planar && gs && gest_planar, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && hw > 0, "bad dims");
    int bx = san_cdiv(hw, kThreads);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(sens_normalize_bwd_kernel, dim3(bx, n), dim3(kThreads), 0, (hipStream_t)stream, est_planar,
                       (const float2*)gs, gest_planar, c, hw);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_rss_bwd(const float* x, const float* y, const float* g, float* gx, int n, int c, int hw, int is_complex,
                void* stream) {
    SAN_CHECK_ARG(x && y && g && gx, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && hw > 0, "bad dims");
    int bx = san_cdiv(hw, kThreads);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(rss_bwd_kernel, dim3(bx, n), dim3(kThreads), 0, (hipStream_t)stream, x, y, g, gx, c, hw,
                       is_complex);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_ssim_loss_bwd(const float* x, const float* y, float* gy, float gscale, int n, int h, int w, float* ws,
                      void* stream) {
    return san_ssim_loss_bwd_dev(x, y, gy, gscale, nullptr, n, h, w, ws, stream);
}

int san_ssim_loss_bwd_dev(const float* x, const float* y, float* gy, float gscale, const float* gscale_dev, int n, int h,
                          int w, float* ws, void* stream) {
    SAN_CHECK_ARG(x && y && gy && ws, "null pointer");
    SAN_CHECK_ARG(n > 0 && h >= 7 && w >= 7, "image smaller than the 7x7 window");
    const int oh = h - 6, ow = w - 6;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ssim_bwd_coef_kernel, dim3(san_cdiv(ow, 32), san_cdiv(oh, 8), n), dim3(kThreads), 0, s, x, y, ws,
                       h, w, oh, ow);
    SAN_LAUNCH_CHECK();
    int bx = san_cdiv(h * w, kThreads);
    if (bx > 512) bx = 512;
    // loss = 1 - mean(S)  ->  dL/dS = -gscale [* gscale_dev[0]] / (n*oh*ow), formed in the kernel so that the host-scalar and
    // the device-scalar (autograd grad_output) forms give the same bits
    hipLaunchKernelGGL(ssim_bwd_gather_kernel, dim3(bx, n), dim3(kThreads), 0, s, x, y, ws, gy, n, h, w, oh, ow, gscale,
                       gscale_dev, (float)n * oh * ow);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

static int act_bwd_coef_impl(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff, const float* sc,
                             const float* sh, float slope, const float* coef, float* dy, int d_ctot, int d_coff, int n, int c,
                             int hw, unsigned* amax, void* stream) {
    SAN_CHECK_ARG(g && y && dy && coef, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && hw > 0, "bad dims");
    SAN_CHECK_ARG((sc == nullptr) == (sh == nullptr), "scale/shift must come together");
    SAN_CHECK_ARG(g_coff >= 0 && g_coff + c <= g_ctot && y_coff >= 0 && y_coff + c <= y_ctot && d_coff >= 0 &&
                      d_coff + c <= d_ctot, "bad channel view");
    int bx = san_cdiv(hw, kThreads * 4);
    long cap = 4096 / ((long)c * n);
    if (cap < 1) cap = 1;
    if (bx > cap) bx = (int)cap;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(act_bwd_coef_kernel, dim3(bx, c, n), dim3(kThreads), 0, (hipStream_t)stream, g, g_ctot, g_coff, y,
                       y_ctot, y_coff, sc, sh, slope, coef, dy, d_ctot, d_coff, c, hw, amax);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_act_bwd_coef(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff, const float* sc,
                     const float* sh, float slope, const float* coef, float* dy, int d_ctot, int d_coff, int n, int c,
                     int hw, void* stream) {
    return act_bwd_coef_impl(g, g_ctot, g_coff, y, y_ctot, y_coff, sc, sh, slope, coef, dy, d_ctot, d_coff, n, c, hw, nullptr, stream);
}

int san_act_bwd_coef_amax(const float* g, int g_ctot, int g_coff, const float* y, int y_ctot, int y_coff, const float* sc,
                          const float* sh, float slope, const float* coef, float* dy, int d_ctot, int d_coff, void* amax,
                          int n, int c, int hw, void* stream) {
    return act_bwd_coef_impl(g, g_ctot, g_coff, y, y_ctot, y_coff, sc, sh, slope, coef, dy, d_ctot, d_coff, n, c, hw,
                             static_cast<unsigned*>(amax), stream);
}

int san_warp_bwd_grid(const float* img, const float* grid, const float* g, float* g_off, int n, int c, int h, int w,
                      void* stream) {
    SAN_CHECK_ARG(img && grid && g && g_off, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && h > 0 && w > 0, "bad dims");
    int bx = san_cdiv(h * w, kThreads);
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(warp_bwd_grid_kernel, dim3(bx, n), dim3(kThreads), 0, (hipStream_t)stream, img, grid, g, g_off, c,
                       h, w);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_gradient_loss_bwd(const float* offset, float* g, float gscale, int accumulate, int n, int h, int w,
                          void* stream) {
    return san_gradient_loss_bwd_dev(offset, g, gscale, nullptr, accumulate, n, h, w, stream);
}

int san_gradient_loss_bwd_dev(const float* offset, float* g, float gscale, const float* gscale_dev, int accumulate, int n,
                              int h, int w, void* stream) {
    SAN_CHECK_ARG(offset && g, "null pointer");
    SAN_CHECK_ARG(n > 0 && h > 1 && w > 1, "bad dims");
    // loss = gscale/2 * (sum dx^2 / cnt_x + sum dy^2 / cnt_y);  d(dx^2)/ds = 2 dx
    int bx = san_cdiv(h * w, kThreads);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(gradient_loss_bwd_kernel, dim3(bx, n * 2), dim3(kThreads), 0, (hipStream_t)stream, offset, g, h, w,
                       gscale, gscale_dev, (float)n * h * (w - 1) * 2, (float)n * (h - 1) * w * 2, accumulate);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_bn_bwd_finalize(const float* part, const float* gamma, const float* beta, float* dgamma, float* dbeta, float* coef,
                        int n, int c, int tiles, double cnt, void* stream) {
    SAN_CHECK_ARG(part && gamma && beta && dgamma && dbeta && coef, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && tiles > 0 && cnt > 0, "bad dims");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(64), 0, (hipStream_t)stream, part, gamma, beta,
                       dgamma, dbeta, coef, n, c, tiles, cnt);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_bias_grad_from_stats(const float* part, float* db, int n, int c, int tiles, void* stream) {
    SAN_CHECK_ARG(part && db, "null pointer");
    SAN_CHECK_ARG(n > 0 && c > 0 && tiles > 0, "bad dims");
    hipLaunchKernelGGL(bias_grad_kernel, dim3(c), dim3(64), 0, (hipStream_t)stream, part, db, n, c, tiles);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_normunet_bwd_coefs(const float* part_b, const float* part_a, int tiles, const float* scale, const float* shift,
                           int x_ctot, const float* stdv, double nel, float* a_sc, float* a_sh, int g_ctot, float* m_sc,
                           float* m_sh, int b, void* stream) {
    SAN_CHECK_ARG(part_b && part_a && scale && shift && stdv && a_sc && a_sh && m_sc && m_sh, "null pointer");
    SAN_CHECK_ARG(b > 0 && tiles > 0 && x_ctot >= 2 && g_ctot >= 2 && nel > 1, "bad dims");
    const int width = g_ctot > x_ctot ? g_ctot : x_ctot;
    hipLaunchKernelGGL(normunet_bwd_coefs_kernel, dim3(san_cdiv(b * width, 64)), dim3(64), 0, (hipStream_t)stream, part_b,
                       part_a, tiles, scale, shift, x_ctot, stdv, nel, a_sc, a_sh, g_ctot, m_sc, m_sh, b);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

// NormUnet backward head / tail (normunet_bwd_head_kernel, normunet_bwd_tail_kernel): see the kernels.  part_b: fp32 [b, 2,
// san_bwd_stat_tiles(hw), 2]; part_x: [b, xc, tiles, 2] = san_plane_dot_stats over the first xc (2, or 3 with g_ref) channels of
// (g_xh, xin); gs / r_planar / t1 / xs may be NULL together (no sensitivity-map gradient), g_ref / db / dcw_part may be NULL.
int san_normunet_bwd_head(const float* g_out, const float* out_planar, const float* isd, const float* nshift, const float* stdv,
                          float* g_u, float* part_b, int b, int hw, void* stream) {
    SAN_CHECK_ARG(g_out && out_planar && isd && nshift && stdv && g_u && part_b, "null pointer");
    SAN_CHECK_ARG(b > 0 && hw > 0, "bad dims");
    const int tiles = san_bwd_stat_tiles(hw);
    hipLaunchKernelGGL(normunet_bwd_head_kernel, dim3(tiles, 2, b), dim3(kThreads), 0, (hipStream_t)stream, g_out, out_planar, isd,
                       nshift, stdv, g_u, part_b, hw, tiles);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_normunet_bwd_tail(const float* part_b, const float* part_x, int xc, const float* xin, int x_ctot, const float* x_scale,
                          const float* x_shift, const float* stdv, double nel, const float* g_xh, int g_ctot, float* gd,
                          const float* sens, float* gs, const float* r_planar, const float* t1, const float* xs, float sign1,
                          float* g_ref, int ref_accumulate, float* db, const float* dcw_part, int dcw_count, float dcw_scale,
                          float* dcw, int b, int c, int hw, void* stream) {
    SAN_CHECK_ARG(part_b && part_x && xin && x_scale && x_shift && stdv && g_xh && gd && sens, "null pointer");
    SAN_CHECK_ARG(!gs || (r_planar && t1 && xs), "the sensitivity-map accumulation needs r, t1 and x");
    SAN_CHECK_ARG(b > 0 && c > 0 && hw > 0 && x_ctot >= 2 && g_ctot >= 2 && (xc == 2 || xc == 3), "bad dims");
    SAN_CHECK_ARG(!g_ref || (xc == 3 && x_ctot >= 3 && g_ctot >= 3), "the reference channel's gradient needs three channels");
    SAN_CHECK_ARG(!dcw_part || (dcw && dcw_count > 0), "dc_weight partials without a destination");
    NuTailArgs a{};
    a.part_b = part_b;
    a.part_x = part_x;
    a.x = xin;
    a.x_sc = x_scale;
    a.x_sh = x_shift;
    a.stdv = stdv;
    a.gxh = g_xh;
    a.gd = (float2*)gd;
    a.S = (const float2*)sens;
    a.gS = (float2*)gs;
    a.r = r_planar;
    a.t1 = (const float2*)t1;
    a.xs = (const float2*)xs;
    a.g_ref = g_ref;
    a.db = db;
    a.dcw_part = dcw_part;
    a.dcw = dcw;
    a.nel = nel;
    a.sign1 = sign1;
    a.dcw_scale = dcw_scale;
    a.tiles = san_bwd_stat_tiles(hw);
    a.x_ctot = x_ctot;
    a.g_ctot = g_ctot;
    a.xc = xc;
    a.C = c;
    a.hw = hw;
    a.b = b;
    a.dcw_count = dcw_count;
    a.ref_acc = ref_accumulate;
    int bx = san_cdiv(hw, kThreads);
    const int cap = c > 1 ? 2048 : 256;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(normunet_bwd_tail_kernel, dim3(bx, b, c > 64 ? 64 : c), dim3(kThreads), 0, (hipStream_t)stream, a);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_bn_update_running(float* rmean, float* rvar, long long* num_batches_tracked, const float* bmean, const float* bvar,
                          int c, float momentum, float var_factor, void* stream) {
    SAN_CHECK_ARG(rmean && rvar && bmean && bvar, "null pointer");
    SAN_CHECK_ARG(c > 0, "bad dims");
    hipLaunchKernelGGL(bn_update_running_kernel, dim3(san_cdiv(c, 64)), dim3(64), 0, (hipStream_t)stream, rmean, rvar,
                       num_batches_tracked, bmean, bvar, c, momentum, var_factor);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_partials_add(const float* part, int count, float scale, float* dst, void* stream) {
    SAN_CHECK_ARG(part && dst && count > 0, "bad arguments");
    hipLaunchKernelGGL(partials_add_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, count, scale, dst);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

}  // extern "C"
