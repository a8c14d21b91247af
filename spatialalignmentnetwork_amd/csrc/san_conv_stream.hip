// 3x3 convolution, "stream" form (round 4): PERSISTENT workgroups for the high-resolution, few-channel layers
// (18 / 36 channels at 320^2 / 160^2: 60 % of a cascade's convolution time), where the one-tile-per-workgroup kernel of
// san_conv_bf16.hip spends its life in launch / prologue / load latency / barriers rather than in the matrix cores.
//
// Same arithmetic as conv_bf16x3_kernel<.., F16 = true, SWAP = true>: two fp16 parts per fp32 operand, three products per MAC on
// v_mfma_f32_16x16x32_f16 (fp32-equivalent: DESIGN.md 3.1), 32 x 8 output tiles, 4 waves, 24-channel chunks, 7 K-steps per
// chunk, the packed weight image of san_conv_bf16x3_pack (mode + 16), the same per-wave statistics records.  What changes:
//   * a workgroup walks a SEQUENCE of tiles (grid = resident workgroups only); index arithmetic, operand addressing and the
//     weight fetch happen once per workgroup instead of once per tile;
//   * the packed weights of ALL chunks live in LDS for the workgroup's lifetime (the partial last block of 16 output channels
//     -- 18 = 16 + 2, 36 = 32 + 4 -- is stored compacted: lanes of absent channels read one shared zero slot);
//   * the packed weights live in LDS: for the workgroup's lifetime when the layer has one 24-channel chunk, otherwise one chunk at
//     a time, brought in by 16-byte LDS-DMA (buffer_load_dwordx4 ... lds, from inline asm) behind the item's first barrier;
//   * the NEXT work item's raw fp32 input (32 values per thread) is requested into registers right after the staging barrier,
//     a whole K-loop + epilogue ahead of its use; with the weights out of the vector-memory queue nothing queues behind it.
// (v1 of this file brought the raw tile in by 4-byte LDS-DMA: 30 DMA instructions per wave and tile at ~100 cycles of issue
// each, plus 32 LDS reads per thread, cost more than the prefetch returned: scratch/attempts/r4_stream_v1_ablation.txt.)
// Reference work replaced: nn.Conv2d(3x3, padding 1, bias False) of varnet.py:139-146 (ConvBlock) and its autograd data gradient.
#include "san_common.h"

#include <cstdint>
#include <cstdlib>

namespace {

constexpr int kT = 256;
constexpr int kTW = 32, kTH = 8;
constexpr int kHP = kTW + 2, kHH = kTH + 2;      // halo tile 34 x 10
constexpr int kNP = kHP * kHH;                   // 340 staged pixels
constexpr int kCKC = 24;                         // input channels per chunk
constexpr int kPS = 48;                          // bytes per staged pixel per part
constexpr int kPartB = kNP * kPS;                // 16,320 B per part
constexpr int kSteps = 7;
constexpr int kUnits = 4;                        // (pixel, channel group) staging units per thread

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float fl2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(2))) _Float16 hf2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
union Frag {
    uint4 u;
    h8 h;
    bf8 v;
};

struct SArgs {
    const float* x;
    const float* in_scale;
    const float* in_shift;
    const uint4* wp;           // packed weights [chunk][step][block (nblkp)][part (3)][64 lanes] x 16 B, fp16 parts in 0, 1
    const float* bias;
    float* y;
    float* part;               // statistics [n][cout][tiles * 4][3] or null
    const uint32_t* amax;      // gradient input: amax record (san_common.h) or null
    const float* w_tail;       // fp16-format weights: {S_w, 1 / S_w} behind the packed image (NPRT = 2), else null
    float in_slope;
    int x_ctot, x_coff, cin;
    int y_ctot, y_coff, cout;
    int N, H, W;
    int tiles_x, tiles_y, chunks, nblkp;
    int total;                 // N * tiles_x * tiles_y
    unsigned x_bytes;          // extent of the whole input tensor (descriptor range)
    unsigned w_bytes;          // extent of the packed weight image
    unsigned m_nt, m_tx;       // multiply-high division by tiles_x * tiles_y and tiles_x (0: plain division)
    int stag, stag_mode;       // de-phasing of the workgroups that share a CU: initial delay = phase x stag x 64 cycles; phase from the
                               // launch order (mode 0: blockIdx / 256) or from the hardware wave slot (mode 1)
    int abl;                   // tuning hook (SAN_CONV_STREAM_ABL, results WRONG): 1 no DMA, 2 no staging arithmetic, 4 no K-loop, 8 no statistics, 16 no stores
};

__device__ __forceinline__ int fdiv(int x, unsigned m, int d) {
    if (d == 1) return x;
    return m ? (int)__umulhi((unsigned)x, m) : x / d;
}

__device__ __forceinline__ uint32_t cvt_pk_h(float f0, float f1) {
    const fl2 v = {f0, f1};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hf2));
}
__device__ __forceinline__ uint32_t cvt_pk_bf(float f0, float f1) {
    const fl2 v = {f0, f1};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
}
// a = a1 + a2 with a1 = fp16(a), a2 = fp16(a - a1) (san_conv_bf16.hip: split2h_pair)
__device__ __forceinline__ void split2h_pair(float f0, float f1, uint32_t& p1, uint32_t& p2) {
    p1 = cvt_pk_h(f0, f1);
    const hf2 h = __builtin_bit_cast(hf2, p1);
    p2 = cvt_pk_h(__builtin_fmaf((float)h[0], -1.f, f0), __builtin_fmaf((float)h[1], -1.f, f1));
}

// One LDS-DMA piece: 64 lanes x 16 bytes, lane l's 16 bytes land at LDS byte address lds_addr + 16 l.  Issued from asm: the
// compiler does not know about it (no vmcnt bookkeeping, no drain at barriers); M0 is saved and restored around it.
__device__ __forceinline__ void dma_x4(uint32_t lds_addr, uint32_t voff, v4i rs, uint32_t soff) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MBF full blocks of 16 output channels + (REM > 0) one partial block of REM channels; OCC = resident waves per SIMD the
// register allocation aims at
// NPRT = 2: two fp16 parts per operand, three products per MAC (the fp32-equivalent mode); NPRT = 1: ONE bf16 part, one product
// (san_set_conv_precision(1): BASELINE configs[1] as written; the packed image's leading bf16 part, activations rounded to bf16)
template <int MBF, int REM, int OCC, int NPRT = 2>
__global__ void __launch_bounds__(kT, OCC) conv3x3_stream_kernel(const SArgs a) {
    constexpr int MB = MBF + (REM > 0 ? 1 : 0);
    constexpr int RS = MBF * 1024 + (REM > 0 ? REM * 64 + 16 : 0);      // LDS bytes of one (step, part) weight region
    constexpr int RSL = RS / 16;                                        // ... in 16-byte slots
    constexpr int NSLOT = kSteps * NPRT * RSL;                          // slots of one chunk's weight image
    constexpr int NWI = (NSLOT + 63) / 64;                              // DMA instructions per chunk
    constexpr int NWK = (NWI + kT / 64 - 1) / (kT / 64);                // ... per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_a = smem;
    float* lds_aff = reinterpret_cast<float*>(smem + 2 * kPartB);
    unsigned char* lds_w = smem + 2 * kPartB + a.chunks * 192;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int nn = lane & 15, kg = lane >> 4;
    const int H = a.H, W = a.W, HWp = H * W;
    const int ntile = a.tiles_x * a.tiles_y;
    const bool has_aff = a.in_scale != nullptr;
    const bool multi = a.chunks > 1;                    // the weights are then streamed chunk by chunk
    const float lrelu_c = a.in_slope <= 1.f ? __builtin_inff() : -__builtin_inff();

    // ---- this workgroup's tiles: XCD x (= id & 7, the observed placement: speed only) owns the contiguous tile range
    // [x total / 8, (x + 1) total / 8); its workgroups take tiles round-robin, so that at any time an XCD's L2 holds a compact band
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int t_begin = (int)(((long long)a.total * xcd) >> 3), t_end = (int)(((long long)a.total * (xcd + 1)) >> 3);
    int t = t_begin + slot;
    if (t >= t_end) return;                             // (before any barrier: the whole workgroup leaves)

    // ---- co-resident workgroups start out of phase: identical workgroups otherwise march in lockstep through load / stage /
    // matrix / store phases (measured: the phases of a launch ADD UP), and a persistent workgroup keeps the offset it starts with
    if (a.stag > 0) {
        const int phase = a.stag_mode == 0 ? (int)(blockIdx.x >> 8) : (int)(__builtin_amdgcn_s_getreg((3 << 11) | 4) % (unsigned)a.stag_mode);   // HW_ID[3:0] = wave slot
        for (int i = 0; i < phase * a.stag; ++i) __builtin_amdgcn_s_sleep(1);
    }

    // ---- staging units: slots 0..2 = pixel tid of channel group s (wave-uniform group); slot 3 = pixels 256..339 x 3 groups
    const int pr0 = tid / kHP, pc0 = tid - pr0 * kHP;
    const bool u3 = tid < 3 * (kNP - kT);
    const int g3 = u3 ? tid / (kNP - kT) : 0;
    const int p3 = u3 ? kT + tid - g3 * (kNP - kT) : 0;
    const int pr3 = p3 / kHP, pc3 = p3 - pr3 * kHP;

    // whole-tensor buffer descriptor: load address = base + SGPR offset (the channel plane: uniform for slots 0..2) + per-lane pixel offset
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const unsigned hw4 = (unsigned)HWp * 4u;

    int n = fdiv(t, a.m_nt, ntile);
    int tile = t - n * ntile;
    int ty = fdiv(tile, a.m_tx, a.tiles_x), tx = tile - ty * a.tiles_x;
    int chunk = 0;

    // ---- prefetch registers: the raw input of the NEXT item (st), and whether this thread's two pixels lie inside the image
    float st[kUnits][8];
    bool in0 = false, in3 = false;
    auto prefetch = [&](int n_, int ty_, int tx_, int chunk_) {
        const int gy0 = ty_ * kTH - 1 + pr0, gx0 = tx_ * kTW - 1 + pc0;
        const int gy3 = ty_ * kTH - 1 + pr3, gx3 = tx_ * kTW - 1 + pc3;
        in0 = gy0 >= 0 && gy0 < H && gx0 >= 0 && gx0 < W;
        in3 = u3 && gy3 >= 0 && gy3 < H && gx3 >= 0 && gx3 < W;
        const unsigned v0 = in0 ? (unsigned)(gy0 * W + gx0) * 4u : 0u, v3 = in3 ? (unsigned)(gy3 * W + gx3) * 4u : 0u;
        const int nch = min(kCKC, a.cin - chunk_ * kCKC);
        const unsigned cbase = (unsigned)(n_ * a.x_ctot + a.x_coff + chunk_ * kCKC);
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                if (s * 8 + i < nch) {                  // (wave-uniform; nch is even for every layer of the network, odd: clamped)
                    st[s][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)v0, (int)((cbase + s * 8 + i) * hw4), 0));
                    st[s][i + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)v0, (int)((cbase + min(s * 8 + i + 1, nch - 1)) * hw4), 0));
                } else {
                    st[s][i] = 0.f;
                    st[s][i + 1] = 0.f;
                }
            }
#pragma unroll
        for (int i = 0; i < 8; ++i)                     // (per-lane group: channels past the chunk's last read its last one and meet zero weights)
            st[3][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(v3 + (unsigned)min(g3 * 8 + i, nch - 1) * hw4), (int)(cbase * hw4), 0));
    };
    prefetch(n, ty, tx, 0);

    // ---- once per workgroup: zero the operand image (absent channel groups stay zero = finite)
    for (int i = tid; i < 2 * kPartB / 16; i += kT) reinterpret_cast<uint4*>(lds_a)[i] = make_uint4(0u, 0u, 0u, 0u);

    // ---- weights: LDS image of a chunk = 14 (step, part) regions of RSL slots: the MBF full blocks lane for lane, then the
    // partial block's REM channels x 4 k-groups, then one zero slot that the lanes of its absent channels read.  Slot q of the
    // image is fetched by DMA instruction q / 64, lane q % 64, from the packed image's 16-byte element woff (within the chunk).
    const uint64_t wa_ = reinterpret_cast<uint64_t>(a.wp);
    const v4i wrs = {__builtin_amdgcn_readfirstlane((int)(uint32_t)wa_), __builtin_amdgcn_readfirstlane((int)(uint32_t)(wa_ >> 32)),
                     __builtin_amdgcn_readfirstlane((int)a.w_bytes), 0x00020000};
    const uint32_t w_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_w;
    uint32_t wvoff[NWK];
    bool wok[NWK];
#pragma unroll
    for (int k = 0; k < NWK; ++k) {
        const int q = (wave + k * (kT / 64)) * 64 + lane;
        wok[k] = q < NSLOT;
        const int qq = min(q, NSLOT - 1);
        const int r = qq / RSL, j = qq - r * RSL;
        const int step = r / NPRT, part = r - step * NPRT;
        int m, l;
        if (j < MBF * 64) {
            m = j >> 6;
            l = j & 63;
        } else if (REM > 0 && j < MBF * 64 + 4 * REM) {
            const int jj = j - MBF * 64;
            m = MBF;
            l = (jj / (REM > 0 ? REM : 1)) * 16 + jj % (REM > 0 ? REM : 1);
        } else {                                        // the zero slot: a lane of an absent channel of the partial block
            m = MBF;
            l = 15;
        }
        wvoff[k] = (uint32_t)(((step * a.nblkp + m) * 3 + part) * 64 + l) * 16u;
    }
    const uint32_t wchunk_bytes = (uint32_t)(kSteps * a.nblkp * 3 * 64) * 16u;
    auto issue_w = [&](int chunk_) {
        const uint32_t soff = (uint32_t)chunk_ * wchunk_bytes;
#pragma unroll
        for (int k = 0; k < NWK; ++k)
            if (wave + k * (kT / 64) < NWI) {           // (wave-uniform)
                if (wok[k]) dma_x4(w_lds + (uint32_t)(wave + k * (kT / 64)) * 1024u, wvoff[k], wrs, soff);
            }
    };
    if (!multi) issue_w(0);

    // gradient input in the fp16 format: x S rides in the affine table, the accumulators get 1 / S (san_conv_bf16.hip)
    float inS = 1.f, inInvS = 1.f;
    if (a.amax) {
        const uint32_t b = san_amax_read(a.amax);
        int e = (int)((b >> 23) & 255u);
        if (b != 0u) {
            e = e < 14 ? 14 : (e > 250 ? 250 : e);
            inS = __builtin_bit_cast(float, (uint32_t)(267 - e) << 23);
            inInvS = __builtin_bit_cast(float, (uint32_t)(e - 13) << 23);
        }
    }
    if (NPRT == 2 && a.w_tail) inInvS *= a.w_tail[1];    // fp16-format weights may be stored x S_w (round 6): exact power of two
    // lazy-affine table of image n: [chunk][scale 24 | shift 24]
    auto load_aff = [&](int n_) {
        if (tid < a.chunks * 48) {
            const int c = tid / 48, j = tid - c * 48;
            const int ci = min(c * kCKC + (j < 24 ? j : j - 24), a.cin - 1);
            const float* src = j < 24 ? a.in_scale : a.in_shift;
            lds_aff[tid] = has_aff ? src[n_ * a.x_ctot + a.x_coff + ci] * inS : (j < 24 ? inS : 0.f);
        }
    };
    load_aff(n);

    // ---- K-loop operand addressing (as conv_bf16x3_kernel, 32 x 8 tile)
    int tapoff[kSteps];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
        const int g = min(4 * s + kg, 26);
        const int tap = g / 3, chg = g - 3 * tap;
        const int ky = tap / 3, kx = tap - 3 * ky;
        tapoff[s] = (ky * kHP + kx) * kPS + chg * 16;
    }
    int boff[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) boff[b] = ((2 * wave + (b >> 1)) * kHP + 16 * (b & 1) + nn) * kPS;
    const int wl_full = lane * 16;
    const int wl_rem = REM > 0 ? (nn < REM ? MBF * 1024 + (kg * REM + nn) * 16 : MBF * 1024 + REM * 64) : 0;
    // The partial block's products (x hi, w hi) and (x hi, w lo) share ONE MFMA: columns 0 .. REM-1 of its weight operand are the
    // channels' hi parts, columns REM .. 2 REM-1 their lo parts (read from the part-1 region: + RS), the rest the zero slot; the
    // epilogue adds column c + REM into column c.  2 instead of 3 MFMAs per K-step and pixel block for those channels.
    const int wl_mix = (REM > 0 && NPRT == 2) ? (nn < REM ? wl_rem : (nn < 2 * REM ? RS + MBF * 1024 + (kg * REM + nn - REM) * 16 : MBF * 1024 + REM * 64)) : 0;
    static_assert(2 * REM <= 16, "hi and lo columns of the partial block must fit one MFMA");

    f4 acc[MB][4];
    float biasv[MB];                                    // this lane's channel of every block: fetched once, not once per tile
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        biasv[m] = (a.bias && 16 * m + nn < a.cout) ? a.bias[16 * m + nn] : 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[m][b] = f4{0.f, 0.f, 0.f, 0.f};
    }

    if (!multi) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the resident weight image (and, once, the first prefetch)
    for (;;) {
        // ---- every wave has left the previous item's K-loop: the operand image (and the streamed weight chunk) may be rewritten
        lds_barrier();
        if (multi) issue_w(chunk);
        const int nch = min(kCKC, a.cin - chunk * kCKC);
        const float* afc = lds_aff + chunk * 48;
        // registers -> LDS: lazy affine + LeakyReLU (one v_med3 per element), split into two fp16 parts, [pixel][channel] image
        auto stage_unit = [&](const float (&raw)[8], int g, int p, bool in, int npair) {
            unsigned char* dst = lds_a + p * kPS + g * 16;
            if (in) {
                const f4 sc0 = *reinterpret_cast<const f4*>(afc + g * 8), sc1 = *reinterpret_cast<const f4*>(afc + g * 8 + 4);
                const f4 sh0 = *reinterpret_cast<const f4*>(afc + 24 + g * 8), sh1 = *reinterpret_cast<const f4*>(afc + 24 + g * 8 + 4);
                uint32_t q1[4], q2[4];
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    q1[i >> 1] = 0u;
                    q2[i >> 1] = 0u;
                    if ((i >> 1) < npair) {
                        const float a0 = __builtin_fmaf(raw[i], i < 4 ? sc0[i] : sc1[i - 4], i < 4 ? sh0[i] : sh1[i - 4]);
                        const float a1 = __builtin_fmaf(raw[i + 1], i < 4 ? sc0[i + 1] : sc1[i - 3], i < 4 ? sh0[i + 1] : sh1[i - 3]);
                        const float v0 = __builtin_amdgcn_fmed3f(a0, a0 * a.in_slope, lrelu_c);
                        const float v1 = __builtin_amdgcn_fmed3f(a1, a1 * a.in_slope, lrelu_c);
                        if constexpr (NPRT == 2)
                            split2h_pair(v0, v1, q1[i >> 1], q2[i >> 1]);
                        else
                            q1[i >> 1] = cvt_pk_bf(v0, v1);
                    }
                }
                *reinterpret_cast<uint4*>(dst) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
                if constexpr (NPRT == 2) *reinterpret_cast<uint4*>(dst + kPartB) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
            } else {
                *reinterpret_cast<uint4*>(dst) = make_uint4(0u, 0u, 0u, 0u);
                if constexpr (NPRT == 2) *reinterpret_cast<uint4*>(dst + kPartB) = make_uint4(0u, 0u, 0u, 0u);
            }
        };
        if (!(a.abl & 2)) {
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (s * 8 < nch) stage_unit(st[s], s, tid, in0, min(4, (nch - s * 8 + 1) >> 1));     // (wave-uniform pair count)
            if (u3 && g3 * 8 < nch) stage_unit(st[3], g3, p3, in3, 4);
        }
        if (multi) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the weight chunk have landed
        lds_barrier();

        // ---- the staging registers are free: request the next item's input
        const int y0 = ty * kTH, x0 = tx * kTW;
        int nn_ = n, nty = ty, ntx = tx, nchunk = chunk + 1;
        bool more = true;
        if (nchunk == a.chunks) {
            nchunk = 0;
            t += per_xcd;
            more = t < t_end;
            if (more) {
                nn_ = fdiv(t, a.m_nt, ntile);
                const int tl = t - nn_ * ntile;
                nty = fdiv(tl, a.m_tx, a.tiles_x);
                ntx = tl - nty * a.tiles_x;
                if (nn_ != n) load_aff(nn_);            // (every wave is past its last read of the table: the barrier above)
            }
        }
        if (more && !(a.abl & 1)) prefetch(nn_, nty, ntx, nchunk);

        // ---- 7 K-steps on the staged image; operands of step s + 1 are requested before the MFMAs of step s.  Weight operands
        // of a step: [m][0] = hi parts, [m][1] = lo parts of the full blocks; the partial block has [MBF][0] = hi and [MBF][1] =
        // the mixed hi | lo operand (wl_mix)
        if (!(a.abl & 4)) {
            Frag wa[2][MB][NPRT], xa[2][4][NPRT];
            auto load_w = [&](int s, Frag (&wq)[MB][NPRT]) {
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int p = 0; p < NPRT; ++p)
                        wq[m][p].u = *reinterpret_cast<const uint4*>(lds_w + (m < MBF ? (s * NPRT + p) * RS + m * 1024 + wl_full
                                                                                        : s * NPRT * RS + (p == 0 ? wl_rem : wl_mix)));
            };
            auto load_x = [&](int s, Frag (&xq)[4][NPRT]) {
                const int to = tapoff[s];
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int p = 0; p < NPRT; ++p) xq[b][p].u = *reinterpret_cast<const uint4*>(lds_a + p * kPartB + boff[b] + to);
            };
            load_w(0, wa[0]);
            load_x(0, xa[0]);
#pragma unroll
            for (int s = 0; s < kSteps; ++s) {
                if (s + 1 < kSteps) {
                    load_w(s + 1, wa[(s + 1) & 1]);
                    load_x(s + 1, xa[(s + 1) & 1]);
                }
                if constexpr (NPRT == 1) {
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[s & 1][b][0].v, wa[s & 1][m][0].v, acc[m][b], 0, 0, 0);
                } else {
                    // (x hi, w hi | mixed), (x lo, w hi), (x hi, w lo of the full blocks)
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[s & 1][b][0].h, wa[s & 1][m][m < MBF ? 0 : NPRT - 1].h, acc[m][b], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[s & 1][b][NPRT - 1].h, wa[s & 1][m][0].h, acc[m][b], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MBF; ++m)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[s & 1][b][0].h, wa[s & 1][m][NPRT - 1].h, acc[m][b], 0, 0, 0);
                }
            }
        }

        // ---- last chunk of the tile: epilogue.  acc[m][b][r] = channel 16 m + nn at tile pixel 64 wave + 16 b + 4 kg + r
        if (chunk == a.chunks - 1) {
            if constexpr (REM > 0 && NPRT == 2) {       // column c + REM (x hi . w lo) into column c: a row shift by REM lanes
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    // (scalars, each shift behind an opaque barrier: on the vector elements hipcc 7.0 shifted element 0 four times)
                    float e[4] = {acc[MBF][b][0], acc[MBF][b][1], acc[MBF][b][2], acc[MBF][b][3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        asm volatile("" : "+v"(e[r]));
                        const float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, e[r]), 0x100 + REM, 0xf, 0xf, false));
                        e[r] += t;
                    }
                    acc[MBF][b] = f4{e[0], e[1], e[2], e[3]};
                }
            }
            if (a.amax || a.w_tail) {                   // (1 / S of a gradient input and / or 1 / S_w of scaled fp16 weights)
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[m][b] = acc[m][b] * inInvS;
            }
            if (a.bias) {
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[m][b] += f4{biasv[m], biasv[m], biasv[m], biasv[m]};
            }
            if (a.part && !(a.abl & 8)) {
                const int tiles = ntile * 4;
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const float pilot = __shfl(acc[m][0][0], nn, 64);
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = acc[m][b][r] - pilot;
                            s1 += e;
                            s2 = fmaf(e, e, s2);
                        }
                    s1 += __shfl_xor(s1, 16, 64);
                    s2 += __shfl_xor(s2, 16, 64);
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    const int co = 16 * m + nn;
                    if (kg == 0 && co < a.cout) {
                        float* o = a.part + ((size_t)(n * a.cout + co) * tiles + tile * 4 + wave) * 3;
                        o[0] = 64.f;
                        o[1] = pilot + s1 * (1.f / 64.f);
                        o[2] = fmaxf(s2 - s1 * s1 * (1.f / 64.f), 0.f);
                    }
                }
            }
            float* ybase = a.y + (size_t)(n * a.y_ctot + a.y_coff) * HWp;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const int co = 16 * m + nn;
                float* dst = ybase + (size_t)co * HWp;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int q = 64 * wave + 16 * b + 4 * kg;
                    if (co < a.cout && !(a.abl & 16)) *reinterpret_cast<f4*>(dst + (y0 + (q >> 5)) * W + x0 + (q & 31)) = acc[m][b];
                    acc[m][b] = f4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (!more) break;
        n = nn_;
        ty = nty;
        tx = ntx;
        tile = ty * a.tiles_x + tx;
        chunk = nchunk;
    }
}

int g_stream = 1;              // SAN_CONV_STREAM=0: off
int g_stream_wgs = 0;          // SAN_CONV_STREAM_WGS: workgroups per CU (0: as many as the LDS allows, at most 2)
int g_stream_abl = 0;
int g_stream_stag = 0, g_stream_stag_mode = 0;
struct StreamEnv {
    StreamEnv() {
        if (const char* e = getenv("SAN_CONV_STREAM")) g_stream = atoi(e);
        if (const char* e = getenv("SAN_CONV_STREAM_WGS")) g_stream_wgs = atoi(e);
        if (const char* e = getenv("SAN_CONV_STREAM_ABL")) g_stream_abl = atoi(e);
        if (const char* e = getenv("SAN_CONV_STREAM_STAG")) g_stream_stag = atoi(e);
        if (const char* e = getenv("SAN_CONV_STREAM_STAGMODE")) g_stream_stag_mode = atoi(e);
    }
} g_stream_env;

size_t stream_lds_bytes(int cin, int cout, int nprt = 2) {
    const int chunks = san_cdiv(cin, kCKC);
    const int mbf = cout / 16, rem = cout % 16;
    const size_t rs = (size_t)mbf * 1024 + (rem ? rem * 64 + 16 : 0);
    return 2 * (size_t)kPartB + (size_t)chunks * 192 + (size_t)kSteps * nprt * rs;     // operand image, affine table, one chunk of weights
}

template <int MBF, int REM, int OCC, int NPRT = 2>
int launch_stream(const SArgs& a, hipStream_t s) {
    const size_t lds = stream_lds_bytes(a.cin, a.cout, NPRT);
    static size_t configured_on[128] = {};      // per device: the attribute is a per-device setting (round 6)
    const int dev__ = san_current_device();
    size_t& configured = configured_on[dev__ >= 0 && dev__ < 128 ? dev__ : 0];
    if (lds > configured || dev__ < 0 || dev__ >= 128) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_stream_kernel<MBF, REM, OCC, NPRT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            san_set_error("cannot reserve %d bytes of LDS for the stream convolution", (int)lds);
            return SAN_E_UNSUPPORTED;
        }
        configured = lds;
    }
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > OCC) per_cu = OCC;
    if (g_stream_wgs > 0 && g_stream_wgs < per_cu) per_cu = g_stream_wgs;
    int grid = 256 * per_cu;
    const int need = ((a.total + 7) / 8) * 8;           // (a multiple of 8: the tile ranges are per XCD)
    if (grid > need) grid = need;
    hipLaunchKernelGGL((conv3x3_stream_kernel<MBF, REM, OCC, NPRT>), dim3(grid), dim3(kT), lds, s, a);
    return SAN_OK;
}

}  // namespace

extern "C" {

// tests / tuning: 0 = every layer on the one-tile-per-workgroup kernel, 1 = the stream form where eligible
int san_conv_stream_set_tuning(int on) {
    g_stream = on ? 1 : 0;
    return SAN_OK;
}

// 1 when the stream form takes this fp16-format 3x3 layer (called by conv_bf16x3_run, san_conv_bf16.hip)
int san_conv_stream_eligible(int n, int h, int w, int cin, int cout, int x_ctot) {
    if (!g_stream) return 0;
    if ((w % kTW) != 0 || (h % kTH) != 0 || w < 64 || h * w < 160 * 160) return 0;
    if (cin > 4 * kCKC) return 0;
    // (round 6: 2 / 3 output channels as a partial block ALONE (MBF = 0) -- the data gradient of a cascade's input convolution,
    //  18 -> 3 @320^2: 30.8 us against 39.5 (50 with a cold input) on the direct fp32 kernel; SAN_STREAM_SMALL_COUT=0: off)
    static const bool small = !(getenv("SAN_STREAM_SMALL_COUT") && atoi(getenv("SAN_STREAM_SMALL_COUT")) == 0);
    if (!(cout == 18 || cout == 32 || cout == 36 || cout == 16 || cout == 48 || (small && (cout == 2 || cout == 3)))) return 0;
    if ((unsigned long long)n * x_ctot * h * w * 4ull >= 0x7fffffffull) return 0;
    return stream_lds_bytes(cin, cout) <= 160 * 1024 ? 1 : 0;      // everything resident
}

}  // extern "C"

int san_conv_stream_run(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift, float in_slope,
                        const void* w_packed, int nblkp, const float* bias, float* y, int y_ctot, int y_coff, int cout, float* part_stats,
                        const void* amax, int n, int h, int w, void* stream, int nprt, const float* w_tail) {
    SArgs a{};
    a.w_tail = w_tail;
    a.x = x;
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_slope = in_slope;
    a.wp = static_cast<const uint4*>(w_packed);
    a.bias = bias;
    a.y = y;
    a.part = part_stats;
    a.amax = static_cast<const uint32_t*>(amax);
    a.x_ctot = x_ctot;
    a.x_coff = x_coff;
    a.cin = cin;
    a.y_ctot = y_ctot;
    a.y_coff = y_coff;
    a.cout = cout;
    a.N = n;
    a.H = h;
    a.W = w;
    a.tiles_x = w / kTW;
    a.tiles_y = h / kTH;
    a.chunks = san_cdiv(cin, kCKC);
    a.nblkp = nblkp;
    a.total = n * a.tiles_x * a.tiles_y;
    a.x_bytes = (unsigned)((size_t)n * x_ctot * h * w * 4);
    a.w_bytes = (unsigned)((size_t)a.chunks * kSteps * nblkp * 3 * 64 * 16);
    auto magic = [&](int d) -> unsigned {
        return (d > 1 && (unsigned long long)a.total * (unsigned long long)d < 0xffffffffull) ? (unsigned)(0x100000000ull / (unsigned)d + 1) : 0u;
    };
    a.m_nt = magic(a.tiles_x * a.tiles_y);
    a.m_tx = magic(a.tiles_x);
    a.abl = g_stream_abl;
    a.stag = g_stream_stag;
    a.stag_mode = g_stream_stag_mode;
    SAN_CHECK_ARG((reinterpret_cast<uintptr_t>(y) & 15) == 0, "stream convolution: output must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (nprt == 1) {                                    // one bf16 part (the layers of the network only)
        switch (cout) {
            case 18: rc = launch_stream<1, 2, 2, 1>(a, s); break;
            case 36: rc = launch_stream<2, 4, 2, 1>(a, s); break;
            default: san_set_error("stream convolution: unsupported cout %d for the one-part form", cout); return SAN_E_UNSUPPORTED;
        }
    } else {
        switch (cout) {
            case 2: rc = launch_stream<0, 2, 2>(a, s); break;
            case 3: rc = launch_stream<0, 3, 2>(a, s); break;
            case 16: rc = launch_stream<1, 0, 2>(a, s); break;
            case 18: rc = launch_stream<1, 2, 2>(a, s); break;
            case 32: rc = launch_stream<2, 0, 2>(a, s); break;
            case 36: rc = launch_stream<2, 4, 2>(a, s); break;
            case 48: rc = launch_stream<3, 0, 2>(a, s); break;
            default: san_set_error("stream convolution: unsupported cout %d", cout); return SAN_E_UNSUPPORTED;
        }
    }
    if (rc != SAN_OK) return rc;
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}
