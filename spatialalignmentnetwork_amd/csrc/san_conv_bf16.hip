// 3x3 convolution on the CDNA4 bf16 matrix cores with fp32-level accuracy ("bf16x3"): every fp32
// operand is split into three bf16 parts a = a1 + a2 + a3 (each the round-to-nearest bf16 of what is
// left), and a*w is evaluated as the six products a1w1 + a1w2 + a2w1 + a1w3 + a2w2 + a3w1 with
// v_mfma_f32_16x16x32_bf16, accumulated in fp32.  The dropped terms are O(2^-24): measured on the
// whole 12-cascade network the result moves 2.2e-5 (relative L2) from the fp32 path, inside the
// reference's own fp32-vs-fp64 distance (scratch/study/split_bf16_accuracy.py).  Six bf16 MFMAs cost
// 6 x 16 cycles for 16 x 16 x 32 MACs = 2.7x less matrix-pipe time than the fp32 4x4x1 form, and the
// 16-cycle instructions leave the LDS / VALU work room (scratch/probe/mfma_bf16.hip: 290-370 TF
// fp32-equivalent for the K-step below, against 137 TF for fp32 MFMAs alone).
//
// GEMM view (im2col never materialised):  D[co][pixel] += sum_k W[co][k] * A[k][pixel],
//   M = 16 output channels per MFMA (rows), N = 16 pixels of one image row (columns),
//   k = (tap, input channel), consumed in GROUPS of 8 consecutive channels of one tap: a lane's
//   8 bf16 operand elements are 8 channels of one pixel, i.e. one 16-byte LDS read from a
//   [pixel][channel] image of the staged input tile.
// Used for every 3x3 layer from 18 -> 18 channels upwards (san_conv_bf16x3_eligible) and, as KS = 1, for the 1x1
// convolutions, the 2x2 stride-2 transposed convolutions (1x1 to 4 cout virtual channels + pixel shuffle in the
// epilogue) and their data gradients; the 2-/3-/8-channel layers stay on the fp32 4x4x1 kernel (san_conv_mfma.hip).
//
// Workgroup = 4 waves, output tile 32 x 8 pixels, MB blocks of 16 output channels.  Wave w owns
// rows 2w, 2w+1 = four 16-pixel blocks and all MB channel blocks: 4*MB accumulator tiles.  Input
// channels are staged 24 at a time (three groups of 8): 3 parts x 340 halo pixels x 48 B in LDS
// (pixel stride 48 B = 3 x 16 B: conflict-free b128 reads and writes), next to the chunk's packed
// weights, which are stored in HBM exactly as the lanes read them.  A chunk = 27 (tap, group) pairs =
// 7 K-steps of 4 groups (one zero group of padding).  In the weights-direct form (WD, MB <= 4) the weights skip
// LDS and are read by the K-steps straight from L2, which leaves room for three workgroups per CU.
#include "san_common.h"

#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

void san_wgrad_set_parts(int parts);             // san_wgrad_bf16.hip
// san_conv_stream.hip: the persistent form for the high-resolution few-channel layers (fp16-format weights)
int san_conv_stream_run(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift, float in_slope,
                        const void* w_packed, int nblkp, const float* bias, float* y, int y_ctot, int y_coff, int cout, float* part_stats,
                        const void* amax, int n, int h, int w, void* stream, int nprt, const float* w_tail);

// Tuning switches, both measured and left off (scratch/README.md, round 2): HALFX = the K-loop's activation operand reads run
// half a K-step ahead (two of a wave's four pixel blocks per set: 32 operand registers fewer) -- neutral, the register peak is
// in the staging phase; PIN4 = register allocation pinned to four resident workgroups per CU for the MB = 2 weights-direct
// forms -- 18->18 @320^2 -3 %, every multi-chunk layer +8..25 % slower.
#ifndef SAN_B16_HALFX
#define SAN_B16_HALFX 0
#endif
#ifndef SAN_B16_PIN4
#define SAN_B16_PIN4 0
#endif
#if SAN_B16_PIN4
// resident workgroups per CU the register allocation aims at: four for the small weights-direct forms (<= 128 registers per lane)
#define SAN_B16_MINWG(MB, WD, KS, NP) (((WD) && (MB) == 2 && (NP) <= 2) ? 4 : 1)
#else
// SAN_B16_VGPRFORM=1: promise two resident workgroups for the weights-direct forms, so that (register budget <= 256 per lane)
// the compiler selects the VGPR form of the MFMAs and the accumulators never visit AGPRs.  Measured (round 3, same box, N = 8):
// 5-18 % SLOWER on every layer (18->18 @320^2 54.5 -> 57.4 us, 128->64 @160^2 123.6 -> 145.7) although the wave executes 17 % fewer
// VALU instructions: with the accumulators in AGPRs the matrix pipe does not compete with the other waves' VALU operand
// reads for VGPR ports.  Left off; the epilogue's redundant AGPR passes were removed at the source instead.
#ifndef SAN_B16_VGPRFORM
#define SAN_B16_VGPRFORM 0
#endif
#define SAN_B16_MINWG(MB, WD, KS, NP) ((SAN_B16_VGPRFORM && (WD) && (MB) <= 4) ? 2 : 1)
#endif

namespace {

constexpr int kT = 256;
constexpr int kTW = 32, kTH = 8;                 // output tile
constexpr int kHW_ = kTW + 2, kHH = kTH + 2;     // halo tile 34 x 10
constexpr int kNP = kHW_ * kHH;                  // 340 staged pixels
constexpr int kCKC = 24;                         // input channels per chunk (3 groups of 8)
constexpr int kPS = 48;                          // bytes per staged pixel per part
constexpr int kPartB = kNP * kPS;                // 16,320 B per part
constexpr int kSteps = 7;                        // K-steps of 4 groups per chunk (27 + 1 pad)
constexpr int kUnits = (kNP * 3 + kT - 1) / kT;  // (pixel, group) staging units per thread: 4

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
union Frag {
    uint4 u;
    bf8 v;
    h8 h;
    long l[2];                 // l[0]: the 8 fp8 values of the one-part fp8 format
};

struct BArgs {
    const float* x;
    const float* in_scale;
    const float* in_shift;
    int shuffle;               // 1: transposed convolution 2x2 s2 = 1x1 conv to 4 cout virtual channels + pixel shuffle (KS = 1)
    const uint4* wp;           // packed weights [chunk][step][block of 16 couts][part][64 lanes] x 16 B
    const float* bias;
    float* y;
    float* part;               // statistics tiles [n][cout][tiles][3] or null
    float in_slope;
    int x_ctot, x_coff, cin;
    int y_ctot, y_coff, cout;
    int N, H, W;
    int tiles_x, tiles_y, cgs, chunks, nblkp;   // nblkp: channel blocks in the packed image (>= cgs * MB)
    int S;                     // split-K: S workgroups share an output tile, each takes a contiguous range of the chunks
    float* ws;                 // [S][N][cout][H][W] partial outputs (S > 1); splitk_reduce_kernel adds them up
    int nbw;                   // host only: 16-pixel blocks per wave of the launch (4, or 3 = the short-tile form, see tile_plan)
    int tw, th, hp, npx;       // tile width / height (tw * th <= 256 pixels, taken in flattened order), halo pitch tw + 2, halo pixels
    // Exact division by run-time constants without the ~25-instruction software divide (eight of them open every workgroup;
    // scratch/probe/wg_floor.hip: a dozen scalar divisions alone cost 1.7 us per 3,200-workgroup launch): q = umulhi(x, m)
    // with m = 2^32 / d + 1, exact while x d < 2^32 (checked by the host, which otherwise leaves m = 0 = "divide").
    // m_*: S, cgs, cgs * ntile, tiles_x, and for the run-time tile shapes hp, npx - 256, tw.
    unsigned m_S, m_cgs, m_cgsnt, m_tx, m_hp, m_rem, m_tw;
    int fmt;                   // operand format of the packed weights / the staging: 0 = bf16 parts, 1 = two fp16 parts
    unsigned long long* dbg;   // tuning hook (san_conv_bf16x3_debug_timeline): per workgroup 8 x u64 = 100 MHz clock at start, first
                               // chunk staged, epilogue start, end; HW_ID; XCC_ID -- null in normal use
    const uint32_t* amax;      // fp16 format on a GRADIENT input: the tensor's amax record (san_common.h); the input is scaled by a power of two
    const float* f8_tail;      // fp8 and fp16 formats: {S_w, 1 / S_w} behind the packed image (the per-tensor power-of-two weight scale)
};

// x / d for 0 <= x, d >= 1 with the host's multiplier m (BArgs.m_*); m == 0: plain division
__device__ __forceinline__ int fdiv(int x, unsigned m, int d) {
    if (d == 1) return x;
    return m ? (int)__umulhi((unsigned)x, m) : x / d;
}

// round-to-nearest-even bf16 of f, returned as the fp32 it represents (upper 16 bits)
__device__ __forceinline__ float bf16_round(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return __builtin_bit_cast(float, u & 0xffff0000u);
}

// f -> three bf16 parts (bit patterns) with p1 + p2 + p3 == f to ~2^-24 relative
__device__ __forceinline__ void split3(float f, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    const float a = bf16_round(f);
    const float r1 = f - a;
    const float b = bf16_round(r1);
    const float c = bf16_round(r1 - b);
    p1 = __builtin_bit_cast(uint32_t, a) >> 16;
    p2 = __builtin_bit_cast(uint32_t, b) >> 16;
    p3 = __builtin_bit_cast(uint32_t, c) >> 16;
}

// two values at once on the hardware converter: v_cvt_pk_bf16_f32 rounds to nearest even and packs (lo = f0, hi = f1);
// the fp32 value of a bf16 is its bits shifted left by 16
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) float fl2;
__device__ __forceinline__ uint32_t cvt_pk(float f0, float f1) {
    const fl2 v = {f0, f1};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
}
__device__ __forceinline__ void split3_pair(float f0, float f1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = cvt_pk(f0, f1);
    const float r0 = f0 - __builtin_bit_cast(float, p1 << 16), r1 = f1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = cvt_pk(r0, r1);
    p3 = cvt_pk(r0 - __builtin_bit_cast(float, p2 << 16), r1 - __builtin_bit_cast(float, p2 & 0xffff0000u));
}

// fp16 operand format ("f16x2"): a = a1 + a2 with a1 = fp16(a), a2 = fp16(a - a1): 22 mantissa bits in two parts, so the three
// products a1w1 + a1w2 + a2w1 are fp32-equivalent (dropped: 2^-22) at HALF the matrix work of the six bf16 products.  The
// matrix cores keep fp16 denormal inputs (scratch/probe/mfma_f16_denorm.hip), so small a2 cost nothing but an absolute
// floor of 2^-25; what fp16 lacks is RANGE (|x| <= 65504, full precision above 6e-5): fine as it is for the forward pass,
// whose operands are normalised activations and weights; GRADIENT inputs (1e-7-sized dy) are multiplied by a power of two
// S = 2^(13 - floor(log2 max|dy|)) while they are staged and the accumulators by 1 / S in the epilogue -- exact, and max|dy|
// is a device scalar recorded by the kernel that wrote dy (san_act_bwd_amax: per-wave maxima + a one-workgroup finalize,
// no atomics), so nothing synchronises with the host.  End to end (scratch/study/split_fp16_accuracy.py): 2.8e-5 from the
// fp32 result, 5.3e-5 from fp64 (fp32 itself: 4.7e-5).
typedef __attribute__((ext_vector_type(2))) _Float16 hf2;
__device__ __forceinline__ uint32_t cvt_pk_h(float f0, float f1) {
    const fl2 v = {f0, f1};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hf2));      // round to nearest even, denormal results kept
}
__device__ __forceinline__ void split2h_pair(float f0, float f1, uint32_t& p1, uint32_t& p2) {
    p1 = cvt_pk_h(f0, f1);
    const hf2 h = __builtin_bit_cast(hf2, p1);
    // f - (float)h as fma((float)h, -1, f): exact either way (the difference of two floats 2^-11 apart is representable),
    // but this form selects v_fma_mix_f32, which reads the fp16 half directly: one instruction instead of convert + subtract
    p2 = cvt_pk_h(__builtin_fmaf((float)h[0], -1.f, f0), __builtin_fmaf((float)h[1], -1.f, f1));
}
__device__ __forceinline__ void split2h(float f, uint32_t& p1, uint32_t& p2) {
    const _Float16 a = (_Float16)f;
    const _Float16 b = (_Float16)(f - (float)a);
    p1 = __builtin_bit_cast(uint16_t, a);
    p2 = __builtin_bit_cast(uint16_t, b);
}

// fp8 operand format ("fp8", BASELINE config 5): ONE OCP e4m3 part per operand on v_mfma_f32_16x16x32_fp8_fp8 (same rate as the
// bf16 instruction, half the operand bytes).  e4m3 holds 4 significant bits between 2^-6 and 448, so both operands are brought
// into range by powers of two (exact): activations (normalised by the lazy affine: O(1)) x 8, clamped to +-448; weights x
// S_w = 2^(7 - floor(log2 max |w|)) per tensor (fp8_wscale_*_kernel, next to the packing); the accumulators get 1 / (8 S_w).
constexpr float kF8ActScale = 8.f;
__device__ __forceinline__ uint32_t cvt_f8x4(float f0, float f1, float f2, float f3) {
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f0, -448.f, 448.f), __builtin_amdgcn_fmed3f(f1, -448.f, 448.f), 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f2, -448.f, 448.f), __builtin_amdgcn_fmed3f(f3, -448.f, 448.f), p, true);
    return (uint32_t)p;
}

// WD ("weights direct"): the K-steps read their weight operands straight from the packed image in L2 instead of
// staging each chunk's weights through LDS.  All four waves then fetch the same weights (4x the L1 traffic), but the
// workgroup needs 49 KB of LDS instead of 92-156 KB, so three of them share a CU: measured +17..30 % for MB <= 4 with
// up to 6 chunks (64->64 @160^2, 96->32 @320^2, 72->36 @160^2), -7..33 % for MB = 5 or deep K (288->288 @20^2).
// KS = 1: the same kernel as a 1x1 convolution (the alignment net's 1x1 layers, the data gradient of the transposed
// convolutions): one K-step per 24-channel chunk (three channel groups at the centre tap + the zero group), always WD;
// only the tile's interior is staged.
// FLAT: run-time tile shape (narrow images: full-width rows, see tile_geom); otherwise the 32 x 8 tile with
// compile-time offsets between a wave's blocks (immediate LDS offsets in the K-loop).
// NP: operand parts used.  3 = the fp32-equivalent form (six products); 2 = a1 + a2 (16 mantissa bits, three products
// a1w1 + a1w2 + a2w1); 1 = plain bf16 (one product).  The reduced forms are the narrow-precision modes selected with
// san_set_conv_precision (judged by PSNR, not by the 1e-4 parity bar); the packed weight image is the same for all.
// SWAP: the ACTIVATIONS are the MFMA's A operand (16 pixels = rows of D) and the weights its B operand (16 channels = columns):
// a lane ends with FOUR CONSECUTIVE PIXELS of ONE channel per accumulator tile (D[4 kg + r][nn]), so the epilogue stores 16
// bytes per lane and the plane statistics are 16 in-lane values + two cross-row steps per channel.  Not for the pixel-shuffle
// epilogue of the transposed form (there a lane must hold the four virtual channels of one real channel).
// F8: the one-part fp8 e4m3 format (see kF8ActScale).
// NBW: 16-pixel blocks per wave (4: the 256-pixel tile; 3: full-width FLAT tiles of at most 192 pixels whose halo fits 256 staging
// units -- round 6, for the 40^2 level: 40 x 4 tiles make 240 workgroups of three blocks per wave out of 168 of four, see tile_plan).
template <int MB, bool WD, int KS, bool FLAT, int NP, bool SWAP, bool F16, bool F8, int NBW>
__global__ void __launch_bounds__(kT, SAN_B16_MINWG(MB, WD, KS, NP)) conv_bf16x3_kernel(const BArgs a) {
    static_assert(NBW == 4 || (FLAT && KS == 3 && (NBW == 3 || NBW == 2)), "short tiles are a FLAT 3x3 form");
    constexpr int kU = NBW == 4 ? kUnits : 3;          // staging units per thread (short tiles: halo <= 256 pixels, no fourth slot)
    static_assert(!F16 || NP == 2, "the fp16 format has two parts");
    static_assert(!F8 || (NP == 1 && !F16), "the fp8 format has one part");
    constexpr int kSteps = KS == 3 ? 7 : 1;            // (shadows the 3x3 constant)
    static_assert(KS == 3 || WD, "the 1x1 form reads its weights directly");
    constexpr int WCH = kSteps * MB * 3 * 64;          // uint4 per (cg, chunk) weight image
    constexpr int NWU = kSteps * MB * 3;               // ... = NWU wave-wide pieces (64 lanes x 16 B), one per (step, block, part)
    constexpr int WSL = (NWU + kT / 64 - 1) / (kT / 64); // pieces per wave: wave w moves pieces w, w + 4, ...
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_a = smem;                       // NP parts x kPartB
    uint4* lds_w = reinterpret_cast<uint4*>(smem + NP * kPartB);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int H = a.H, W = a.W;
    const int HWp = H * W;

    // XCD-aware order: consecutive logical ids (channel group fastest, then tile) on one XCD
    int lin;
    {
        const int total = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3;
        lin = xcd * (total >> 3) + min(xcd, total & 7) + slot;
    }
    // (compiled in with -DSAN_B16_TIMELINE only: the marks cost ~10 VGPRs, i.e. a resident workgroup for some forms)
    auto mark = [&](int i) {
#ifdef SAN_B16_TIMELINE
        if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 8 + i] = __builtin_amdgcn_s_memrealtime();
#else
        (void)i;
#endif
    };
    mark(0);
    const int ntile = a.tiles_x * a.tiles_y;
    const int group = fdiv(lin, a.m_S, a.S);            // (n, tile, cg)
    const int sk = lin - group * a.S;                   // split-K part (fastest: the S parts of a tile share its input)
    const int gq = fdiv(group, a.m_cgs, a.cgs);
    const int cg = group - gq * a.cgs;
    const int n = fdiv(group, a.m_cgsnt, a.cgs * ntile);
    const int tile = gq - n * ntile;
    const int c0 = fdiv(a.chunks * sk, a.m_S, a.S), c1 = fdiv(a.chunks * (sk + 1), a.m_S, a.S);     // this workgroup's chunks
    const int ty = fdiv(tile, a.m_tx, a.tiles_x), tx = tile - ty * a.tiles_x;
    const int tw = FLAT ? a.tw : kTW, th = FLAT ? a.th : kTH, hp = FLAT ? a.hp : kHW_, npx = FLAT ? a.npx : kNP;
    const int x0 = tx * tw, y0 = ty * th;

    // ---- staging units (pixel p, channel group chg): slots 0..2 = pixel tid of group s (the group, hence the
    // lazy-affine entries, is wave-uniform there), slot 3 = the remaining 84 pixels x 3 groups
    int s_goff[kU], s_loff[kU], s_chg[kU];
    bool s_in[kU], s_used[kU];
#pragma unroll
    for (int s = 0; s < kU; ++s) {
        int p, chg;
        bool used = true;
        if (s < 3) {
            p = tid;
            chg = s;
            used = tid < npx;
        } else {
            const int rem = npx - kT;                   // halo pixels beyond the first 256 (<= 84)
            used = rem > 0 && tid < 3 * rem;
            chg = used ? (FLAT ? fdiv(tid, a.m_rem, rem) : tid / rem) : 0;
            p = used ? kT + tid - chg * rem : 0;
        }
        const int pr = FLAT ? fdiv(p, a.m_hp, hp) : p / hp, pc = p - pr * hp;
        if (KS == 1) used = used && pr >= 1 && pr <= th && pc >= 1 && pc <= tw;       // no halo
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        s_in[s] = used && gy >= 0 && gy < H && gx >= 0 && gx < W;
        s_used[s] = used;
        s_goff[s] = s_in[s] ? gy * W + gx : 0;
        s_loff[s] = used ? p * kPS + chg * 16 : -1;
        s_chg[s] = chg;
    }
    // The zero frame (halo pixels outside the image) is the same for every chunk: written ONCE here, and the staging
    // loop below stores in-image units only -- no per-element select (8 per unit per chunk before).
#pragma unroll
    for (int s = 0; s < kU; ++s)
        if (s_used[s] && !s_in[s]) {
            if constexpr (F8) {
                *reinterpret_cast<uint2*>(smem + s_loff[s]) = make_uint2(0u, 0u);
            } else {
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<uint4*>(smem + p * kPartB + s_loff[s]) = make_uint4(0u, 0u, 0u, 0u);
            }
        }

    float st[kU][8];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 wst[WSL];                                   // (a plain vector type: the struct uint4 copy kept this array in scratch)
    // The chunk's lazy affine (24 scales + 24 shifts of this sample) travels through a double-buffered LDS
    // table: thread t < 48 fetches ONE value with the tile and stores it during the previous chunk's staging
    // phase (two barriers before anyone reads it) -- not 64 loads per thread per chunk.
    float* lds_aff = reinterpret_cast<float*>(smem + NP * kPartB + (WD ? (size_t)0 : (size_t)WCH * 16));
    const bool has_aff = a.in_scale != nullptr;
    const float lrelu_c = a.in_slope <= 1.f ? __builtin_inff() : -__builtin_inff();     // see the staging loop
    float my_aff = 0.f;
    // gradient input in the fp16 format: x S (S = 2^(13 - floor(log2 max |x|)), exact) rides in the affine table, the
    // accumulators get 1 / S at the end
    float inS = 1.f, inInvS = 1.f;
    if constexpr (F16) {
        if (a.amax) {
            const uint32_t b = san_amax_read(a.amax);
            int e = (int)((b >> 23) & 255u);
            if (b != 0u) {
                e = e < 14 ? 14 : (e > 250 ? 250 : e);
                inS = __builtin_bit_cast(float, (uint32_t)(267 - e) << 23);
                inInvS = __builtin_bit_cast(float, (uint32_t)(e - 13) << 23);
            }
        }
    }
    if constexpr (F16) inInvS *= a.f8_tail[1];          // the fp16-format weights' 1 / S_w (round 6; 1 unless the scale is switched on)
    if constexpr (F8) {
        inS = kF8ActScale;
        inInvS = a.f8_tail[1] * (1.f / kF8ActScale);
    }
    auto fetch_aff = [&](int chunk) {
        if (tid < 48) {
            const int ci = min(chunk * kCKC + (tid < 24 ? tid : tid - 24), a.cin - 1);
            const float* src = tid < 24 ? a.in_scale : a.in_shift;
            my_aff = has_aff ? src[n * a.x_ctot + a.x_coff + ci] * inS : (tid < 24 ? inS : 0.f);
        }
    };
    // Loads go through a buffer descriptor of this sample's view (base = its first channel): the address of a load is
    // descriptor base + SGPR offset (the channel: wave-uniform for slots 0..2, so it costs one scalar multiply) + 32-bit
    // per-lane offset (the pixel) -- no 64-bit vector address arithmetic per load (it was one v_lshl_add_u64 each, 32 per chunk).
    const unsigned hw4 = (unsigned)HWp * 4u;
    __amdgpu_buffer_rsrc_t xrs;
    {
        const size_t ext = (size_t)(a.x_ctot - a.x_coff) * (size_t)HWp * 4;            // to the end of this sample
        xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (size_t)(n * a.x_ctot + a.x_coff) * HWp), 0,
                                                (int)(ext > 0x7fffffffu ? 0x7fffffffu : ext), 0x00020000);
    }
    unsigned s_boff[kU];
#pragma unroll
    for (int s = 0; s < kU; ++s) s_boff[s] = (unsigned)s_goff[s] * 4u;
    auto prefetch = [&](int chunk) {
#pragma unroll
        for (int s = 0; s < kU; ++s) {
            if (s < 3) {
                const int c0 = chunk * kCKC + s * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    st[s][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             xrs, (int)s_boff[s], (int)((unsigned)min(c0 + i, a.cin - 1) * hw4), 0));
            } else {
                const int c0 = chunk * kCKC + s_chg[s] * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    st[s][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             xrs, (int)(s_boff[s] + (unsigned)min(c0 + i, a.cin - 1) * hw4), 0, 0));
            }
        }
        fetch_aff(chunk + 1);                         // stored as the next table during this chunk's staging phase
        if constexpr (!WD) {
            // this group's MB blocks of every K-step: 7 contiguous runs of MB*3 pieces.  The piece index is wave-uniform,
            // so its step / offset arithmetic runs on the scalar unit and the load is base + lane.
            const u32x4* src = reinterpret_cast<const u32x4*>(a.wp + ((size_t)chunk * kSteps * a.nblkp + (size_t)cg * MB) * 192 + lane);
    #pragma unroll
            for (int q = 0; q < WSL; ++q) {
                const int u = min(wave + q * (kT / 64), NWU - 1);
                const int st_ = u / (MB * 3);
                wst[q] = src[(st_ * a.nblkp * 3 + (u - st_ * (MB * 3))) * 64];
            }
        }
    };

    f4 acc[MB][NBW];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int b = 0; b < NBW; ++b) acc[m][b] = f4{0.f, 0.f, 0.f, 0.f};

    // ---- per-lane operand addressing.  B (activations): lane = (pixel nn = lane & 15, k-group kg = lane >> 4);
    // k-group g = 4 step + kg of the chunk is (tap = g / 3, channel group g % 3); group 27 is padding (zero weights).
    const int nn = lane & 15, kg = lane >> 4;
    int tapoff[kSteps];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
        if (KS == 3) {
            const int g = min(4 * s + kg, 26);
            const int tap = g / 3, chg = g - 3 * tap;
            const int ky = tap / 3, kx = tap - 3 * ky;
            tapoff[s] = (ky * hp + kx) * kPS + chg * 16;
        } else {
            tapoff[s] = (hp + 1) * kPS + min(kg, 2) * 16;          // centre tap; group 3 meets zero weights
        }
    }
    // block b of wave w = the 16 pixels 64 w + 16 b .. of the tile in flattened (row-major) order: any tile shape with
    // tw * th <= 256 works, a block may wrap from one tile row into the next (32 x 8: rows 2w, 2w+1 as before)
    int boff[NBW], trow[NBW], tcol[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        if constexpr (FLAT) {
            const int q = min(16 * NBW * wave + 16 * b + nn, tw * th - 1);
            trow[b] = fdiv(q, a.m_tw, tw);
            tcol[b] = q - trow[b] * tw;
        } else {
            trow[b] = 2 * wave + (b >> 1);
            tcol[b] = 16 * (b & 1) + nn;
        }
        boff[b] = (trow[b] * hp + tcol[b]) * kPS;
    }

    fetch_aff(c0);
    if (tid < 48) lds_aff[(c0 & 1) * 48 + tid] = my_aff;       // the first chunk's table; visible after the loop's first barrier
    prefetch(c0);
    for (int chunk = c0; chunk < c1; ++chunk) {
        __syncthreads();
#if SAN_B16_HALFX
        Frag wa[2][MB][3], xa[2][2][3];              // activations: two half-sets (2 of a wave's 4 pixel blocks each)
#else
        Frag wa[2][MB][3], xa[2][NBW][3];
#endif
        const uint4* wsrc = a.wp + ((size_t)chunk * kSteps * a.nblkp + (size_t)cg * MB) * 192 + lane;      // (WD)
        auto load_w = [&](int s, Frag (&wq)[MB][3]) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if constexpr (F8) {             // the low 8 bytes of the part-0 slot
                        if constexpr (WD) wq[m][p].l[0] = *reinterpret_cast<const long*>(&wsrc[((s * a.nblkp + m) * 3 + p) * 64]);
                        else wq[m][p].l[0] = *reinterpret_cast<const long*>(&lds_w[((s * MB + m) * 3 + p) * 64 + lane]);
                    } else if constexpr (WD) wq[m][p].u = wsrc[((s * a.nblkp + m) * 3 + p) * 64];
                    else wq[m][p].u = lds_w[((s * MB + m) * 3 + p) * 64 + lane];
                }
        };
        // KS = 1 has one K-step per chunk: its weights do not depend on the staged tile, so they are fetched here,
        // a whole staging phase ahead of their use
        if constexpr (KS == 1) load_w(0, wa[0]);
        // ---- registers -> LDS: lazy activation, split into three bf16 parts, [pixel][channel] image.  Channels past
        // cin were loaded from a clamped (real) channel and meet zero weights: no select needed for them.
        const float* afc = lds_aff + (chunk & 1) * 48;
        if (tid < 48) lds_aff[((chunk + 1) & 1) * 48 + tid] = my_aff;      // the NEXT chunk's table (prefetched below)
#pragma unroll
        for (int s = 0; s < kU; ++s) {
            uint32_t q1[4], q2[4], q3[4];
            float vv[8];
            const f4 sc0 = *reinterpret_cast<const f4*>(afc + s_chg[s] * 8), sc1 = *reinterpret_cast<const f4*>(afc + s_chg[s] * 8 + 4);
            const f4 sh0 = *reinterpret_cast<const f4*>(afc + 24 + s_chg[s] * 8), sh1 = *reinterpret_cast<const f4*>(afc + 24 + s_chg[s] * 8 + 4);
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                // two elements at a time on the packed fp32 pipe: affine (v_pk_fma_f32), slope product (v_pk_mul_f32), then
                // LeakyReLU as one v_med3_f32 per element: med3(v, slope v, +inf) = max(v, slope v) for slope <= 1,
                // med3(v, slope v, -inf) = min(v, slope v) for slope > 1 -- the reference's leaky_relu for every slope
                const fl2 xv = {st[s][i], st[s][i + 1]};
                const fl2 scv = {i < 4 ? sc0[i] : sc1[i - 4], i < 4 ? sc0[i + 1] : sc1[i - 3]};
                const fl2 shv = {i < 4 ? sh0[i] : sh1[i - 4], i < 4 ? sh0[i + 1] : sh1[i - 3]};
                const fl2 av = __builtin_elementwise_fma(xv, scv, shv);
                const fl2 sv = av * fl2{a.in_slope, a.in_slope};
                const float v0 = __builtin_amdgcn_fmed3f(av[0], sv[0], lrelu_c);
                const float v1 = __builtin_amdgcn_fmed3f(av[1], sv[1], lrelu_c);
                if constexpr (F8) {
                    vv[i] = v0;
                    vv[i + 1] = v1;
                } else if constexpr (F16) {
                    split2h_pair(v0, v1, q1[i >> 1], q2[i >> 1]);
                    q3[i >> 1] = 0u;
                } else {
                    split3_pair(v0, v1, q1[i >> 1], q2[i >> 1], q3[i >> 1]);
                }
            }
            if constexpr (F8) {
                if (s_in[s])
                    *reinterpret_cast<uint2*>(lds_a + s_loff[s]) = make_uint2(cvt_f8x4(vv[0], vv[1], vv[2], vv[3]), cvt_f8x4(vv[4], vv[5], vv[6], vv[7]));
            } else if (s_in[s]) {
                *reinterpret_cast<uint4*>(lds_a + s_loff[s]) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
                if constexpr (NP > 1) *reinterpret_cast<uint4*>(lds_a + kPartB + s_loff[s]) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
                if constexpr (NP > 2) *reinterpret_cast<uint4*>(lds_a + 2 * kPartB + s_loff[s]) = make_uint4(q3[0], q3[1], q3[2], q3[3]);
            }
        }
        if constexpr (!WD) {
#pragma unroll
            for (int q = 0; q < WSL; ++q) {
                const int u = wave + q * (kT / 64);
                if (u < NWU) *reinterpret_cast<u32x4*>(lds_w + u * 64 + lane) = wst[q];
            }
        }
        // ---- 7 K-steps: (MB + 4) x 3 sixteen-byte operand reads feed 6 x MB x 4 MFMAs; the reads of step s+1
        // are issued before the MFMAs of step s (two operand sets, compile-time indices after unrolling)
#if SAN_B16_HALFX
        // half-step hq = 2 s + hb: blocks 2 hb, 2 hb + 1 of K-step s
        auto load_xh = [&](int hq, Frag (&xq)[2][3]) {
            const int to = tapoff[hq >> 1];
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if constexpr (F8) xq[bb][p].l[0] = *reinterpret_cast<const long*>(lds_a + boff[2 * (hq & 1) + bb] + to);
                    else xq[bb][p].u = *reinterpret_cast<const uint4*>(lds_a + p * kPartB + boff[2 * (hq & 1) + bb] + to);
                }
        };
#else
        auto load_x = [&](int s, Frag (&xq)[NBW][3]) {
            const int to = tapoff[s];
#pragma unroll
            for (int b = 0; b < NBW; ++b)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if constexpr (F8) xq[b][p].l[0] = *reinterpret_cast<const long*>(lds_a + boff[b] + to);
                    else xq[b][p].u = *reinterpret_cast<const uint4*>(lds_a + p * kPartB + boff[b] + to);
                }
        };
#endif
        __syncthreads();
        if (chunk == c0) mark(1);
        if (chunk + 1 < c1) prefetch(chunk + 1);
        if constexpr (KS == 3) load_w(0, wa[0]);
#if SAN_B16_HALFX
        // operand reads run one HALF-step ahead of the MFMAs (two pixel blocks x NP parts: half the operand registers of a
        // full-step ring), the weights one step ahead as before
        load_xh(0, xa[0]);
#pragma unroll
        for (int hq = 0; hq < 2 * kSteps; ++hq) {
            const int s = hq >> 1;
            if ((hq & 1) == 0 && s + 1 < kSteps) load_w(s + 1, wa[(s + 1) & 1]);
            if (hq + 1 < 2 * kSteps) load_xh(hq + 1, xa[(hq + 1) & 1]);
#pragma unroll
            for (int pw = 0; pw < NP; ++pw)
#pragma unroll
                for (int px = 0; px < NP - pw; ++px)
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int bb = 0; bb < 2; ++bb) {
                            const int b = 2 * (hq & 1) + bb;
                            Frag& X = xa[hq & 1][bb][px];
                            Frag& Wv = wa[s & 1][m][pw];
                            if constexpr (F8) {
                                if constexpr (SWAP) acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(X.l[0], Wv.l[0], acc[m][b], 0, 0, 0);
                                else acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(Wv.l[0], X.l[0], acc[m][b], 0, 0, 0);
                            } else if constexpr (F16) {
                                if constexpr (SWAP) acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(X.h, Wv.h, acc[m][b], 0, 0, 0);
                                else acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wv.h, X.h, acc[m][b], 0, 0, 0);
                            } else if constexpr (SWAP)
                                acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X.v, Wv.v, acc[m][b], 0, 0, 0);
                            else
                                acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wv.v, X.v, acc[m][b], 0, 0, 0);
                        }
        }
    }
#else
        load_x(0, xa[0]);
#pragma unroll
        for (int s = 0; s < kSteps; ++s) {
            if (s + 1 < kSteps) {
                load_w(s + 1, wa[(s + 1) & 1]);
                load_x(s + 1, xa[(s + 1) & 1]);
            }
#pragma unroll
            for (int pw = 0; pw < NP; ++pw)
#pragma unroll
                for (int px = 0; px < NP - pw; ++px)
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int b = 0; b < NBW; ++b) {
                            if constexpr (F8) {
                                if constexpr (SWAP)
                                    acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(xa[s & 1][b][px].l[0], wa[s & 1][m][pw].l[0], acc[m][b], 0, 0, 0);
                                else
                                    acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wa[s & 1][m][pw].l[0], xa[s & 1][b][px].l[0], acc[m][b], 0, 0, 0);
                            } else if constexpr (F16) {
                                if constexpr (SWAP)
                                    acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[s & 1][b][px].h, wa[s & 1][m][pw].h, acc[m][b], 0, 0, 0);
                                else
                                    acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[s & 1][m][pw].h, xa[s & 1][b][px].h, acc[m][b], 0, 0, 0);
                            } else if constexpr (SWAP)
                                acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[s & 1][b][px].v, wa[s & 1][m][pw].v, acc[m][b], 0, 0, 0);
                            else
                                acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s & 1][m][pw].v, xa[s & 1][b][px].v, acc[m][b], 0, 0, 0);
                        }
#ifdef SAN_B16_TIMELINE
            if (chunk == c0 && s == 0) mark(6);
            if (chunk == c0 && s == 3) mark(7);
#endif
        }
    }
#endif

    // ------------------------------------------------------------ epilogue
    mark(2);
#ifdef SAN_B16_TIMELINE
    if (a.dbg && tid == 0) {
        a.dbg[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
        a.dbg[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
    }
#endif
    // The matrix cores leave the accumulators in AGPRs.  The power-of-two rescale of the fp16 / fp8 formats is applied
    // UNCONDITIONALLY (x 1 where there is none: exact): a run-time `if` around an in-place update made the compiler keep the
    // array in AGPRs across the whole epilogue, and every later use (bias, statistics, stores) paid a v_accvgpr_read pass
    // plus a write-back (four reads + two writes over all accumulators); now there is one read pass.
    {
        float osc = 1.f;
        if constexpr (F16 || F8) osc = inInvS;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int b = 0; b < NBW; ++b) acc[m][b] = acc[m][b] * osc;
    }
    if constexpr (SWAP) {
        // acc[m][b][r] = output channel (cg MB + m) 16 + nn at the tile pixel r places after pixel 64 wave + 16 b + 4 kg
        const int cb0 = cg * MB * 16 + nn;
        const bool quad_ok = !FLAT || (tw & 3) == 0;    // FLAT tiles are full-width rows: a quad stays inside one row
        int qy[NBW], qx[NBW];
        bool vq[NBW], v1[NBW][4];
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            const int q = 16 * NBW * wave + 16 * b + 4 * kg;
            int orow, ocol;
            bool oin = true;
            if constexpr (FLAT) {
                const int qq = min(q, tw * th - 1);
                orow = qq / tw;
                ocol = qq - orow * tw;
                oin = q < tw * th;
            } else {
                orow = q >> 5;
                ocol = q & 31;
            }
            qy[b] = y0 + orow;
            qx[b] = x0 + ocol;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (quad_ok) {
                    v1[b][r] = oin && qy[b] < H && qx[b] + r < W;
                } else {
                    const int q1 = q + r, rr = q1 / tw, cc = q1 - rr * tw;
                    v1[b][r] = q1 < tw * th && y0 + rr < H && x0 + cc < W;
                }
            }
            vq[b] = quad_ok && v1[b][3] && (W & 3) == 0 && (reinterpret_cast<uintptr_t>(a.S > 1 ? a.ws : a.y) & 15) == 0;
        }
        float* ybase = a.S > 1 ? a.ws + ((size_t)(sk * a.N + n) * a.cout) * HWp : a.y + (size_t)(n * a.y_ctot + a.y_coff) * HWp;
        if (a.S == 1) {
            if (a.bias) {
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const int co = cb0 + 16 * m;
                    const float bv = co < a.cout ? a.bias[co] : 0.f;
#pragma unroll
                    for (int b = 0; b < NBW; ++b) acc[m][b] += f4{bv, bv, bv, bv};
                }
            }
            if (a.part) {
                float cnt = 0.f;
#pragma unroll
                for (int b = 0; b < NBW; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cnt += v1[b][r] ? 1.f : 0.f;
                cnt += __shfl_xor(cnt, 16, 64);
                cnt += __shfl_xor(cnt, 32, 64);
                const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
                const int tiles = ntile * 4;
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const float pilot = __shfl(acc[m][0][0], nn, 64);      // the wave's first pixel of this channel
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int b = 0; b < NBW; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = v1[b][r] ? acc[m][b][r] - pilot : 0.f;
                            s1 += e;
                            s2 = fmaf(e, e, s2);
                        }
                    s1 += __shfl_xor(s1, 16, 64);
                    s2 += __shfl_xor(s2, 16, 64);
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    const int co = cb0 + 16 * m;
                    if (kg == 0 && co < a.cout) {
                        float* o = a.part + ((size_t)(n * a.cout + co) * tiles + tile * 4 + wave) * 3;
                        o[0] = cnt;
                        o[1] = cnt > 0.f ? pilot + s1 * inv : 0.f;
                        o[2] = cnt > 0.f ? fmaxf(s2 - s1 * s1 * inv, 0.f) : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int co = cb0 + 16 * m;
            if (co < a.cout) {
                float* dst = ybase + (size_t)co * HWp;
#pragma unroll
                for (int b = 0; b < NBW; ++b) {
                    if (vq[b]) {
                        *reinterpret_cast<f4*>(dst + qy[b] * W + qx[b]) = acc[m][b];
                    } else if (quad_ok) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (v1[b][r]) dst[qy[b] * W + qx[b] + r] = acc[m][b][r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (v1[b][r]) {
                                const int q1 = 16 * NBW * wave + 16 * b + 4 * kg + r, rr = q1 / tw, cc = q1 - rr * tw;
                                dst[(y0 + rr) * W + x0 + cc] = acc[m][b][r];
                            }
                    }
                }
            }
        }
        mark(3);
        return;
    }
    // acc[m][b][r] = output channel co = (cg MB + m) 16 + 4 (lane >> 4) + r at tile pixel (trow[b], tcol[b])
    const int cbase = cg * MB * 16 + 4 * kg;
    int oy[NBW], ox[NBW];
    bool valid[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        oy[b] = y0 + trow[b];
        ox[b] = x0 + tcol[b];
        valid[b] = (!FLAT || 16 * NBW * wave + 16 * b + nn < tw * th) && oy[b] < H && ox[b] < W;
    }
    if (a.S > 1) {
        // split-K: this workgroup covered only chunks [c0, c1): its partial sums go to slice sk of the scratch tensor;
        // bias, statistics and the real store happen in splitk_reduce_kernel (next launch: no device-scope fences here)
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cbase + 16 * m + r;
                if (co < a.cout) {
                    float* dst = a.ws + ((size_t)(sk * a.N + n) * a.cout + co) * HWp;
#pragma unroll
                    for (int b = 0; b < NBW; ++b)
                        if (valid[b]) dst[oy[b] * W + ox[b]] = acc[m][b][r];
                }
            }
        return;
    }
    if (a.bias) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cbase + 16 * m + r;
                const float bv = co < a.cout ? a.bias[co] : 0.f;
#pragma unroll
                for (int b = 0; b < NBW; ++b) acc[m][b][r] += bv;
            }
    }
    if (a.part) {
        // per-wave (count, mean, M2) of every channel over the wave's 64 pixels: pilot-shifted single pass; the 16
        // lanes of a DPP row hold the 16 pixels of a block, so four row-local DPP steps finish the sum
        float cnt = 0.f;
#pragma unroll
        for (int b = 0; b < NBW; ++b) cnt += valid[b] ? 1.f : 0.f;
        cnt += san_dpp_get<0xB1, 0xf>(cnt);
        cnt += san_dpp_get<0x4E, 0xf>(cnt);
        cnt += san_dpp_get<0x141, 0xf>(cnt);
        cnt += san_dpp_get<0x140, 0xf>(cnt);
        const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
        const int tiles = ntile * 4;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // pilot: block 0, first pixel of the row of lanes (valid whenever the wave has any valid pixel)
                const float pilot = __shfl(acc[m][0][r], lane & 48, 64);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int b = 0; b < NBW; ++b) {
                    const float e = valid[b] ? acc[m][b][r] - pilot : 0.f;
                    s1 += e;
                    s2 = fmaf(e, e, s2);
                }
                s1 += san_dpp_get<0xB1, 0xf>(s1);
                s2 += san_dpp_get<0xB1, 0xf>(s2);
                s1 += san_dpp_get<0x4E, 0xf>(s1);
                s2 += san_dpp_get<0x4E, 0xf>(s2);
                s1 += san_dpp_get<0x141, 0xf>(s1);
                s2 += san_dpp_get<0x141, 0xf>(s2);
                s1 += san_dpp_get<0x140, 0xf>(s1);
                s2 += san_dpp_get<0x140, 0xf>(s2);
                const int co = cbase + 16 * m + r;
                if (nn == 0 && co < a.cout) {
                    // transposed convolution: the 4 virtual channels of a real channel are 4 more statistics tiles
                    float* o = a.shuffle ? a.part + ((size_t)(n * (a.cout >> 2) + (co >> 2)) * (tiles * 4) + (tile * 4 + wave) * 4 + (co & 3)) * 3
                                         : a.part + ((size_t)(n * a.cout + co) * tiles + tile * 4 + wave) * 3;
                    o[0] = cnt;
                    o[1] = cnt > 0.f ? pilot + s1 * inv : 0.f;
                    o[2] = cnt > 0.f ? fmaxf(s2 - s1 * s1 * inv, 0.f) : 0.f;
                }
            }
    }
    if (a.shuffle) {
        // virtual channel 4 c + 2 dy + dx of input pixel (y, x) is output pixel (2 y + dy, 2 x + dx) of channel c; a lane
        // holds r = 0..3 = the four positions of one real channel: two 8-byte stores, 128 contiguous bytes per 16 lanes
        typedef float fl2v __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int co = cbase + 16 * m;
            if (co < a.cout) {
                float* dst = a.y + (size_t)(n * a.y_ctot + a.y_coff + (co >> 2)) * (4 * (size_t)HWp);
#pragma unroll
                for (int b = 0; b < NBW; ++b)
                    if (valid[b]) {
                        float* q = dst + (size_t)(2 * oy[b]) * (2 * W) + 2 * ox[b];
                        *reinterpret_cast<fl2v*>(q) = fl2v{acc[m][b][0], acc[m][b][1]};
                        *reinterpret_cast<fl2v*>(q + 2 * W) = fl2v{acc[m][b][2], acc[m][b][3]};
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = cbase + 16 * m + r;
            if (co < a.cout) {
                float* dst = a.y + (size_t)(n * a.y_ctot + a.y_coff + co) * HWp;
#pragma unroll
                for (int b = 0; b < NBW; ++b)
                    if (valid[b]) dst[oy[b] * W + ox[b]] = acc[m][b][r];
            }
        }
}

// ---------------------------------------------------------------- split-K second pass
// y[n][c][p] = bias[c] + sum_k ws[k][n][c][p] (fixed order), plus the plane's (count, mean, M2) as statistics tile 0
// (the other tiles of the caller's partials array get count 0).  One workgroup per (n, c) plane of at most 4096 pixels.
// fin_scale != null: the plane IS the InstanceNorm statistic, so the lazy affine (scale = rsqrt(var_b + eps), shift = -mean scale:
// exactly san_norm_finalize's arithmetic on a one-tile record) is written here and no finalising launch follows.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, int S, const float* __restrict__ bias,
                                                             float* __restrict__ y, int y_ctot, int y_coff, float* __restrict__ part,
                                                             int tiles, int N, int C, int HW, float* __restrict__ fin_scale,
                                                             float* __restrict__ fin_shift, int fin_ctot, int fin_coff, float fin_eps) {
    __shared__ float red[2][4];
    const int nc = blockIdx.x, n = nc / C, c = nc - n * C;
    const float bv = bias ? bias[c] : 0.f;
    const size_t plane = (size_t)N * C * HW;
    const float* src = ws + (size_t)nc * HW;
    float* dst = y + ((size_t)n * y_ctot + y_coff + c) * HW;
    float v[16];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int p = threadIdx.x + 256 * i;
        v[i] = 0.f;
        if (p < HW) {
            float t = src[p];
            for (int k = 1; k < S; ++k) t += src[(size_t)k * plane + p];
            v[i] = t + bv;
            dst[p] = v[i];
            s1 += v[i];
        }
    }
    if (!part && !fin_scale) return;
    s1 = san_wave_sum(s1);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s1;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)HW;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int p = threadIdx.x + 256 * i;
        if (p < HW) s2 = fmaf(v[i] - mean, v[i] - mean, s2);
    }
    s2 = san_wave_sum(s2);
    if ((threadIdx.x & 63) == 0) red[1][threadIdx.x >> 6] = s2;
    __syncthreads();
    if (fin_scale) {
        if (threadIdx.x == 0) {
            const double m2 = (double)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
            const float sc = (float)(1.0 / sqrt(m2 / (double)HW + (double)fin_eps));
            fin_scale[n * fin_ctot + fin_coff + c] = sc;
            fin_shift[n * fin_ctot + fin_coff + c] = (float)(-(double)mean) * sc;
        }
        return;
    }
    float* o = part + (size_t)nc * tiles * 3;
    for (int t = threadIdx.x; t < tiles; t += 256) {
        o[3 * t] = t == 0 ? (float)HW : 0.f;
        o[3 * t + 1] = t == 0 ? mean : 0.f;
        o[3 * t + 2] = t == 0 ? (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]) : 0.f;
    }
}

// ---------------------------------------------------------------- packing
// packed[chunk][step][blk][part][lane][i] (bf16), blk over nblkp = ceil(cout/16) + 4 blocks (zero padded so any
// grouping of up to 5 blocks per workgroup stays inside): lane = (co16 = lane & 15, kg = lane >> 4);
// co = 16 blk + co16; group g = 4 step + kg -> tap = g / 3, ci = chunk 24 + (g % 3) 8 + i.
// mode 0: w is the forward weight [cout][cin][3][3]; mode 2 (data gradient): (cout, cin) are those of the
// data-gradient convolution and w is the forward weight of the layer being differentiated: value = w[ci][co][8 - tap].
// ks = 1: one step per chunk, group kg < 3 -> channels chunk 24 + 8 kg + i of the single tap, kg = 3 zero;
// mode 0: w[co][ci], mode 2: w[ci][co].
// One thread per (chunk, step, blk, lane): its 8 values (consecutive input channels of one tap) are gathered once,
// split three ways and written as three 16-byte vectors (one per part).
// FULL = false (the batched re-pack of every training step): units that are zero for the image's whole life -- channels past cout
// in the padded blocks, the fourth lane row of the last K-step, the unused part slots of the fp16 / fp8 formats -- are NOT written:
// the caller zero-fills an image once (round 5: they were 55-90 % of the bytes the batch stored, 0.3 ms per step).
template <bool FULL>
__device__ __forceinline__ void pack_unit(const float* __restrict__ w, uint16_t* __restrict__ packed, size_t u, int cout,
                                          int cin, int nblkp, int mode_, int ks, const float* __restrict__ wscale) {
    const bool f16 = (mode_ & 16) != 0;            // mode + 16: two fp16 parts (parts 0, 1 of the image; part 2 zero)
    const bool f8 = (mode_ & 32) != 0;             // mode + 32: one fp8 e4m3 part x the tensor's scale (low 8 bytes of the part-0 slot)
    // fp8 AND (round 6, opt-in: san_conv_f16_wscale_enable) fp16 formats: the tensor's power-of-two scale S_w (exact), written behind
    // the image by the wscale pass ({1, 1} when off).  The fp16 parts then hold w S_w with max |w| S_w in [2^13, 2^14): the second
    // part is a NORMAL fp16 number for every weight within 2^-16 of the largest one -- unscaled, a 0.1-sized weight leaves its second
    // part in the denormals (absolute floor 2^-25: 3e-7 of the weight, the per-layer error of rounds 2-5) and a 1e-4-sized tensor
    // keeps 11 bits.  The kernels multiply the accumulators by 1 / S_w.
    const float wS = (f8 || f16) ? wscale[0] : 1.f;
    const int mode = mode_ & 15;
    const int lane = (int)(u & 63);
    size_t r = u >> 6;
    const int blk = (int)(r % nblkp);
    r /= nblkp;
    const int steps = ks == 1 ? 1 : kSteps;
    const int step = (int)(r % steps);
    const int chunk = (int)(r / steps);
    const int co = blk * 16 + (lane & 15);
    const int g = 4 * step + (lane >> 4);
    const int tap = g / 3;
    const int ci0 = chunk * kCKC + (g - 3 * tap) * 8;
    const bool live = co < cout && (ks == 1 ? tap == 0 : tap < 9) && ci0 < cin;
    if (!FULL && !live) return;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 q[3];
    float v8[8];
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        float v[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ci = ci0 + i + k;
            v[k] = 0.f;
            if (live && ci < cin) {
                if (ks == 1) v[k] = mode == 2 ? w[(size_t)ci * cout + co] : w[(size_t)co * cin + ci];
                else v[k] = mode == 2 ? w[((size_t)ci * cout + co) * 9 + (8 - tap)] : w[((size_t)co * cin + ci) * 9 + tap];
            }
        }
        v8[i] = v[0] * wS;
        v8[i + 1] = v[1] * wS;
        uint32_t a1, a2, a3 = 0, b1, b2, b3 = 0;
        if (f16) {
            split2h(v8[i], a1, a2);
            split2h(v8[i + 1], b1, b2);
        } else {
            split3(v[0], a1, a2, a3);
            split3(v[1], b1, b2, b3);
        }
        q[0][i >> 1] = (a1 & 0xffffu) | (b1 << 16);
        q[1][i >> 1] = (a2 & 0xffffu) | (b2 << 16);
        q[2][i >> 1] = (a3 & 0xffffu) | (b3 << 16);
    }
    // packed[(((chunk steps + step) nblkp + blk) 3 + part) 64 + lane][8]
    uint16_t* o = packed + ((u >> 6) * 3 * 64 + lane) * 8;
    if (f8) {
        q[0] = u32x4{cvt_f8x4(v8[0], v8[1], v8[2], v8[3]), cvt_f8x4(v8[4], v8[5], v8[6], v8[7]), 0u, 0u};
        q[1] = q[2] = u32x4{0u, 0u, 0u, 0u};
    }
    const int parts = FULL ? 3 : (f8 ? 1 : (f16 ? 2 : 3));
#pragma unroll
    for (int p = 0; p < 3; ++p)
        if (p < parts) *reinterpret_cast<u32x4*>(o + (size_t)p * 64 * 8) = q[p];
}

__global__ void pack_bf16x3_kernel(const float* __restrict__ w, uint16_t* __restrict__ packed, size_t total, int cout,
                                   int cin, int nblkp, int mode, int ks) {
    const size_t units = total / 24;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x)
        pack_unit<true>(w, packed, u, cout, cin, nblkp, mode, ks, reinterpret_cast<const float*>(packed + total));
}

// batched: 8 x int64 per job = {w, packed, cout, cin, nblkp, ks (0 = 3), mode, total}
__global__ void pack_bf16x3_batch_kernel(const long long* __restrict__ jobs) {
    const long long* j = jobs + 8 * (size_t)blockIdx.y;
    const float* w = reinterpret_cast<const float*>(j[0]);
    uint16_t* packed = reinterpret_cast<uint16_t*>(j[1]);
    const size_t units = (size_t)j[7] / 24;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x)
        pack_unit<false>(w, packed, u, (int)j[2], (int)j[3], (int)j[4], (int)j[6], j[5] == 1 ? 1 : 3, reinterpret_cast<const float*>(packed + (size_t)j[7]));
}

// fp8 format: the tensor's power-of-two scale {S_w, 1 / S_w}, S_w = 2^(7 - floor(log2 max |w|)) (scaled maximum in [128, 256)),
// written behind the packed image.  One 1024-thread workgroup per tensor (max is order-independent); launched before the packing kernel.
// (round 6: also the fp16-part format, `target` = 13: scaled maximum in [2^13, 2^14), as the gradient inputs' amax scale)
__device__ __forceinline__ void fp8_wscale(const float* __restrict__ w, size_t count, float* __restrict__ tail, int target = 7) {
    __shared__ float red[16];
    float m = 0.f;
    if ((reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        const f4* w4 = reinterpret_cast<const f4*>(w);
        const size_t n4 = count >> 2;
#pragma unroll 4
        for (size_t i = threadIdx.x; i < n4; i += 1024) {
            const f4 v = w4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        for (size_t i = (n4 << 2) + threadIdx.x; i < count; i += 1024) m = fmaxf(m, fabsf(w[i]));
    } else {
#pragma unroll 4
        for (size_t i = threadIdx.x; i < count; i += 1024) m = fmaxf(m, fabsf(w[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) m = fmaxf(m, red[k]);
        int e = (int)((__builtin_bit_cast(uint32_t, m) >> 23) & 255u);       // biased exponent of the maximum
        float S = 1.f, inv = 1.f;
        if (m > 0.f && m < __builtin_inff()) {
            e = e < 14 ? 14 : (e > 240 ? 240 : e);
            S = __builtin_bit_cast(float, (uint32_t)(127 + target - (e - 127)) << 23);
            inv = __builtin_bit_cast(float, (uint32_t)(127 - target + (e - 127)) << 23);
        }
        tail[0] = S;
        tail[1] = inv;
    }
}

__global__ void __launch_bounds__(1024) fp8_wscale_kernel(const float* __restrict__ w, size_t count, float* __restrict__ tail, int target) {
    fp8_wscale(w, count, tail, target);
}

__global__ void __launch_bounds__(1024) fp8_wscale_batch_kernel(const long long* __restrict__ jobs) {
    const long long* j = jobs + 8 * (size_t)blockIdx.x;
    const int m = (int)j[6];
    if ((m & 32) == 0 && !((m & 16) && (m & 256))) return;      // fp8 images, and fp16 images that asked for a scale (bit 8)
    fp8_wscale(reinterpret_cast<const float*>(j[0]), (size_t)j[2] * (size_t)j[3] * (j[5] == 1 ? 1 : 9),
               reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(j[1]) + (size_t)j[7]), (m & 32) ? 7 : 13);
}

int g_b16_wd = -1;             // tuning hook (SAN_B16_WD=0/1 at first use): force the weights-direct choice
int g_b16_mb = -1;             // tuning hook (SAN_B16_MB=2..5): force the channel blocks per workgroup
int g_b16_flat = 1;            // tuning hook (SAN_B16_FLAT=0): always 32 x 8 tiles
int g_b16_splitk = 4;          // tuning hook (SAN_B16_SPLITK=1 disables split-K, 2 / 4 = most parts per tile)
int g_b16_splitcap = 1024;     // tuning hook (SAN_B16_SPLITCAP): most workgroups a split launch may have
unsigned long long* g_b16_dbg = nullptr;
int g_b16_wd_cold = 1;         // tuning hook (SAN_B16_WD_COLD=0): weights-direct also for launches of at most 256 workgroups (the round-5 choice)
int g_b16_nbw = 0;             // tuning hook (SAN_B16_NBW): 0 automatic, 4 never the short-tile form, 3 wherever the geometry allows it
int g_b16_swap = 1;            // tuning hook (SAN_B16_SWAP=0): the channel-per-register accumulator layout everywhere
struct B16Env {
    B16Env() {
        if (const char* e = getenv("SAN_B16_WD")) g_b16_wd = atoi(e);
        if (const char* e = getenv("SAN_B16_MB")) g_b16_mb = atoi(e);
        if (const char* e = getenv("SAN_B16_FLAT")) g_b16_flat = atoi(e);
        if (const char* e = getenv("SAN_B16_SPLITK")) g_b16_splitk = atoi(e);
        if (const char* e = getenv("SAN_B16_SPLITCAP")) g_b16_splitcap = atoi(e);
        if (const char* e = getenv("SAN_B16_SWAP")) g_b16_swap = atoi(e);
        if (const char* e = getenv("SAN_B16_NBW")) g_b16_nbw = atoi(e);
        if (const char* e = getenv("SAN_B16_WD_COLD")) g_b16_wd_cold = atoi(e);
    }
} g_b16_env;

// Split-K: for 3x3 layers whose (image, tile, channel group) count leaves CUs idle and whose K is a long serial chain of
// chunks (288 -> 288 @20^2: 64 tiles, 12 chunks), up to 4 workgroups share a tile, each taking a contiguous range of
// the chunks (at least 2, at most ~4 workgroups per CU in the launch) and writing a partial output; splitk_reduce_kernel then adds the partials in a fixed order,
// applies the bias, stores and takes the statistics.  (An in-kernel last-arriver join was tried first: its device-scope
// fences write back / invalidate the whole L2 of every XCD and made the layer 1.7x SLOWER.)
// out_elems = n * cout * hw: every part writes (and the join reads) that many partial sums, so on a large output a split pays only
// while each part keeps a long chain of chunks (round 4, re-measured: 144 -> 288 @40^2 59 us split in two, 40 unsplit; 144 -> 144
// @40^2 38 / 35; 288 -> 144 @40^2 stays split: 54 / 61; the 20^2 layers, under a million outputs, keep up to four parts)
int splitk_parts(int groups, int chunks, int ks, int hw, long long out_elems) {
    if (ks != 3 || hw > 4096) return 1;
    const int min_chain = out_elems > 1000000ll ? 4 : 2;
    int S = 1;
    while (S * 2 <= g_b16_splitk && groups * S * 2 <= g_b16_splitcap && chunks / (S * 2) >= min_chain) S *= 2;
    return S;
}

size_t splitk_bytes(int S, int n, int cout, int hw) { return (size_t)S * n * cout * hw * sizeof(float); }

// Tile shape: 32 x 8 unless a narrow image fills more of the 256-pixel tile as full-width rows (W <= ~48: 20 x 12,
// 40 x 6 ...; the halo must fit the 340-pixel LDS image).  Blocks are flattened 16-pixel runs, so any shape works.
struct TileGeom {
    int tw, th, tiles_x, tiles_y;
};

TileGeom tile_geom(int h, int w) {
    TileGeom g{kTW, kTH, san_cdiv(w, kTW), san_cdiv(h, kTH)};
    if (g_b16_flat == 0) return g;
    long long best = (long long)g.tiles_x * g.tiles_y;             // workgroups per image: fewer = fuller tiles
    int th = 256 / w;
    if (th > h) th = h;
    while (th > 1 && (w + 2) * (th + 2) > kNP) --th;
    if (th >= 1 && w * th <= 256 && (w + 2) * (th + 2) <= kNP) {
        // rows balanced over the image (20 rows: 10 + 10 rather than 12 + 8)
        const int ty = san_cdiv(h, th);
        const int thb = san_cdiv(h, ty);
        if ((long long)ty < best) g = TileGeom{w, thb, 1, ty};
    }
    return g;
}

struct BPlan {
    int nblkp, chunks;
    size_t packed_elems;       // bf16 elements
};

BPlan bplan(int cout, int cin, int ks = 3) {
    BPlan p{};
    p.nblkp = san_cdiv(cout, 16) + 4;
    p.chunks = san_cdiv(cin, kCKC);
    p.packed_elems = (size_t)p.chunks * (ks == 1 ? 1 : kSteps) * p.nblkp * 3 * 64 * 8;
    return p;
}

// Output-channel blocks per workgroup for this launch: the LDS budget allows one workgroup per CU, so
// prefer the largest MB (input tile staged once for more channels) that still gives every CU a workgroup;
// cost model = dispatch rounds x (MB + ~1.5 blocks' worth of per-tile staging).
int pick_mb(int cout, int tiles) {
    const int nblk = san_cdiv(cout, 16);
    int best = 2;
    float best_c = 1e30f;
    for (int mb = 5; mb >= 2; --mb) {
        const int m = mb < nblk ? mb : (nblk < 2 ? 2 : nblk);
        const int wgs = tiles * san_cdiv(nblk, m);
        const float c = (float)san_cdiv(wgs, 256) * ((float)m + 1.5f);
        if (c < best_c - 1e-6f) {
            best_c = c;
            best = m;
        }
    }
    return best > 5 ? 5 : best;
}

// Block count for the fp16-part format (3x3).  Measured (scratch/bench_layers.py, N = 8): with half the matrix work per chunk the
// kernel is resident three workgroups deep up to 3 blocks, so many small workgroups win: the largest count of 2..4 that still
// gives >= 768 workgroups (256 CUs x 3), and 2 whenever the reduction is deep enough (cin >= 64) for the matrix part to dominate.
int pick_mb_f16(int cin, int cout, int tiles) {
    const int nblk = san_cdiv(cout, 16);
    // three blocks where they cover the layer exactly and two would compute a block that does not exist (cout 36 / 48: 4 blocks
    // for 3; cout 144: 10 for 9): 72->36 @80^2 23.5 -> 20.5 us, 72->144 @80^2 50 -> 48, 144->144 @40^2 34.8 -> 32.6 (not the tiny
    // 144->144 @20^2: 20 -> 23)
    if (nblk == 3 || (nblk == 9 && (tiles >= 32 || cin >= 288))) return 3;
    if (nblk <= 2 || cin >= 64) return 2;
    // (round 4, re-measured on the final kernels: four blocks per workgroup lose everywhere -- 36->72 @160^2, the data gradient of
    // the decoder's 72->36: 97 us at 4, 74 at 2; 32->64 @160^2: 59 vs 50 -- three win only where they cover the layer exactly)
    return 2;
}

int g_conv_np = 3;             // operand parts of the bf16 convolutions / weight gradients (san_set_conv_precision)

// Tile geometry of one 3x3 launch.  The default is tile_geom's (up to 256 pixels, four 16-pixel blocks per wave).  The SHORT form
// (round 6, VERDICT r5 #1d: the 40^2 level leaves a third of the chip idle): full-width tiles of at most 192 pixels = three blocks per
// wave, halo <= 256 staging units (three instead of four load slots per thread), chosen where the default launch does not fill the
// 256 compute units and the short one has more workgroups that still run in one round -- 144 -> 144 @40^2, N = 8: 168 workgroups of
// 40 x 6 -> 240 of 40 x 4, each with 3/4 of the matrix work and of the staging loads.  Two-fp16-part format only (the default mode).
struct TilePlan {
    TileGeom tg;
    int nbw;
};

TilePlan tile_plan(int n, int h, int w, int cin, int cout, int ks, bool f16fmt, bool shuffle) {
    TilePlan p{tile_geom(h, w), 4};
    if (ks != 3 || !f16fmt || shuffle || g_conv_np != 3 || g_b16_nbw == 4) return p;
    if (p.tg.tiles_x != 1 || p.tg.tw != w || (p.tg.tw == kTW && p.tg.th == kTH)) return p;      // only where the default is FLAT
    int th = 192 / w;
    if (th > h) th = h;
    while (th > 1 && (w + 2) * (th + 2) > kT) --th;
    if (th < 1 || w * th > 192 || (w + 2) * (th + 2) > kT) return p;
    const int ty = san_cdiv(h, th), thb = san_cdiv(h, ty);
    const int tiles4 = p.tg.tiles_x * p.tg.tiles_y;
    if (ty <= tiles4) return p;
    const int cgs = san_cdiv(san_cdiv(cout, 16), pick_mb_f16(cin, cout, tiles4 * n));
    const long long wgs4 = (long long)tiles4 * n * cgs, wgs3 = (long long)ty * n * cgs;
    if (g_b16_nbw == 3 || (wgs4 < 256 && wgs3 <= 256)) p = TilePlan{TileGeom{w, thb, 1, ty}, 3};
    return p;
}

// Per-tensor power-of-two scale of the fp16-format weight images (round 6, san_conv_f16_wscale_enable / SAN_F16_WSCALE=1).  OFF by
// default: it costs a max pass over every weight per optimiser step (+0.6 ms of a 41 ms step as measured) and buys 20 % of the
// end-to-end error (12 cascades, 320^2: 7.8e-5 -> 6.1e-5 vs float64; per layer 3e-7 -> 1e-7) and weights of ANY magnitude.
int g_f16_wscale = (getenv("SAN_F16_WSCALE") && atoi(getenv("SAN_F16_WSCALE")) == 1) ? 1 : 0;

// Which packed weight images hold two fp16 parts (packed with mode + 16) instead of bf16 parts: recorded by the pack entry
// points (host side), looked up by the launchers, so the convolution entry points need no format argument.
std::mutex g_fmt_mu;
std::unordered_map<const void*, int> g_fmt;
void note_format(const void* packed, int mode) {
    std::lock_guard<std::mutex> lk(g_fmt_mu);
    if (mode & 32) g_fmt[packed] = 2;
    else if (mode & 16) g_fmt[packed] = 1;
    else g_fmt.erase(packed);
}
int format_of(const void* packed) {
    std::lock_guard<std::mutex> lk(g_fmt_mu);
    auto it = g_fmt.find(packed);
    return it == g_fmt.end() ? 0 : it->second;
}

template <int MB, bool WD, int KS, bool FLAT, int NP, bool SWAP, bool F16 = false, bool F8 = false, int NBW = 4>
int launch_bfns(const BArgs& a, hipStream_t s) {
    // (only the parts in use are allocated: 33 KB for the two-fp16-part form, 17 KB for one part -- LDS never limits residency)
    constexpr size_t lds = NP * (size_t)kPartB + (WD ? (size_t)0 : (size_t)kSteps * MB * 3 * 64 * 16) + 2 * 48 * sizeof(float);
    static SanPerDevice configured;
    const int dev__ = san_current_device();
    if (!configured.has(dev__)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_kernel<MB, WD, KS, FLAT, NP, SWAP, F16, F8, NBW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            san_set_error("cannot reserve %d bytes of LDS for the bf16x3 convolution", (int)lds);
            return SAN_E_UNSUPPORTED;
        }
        configured.mark(dev__);
    }
    const int total = a.tiles_x * a.tiles_y * a.cgs * a.N * a.S;
    hipLaunchKernelGGL((conv_bf16x3_kernel<MB, WD, KS, FLAT, NP, SWAP, F16, F8, NBW>), dim3(total), dim3(kT), lds, s, a);
    return SAN_OK;
}

template <int MB, bool WD, int KS, bool FLAT, int NP>
int launch_bfn(const BArgs& a, hipStream_t s) {
    // the pixel-shuffle epilogue of the transposed form needs the channel-per-register layout
    // (measured: +3-5 % on the 32 x 8-tile wide layers, neutral on the 18- / 36-channel ones, -2-3 % on the full-width FLAT
    // tiles of the 20^2 / 40^2 levels, whose quads need per-lane row / column arithmetic)
    if (g_b16_swap && !a.shuffle && !FLAT) return launch_bfns<MB, WD, KS, FLAT, NP, true>(a, s);
    return launch_bfns<MB, WD, KS, FLAT, NP, false>(a, s);
}

template <int MB, bool WD, int KS, bool FLAT>
int launch_bf(const BArgs& a, hipStream_t s) {
    if (a.fmt == 1) {                               // two fp16 parts (forward operands, fp32-equivalent mode only)
        if constexpr (FLAT && KS == 3) {
            if (a.nbw == 3) return launch_bfns<MB, WD, KS, FLAT, 2, false, true, false, 3>(a, s);
        }
        if (g_b16_swap && !a.shuffle && !FLAT) return launch_bfns<MB, WD, KS, FLAT, 2, true, true>(a, s);
        return launch_bfns<MB, WD, KS, FLAT, 2, false, true>(a, s);
    }
    if (a.fmt == 2) {                               // one fp8 part (forward operands of the "fp8" mode)
        if (g_b16_swap && !a.shuffle && !FLAT) return launch_bfns<MB, WD, KS, FLAT, 1, true, false, true>(a, s);
        return launch_bfns<MB, WD, KS, FLAT, 1, false, false, true>(a, s);
    }
    switch (g_conv_np) {
        case 1: return launch_bfn<MB, WD, KS, FLAT, 1>(a, s);
        case 2: return launch_bfn<MB, WD, KS, FLAT, 2>(a, s);
        default: return launch_bfn<MB, WD, KS, FLAT, 3>(a, s);
    }
}

template <int MB, bool WD, int KS = 3>
int launch_b(const BArgs& a, hipStream_t s) {
    return (a.tw == kTW && a.th == kTH) ? launch_bf<MB, WD, KS, false>(a, s) : launch_bf<MB, WD, KS, true>(a, s);
}



}  // namespace

extern "C" {

// Narrow-precision modes of every bf16 matrix-core convolution and weight gradient (BASELINE configs 2 / 5 name bf16 /
// fp8 U-Net convolutions judged by PSNR): parts = 3 fp32-equivalent (default), 2 = 16 mantissa bits (three products),
// 1 = plain bf16 (one product).  FFT, data consistency, normalisation statistics and losses stay fp32 in every mode.
// 1 / 0: fp16-format weight images packed FROM NOW ON carry / do not carry the tensor's power-of-two scale (see g_f16_wscale);
// on < 0: query.  Returns the previous setting.  Images packed earlier keep what they have (re-pack them: the registry's epoch).
int san_conv_f16_wscale_enable(int on) {
    const int prev = g_f16_wscale;
    if (on >= 0) g_f16_wscale = on ? 1 : 0;
    return prev;
}

int san_set_conv_precision(int parts) {
    SAN_CHECK_ARG(parts >= 1 && parts <= 3, "parts must be 1, 2 or 3");
    g_conv_np = parts;
    san_wgrad_set_parts(parts);
    return SAN_OK;
}

int san_get_conv_precision(void) { return g_conv_np; }

// tuning / test hook: wd = -1 automatic, 0 LDS-staged weights, 1 weights-direct where MB <= 4; mb = -1 automatic or 2..5
int san_conv_bf16x3_set_tuning(int wd, int mb) {
    g_b16_wd = wd;
    g_b16_mb = mb;
    return SAN_OK;
}

// tuning / test hook of the round-6 launch-plan choices: nbw = 0 automatic, 4 never the short-tile form, 3 wherever it fits;
// wd_cold = 1 LDS-staged weights for launches of at most 256 workgroups (default), 0 the weights-direct form there too
int san_conv_bf16x3_tile_set_tuning(int nbw, int wd_cold) {
    g_b16_nbw = nbw;
    g_b16_wd_cold = wd_cold;
    return SAN_OK;
}

// 1 when san_conv2d_bf16x3_fwd takes this layer (3x3, channel counts that fill 16-wide tiles); the caller then
// packs the weights with san_conv_bf16x3_pack and sizes statistics with san_conv_bf16x3_stat_tiles.
// Tuning hook (builds with -DSAN_B16_TIMELINE only; -1 otherwise): buf = device array of 8 x u64 per workgroup of the NEXT
// bf16x3 / fp16-part convolution launches (null: off).
// Each workgroup records the 100 MHz clock at its start, after its first chunk is staged, at the epilogue start and at its
// end (entries 0-3), and HW_ID / XCC_ID (4, 5): scratch/conv_timeline.py turns that into per-CU occupancy and phase times.
int san_conv_bf16x3_debug_timeline(void* buf) {
#ifdef SAN_B16_TIMELINE
    g_b16_dbg = static_cast<unsigned long long*>(buf);
    return 0;
#else
    (void)buf;
    san_set_error("san_conv_bf16x3_debug_timeline: build with SAN_EXTRA_HIPCC_FLAGS=-DSAN_B16_TIMELINE");
    return -1;
#endif
}

int san_conv_bf16x3_eligible(int cin, int cout, int h, int w, int ks) {
    if (ks != 3) return 0;
    // measured against the fp32 4x4x1 kernel (N = 8): faster from 18 -> 18 (66 vs 92 us @320^2) and 24 -> 24 upwards
    // since the weights-direct form; 16 -> 16 ties, below that the 16-wide tiles are mostly padding (8 -> 8: 67 vs 33 us)
    const int lo = cin < cout ? cin : cout, hi = cin < cout ? cout : cin;
    if (lo < 16 || hi < 18) return 0;
    if (h < 8 || w < 16) return 0;
    return 1;
}

// (+ 16 bytes behind the image: the fp8 format keeps its per-tensor weight scale there)
size_t san_conv_bf16x3_packed_bytes_ks(int cout, int cin, int ks) { return bplan(cout, cin, ks).packed_elems * 2 + 16; }
size_t san_conv_bf16x3_packed_bytes(int cout, int cin) { return san_conv_bf16x3_packed_bytes_ks(cout, cin, 3); }

int san_conv_bf16x3_stat_tiles(int n, int h, int w) {
    (void)n;
    const TileGeom tg = tile_geom(h, w);
    return tg.tiles_x * tg.tiles_y * 4;
}

int san_conv3x3_bf16x3_stat_tiles(int n, int h, int w, int cin, int cout, int f16_format) {
    const TileGeom tg = tile_plan(n, h, w, cin, cout, 3, f16_format != 0, false).tg;
    return tg.tiles_x * tg.tiles_y * 4;
}

int san_conv_bf16x3_pack_ks(const float* w, void* packed, int cout, int cin, int mode, int ks, void* stream) {
    SAN_CHECK_ARG(w && packed, "null pointer");
    SAN_CHECK_ARG(cout > 0 && cin > 0 && ((mode & 15) == 0 || (mode & 15) == 2) && (mode & ~48) == (mode & 15) && (mode & 48) != 48 && (ks == 1 || ks == 3), "bad dims / mode / ks");
    // mode 2: `cout`, `cin` are those of the DATA-GRADIENT convolution (cout = forward cin, cin = forward cout)
    const BPlan p = bplan(cout, cin, ks);
    note_format(packed, mode);
    if ((mode & 32) || ((mode & 16) && g_f16_wscale)) {
        hipLaunchKernelGGL(fp8_wscale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (size_t)cout * cin * (ks == 1 ? 1 : 9),
                           reinterpret_cast<float*>(static_cast<uint16_t*>(packed) + p.packed_elems), (mode & 32) ? 7 : 13);
        SAN_LAUNCH_CHECK();
    }
    size_t blocks = (p.packed_elems + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)packed,
                       p.packed_elems, cout, cin, p.nblkp, mode, ks);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int san_conv_bf16x3_pack(const float* w, void* packed, int cout, int cin, int mode, void* stream) {
    return san_conv_bf16x3_pack_ks(w, packed, cout, cin, mode, 3, stream);
}

int san_conv_bf16x3_pack_job_ks(long long* job8, const float* w, void* packed, int cout, int cin, int mode, int ks) {
    SAN_CHECK_ARG(job8 && w && packed, "null pointer");
    SAN_CHECK_ARG(cout > 0 && cin > 0 && ((mode & 15) == 0 || (mode & 15) == 2) && (mode & ~48) == (mode & 15) && (mode & 48) != 48 && (ks == 1 || ks == 3), "bad dims / mode / ks");
    const BPlan p = bplan(cout, cin, ks);
    note_format(packed, mode);
    job8[0] = (long long)(uintptr_t)w;
    job8[1] = (long long)(uintptr_t)packed;
    job8[2] = cout;
    job8[3] = cin;
    job8[4] = p.nblkp;
    job8[5] = ks;
    job8[6] = mode | (((mode & 16) && g_f16_wscale) ? 256 : 0);       // (bit 8: this fp16 image carries a per-tensor scale)
    job8[7] = (long long)p.packed_elems;
    return SAN_OK;
}

int san_conv_bf16x3_pack_job(long long* job8, const float* w, void* packed, int cout, int cin, int mode) {
    return san_conv_bf16x3_pack_job_ks(job8, w, packed, cout, cin, mode, 3);
}

int san_conv_bf16x3_pack_batch(const long long* jobs_dev, int njobs, void* stream) {
    return san_conv_bf16x3_pack_batch_grid(jobs_dev, njobs, 64, 1, stream);
}

// The same with `blocks` workgroups per job (1..64) and the fp8 scale pass only when some job of the run needs it.  The caller sorts
// its jobs by size and packs each size class with a grid that fits it (round 5): with 64 workgroups for every one of ~660 jobs, most of
// them a few hundred weights, the launch spent its 0.3 ms dispatching 42,000 workgroups that had nothing to do.
int san_conv_bf16x3_pack_batch_grid(const long long* jobs_dev, int njobs, int blocks, int fp8, void* stream) {
    SAN_CHECK_ARG(jobs_dev && njobs > 0, "empty job table");
    SAN_CHECK_ARG(blocks >= 1 && blocks <= 64, "1..64 workgroups per job");
    if (fp8) hipLaunchKernelGGL(fp8_wscale_batch_kernel, dim3(njobs), dim3(1024), 0, (hipStream_t)stream, jobs_dev);    // (returns at once for other formats)
    hipLaunchKernelGGL(pack_bf16x3_batch_kernel, dim3(blocks, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

static int conv_bf16x3_run(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                           float in_slope, const void* w_packed, const float* bias, float* y, int y_ctot, int y_coff, int cout,
                           float* part_stats, int n, int h, int w, int ks, void* stream, int shuffle = 0,
                           void* ws = nullptr, size_t ws_bytes = 0, const void* amax = nullptr, float* fin_scale = nullptr,
                           float* fin_shift = nullptr, int fin_ctot = 0, int fin_coff = 0, float fin_eps = 0.f) {
    SAN_CHECK_ARG(x && w_packed && y, "null pointer");
    SAN_CHECK_ARG(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "bad dims");
    SAN_CHECK_ARG(x_coff >= 0 && x_coff + cin <= x_ctot && y_coff >= 0 && y_coff + (shuffle ? cout / 4 : cout) <= y_ctot, "bad channel view");
    SAN_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "in_scale/in_shift must come together");
    SAN_CHECK_ARG(!shuffle || (ks == 1 && cout % 4 == 0 && (((uintptr_t)y) & 7) == 0), "transposed form: 1x1, 4 virtual channels per channel, 8-byte aligned output");
    const BPlan p = bplan(cout, cin, ks);
    BArgs a{};
    a.x = x;
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_slope = in_slope;
    a.wp = (const uint4*)w_packed;
    a.fmt = format_of(w_packed);
    SAN_CHECK_ARG(a.fmt != 1 || g_conv_np == 3, "fp16-format weights are for the fp32-equivalent mode only");
    SAN_CHECK_ARG(a.fmt != 2 || g_conv_np == 1, "fp8-format weights are for the one-part mode only (san_set_conv_precision(1))");
    a.amax = a.fmt == 1 ? static_cast<const uint32_t*>(amax) : nullptr;
    a.f8_tail = a.fmt >= 1 ? reinterpret_cast<const float*>(static_cast<const uint16_t*>(w_packed) + p.packed_elems) : nullptr;     // {S_w, 1 / S_w}
    // the persistent form: two fp16 parts (fp32-equivalent mode) or one bf16 part (cout 18 / 36: the network's layers)
    const bool stream_f16 = a.fmt == 1 && g_conv_np == 3, stream_bf1 = a.fmt == 0 && g_conv_np == 1 && (cout == 18 || cout == 36);
    if (ks == 3 && (stream_f16 || stream_bf1) && !shuffle && g_b16_mb < 0 && g_b16_wd < 0 &&
        san_conv_stream_eligible(n, h, w, cin, cout, x_ctot))
        return san_conv_stream_run(x, x_ctot, x_coff, cin, in_scale, in_shift, in_slope, w_packed, p.nblkp, bias, y, y_ctot, y_coff, cout,
                                   part_stats, a.amax, n, h, w, stream, stream_f16 ? 2 : 1, stream_f16 ? a.f8_tail : nullptr);
    if (ks == 1 && ((a.fmt == 1 && g_conv_np == 3) || (a.fmt == 0 && g_conv_np == 1)) && g_b16_mb < 0 && san_gemm1x1_enabled()) {
        // round 5: the whole K range staged once, no barrier between K-steps (san_conv1x1.hip)
        const TileGeom tg1 = tile_geom(h, w);
        SanGemm1x1Args g{};
        g.x = x;
        g.in_scale = in_scale;
        g.in_shift = in_shift;
        g.in_slope = in_slope;
        g.wp = w_packed;
        g.bias = bias;
        g.y = y;
        g.part = part_stats;
        g.amax = a.amax;
        g.x_ctot = x_ctot;
        g.x_coff = x_coff;
        g.cin = cin;
        g.y_ctot = y_ctot;
        g.y_coff = y_coff;
        g.cout = cout;
        g.N = n;
        g.H = h;
        g.W = w;
        g.chunks = p.chunks;
        g.nblkp = p.nblkp;
        g.shuffle = shuffle;
        g.bf1 = a.fmt == 0 ? 1 : 0;
        g.w_tail = a.fmt == 1 ? a.f8_tail : nullptr;
        g.slots = tg1.tiles_x * tg1.tiles_y * 4;
        return san_gemm1x1_f16_run(g, stream);
    }
    a.dbg = g_b16_dbg;
    a.shuffle = shuffle;
    a.bias = bias;
    a.y = y;
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
    const TilePlan tpl = tile_plan(n, h, w, cin, cout, ks, a.fmt == 1, shuffle != 0);
    const TileGeom tg = tpl.tg;
    a.nbw = tpl.nbw;
    a.tiles_x = tg.tiles_x;
    a.tiles_y = tg.tiles_y;
    a.tw = tg.tw;
    a.th = tg.th;
    a.hp = tg.tw + 2;
    a.npx = (tg.tw + 2) * (tg.th + 2);
    // (the one-part bf16 mode has even less matrix work per chunk than the fp16-part format: the same small-workgroup choice --
    // 72->144 @80^2 45.7 us with pick_mb's five blocks, 28-34 with two / three; 36->72 @160^2 66.6 -> 41.8)
    int mb = (a.fmt >= 1 || g_conv_np == 1) && ks == 3 ? pick_mb_f16(cin, cout, a.tiles_x * a.tiles_y * n) : pick_mb(cout, a.tiles_x * a.tiles_y * n);
    if (g_b16_mb >= 2 && g_b16_mb <= 5 && g_b16_mb <= san_cdiv(cout, 16)) mb = g_b16_mb;
    a.cgs = san_cdiv(san_cdiv(cout, 16), mb);
    a.chunks = p.chunks;
    a.nblkp = p.nblkp;
    a.S = 1;
    if (ws && !shuffle) {
        const int S = splitk_parts(a.tiles_x * a.tiles_y * n * a.cgs, p.chunks, ks, h * w, (long long)n * cout * h * w);
        if (S > 1 && ws_bytes >= splitk_bytes(S, n, cout, h * w)) {
            a.S = S;
            a.ws = static_cast<float*>(ws);
        }
    }
    {
        const unsigned long long total = (unsigned long long)a.tiles_x * a.tiles_y * a.cgs * a.N * a.S;
        const int ntile = a.tiles_x * a.tiles_y;
        // every dividend is below `total` (or chunks (S + 1), or 256 + 3 * 84 for the per-thread ones): exact while x d < 2^32
        auto magic = [&](int d, unsigned long long xmax) -> unsigned {
            return (d > 1 && xmax * (unsigned long long)d < 0xffffffffull) ? (unsigned)(0x100000000ull / (unsigned)d + 1) : 0u;
        };
        const unsigned long long xm = total > (unsigned long long)a.chunks * (a.S + 1) ? total : (unsigned long long)a.chunks * (a.S + 1);
        a.m_S = magic(a.S, xm);
        a.m_cgs = magic(a.cgs, xm);
        a.m_cgsnt = magic(a.cgs * ntile, xm);
        a.m_tx = magic(a.tiles_x, xm);
        a.m_hp = magic(a.hp, 1024);
        a.m_rem = magic(a.npx - kT > 0 ? a.npx - kT : 1, 1024);
        a.m_tw = magic(a.tw, 1024);
    }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (ks == 1) {
        switch (mb) {
            case 2: rc = launch_b<2, true, 1>(a, s); break;
            case 3: rc = launch_b<3, true, 1>(a, s); break;
            case 4: rc = launch_b<4, true, 1>(a, s); break;
            default: rc = launch_b<5, true, 1>(a, s); break;
        }
    } else {
        int wd = (mb <= 4 && p.chunks <= 6 && h * w >= 1600) ? 1 : 0;        // see the WD note at the kernel
        if (a.fmt >= 1 && mb <= 4) wd = 1;      // fp16 parts: half the matrix work per chunk, residency wins at every depth
        // ... except where the launch has at most one workgroup per compute unit anyway (the 40^2 level, the 36-channel data gradient
        // at 80^2): nothing shares the unit, and the weights-direct K-steps run one step (0.4 us) ahead of operands that, INSIDE THE
        // STEP, come from HBM (every layer's image was packed at the start of the step, ~100 MB of traffic ago) -- the LDS-staged form
        // requests a chunk's weights a whole chunk ahead.  Round 6, rocprofv3 averages with a different weight image and input tensor per
        // launch (scratch/r6_cold_probe.py; L2-hot figures in brackets): 144->144 @40^2 42.1 -> 32.3 us (28.8 -> 31.0), 72->144 @40^2
        // 24.6 -> 20.3 (18.2 -> 19.6), 72->36 @80^2 26.4 -> 22.4; with more workgroups than units the direct form stays ahead
        // (288->144 @40^2 split in two: 47.4 vs 58.9, 72->72 @80^2 31.6 vs 37.6).
        if (a.fmt >= 1 && wd && g_b16_wd_cold && (long long)a.tiles_x * a.tiles_y * a.cgs * a.N * a.S <= 256) wd = 0;
        if (g_b16_wd >= 0) wd = g_b16_wd && mb <= 4;
        switch (mb * 2 + wd) {
            case 4: rc = launch_b<2, false>(a, s); break;
            case 5: rc = launch_b<2, true>(a, s); break;
            case 6: rc = launch_b<3, false>(a, s); break;
            case 7: rc = launch_b<3, true>(a, s); break;
            case 8: rc = launch_b<4, false>(a, s); break;
            case 9: rc = launch_b<4, true>(a, s); break;
            default: rc = launch_b<5, false>(a, s); break;
        }
    }
    if (rc != SAN_OK) return rc;
    SAN_LAUNCH_CHECK();
    if (a.S > 1) {
        const TileGeom tgs = tg;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(n * cout), dim3(256), 0, s, a.ws, a.S, bias, y, y_ctot, y_coff, part_stats,
                           tgs.tiles_x * tgs.tiles_y * 4, n, cout, h * w, fin_scale, fin_shift, fin_ctot, fin_coff, fin_eps);
        SAN_LAUNCH_CHECK();
        return fin_scale ? 1 : SAN_OK;                 // 1: the InstanceNorm affine was finalised in the reduction
    }
    return SAN_OK;
}

int san_conv2d_bf16x3_fwd(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                          float in_slope, const void* w_packed, const float* bias, float* y, int y_ctot, int y_coff,
                          int cout, float* part_stats, int n, int h, int w, void* stream) {
    return conv_bf16x3_run(x, x_ctot, x_coff, cin, in_scale, in_shift, in_slope, w_packed, bias, y, y_ctot, y_coff, cout,
                           part_stats, n, h, w, 3, stream);
}

// 1x1 form: weights packed with san_conv_bf16x3_pack_ks(..., ks = 1); otherwise the contract of san_conv2d_bf16x3_fwd
int san_conv1x1_bf16x3_eligible(int cin, int cout, int h, int w) {
    if (cin < 16 || cout < 16) return 0;
    if (h < 8 || w < 16) return 0;
    return 1;
}

int san_conv1x1_bf16x3_fwd(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                           float in_slope, const void* w_packed, const float* bias, float* y, int y_ctot, int y_coff,
                           int cout, float* part_stats, int n, int h, int w, void* stream) {
    return conv_bf16x3_run(x, x_ctot, x_coff, cin, in_scale, in_shift, in_slope, w_packed, bias, y, y_ctot, y_coff, cout,
                           part_stats, n, h, w, 1, stream);
}

// ConvTranspose2d(2x2, stride 2, no bias) (varnet.py:159-192) on the same kernel: a 1x1 convolution to 4 cout virtual
// channels whose epilogue writes the pixel shuffle.  x [n, x_ctot, h, w] -> y [n, y_ctot, 2h, 2w]; weights: the
// [Cin, Cout, 2, 2] tensor packed with san_conv_bf16x3_pack_ks(w, packed, 4 * cout, cin, mode 2, ks 1);
// part_stats [n, cout, 4 * san_conv_bf16x3_stat_tiles(n, h, w), 3].
int san_tconv2x2_bf16x3_eligible(int cin, int cout, int h, int w) { return san_conv1x1_bf16x3_eligible(cin, 4 * cout, h, w); }

int san_tconv2x2_bf16x3_fwd(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                            float in_slope, const void* w_packed, float* y, int y_ctot, int y_coff, int cout,
                            float* part_stats, int n, int h, int w, void* stream) {
    return conv_bf16x3_run(x, x_ctot, x_coff, cin, in_scale, in_shift, in_slope, w_packed, nullptr, y, y_ctot, y_coff,
                           4 * cout, part_stats, n, h, w, 1, stream, 1);
}

// Split-K form of san_conv2d_bf16x3_fwd for deep-K layers on small images (288 -> 288 @20^2 ...): ws = device scratch of
// san_conv_bf16x3_ws_bytes(...) bytes (0: the layer is not split; the plain entry point does the same work).  The partial
// outputs are added in a fixed order by a second launch: deterministic.
size_t san_conv_bf16x3_ws_bytes(int n, int h, int w, int cin, int cout, int ks) {
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || ks != 3) return 0;
    const BPlan p = bplan(cout, cin, ks);
    const TileGeom tg = tile_geom(h, w);
    // the run picks its block count per operand format; size the scratch for whichever of the two splits further
    int S = 1;
    for (int f = 0; f < 2; ++f) {
        int mb = f ? pick_mb_f16(cin, cout, tg.tiles_x * tg.tiles_y * n) : pick_mb(cout, tg.tiles_x * tg.tiles_y * n);
        if (g_b16_mb >= 2 && g_b16_mb <= 5 && g_b16_mb <= san_cdiv(cout, 16)) mb = g_b16_mb;
        const int groups = tg.tiles_x * tg.tiles_y * n * san_cdiv(san_cdiv(cout, 16), mb);
        const int Sf = splitk_parts(groups, p.chunks, ks, h * w, (long long)n * cout * h * w);
        if (Sf > S) S = Sf;
    }
    return S > 1 ? splitk_bytes(S, n, cout, h * w) : 0;
}

int san_conv2d_bf16x3_fwd_ws(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                             float in_slope, const void* w_packed, const float* bias, float* y, int y_ctot, int y_coff,
                             int cout, float* part_stats, int n, int h, int w, void* ws, size_t ws_bytes, void* stream) {
    return conv_bf16x3_run(x, x_ctot, x_coff, cin, in_scale, in_shift, in_slope, w_packed, bias, y, y_ctot, y_coff, cout,
                           part_stats, n, h, w, 3, stream, 0, ws, ws_bytes);
}

// san_conv2d_bf16x3_fwd_ws for a layer followed by InstanceNorm2d: when the launch is split over K, the second pass sees every
// (sample, channel) plane whole and writes the lazy affine itself (scale / shift views [n, sc_ctot] at channel offset sc_coff,
// exactly san_norm_finalize(SAN_NORM_INSTANCE)'s values): *finalised = 1 and part_stats is left untouched -- the caller skips
// its san_norm_finalize launch.  Otherwise *finalised = 0 and part_stats holds the per-tile partials as usual.
int san_conv2d_bf16x3_fwd_ws_in(const float* x, int x_ctot, int x_coff, int cin, const float* in_scale, const float* in_shift,
                                float in_slope, const void* w_packed, const float* bias, float* y, int y_ctot, int y_coff,
                                int cout, float* part_stats, int n, int h, int w, void* ws, size_t ws_bytes, float* scale,
                                float* shift, int sc_ctot, int sc_coff, float eps, int* finalised, void* stream) {
    SAN_CHECK_ARG(scale && shift && finalised && part_stats, "null pointer");
    SAN_CHECK_ARG(sc_coff >= 0 && sc_coff + cout <= sc_ctot, "bad scale/shift view");
    const int rc = conv_bf16x3_run(x, x_ctot, x_coff, cin, in_scale, in_shift, in_slope, w_packed, bias, y, y_ctot, y_coff, cout,
                                   part_stats, n, h, w, 3, stream, 0, ws, ws_bytes, nullptr, scale, shift, sc_ctot, sc_coff, eps);
    *finalised = rc == 1 ? 1 : 0;
    return rc == 1 ? SAN_OK : rc;
}

// Data gradient on fp16-format weights (packed with mode 2 + 16): x = dy [n, cout_fwd, h, w] materialised, amax = device
// pointer to the bits of max |dy| (san_act_bwd* maintain it).  ks = 3 or 1; ws / ws_bytes as in san_conv2d_bf16x3_fwd_ws
// (may be NULL / 0).  With bf16-format weights or amax == NULL this is san_conv2d_bf16x3_fwd_ws / san_conv1x1_bf16x3_fwd.
int san_conv_bf16x3_dgrad_amax(const float* dy, int dy_ctot, int dy_coff, int cin, const void* w_packed, float* dx, int dx_ctot,
                               int dx_coff, int cout, const void* amax, int n, int h, int w, int ks, void* ws, size_t ws_bytes,
                               void* stream) {
    SAN_CHECK_ARG(ks == 1 || ks == 3, "ks must be 1 or 3");
    return conv_bf16x3_run(dy, dy_ctot, dy_coff, cin, nullptr, nullptr, 1.f, w_packed, nullptr, dx, dx_ctot, dx_coff, cout, nullptr,
                           n, h, w, ks, stream, 0, ks == 3 ? ws : nullptr, ks == 3 ? ws_bytes : 0, amax);
}

}  // extern "C"
