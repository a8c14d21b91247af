// 2-D complex FFT with fused k-space prologues/epilogues for gfx950.
//
// A 2-D transform of an H x W plane is two batched 1-D passes.  One plane
// (320 x 320 x 8 B = 800 KiB) does not fit a CU's 160 KiB LDS, so each pass
// stages a batch of B lines in LDS, runs a mixed-radix Stockham autosort FFT on
// them (twiddles staged in LDS, ping-pong buffers), and touches global memory
// exactly once on the way in and once on the way out.  Everything the reference
// does around its torch.fft calls (sensitivity multiply, coil sum, soft data
// consistency, root-sum-of-squares, column masks, planar<->complex) is folded
// into those two touches, so the only extra HBM traffic is the inter-pass
// intermediate, which is L2/MALL sized (N*C*H*W*8 B).
//
//   row pass   : B full rows per workgroup, contiguous 8-byte-per-lane accesses
//   column pass: a tile of B adjacent columns x H rows per workgroup; B*8-byte
//                contiguous segments per row (64 or 128 B), LDS pitch H+1 to
//                keep the transposing stores conflict-free.
#include "san_common.h"

#include <cmath>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxRad = 12;

enum RowPro { RP_NONE = 0, RP_PLANAR_MUL_SENS = 1 };
enum RowEpi { RE_STORE = 0, RE_STORE_PLANAR = 1, RE_CONJ_SENS_SUM_PLANAR = 2, RE_RSS = 3 };
enum ColEpi { CE_STORE = 0, CE_DC = 1, CE_DC_NEXT = 2 };   // _NEXT: also the next cascade's inverse column pass

struct FftArgs {
    const float2* in;
    float2* out;
    const float* in_planar;  // RP_PLANAR_MUL_SENS: [n, 2, H, W]
    const float2* sens;      // [n, C, H, W]
    const float2* k;         // CE_DC
    const float2* k0;        // CE_DC
    const float* mask;       // CE_DC: [W]
    const float* dcw;        // CE_DC: 1 float
    float2* out2;            // CE_DC_NEXT: inverse column transform of the DC result (the next sens_reduce's first pass)
    float* out_real;         // planar / rss output
    const float* cm_in;      // optional column mask on load  [W]
    const float* cm_out;     // optional column mask on store [W]
    const float2* tw;        // W_len^k, k in [0, len)
    int C, H, W;
    int len;                 // transform length of this pass
    int B;                   // lines per workgroup (power of two, <= 256)
    int logB;
    int pitch;               // LDS pitch in float2
    int inner;               // coil-loop count (reduce epilogues) else 1
    int out_ctot;            // planar outputs: channels per sample in the destination (>= 2)
    int nrad;
    int rad[kMaxRad];
    int ns_shift[kMaxRad];   // log2(Ns) if Ns is a power of two else -1
    float scale;
    float sgn;               // +1 forward, -1 inverse
    int pro, epi;
    // image-domain cascade boundary (dc_rows kernels)
    float2* dk_out;          // forward: mask * (fft_x(x) - k0x), kept for the dc_weight gradient (or null)
    const float2* dk_in;     // backward: that tensor of the forward pass
    float* dcw_part;         // backward: one partial of Re sum conj(fft_x(g)) dk_in per workgroup
    float m_scale;           // factor on the coil-combined planar output (-1 in the backward form)
    float* m_stats;          // forward, 320-wide kernel: [n][2][workgroups per sample][3] = (count, mean, M2) of this workgroup's rows of the
                             // two planes of the coil-combined output (the next cascade's NormUnet statistics, varnet.py:262-273), or null
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// One Stockham pass of radix R over `nseq` sequences (one per thread group of
// `tps` threads).  src/dst are the sequence bases of THIS thread's sequence.
template <int R>
__device__ __forceinline__ void stockham_pass_reg(const float2* __restrict__ src, float2* __restrict__ dst,
                                                  const float2* __restrict__ tw, int n, int Ns, int ns_shift,
                                                  int lane, int tps, float sgn) {
    const int m = n / R;           // butterflies per sequence
    const int ts = n / (Ns * R);   // twiddle stride
    const int wr = n / R;          // W_R^1 = tw[n/R]
    for (int j = lane; j < m; j += tps) {
        int kk, jq;
        if (ns_shift >= 0) {
            kk = j & (Ns - 1);
            jq = j >> ns_shift;
        } else {
            jq = j / Ns;
            kk = j - jq * Ns;
        }
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = src[j + r * m];
#pragma unroll
        for (int r = 1; r < R; ++r) v[r] = cmul(v[r], tw[r * kk * ts]);
        const int j0 = jq * Ns * R + kk;
        if (R == 2) {
            dst[j0] = cadd(v[0], v[1]);
            dst[j0 + Ns] = csub(v[0], v[1]);
        } else if (R == 4) {
            float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
            float2 c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
            float2 dr = make_float2(sgn * d.y, -sgn * d.x);   // d * (-i) forward, d * (+i) inverse
            dst[j0] = cadd(a, c);
            dst[j0 + Ns] = cadd(b, dr);
            dst[j0 + 2 * Ns] = csub(a, c);
            dst[j0 + 3 * Ns] = csub(b, dr);
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) {
                float2 acc = v[0];
#pragma unroll
                for (int r = 1; r < R; ++r) acc = cadd(acc, cmul(v[r], tw[((q * r) % R) * wr]));
                dst[j0 + q * Ns] = acc;
            }
        }
    }
}

// Odd prime radix R in registers, with the conjugate symmetry of the DFT matrix: pairing inputs r and R - r,
//   a_r = v_r + v_(R-r), b_r = v_r - v_(R-r):  out_q = v_0 + sum_r a_r cos_qr + i sum_r b_r sin_qr,  out_(R-q) = v_0 + sum - i sum
// (r = 1 .. (R-1)/2; cos / sin of 2 pi k / R with the transform direction come from the twiddle table).  (R-1)^2 real
// multiply-adds per butterfly instead of the 4 (R-1)^2 of the direct form: the radix-23
// pass of the 368-wide multi-coil planes (368 = 4 x 4 x 23) went from ~45 % of the kernel's time to a few per cent.
template <int R>
__device__ __forceinline__ void stockham_pass_odd(const float2* __restrict__ src, float2* __restrict__ dst,
                                                  const float2* __restrict__ tw, int n, int Ns, int ns_shift,
                                                  int lane, int tps) {
    constexpr int HR = (R - 1) / 2;
    const int m = n / R;
    const int ts = n / (Ns * R);
    const int wr = n / R;
    for (int j = lane; j < m; j += tps) {
        int kk, jq;
        if (ns_shift >= 0) {
            kk = j & (Ns - 1);
            jq = j >> ns_shift;
        } else {
            jq = j / Ns;
            kk = j - jq * Ns;
        }
        float2 a[HR], b[HR];
        const float2 v0 = src[j];
        float2 s0 = v0;
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            const float2 lo = cmul(src[j + (r + 1) * m], tw[(r + 1) * kk * ts]);
            const float2 hi = cmul(src[j + (R - 1 - r) * m], tw[(R - 1 - r) * kk * ts]);
            a[r] = cadd(lo, hi);
            b[r] = csub(lo, hi);
            s0 = cadd(s0, a[r]);
        }
        const int j0 = jq * Ns * R + kk;
        dst[j0] = s0;
        // one output pair per iteration, NOT unrolled: with all (R-1)/2 independent sums in flight the radix-23 pass alone
        // takes > 256 registers (one wave per SIMD for every kernel that inlines it).  The coefficients W^(q r) are
        // wave-uniform LDS reads of the twiddle table (cos, direction * sin).
#pragma unroll 1
        for (int q = 1; q <= HR; ++q) {
            float2 P = v0, Q = make_float2(0.f, 0.f);
            int qr = 0;
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                qr += q;
                if (qr >= R) qr -= R;
                const float2 t = tw[qr * wr];
                P.x = fmaf(a[r].x, t.x, P.x);
                P.y = fmaf(a[r].y, t.x, P.y);
                Q.x = fmaf(b[r].x, t.y, Q.x);
                Q.y = fmaf(b[r].y, t.y, Q.y);
            }
            // i Q = (-Q.y, Q.x)
            dst[j0 + q * Ns] = make_float2(P.x - Q.y, P.y + Q.x);
            dst[j0 + (R - q) * Ns] = make_float2(P.x + Q.y, P.y - Q.x);
        }
    }
}

// Any other (prime) radix: twiddle the R inputs in place (each butterfly owns
// its inputs), then a direct DFT reading them back from LDS.
__device__ __forceinline__ void stockham_pass_any(float2* __restrict__ src, float2* __restrict__ dst,
                                                  const float2* __restrict__ tw, int n, int R, int Ns,
                                                  int ns_shift, int lane, int tps) {
    const int m = n / R;
    const int ts = n / (Ns * R);
    const int wr = n / R;
    for (int j = lane; j < m; j += tps) {
        int kk, jq;
        if (ns_shift >= 0) {
            kk = j & (Ns - 1);
            jq = j >> ns_shift;
        } else {
            jq = j / Ns;
            kk = j - jq * Ns;
        }
        for (int r = 1; r < R; ++r) src[j + r * m] = cmul(src[j + r * m], tw[r * kk * ts]);
        const int j0 = jq * Ns * R + kk;
        for (int q = 0; q < R; ++q) {
            float2 acc = src[j];
            int qr = 0;
            for (int r = 1; r < R; ++r) {
                qr += q;
                if (qr >= R) qr -= R;
                acc = cadd(acc, cmul(src[j + r * m], tw[qr * wr]));
            }
            dst[j0 + q * Ns] = acc;
        }
    }
}

// Runs all passes; returns the buffer that holds the result.  flat_rows > 0: the butterflies of the prime radices above 5
// are numbered over the whole workgroup (flat_rows active sequences) instead of per sequence.
__device__ __forceinline__ float2* run_fft(const FftArgs& a, float2* bufA, float2* bufB, const float2* tw,
                                           int seq, int lane, int tps, bool active, float sgn, int flat_rows = 0) {
    float2* src = bufA;
    float2* dst = bufB;
    int Ns = 1;
    for (int p = 0; p < a.nrad; ++p) {
        const int R = a.rad[p];
        if (flat_rows > 0 && (R == 7 || R == 11 || R == 13 || R == 23)) {
            const int m = a.len / R;
            for (int g = threadIdx.x; g < flat_rows * m; g += kThreads) {
                const int sq = g / m, j = g - sq * m;
                float2* s = src + sq * a.pitch;
                float2* d = dst + sq * a.pitch;
                switch (R) {
                    case 7: stockham_pass_odd<7>(s, d, tw, a.len, Ns, a.ns_shift[p], j, m); break;
                    case 11: stockham_pass_odd<11>(s, d, tw, a.len, Ns, a.ns_shift[p], j, m); break;
                    case 13: stockham_pass_odd<13>(s, d, tw, a.len, Ns, a.ns_shift[p], j, m); break;
                    default: stockham_pass_odd<23>(s, d, tw, a.len, Ns, a.ns_shift[p], j, m); break;
                }
            }
        } else if (active) {
            float2* s = src + seq * a.pitch;
            float2* d = dst + seq * a.pitch;
            switch (R) {
                case 2: stockham_pass_reg<2>(s, d, tw, a.len, Ns, a.ns_shift[p], lane, tps, sgn); break;
                case 3: stockham_pass_reg<3>(s, d, tw, a.len, Ns, a.ns_shift[p], lane, tps, sgn); break;
                case 4: stockham_pass_reg<4>(s, d, tw, a.len, Ns, a.ns_shift[p], lane, tps, sgn); break;
                case 5: stockham_pass_reg<5>(s, d, tw, a.len, Ns, a.ns_shift[p], lane, tps, sgn); break;
                case 7: stockham_pass_odd<7>(s, d, tw, a.len, Ns, a.ns_shift[p], lane, tps); break;
                case 11: stockham_pass_odd<11>(s, d, tw, a.len, Ns, a.ns_shift[p], lane, tps); break;
                case 13: stockham_pass_odd<13>(s, d, tw, a.len, Ns, a.ns_shift[p], lane, tps); break;
                case 23: stockham_pass_odd<23>(s, d, tw, a.len, Ns, a.ns_shift[p], lane, tps); break;
                default: stockham_pass_any(s, d, tw, a.len, R, Ns, a.ns_shift[p], lane, tps); break;
            }
        }
        __syncthreads();
        Ns *= R;
        float2* t = src;
        src = dst;
        dst = t;
    }
    return src;
}

extern __shared__ __attribute__((aligned(16))) char smem_raw[];
#ifndef SAN_DC_PRE
#define SAN_DC_PRE 1
#endif
constexpr int kDcPre = SAN_DC_PRE;   // dc_rows_kernel: S and r requested before (1) or after (0) the inverse transform

// ------------------------------------------------------------------ row pass
// grid: (ceil(H/B), outer); plane = outer*inner + l for l in [0, inner)
__global__ void __launch_bounds__(kThreads) fft_rows_kernel(const FftArgs a) {
    float2* tw = reinterpret_cast<float2*>(smem_raw);
    float2* bufA = tw + a.len;
    float2* bufB = bufA + a.B * a.pitch;
    const int tid = threadIdx.x;
    const int W = a.W, H = a.H;
    const int h0 = blockIdx.x * a.B;
    const int rows = min(a.B, H - h0);
    const int cnt = rows * W;                 // contiguous floats2 in global for this row group
    const int tps = kThreads >> a.logB;
    const int seq = tid / tps;
    const int lane = tid - seq * tps;

    for (int i = tid; i < a.len; i += kThreads) {
        float2 t = a.tw[i];
        t.y *= a.sgn;
        tw[i] = t;
    }

    const int outer = blockIdx.y;
    // per-thread accumulators for the coil-reducing epilogues: element e = tid + it*kThreads
    // (B*W/256 <= 8*640/256 = 20 per thread at the largest supported row)
    constexpr int kMaxAcc = 24;
    float2 acc[kMaxAcc];
#pragma unroll
    for (int i = 0; i < kMaxAcc; ++i) acc[i] = make_float2(0.f, 0.f);

    for (int l = 0; l < a.inner; ++l) {
        const int plane = outer * a.inner + l;
        const size_t base = ((size_t)plane * H + h0) * W;
        __syncthreads();   // bufA free (previous iteration's epilogue done), tw visible
        // ---- load (pitch == W for the row pass, so LDS index == e)
        if (a.pro == RP_NONE) {
            int wi = tid % W;
            for (int e = tid; e < a.B * W; e += kThreads) {
                float2 v = make_float2(0.f, 0.f);
                if (e < cnt) {
                    v = a.in[base + e];
                    if (a.cm_in) {
                        float m = a.cm_in[wi];
                        v.x *= m;
                        v.y *= m;
                    }
                }
                bufA[e] = v;
                wi += kThreads % W;
                if (wi >= W) wi -= W;
            }
        } else {  // RP_PLANAR_MUL_SENS: plane = n*C + c ; r planar [n,2,H,W]
            const int n = plane / a.C;
            const size_t rbase = ((size_t)n * 2 * H + h0) * W;
            const size_t ibase = rbase + (size_t)H * W;
            for (int e = tid; e < a.B * W; e += kThreads) {
                float2 v = make_float2(0.f, 0.f);
                if (e < cnt) {
                    float2 r = make_float2(a.in_planar[rbase + e], a.in_planar[ibase + e]);
                    v = cmul(r, a.sens[base + e]);
                }
                bufA[e] = v;
            }
        }
        __syncthreads();
        float2* res = run_fft(a, bufA, bufB, tw, seq, lane, tps, seq < rows, a.sgn);
        // ---- epilogue
        if (a.epi == RE_STORE) {
            int wi = tid % W;
            for (int e = tid; e < cnt; e += kThreads) {
                float2 v = res[e];
                float s = a.scale;
                if (a.cm_out) s *= a.cm_out[wi];
                a.out[base + e] = make_float2(v.x * s, v.y * s);
                wi += kThreads % W;
                if (wi >= W) wi -= W;
            }
        } else if (a.epi == RE_STORE_PLANAR) {
            const size_t rbase = ((size_t)plane * a.out_ctot * H + h0) * W;
            const size_t ibase = rbase + (size_t)H * W;
            for (int e = tid; e < cnt; e += kThreads) {
                float2 v = res[e];
                a.out_real[rbase + e] = v.x * a.scale;
                a.out_real[ibase + e] = v.y * a.scale;
            }
        } else if (a.epi == RE_CONJ_SENS_SUM_PLANAR) {
#pragma unroll
            for (int it = 0; it < kMaxAcc; ++it) {
                const int e = tid + it * kThreads;
                if (e < cnt) {
                    float2 s = a.sens[base + e];
                    float2 v = res[e];
                    // v * conj(s)
                    acc[it].x += v.x * s.x + v.y * s.y;
                    acc[it].y += v.y * s.x - v.x * s.y;
                }
            }
        } else {  // RE_RSS
#pragma unroll
            for (int it = 0; it < kMaxAcc; ++it) {
                const int e = tid + it * kThreads;
                if (e < cnt) {
                    float2 v = res[e];
                    acc[it].x += v.x * v.x + v.y * v.y;
                }
            }
        }
    }
    if (a.epi == RE_CONJ_SENS_SUM_PLANAR) {
        const size_t rbase = ((size_t)outer * a.out_ctot * H + h0) * W;
        const size_t ibase = rbase + (size_t)H * W;
#pragma unroll
        for (int it = 0; it < kMaxAcc; ++it) {
            const int e = tid + it * kThreads;
            if (e < cnt) {
                a.out_real[rbase + e] = acc[it].x * a.scale;
                a.out_real[ibase + e] = acc[it].y * a.scale;
            }
        }
    } else if (a.epi == RE_RSS) {
        const size_t obase = ((size_t)outer * H + h0) * W;
#pragma unroll
        for (int it = 0; it < kMaxAcc; ++it) {
            const int e = tid + it * kThreads;
            if (e < cnt) a.out_real[obase + e] = sqrtf(acc[it].x) * a.scale;
        }
    }
}

// --------------------------------------------------------------- column pass
// grid: (ceil(W/B), planes).  Transform length = H.
__global__ void __launch_bounds__(kThreads) fft_cols_kernel(const FftArgs a) {
    float2* tw = reinterpret_cast<float2*>(smem_raw);
    float2* bufA = tw + a.len;
    float2* bufB = bufA + a.B * a.pitch;
    const int tid = threadIdx.x;
    const int W = a.W, H = a.H, B = a.B;
    const int w0 = blockIdx.x * B;
    const int cols = min(B, W - w0);
    const int plane = blockIdx.y;
    const size_t base = (size_t)plane * H * W + w0;
    const int tps = kThreads >> a.logB;
    const int seq = tid / tps;
    const int lane = tid - seq * tps;

    for (int i = tid; i < a.len; i += kThreads) {
        float2 t = a.tw[i];
        t.y *= a.sgn;
        tw[i] = t;
    }
    for (int e = tid; e < H * B; e += kThreads) {
        const int h = e >> a.logB;
        const int j = e & (B - 1);
        float2 v = make_float2(0.f, 0.f);
        if (j < cols) {
            v = a.in[base + (size_t)h * W + j];
            if (a.cm_in) {
                float m = a.cm_in[w0 + j];
                v.x *= m;
                v.y *= m;
            }
        }
        bufA[j * a.pitch + h] = v;
    }
    __syncthreads();
    float2* res = run_fft(a, bufA, bufB, tw, seq, lane, tps, seq < cols, a.sgn);
    if (a.epi == CE_STORE) {
        for (int e = tid; e < H * B; e += kThreads) {
            const int h = e >> a.logB;
            const int j = e & (B - 1);
            if (j < cols) {
                float2 v = res[j * a.pitch + h];
                a.out[base + (size_t)h * W + j] = make_float2(v.x * a.scale, v.y * a.scale);
            }
        }
    } else {  // CE_DC: k_out = (k - dcw*mask*(k-k0)) - R
        const float dcw = a.dcw[0];
        for (int e = tid; e < H * B; e += kThreads) {
            const int h = e >> a.logB;
            const int j = e & (B - 1);
            if (j < cols) {
                const size_t idx = base + (size_t)h * W + j;
                float2 R = res[j * a.pitch + h];
                R.x *= a.scale;
                R.y *= a.scale;
                const float2 kk = a.k[idx];
                const float2 k0 = a.k0[idx];
                const float m = a.mask[w0 + j];
                float2 dc = make_float2((kk.x - k0.x) * m * dcw, (kk.y - k0.y) * m * dcw);
                a.out[idx] = make_float2((kk.x - dc.x) - R.x, (kk.y - dc.y) - R.y);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Fast path for length-320 transforms (the benchmark size): one 64-lane wave per
// workgroup owns 4 lines = 1280 complex values, 20 per lane, held in REGISTERS
// across the four Stockham passes (radix 4, 4, 4, 5).  The first pass reads its
// butterfly inputs straight from global memory and the last pass writes its
// outputs straight to global memory (with the fused prologue / epilogue), so
// only the three inter-pass exchanges go through LDS (10 KiB per wave), and a
// single-wave workgroup needs no cross-wave barrier.  640 workgroups per pass
// at N=8 -> 2.5 independent waves per CU whose load / compute / store phases
// overlap, instead of 1.25 four-wave workgroups in lock step.
//
// MAP 0 (rows):    line = one image row,    lanes run along the row (8-byte coalesced)
// MAP 1 (columns): line = one image column, lanes = 16 rows x 4 adjacent columns
constexpr int kN320 = 320;
constexpr int kL320 = 4;           // lines per wave
constexpr int kDcL320 = 2;         // ... of the cascade kernel (dc_rows320_kernel), see Map320
constexpr int kP320 = 321;         // LDS pitch (float2)

// L: lines per wave.  4 everywhere (5 radix-4 butterflies per lane and pass) except the cascade kernel dc_rows320_kernel, which
// runs 2 (three rounds, the last one half empty: its upper lanes repeat their previous round's butterflies -- same loads, same
// LDS stores): half the serial chain per wave and twice the waves, on a launch that has only 2.5 waves per CU at N = 8.
template <int MAP, int L = 4>
struct Map320 {
    static constexpr int RA = (L * 80 + 63) / 64;      // rounds of the radix-4 passes
    // radix-4 passes: butterfly t of this lane -> (line, j), j in [0, 80)
    __device__ static void r4(int lane, int t, int& line, int& j) {
        if (MAP == 0) {
            int b = lane + 64 * t;
            if (b >= L * 80) b -= 64;                   // (last round: the lane repeats its previous butterfly)
            line = b / 80;
            j = b - line * 80;
        } else {
            line = lane & 3;
            j = (lane >> 2) + 16 * t;
        }
    }
    // radix-5 pass: butterfly t -> (line, j), j in [0, 64)
    __device__ static void r5(int lane, int t, int& line, int& j) {
        if (MAP == 0) {
            line = t;
            j = lane;
        } else {
            line = lane & 3;
            j = (lane >> 2) + 16 * t;
        }
    }
};

__device__ __forceinline__ void bfly4(float2 (&v)[4], float sgn) {
    const float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
    const float2 c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
    const float2 dr = make_float2(sgn * d.y, -sgn * d.x);
    v[0] = cadd(a, c);
    v[1] = cadd(b, dr);
    v[2] = csub(a, c);
    v[3] = csub(b, dr);
}

__device__ __forceinline__ void bfly5(float2 (&v)[5], float sgn) {
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    const float2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const float2 o0 = make_float2(v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y);
    const float2 m1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
    const float2 m2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
    const float2 n1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
    const float2 n2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
    // forward: o1 = m1 - i n1, o4 = m1 + i n1, o2 = m2 - i n2, o3 = m2 + i n2  (-i z = (z.y, -z.x))
    const float2 r1 = make_float2(sgn * n1.y, -sgn * n1.x);
    const float2 r2 = make_float2(sgn * n2.y, -sgn * n2.x);
    v[0] = o0;
    v[1] = cadd(m1, r1);
    v[4] = csub(m1, r1);
    v[2] = cadd(m2, r2);
    v[3] = csub(m2, r2);
}

// The FFT core.  in[t][r] : pass-A inputs of butterfly t (element j + 80 r of its line).
// out[t][r]: final outputs of radix-5 butterfly t (element j + 64 r of its line).
template <int MAP, int L = 4, int RA = (L * 80 + 63) / 64>
__device__ __forceinline__ void fft320_core(float2 (&in)[RA][4], float2 (&out)[L][5], float2* lds,
                                            const float2* twg, float sgn, int lane) {
    static_assert(MAP == 0 || L == 4, "the column map is written for four lines");
    int lineA[RA], jA[RA];
#pragma unroll
    for (int t = 0; t < RA; ++t) Map320<MAP, L>::r4(lane, t, lineA[t], jA[t]);
    // ---- pass A: radix 4, Ns = 1 (no twiddles); out index 4 j + r
#pragma unroll
    for (int t = 0; t < RA; ++t) {
        bfly4(in[t], sgn);
        float2* d = lds + lineA[t] * kP320 + 4 * jA[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = in[t][r];
    }
    __syncthreads();
    // ---- pass B: radix 4, Ns = 4; twiddle W^(r * kk * 20); out index jq*16 + kk + 4 r
#pragma unroll
    for (int t = 0; t < RA; ++t) {
        const float2* s = lds + lineA[t] * kP320 + jA[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) in[t][r] = s[80 * r];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RA; ++t) {
        const int kk = jA[t] & 3, jq = jA[t] >> 2;
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            float2 w = twg[r * kk * 20];
            w.y *= sgn;
            in[t][r] = cmul(in[t][r], w);
        }
        bfly4(in[t], sgn);
        float2* d = lds + lineA[t] * kP320 + jq * 16 + kk;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[4 * r] = in[t][r];
    }
    __syncthreads();
    // ---- pass C: radix 4, Ns = 16; twiddle W^(r * kk * 5); out index jq*64 + kk + 16 r
#pragma unroll
    for (int t = 0; t < RA; ++t) {
        const float2* s = lds + lineA[t] * kP320 + jA[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) in[t][r] = s[80 * r];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RA; ++t) {
        const int kk = jA[t] & 15, jq = jA[t] >> 4;
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            float2 w = twg[r * kk * 5];
            w.y *= sgn;
            in[t][r] = cmul(in[t][r], w);
        }
        bfly4(in[t], sgn);
        float2* d = lds + lineA[t] * kP320 + jq * 64 + kk;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[16 * r] = in[t][r];
    }
    __syncthreads();
    // ---- pass D: radix 5, Ns = 64; twiddle W^(r * j); out index j + 64 r
#pragma unroll
    for (int t = 0; t < L; ++t) {
        int line, j;
        Map320<MAP, L>::r5(lane, t, line, j);
        const float2* s = lds + line * kP320 + j;
#pragma unroll
        for (int r = 0; r < 5; ++r) out[t][r] = s[64 * r];
#pragma unroll
        for (int r = 1; r < 5; ++r) {
            float2 w = twg[r * j];
            w.y *= sgn;
            out[t][r] = cmul(out[t][r], w);
        }
        bfly5(out[t], sgn);
    }
}

// rows: grid (H/4, outer); plane = outer*inner + l.  PRO / EPI are compile-time so the load and
// store sections are straight-line code: every global load of a wave is issued before the first
// one is consumed (rows past H are clamped and zeroed by a select, not skipped by a branch --
// a branchy prologue made the planar-x-S variant 8 us slower than the plain one).
template <int PRO, int EPI>
__global__ void __launch_bounds__(64) fft320_rows_kernel(const FftArgs a) {
    __shared__ float2 lds[kL320 * kP320];
    __shared__ float2 tws[kN320];          // LDS-staged twiddles: passes B, C, D read them with LDS latency
    const int lane = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 5; ++r) tws[lane + 64 * r] = a.tw[lane + 64 * r];
    const int W = kN320, H = a.H;
    const int h0 = blockIdx.x * kL320;
    const int outer = blockIdx.y;
    float2 acc[4][5];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 5; ++r) acc[t][r] = make_float2(0.f, 0.f);

    for (int l = 0; l < a.inner; ++l) {
        const int plane = outer * a.inner + l;
        const size_t pbase = (size_t)plane * H * W;
        float2 in[5][4], out[4][5];
        if (l > 0) __syncthreads();
        // ---- pass-A inputs straight from global: butterfly b = lane + 64 t -> row b/80, j = b%80
        float2 sin[5][4];
        float pre[5][4], pim[5][4];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            int line, j;
            Map320<0>::r4(lane, t, line, j);
            const int row = min(h0 + line, H - 1);
            const int e = row * W + j;                        // in-plane offset (H * W < 2^31)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (PRO == RP_NONE) {
                    in[t][r] = a.in[pbase + e + 80 * r];
                } else {                                      // planar r [n,2,H,W] times S[n,c]
                    const int n = plane / a.C;
                    const float* rp = a.in_planar + (size_t)n * 2 * H * W;
                    pre[t][r] = rp[e + 80 * r];
                    pim[t][r] = rp[e + 80 * r + H * W];
                    sin[t][r] = a.sens[pbase + e + 80 * r];
                }
            }
        }
        // epilogue operands are requested BEFORE the transform so their latency hides under it
        float2 sv[4][5];
        if (EPI == RE_CONJ_SENS_SUM_PLANAR) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 5; ++r) sv[t][r] = a.sens[pbase + (size_t)min(h0 + t, H - 1) * W + lane + 64 * r];
        }
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            int line, j;
            Map320<0>::r4(lane, t, line, j);
            const bool ok = (h0 + line) < H;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (PRO != RP_NONE) in[t][r] = cmul(make_float2(pre[t][r], pim[t][r]), sin[t][r]);
                if (!ok) in[t][r] = make_float2(0.f, 0.f);
            }
        }
        fft320_core<0>(in, out, lds, tws, a.sgn, lane);
        // ---- outputs: row t, element lane + 64 r
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (h0 + t >= H) continue;
            const size_t e = pbase + (size_t)(h0 + t) * W + lane;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const float2 v = out[t][r];
                if (EPI == RE_STORE) {
                    float s = a.scale;
                    if (a.cm_out) s *= a.cm_out[lane + 64 * r];
                    a.out[e + 64 * r] = make_float2(v.x * s, v.y * s);
                } else if (EPI == RE_STORE_PLANAR) {
                    const size_t pe = ((size_t)plane * a.out_ctot * H + h0 + t) * W + lane + 64 * r;
                    a.out_real[pe] = v.x * a.scale;
                    a.out_real[pe + (size_t)H * W] = v.y * a.scale;
                } else if (EPI == RE_CONJ_SENS_SUM_PLANAR) {
                    const float2 s = sv[t][r];
                    acc[t][r].x += v.x * s.x + v.y * s.y;
                    acc[t][r].y += v.y * s.x - v.x * s.y;
                } else {
                    acc[t][r].x += v.x * v.x + v.y * v.y;
                }
            }
        }
    }
    if (EPI == RE_CONJ_SENS_SUM_PLANAR || EPI == RE_RSS) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (h0 + t >= H) continue;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                if (EPI == RE_RSS) {
                    a.out_real[((size_t)outer * H + h0 + t) * W + lane + 64 * r] = sqrtf(acc[t][r].x) * a.scale;
                } else {
                    const size_t pe = ((size_t)outer * a.out_ctot * H + h0 + t) * W + lane + 64 * r;
                    a.out_real[pe] = acc[t][r].x * a.scale;
                    a.out_real[pe + (size_t)H * W] = acc[t][r].y * a.scale;
                }
            }
        }
    }
}

// columns: grid (W/4, planes); transform along H == 320
template <int EPI>
__global__ void __launch_bounds__(64) fft320_cols_kernel(const FftArgs a) {
    __shared__ float2 lds[kL320 * kP320];
    __shared__ float2 tws[kN320];
    const int lane = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 5; ++r) tws[lane + 64 * r] = a.tw[lane + 64 * r];
    const int W = a.W;
    // XCD-aware column order: workgroups are dealt round-robin to the 8 XCDs (id % 8), and four
    // adjacent column quads share every 128-byte line of a row; give XCD x the contiguous quads
    // [x*nq/8, (x+1)*nq/8) so each line is filled once into ONE L2 instead of into four.
    const int nq = gridDim.x;
    int cq = blockIdx.x;
    if ((nq & 7) == 0) cq = (cq & 7) * (nq >> 3) + (cq >> 3);
    const int w0 = cq * kL320;
    const int plane = blockIdx.y;
    const int c = lane & 3;
    const bool ok = (w0 + c) < W;
    const size_t base = (size_t)plane * kN320 * W + min(w0 + c, W - 1);      // clamped column: loads are unconditional
    float2 in[5][4], out[4][5];
    float cm = ok ? 1.f : 0.f;
    if (a.cm_in) cm *= a.cm_in[min(w0 + c, W - 1)];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        int line, j;
        Map320<1>::r4(lane, t, line, j);
#pragma unroll
        for (int r = 0; r < 4; ++r) in[t][r] = a.in[base + (size_t)(j + 80 * r) * W];
    }
    // soft-DC operands are requested BEFORE the transform so their latency hides under it
    float2 kv[4][5], k0v[4][5];
    float dcw = 0.f, m = 0.f;
    if (EPI != CE_STORE) {
        dcw = a.dcw[0];
        m = a.mask[min(w0 + c, W - 1)];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int line, j;
            Map320<1>::r5(lane, t, line, j);
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const size_t idx = base + (size_t)(j + 64 * r) * W;
                kv[t][r] = a.k[idx];
                k0v[t][r] = a.k0[idx];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) in[t][r] = make_float2(in[t][r].x * cm, in[t][r].y * cm);
    fft320_core<1>(in, out, lds, tws, a.sgn, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int line, j;
        Map320<1>::r5(lane, t, line, j);
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const size_t idx = base + (size_t)(j + 64 * r) * W;
            float2 R = out[t][r];
            R.x *= a.scale;
            R.y *= a.scale;
            if (EPI == CE_STORE) {
                if (ok) a.out[idx] = R;
            } else {
                const float2 kk = kv[t][r];
                const float2 k0 = k0v[t][r];
                const float2 dc = make_float2((kk.x - k0.x) * m * dcw, (kk.y - k0.y) * m * dcw);
                R = make_float2((kk.x - dc.x) - R.x, (kk.y - dc.y) - R.y);
                if (ok) a.out[idx] = R;
                out[t][r] = ok ? R : make_float2(0.f, 0.f);
            }
        }
    }
    if (EPI == CE_DC_NEXT) {
        // The next cascade starts with the inverse column transform of exactly these columns
        // (sens_reduce, varnet.py:511): run it here while k' is still in registers instead of
        // re-reading it from HBM in a separate launch.  One LDS exchange converts the radix-5
        // output layout (element j + 64 r) into the pass-A input layout (element j + 80 r).
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int line, j;
            Map320<1>::r5(lane, t, line, j);
#pragma unroll
            for (int r = 0; r < 5; ++r) lds[line * kP320 + j + 64 * r] = out[t][r];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            int line, j;
            Map320<1>::r4(lane, t, line, j);
#pragma unroll
            for (int r = 0; r < 4; ++r) in[t][r] = lds[line * kP320 + j + 80 * r];
        }
        __syncthreads();
        fft320_core<1>(in, out, lds, tws, -a.sgn, lane);
        if (!ok) return;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int line, j;
            Map320<1>::r5(lane, t, line, j);
#pragma unroll
            for (int r = 0; r < 5; ++r) a.out2[base + (size_t)(j + 64 * r) * W] = out[t][r];
        }
    }
}


// ---------------------------------------------------------------------------
// Image-domain cascade boundary.  The column mask M of the soft data consistency depends on kx only, so with
// x = ifft2(k) the k-space update  k' = k - w M (k - k0) - fft2(r S)  (varnet.py:514-530) is ROW-LOCAL in the image domain:
//     D(x) = ifft_x( M (fft_x(x) - k0x) ),  k0x = ifft_y(k0)        x' = x - w D(x) - r S        m' = sum_c conj(S_c) x'_c
// (the y transforms cancel: F_y^H (F_y X F_x diag(M)) F_x^H = X F_x diag(M) F_x^H).  One launch per cascade does the
// forward row transform, the masked combine, the inverse row transform, the regulariser term and the coil combination
// for the NEXT cascade, touching HBM once: read x, S, k0x (C planes each) and r (1), write x' (C) and m' (1) =
// (4C + 2) planes instead of the (6C + 2) of the two 2-D transforms, and no column passes at all.
// MODE 0 (forward): as above; optionally stores dk = M (fft_x(x) - k0x) for the dc_weight gradient.
// MODE 1 (backward: the DC term is self-adjoint): g' = g - w ifft_x(M fft_x(g)), h = m_scale sum_c conj(S_c) g_c (the
//        INPUT g: gradient wrt the regulariser output), and one partial of Re sum conj(fft_x(g)) dk per workgroup.
template <int MODE, int L>
__global__ void __launch_bounds__(64) dc_rows320_kernel(const FftArgs a) {
    __shared__ float2 lds[L * kP320];
    __shared__ float2 tws[kN320];
    const int lane = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 5; ++r) tws[lane + 64 * r] = a.tw[lane + 64 * r];
    const int W = kN320, H = a.H;
    constexpr int RA = Map320<0, L>::RA;
    const int h0 = blockIdx.x * L;
    const int n = blockIdx.y;
    const float dcw = a.dcw[0];
    float mk[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) mk[r] = a.mask[lane + 64 * r];
    float2 macc[L][5];
#pragma unroll
    for (int t = 0; t < L; ++t)
#pragma unroll
        for (int r = 0; r < 5; ++r) macc[t][r] = make_float2(0.f, 0.f);
    float wsum = 0.f;
    int lineA[RA], jA[RA];
#pragma unroll
    for (int t = 0; t < RA; ++t) Map320<0, L>::r4(lane, t, lineA[t], jA[t]);

    for (int c = 0; c < a.C; ++c) {
        const size_t pbase = (size_t)(n * a.C + c) * H * W;
        float2 in[RA][4], out[L][5];
        if (c > 0) __syncthreads();
#pragma unroll
        for (int t = 0; t < RA; ++t) {
            const int e = min(h0 + lineA[t], H - 1) * W + jA[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) in[t][r] = a.in[pbase + e + 80 * r];
        }
        // the k_x-domain operand is requested before the transform so that its latency hides under it
        float2 kq[L][5];
#pragma unroll
        for (int t = 0; t < L; ++t) {
            const size_t e = pbase + (size_t)min(h0 + t, H - 1) * W + lane;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                if (MODE == 0) kq[t][r] = a.k0 ? a.k0[e + 64 * r] : make_float2(0.f, 0.f);
                else kq[t][r] = a.dk_in ? a.dk_in[e + 64 * r] : make_float2(0.f, 0.f);
            }
        }
        fft320_core<0, L>(in, out, lds, tws, 1.f, lane);
#pragma unroll
        for (int t = 0; t < L; ++t) {
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const float2 X = make_float2(out[t][r].x * a.scale, out[t][r].y * a.scale);
                float2 d;
                if (MODE == 0) {
                    d = make_float2(mk[r] * (X.x - kq[t][r].x), mk[r] * (X.y - kq[t][r].y));
                    if (a.dk_out && h0 + t < H) a.dk_out[pbase + (size_t)(h0 + t) * W + lane + 64 * r] = d;
                } else {
                    d = make_float2(mk[r] * X.x, mk[r] * X.y);
                    if (h0 + t < H) wsum += X.x * kq[t][r].x + X.y * kq[t][r].y;
                }
                out[t][r] = d;
            }
        }
        // one LDS exchange converts the radix-5 output layout (element lane + 64 r) into the pass-A input layout
        __syncthreads();
#pragma unroll
        for (int t = 0; t < L; ++t)
#pragma unroll
            for (int r = 0; r < 5; ++r) lds[t * kP320 + lane + 64 * r] = out[t][r];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RA; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) in[t][r] = lds[lineA[t] * kP320 + jA[t] + 80 * r];
        __syncthreads();
        // epilogue operands, requested before the inverse transform
        float2 xo[L][5], sv[L][5];
        float rre[L][5], rim[L][5];
#pragma unroll
        for (int t = 0; t < L; ++t) {
            const int row = min(h0 + t, H - 1);
            const size_t e = pbase + (size_t)row * W + lane;
            const size_t pe = ((size_t)n * 2 * H + row) * W + lane;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                xo[t][r] = a.in[e + 64 * r];
                sv[t][r] = a.sens[e + 64 * r];
                if (MODE == 0 && a.in_planar) {
                    rre[t][r] = a.in_planar[pe + 64 * r];
                    rim[t][r] = a.in_planar[pe + 64 * r + (size_t)H * W];
                } else {
                    rre[t][r] = rim[t][r] = 0.f;
                }
            }
        }
        fft320_core<0, L>(in, out, lds, tws, -1.f, lane);
#pragma unroll
        for (int t = 0; t < L; ++t) {
            if (h0 + t >= H) continue;
            const size_t e = pbase + (size_t)(h0 + t) * W + lane;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const float2 s = sv[t][r];
                float2 v = make_float2(xo[t][r].x - dcw * (out[t][r].x * a.scale), xo[t][r].y - dcw * (out[t][r].y * a.scale));
                if (MODE == 0) {
                    v.x -= rre[t][r] * s.x - rim[t][r] * s.y;
                    v.y -= rre[t][r] * s.y + rim[t][r] * s.x;
                }
                if (a.out) a.out[e + 64 * r] = v;
                const float2 q = MODE == 0 ? v : xo[t][r];          // coil combination: conj(S) * q
                macc[t][r].x += q.x * s.x + q.y * s.y;
                macc[t][r].y += q.y * s.x - q.x * s.y;
            }
        }
    }
    if (a.out_real) {
#pragma unroll
        for (int t = 0; t < L; ++t) {
            if (h0 + t >= H) continue;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const size_t pe = ((size_t)n * a.out_ctot * H + h0 + t) * W + lane + 64 * r;
                a.out_real[pe] = macc[t][r].x * a.m_scale;
                a.out_real[pe + (size_t)H * W] = macc[t][r].y * a.m_scale;
            }
        }
    }
    if (MODE == 0 && a.m_stats) {
        // statistics of the rows this wave just produced (the values are still in registers): one (count, mean, M2) record per
        // workgroup and plane, merged by san_norm_finalize like san_plane_stats' chunk records -- that launch is not needed.
        // Pilot-shifted single pass (the convolutions' scheme): the four wave sums are independent chains
        int rows = 0;
#pragma unroll
        for (int t = 0; t < L; ++t) rows += h0 + t < H ? 1 : 0;
        const float cnt = (float)(rows * W);
        const float p0 = __shfl(macc[0][0].x * a.m_scale, 0, 64), p1 = __shfl(macc[0][0].y * a.m_scale, 0, 64);     // row h0 always exists
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < L; ++t)
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const bool ok = h0 + t < H;
                const float e0 = ok ? macc[t][r].x * a.m_scale - p0 : 0.f, e1 = ok ? macc[t][r].y * a.m_scale - p1 : 0.f;
                s1[0] += e0;
                s2[0] = fmaf(e0, e0, s2[0]);
                s1[1] += e1;
                s2[1] = fmaf(e1, e1, s2[1]);
            }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            s1[p] = san_wave_total(s1[p]);
            s2[p] = san_wave_total(s2[p]);
        }
        if (lane == 0) {
            const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float* o = a.m_stats + ((size_t)(n * 2 + p) * gridDim.x + blockIdx.x) * 3;
                o[0] = cnt;
                o[1] = (p ? p1 : p0) + s1[p] * inv;
                o[2] = fmaxf(s2[p] - s1[p] * s1[p] * inv, 0.f);
            }
        }
    }
    if (MODE == 1 && a.dcw_part) {
        wsum = san_wave_total(wsum);
        if (lane == 0) a.dcw_part[blockIdx.y * gridDim.x + blockIdx.x] = wsum;
    }
}

// ---------------------------------------------------------------- 368-wide rows (the multi-coil 640 x 368 planes)
// 368 = 16 x 23, as TWO register passes per transform and no workgroup barrier: one wave owns 4 rows of one coil.
//   n = n1 + 16 n2, k = 23 k1 + k2:   W368^(nk) = W16^(n1 k1) W368^(n1 k2) W23^(n2 k2)
//   "23-side" task (row t, n1): lane = 16 t + n1 holds the 23 image samples n1 + 16 n2 (128-byte aligned runs of 16 lanes)
//   "16-side" task (row t, k2): 92 tasks in two rounds hold the 16 frequencies 23 k1 + k2
// forward: 23-point DFTs over n2 -> LDS -> twiddle, 16-point DFTs over n1 -> mask / k0 in registers -> inverse 16-point
// -> conj twiddle -> LDS (the same slots) -> inverse 23-point DFTs -> epilogue on x, S, r held in registers since the
// start of the kernel / of the inverse transform.  One LDS exchange per transform (the general kernel below: three
// Stockham passes each, eight workgroup barriers, 2-3 workgroups per CU; measured 72 us against this kernel's
// numbers in DESIGN.md section 3.5).
constexpr int kN368 = 368;
constexpr int kL368 = 4;            // rows per wave
constexpr int kP368 = 400;          // LDS row pitch (float2): slot(t, n1, k2) = t * 400 + k2 * 17 + n1

template <int INV>
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = csub(a1, a3);
    const float2 r = INV ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);   // -i d13 forward, +i d13 inverse
    a0 = cadd(s02, s13);
    a1 = cadd(d02, r);
    a2 = csub(s02, s13);
    a3 = csub(d02, r);
}

// v <- v * W16^m (forward) or its conjugate (INV); m a compile-time constant
template <int INV, int M>
__device__ __forceinline__ float2 tw16(float2 v) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    constexpr int m = M & 15;
    if constexpr (m == 0) {
        return v;
    } else if constexpr (m == 4) {
        return INV ? make_float2(-v.y, v.x) : make_float2(v.y, -v.x);
    } else {
        static_assert(m == 1 || m == 2 || m == 3 || m == 6 || m == 9, "twiddle not tabulated");
        constexpr float co = m == 1 ? c1 : m == 2 ? h : m == 3 ? s1 : m == 6 ? -h : -c1;
        constexpr float si = m == 1 ? s1 : m == 2 ? h : m == 3 ? c1 : m == 6 ? h : -s1;
        return cmul(v, make_float2(co, INV ? si : -si));
    }
}

// 16-point DFT in registers: n1 = 4 a + b, k1 = c + 4 d:  W16^(n1 k1) = W4^(a c) W16^(b c) W4^(b d).  o[k1] in natural order.
template <int INV>
__device__ __forceinline__ void dft16(float2 (&v)[16], float2 (&o)[16]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4<INV>(v[b], v[4 + b], v[8 + b], v[12 + b]);          // over a; result index c at 4 c + b
    v[5] = tw16<INV, 1>(v[5]);
    v[6] = tw16<INV, 2>(v[6]);
    v[7] = tw16<INV, 3>(v[7]);
    v[9] = tw16<INV, 2>(v[9]);
    v[10] = tw16<INV, 4>(v[10]);
    v[11] = tw16<INV, 6>(v[11]);
    v[13] = tw16<INV, 3>(v[13]);
    v[14] = tw16<INV, 6>(v[14]);
    v[15] = tw16<INV, 9>(v[15]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dft4<INV>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);                 // over b; result index d at 4 c + d
#pragma unroll
        for (int d = 0; d < 4; ++d) o[c + 4 * d] = v[4 * c + d];
    }
}

// 23-point DFT of u[0..22] (registers), outputs written to dst[q * 17] (this lane's LDS slots).  Conjugate-symmetric
// form: a_r = u_r + u_(23-r), b_r = u_r - u_(23-r); out_q = u_0 + sum a_r cos - i sum b_r sin (forward; + i inverse),
// out_(23-q) its mirror.  Fully unrolled: every coefficient is an instruction constant (as a rolled loop over q the 22
// coefficients of an iteration were dependent scalar loads, one s_waitcnt per multiply-add: 9 of the wave's 27 us); the
// scheduling barrier keeps the eleven output pairs from being interleaved (all sums in flight need > 256 registers).
template <int INV>
__device__ __forceinline__ void dft23_to_lds(const float2 (&u)[23], float2* __restrict__ dst) {
    constexpr float C[23] = {
        1.f, 0.962917268f, 0.85441941f, 0.682553172f, 0.460065037f, 0.203456014f, -0.0682424158f, -0.334879607f, -0.576680303f,
        -0.775711298f, -0.917211294f, -0.99068594f, -0.99068594f, -0.917211294f, -0.775711298f, -0.576680303f, -0.334879607f,
        -0.0682424158f, 0.203456014f, 0.460065037f, 0.682553172f, 0.85441941f, 0.962917268f};
    constexpr float S[23] = {
        0.f, 0.269796759f, 0.519583941f, 0.730835974f, 0.887885213f, 0.979084074f, 0.997668743f, 0.942260921f, 0.816969872f,
        0.631087959f, 0.398401082f, 0.136166647f, -0.136166647f, -0.398401082f, -0.631087959f, -0.816969872f, -0.942260921f,
        -0.997668743f, -0.979084074f, -0.887885213f, -0.730835974f, -0.519583941f, -0.269796759f};
    float2 a[11], b[11];
    float2 s0 = u[0];
#pragma unroll
    for (int r = 0; r < 11; ++r) {
        a[r] = cadd(u[r + 1], u[22 - r]);
        b[r] = csub(u[r + 1], u[22 - r]);
        s0 = cadd(s0, a[r]);
    }
    dst[0] = s0;
#pragma unroll
    for (int q = 1; q <= 11; ++q) {
        float2 P = u[0], Q = make_float2(0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 11; ++r) {
            const float co = C[(q * (r + 1)) % 23], si = S[(q * (r + 1)) % 23];
            P.x = fmaf(a[r].x, co, P.x);
            P.y = fmaf(a[r].y, co, P.y);
            Q.x = fmaf(b[r].x, si, Q.x);
            Q.y = fmaf(b[r].y, si, Q.y);
        }
        // forward: P - iQ at q, P + iQ at 23 - q;  i Q = (-Q.y, Q.x)
        const float2 lo = make_float2(P.x + Q.y, P.y - Q.x), hi = make_float2(P.x - Q.y, P.y + Q.x);
        dst[q * 17] = INV ? hi : lo;
        dst[(23 - q) * 17] = INV ? lo : hi;
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int MODE>
__global__ void __launch_bounds__(64) dc_rows368_kernel(const FftArgs a) {
    __shared__ float2 lds[kL368 * kP368];
    __shared__ float2 tws[kN368];
    __shared__ float msk[kN368];
    const int lane = threadIdx.x;
    const int W = kN368, H = a.H;
    const int h0 = blockIdx.x * kL368;
    const int n = blockIdx.y, c = blockIdx.z;
    const size_t pbase = (size_t)(n * a.C + c) * H * W;
    // 23-side task of this lane
    const int tA = lane >> 4, n1 = lane & 15;
    const int rowA = min(h0 + tA, H - 1);
    const bool liveA = h0 + tA < H;
    const size_t eA = pbase + (size_t)rowA * W + n1;
    float2* slotA = lds + tA * kP368 + n1;                 // + k2 * 17
    // 16-side tasks (two rounds)
    int tB[2], k2B[2];
    bool actB[2], liveB[2];
    size_t eB[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int id = it * 64 + lane;
        actB[it] = id < kL368 * 23;
        tB[it] = actB[it] ? id / 23 : 0;
        k2B[it] = actB[it] ? id - 23 * tB[it] : 0;
        liveB[it] = actB[it] && h0 + tB[it] < H;
        eB[it] = pbase + (size_t)min(h0 + tB[it], H - 1) * W + k2B[it];
    }
    // operands: x now (kept for the epilogue), k0 / dk_in for the mask stage
    float2 xo[23];
#pragma unroll
    for (int q = 0; q < 23; ++q) xo[q] = a.in[eA + 16 * q];
    float2 kq[2][16];
    {
        const float2* kp = MODE == 0 ? a.k0 : a.dk_in;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) kq[it][k1] = (kp && actB[it]) ? kp[eB[it] + 23 * k1] : make_float2(0.f, 0.f);
    }
    for (int i = lane; i < kN368; i += 64) {
        tws[i] = a.tw[i];
        msk[i] = a.mask[i];
    }
    const float dcw = a.dcw[0];
    dft23_to_lds<0>(xo, slotA);
    __syncthreads();
    float wsum = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if (!actB[it]) continue;
        float2* slotB = lds + tB[it] * kP368 + k2B[it] * 17;      // + n1
        float2 v[16], X[16], tw[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            tw[j] = tws[j * k2B[it]];
            v[j] = cmul(slotB[j], tw[j]);
        }
        dft16<0>(v, X);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            const float m = msk[23 * k1 + k2B[it]];
            const float2 Xs = make_float2(X[k1].x * a.scale, X[k1].y * a.scale);
            if (MODE == 0) {
                v[k1] = make_float2(m * (Xs.x - kq[it][k1].x), m * (Xs.y - kq[it][k1].y));
                if (a.dk_out && liveB[it]) a.dk_out[eB[it] + 23 * k1] = v[k1];
            } else {
                v[k1] = make_float2(m * Xs.x, m * Xs.y);
                if (liveB[it]) wsum += Xs.x * kq[it][k1].x + Xs.y * kq[it][k1].y;
            }
        }
        dft16<1>(v, X);
#pragma unroll
        for (int j = 0; j < 16; ++j) slotB[j] = cmul(X[j], make_float2(tw[j].x, -tw[j].y));
    }
    __syncthreads();
    // epilogue operands, requested before the inverse 23-point transforms
    float2 sv[23];
    float rre[23], rim[23];
    {
        const size_t pe = ((size_t)n * 2 * H + rowA) * W + n1;
#pragma unroll
        for (int q = 0; q < 23; ++q) {
            sv[q] = a.sens[eA + 16 * q];
            if (MODE == 0 && a.in_planar) {
                rre[q] = a.in_planar[pe + 16 * q];
                rim[q] = a.in_planar[pe + 16 * q + (size_t)H * W];
            } else {
                rre[q] = rim[q] = 0.f;
            }
        }
    }
    {
        float2 u[23];
#pragma unroll
        for (int q = 0; q < 23; ++q) u[q] = slotA[q * 17];
        dft23_to_lds<1>(u, slotA);
    }
    // (the 23 results are this lane's own slots: no barrier)
    const bool comb = a.out_real && a.C == 1;
    const size_t pr = ((size_t)n * a.out_ctot * H + rowA) * W + n1;
    if (liveA) {
#pragma unroll
        for (int q = 0; q < 23; ++q) {
            const float2 y = slotA[q * 17];
            const float2 s = sv[q];
            float2 v = make_float2(xo[q].x - dcw * (y.x * a.scale), xo[q].y - dcw * (y.y * a.scale));
            if (MODE == 0) {
                v.x -= rre[q] * s.x - rim[q] * s.y;
                v.y -= rre[q] * s.y + rim[q] * s.x;
            }
            if (a.out) a.out[eA + 16 * q] = v;
            if (comb) {
                const float2 qv = MODE == 0 ? v : xo[q];
                a.out_real[pr + 16 * q] = (qv.x * s.x + qv.y * s.y) * a.m_scale;
                a.out_real[pr + 16 * q + (size_t)H * W] = (qv.y * s.x - qv.x * s.y) * a.m_scale;
            }
        }
    }
    if (MODE == 1 && a.dcw_part) {
        wsum = san_wave_total(wsum);
        if (lane == 0) a.dcw_part[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = wsum;
    }
}

// Any row length: B rows of ONE coil per workgroup (grid: row blocks, samples, coils) staged in LDS, both transforms with
// the mixed-radix Stockham passes.  Every element e of the B x W tile belongs to thread e % kThreads in all phases, so the
// operands travel in registers: x (loaded once, used again by the epilogue), k0 / dk_in requested before the forward
// transform, S and r before (PRE) or in one batch after the inverse one -- never one dependent HBM round trip per element
// (measured at 15 x 640 x 368: the per-element load / use loops were ~35 serial round trips per workgroup, 42 us of the
// 72; the transforms 28 us).  Prime radices above 5 run with the workgroup's butterflies spread over ALL threads
// (16 butterflies per 368-row leave half of every wave idle in the per-row mapping).
// With several coils coil_combine_kernel follows; a single-coil launch writes the coil combination itself.
template <int MODE, int EPT, int PRE>
__global__ void __launch_bounds__(kThreads, 2) dc_rows_kernel(const FftArgs a) {
    float2* twf = reinterpret_cast<float2*>(smem_raw);
    float2* twi = twf + a.len;
    float2* bufA = twi + a.len;
    float2* bufB = bufA + a.B * a.pitch;
    float* msk = reinterpret_cast<float*>(bufB + a.B * a.pitch);
    const int tid = threadIdx.x;
    const int W = a.W, H = a.H;
    const int h0 = blockIdx.x * a.B;
    const int rows = min(a.B, H - h0);
    const int cnt = rows * W;
    const int tps = kThreads >> a.logB;
    const int seq = tid / tps;
    const int lane = tid - seq * tps;
    const int n = blockIdx.y, c = blockIdx.z;
    const size_t base = ((size_t)(n * a.C + c) * H + h0) * W;
    float2 xo[EPT], kq[EPT];
    {
        const float2* kp = MODE == 0 ? a.k0 : a.dk_in;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + k * kThreads;
            xo[k] = e < cnt ? a.in[base + e] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + k * kThreads;
            kq[k] = (kp && e < cnt) ? kp[base + e] : make_float2(0.f, 0.f);
        }
    }
    for (int i = tid; i < a.len; i += kThreads) {
        const float2 t = a.tw[i];
        twf[i] = t;
        twi[i] = make_float2(t.x, -t.y);
    }
    for (int i = tid; i < W; i += kThreads) msk[i] = a.mask[i];
    const float dcw = a.dcw[0];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * kThreads;
        if (e < cnt) bufA[e] = xo[k];
    }
    __syncthreads();
    float2* res = run_fft(a, bufA, bufB, twf, seq, lane, tps, seq < rows, 1.f, rows);
    float wsum = 0.f;
    {
        int wi = tid % W;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + k * kThreads;
            if (e < cnt) {
                const float2 X = make_float2(res[e].x * a.scale, res[e].y * a.scale);
                const float m = msk[wi];
                float2 d;
                if (MODE == 0) {
                    d = make_float2(m * (X.x - kq[k].x), m * (X.y - kq[k].y));
                    if (a.dk_out) a.dk_out[base + e] = d;
                } else {
                    d = make_float2(m * X.x, m * X.y);
                    wsum += X.x * kq[k].x + X.y * kq[k].y;
                }
                res[e] = d;
            }
            wi += kThreads % W;
            if (wi >= W) wi -= W;
        }
    }
    __syncthreads();
    const size_t rbase = ((size_t)n * 2 * H + h0) * W;
    const bool with_r = MODE == 0 && a.in_planar;
    float2 sv[EPT];
    float rre[EPT], rim[EPT];
    auto load_epi = [&]() {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + k * kThreads;
            sv[k] = e < cnt ? a.sens[base + e] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + k * kThreads;
            rre[k] = (with_r && e < cnt) ? a.in_planar[rbase + e] : 0.f;
            rim[k] = (with_r && e < cnt) ? a.in_planar[rbase + e + (size_t)H * W] : 0.f;
        }
    };
    if (PRE) load_epi();
    float2* other = res == bufA ? bufB : bufA;
    float2* res2 = run_fft(a, res, other, twi, seq, lane, tps, seq < rows, -1.f, rows);
    if (!PRE) load_epi();
    const bool comb = a.out_real && a.C == 1;
    const size_t rb = ((size_t)n * a.out_ctot * H + h0) * W;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * kThreads;
        if (e < cnt) {
            const float2 s = sv[k];
            float2 v = make_float2(xo[k].x - dcw * (res2[e].x * a.scale), xo[k].y - dcw * (res2[e].y * a.scale));
            if (MODE == 0) {
                v.x -= rre[k] * s.x - rim[k] * s.y;
                v.y -= rre[k] * s.y + rim[k] * s.x;
            }
            if (a.out) a.out[base + e] = v;
            if (comb) {
                const float2 q = MODE == 0 ? v : xo[k];
                a.out_real[rb + e] = (q.x * s.x + q.y * s.y) * a.m_scale;
                a.out_real[rb + e + (size_t)H * W] = (q.y * s.x - q.x * s.y) * a.m_scale;
            }
        }
    }
    if (MODE == 1 && a.dcw_part) {
        wsum = san_wave_total(wsum);
        __syncthreads();                                   // the transforms are done: the twiddle area is free
        float* red = reinterpret_cast<float*>(twf);         // (one __shared__ object only: the dynamic one may then be the full 160 KB)
        if ((tid & 63) == 0) red[tid >> 6] = wsum;
        __syncthreads();
        if (tid == 0) a.dcw_part[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// m[n] = m_scale * sum_c conj(S[n,c]) q[n,c]  (planar into channels 0, 1 of [n, ctot, hw]); one pass over 2C planes
__global__ void __launch_bounds__(kThreads) coil_combine_kernel(const float2* __restrict__ q, const float2* __restrict__ S,
                                                                 float* __restrict__ out, int out_ctot, float m_scale, int C, int HW) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < HW; i += gridDim.x * kThreads) {
        float ar = 0.f, ai = 0.f;
        for (int c = 0; c < C; ++c) {
            const size_t e = ((size_t)n * C + c) * HW + i;
            const float2 v = q[e], s = S[e];
            ar += v.x * s.x + v.y * s.y;
            ai += v.y * s.x - v.x * s.y;
        }
        out[(size_t)n * out_ctot * HW + i] = ar * m_scale;
        out[(size_t)n * out_ctot * HW + HW + i] = ai * m_scale;
    }
}

// --------------------------------------------------------------- host side
struct Plan {
    int n = 0;
    int nrad = 0;
    int rad[kMaxRad];
    int ns_shift[kMaxRad];
    float2* tw = nullptr;
};

std::mutex g_mu;
std::map<std::pair<int, int>, Plan> g_plans;  // (device, n)

bool factorize(int n, Plan& p) {
    p.n = n;
    p.nrad = 0;
    int m = n;
    auto push = [&](int r) {
        if (p.nrad >= kMaxRad) return false;
        p.rad[p.nrad++] = r;
        return true;
    };
    while (m % 4 == 0) {
        if (!push(4)) return false;
        m /= 4;
    }
    while (m % 2 == 0) {
        if (!push(2)) return false;
        m /= 2;
    }
    for (int f = 3; f <= 61 && m > 1; f += 2) {
        while (m % f == 0) {
            if (!push(f)) return false;
            m /= f;
        }
    }
    if (m != 1) return false;   // prime factor > 61: unsupported
    int Ns = 1;
    for (int i = 0; i < p.nrad; ++i) {
        int s = -1;
        if ((Ns & (Ns - 1)) == 0) {
            s = 0;
            while ((1 << s) < Ns) ++s;
        }
        p.ns_shift[i] = s;
        Ns *= p.rad[i];
    }
    return true;
}

int get_plan(int n, Plan* out) {
    if (n < 1 || n > 4096) {
        san_set_error("fft length %d outside [1, 4096]", n);
        return SAN_E_UNSUPPORTED;
    }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_pair(dev, n);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) {
        *out = it->second;
        return SAN_OK;
    }
    Plan p;
    if (!factorize(n, p)) {
        san_set_error("fft length %d has a prime factor > 61", n);
        return SAN_E_UNSUPPORTED;
    }
    std::vector<float2> h(n);
    for (int k = 0; k < n; ++k) {
        // exact-ish angles: reduce k/n to an octant in integer arithmetic first
        long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
        h[k] = make_float2((float)cosl(ang), (float)sinl(ang));
    }
    e = hipMalloc(&p.tw, sizeof(float2) * n);
    if (e != hipSuccess) {
        san_set_error("hipMalloc twiddles: %s", hipGetErrorString(e));
        return (int)e;
    }
    e = hipMemcpy(p.tw, h.data(), sizeof(float2) * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        san_set_error("hipMemcpy twiddles: %s", hipGetErrorString(e));
        return (int)e;
    }
    g_plans[key] = p;
    *out = p;
    return SAN_OK;
}

void fill_pass(FftArgs& a, const Plan& p) {
    a.len = p.n;
    a.nrad = p.nrad;
    for (int i = 0; i < p.nrad; ++i) {
        a.rad[i] = p.rad[i];
        a.ns_shift[i] = p.ns_shift[i];
    }
    a.tw = p.tw;
}

int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

// lines per workgroup: as many as keep LDS <= ~64 KiB and give every sequence >= 8 threads
int pick_batch(int len, int lines, int max_b) {
    int b = max_b;
    while (b > 1 && (size_t)b * (len + 1) * sizeof(float2) * 2 > 60 * 1024) b >>= 1;
    (void)lines;
    return b;
}

size_t lds_bytes(const FftArgs& a) { return sizeof(float2) * ((size_t)a.len + 2 * (size_t)a.B * a.pitch); }

int launch_rows(FftArgs& a, int outer, hipStream_t s) {
    if (a.len == kN320 && a.W == kN320) {
        const dim3 grid(san_cdiv(a.H, kL320), outer);
#define SAN_ROWS320(P, E) hipLaunchKernelGGL((fft320_rows_kernel<P, E>), grid, dim3(64), 0, s, a)
        if (a.pro == RP_NONE) {
            switch (a.epi) {
                case RE_STORE: SAN_ROWS320(RP_NONE, RE_STORE); break;
                case RE_STORE_PLANAR: SAN_ROWS320(RP_NONE, RE_STORE_PLANAR); break;
                case RE_CONJ_SENS_SUM_PLANAR: SAN_ROWS320(RP_NONE, RE_CONJ_SENS_SUM_PLANAR); break;
                default: SAN_ROWS320(RP_NONE, RE_RSS); break;
            }
        } else {
            switch (a.epi) {
                case RE_STORE: SAN_ROWS320(RP_PLANAR_MUL_SENS, RE_STORE); break;
                case RE_STORE_PLANAR: SAN_ROWS320(RP_PLANAR_MUL_SENS, RE_STORE_PLANAR); break;
                case RE_CONJ_SENS_SUM_PLANAR: SAN_ROWS320(RP_PLANAR_MUL_SENS, RE_CONJ_SENS_SUM_PLANAR); break;
                default: SAN_ROWS320(RP_PLANAR_MUL_SENS, RE_RSS); break;
            }
        }
#undef SAN_ROWS320
        SAN_LAUNCH_CHECK();
        return SAN_OK;
    }
    a.B = pick_batch(a.len, a.H, 8);
    // the coil-reducing epilogues keep B*W/256 accumulators per thread (<= 24)
    while (a.B > 1 && a.B * a.W > 24 * kThreads) a.B >>= 1;
    if (a.B * a.W > 24 * kThreads && (a.epi == RE_CONJ_SENS_SUM_PLANAR || a.epi == RE_RSS)) {
        san_set_error("row length %d too long for the reducing epilogue", a.W);
        return SAN_E_UNSUPPORTED;
    }
    a.logB = ilog2(a.B);
    a.pitch = a.len;
    dim3 grid(san_cdiv(a.H, a.B), outer);
    size_t lds = lds_bytes(a);
    if (lds > 160 * 1024) {
        san_set_error("fft row of %d does not fit LDS", a.len);
        return SAN_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(fft_rows_kernel, grid, dim3(kThreads), lds, s, a);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int launch_cols(FftArgs& a, int planes, hipStream_t s) {
    if (a.len == kN320 && a.H == kN320) {
        const dim3 grid(san_cdiv(a.W, kL320), planes);
        if (a.epi == CE_DC_NEXT)
            hipLaunchKernelGGL((fft320_cols_kernel<CE_DC_NEXT>), grid, dim3(64), 0, s, a);
        else if (a.epi == CE_DC)
            hipLaunchKernelGGL((fft320_cols_kernel<CE_DC>), grid, dim3(64), 0, s, a);
        else
            hipLaunchKernelGGL((fft320_cols_kernel<CE_STORE>), grid, dim3(64), 0, s, a);
        SAN_LAUNCH_CHECK();
        return SAN_OK;
    }
    int maxb = 16;
    if ((long)san_cdiv(a.W, 16) * planes < 512) maxb = 8;
    a.B = pick_batch(a.len, a.W, maxb);
    a.logB = ilog2(a.B);
    a.pitch = a.len + 1;
    dim3 grid(san_cdiv(a.W, a.B), planes);
    size_t lds = lds_bytes(a);
    if (lds > 160 * 1024) {
        san_set_error("fft column of %d does not fit LDS", a.len);
        return SAN_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(fft_cols_kernel, grid, dim3(kThreads), lds, s, a);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

int ensure_big_lds() {
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    std::call_once(once, [] {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_rows_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (err == hipSuccess)
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_cols_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    return (int)err;
}

FftArgs base_args(int c, int h, int w) {
    FftArgs a{};
    a.C = c;
    a.H = h;
    a.W = w;
    a.inner = 1;
    a.scale = 1.f;
    a.sgn = 1.f;
    a.out_ctot = 2;
    return a;
}

int common_checks(const void* p0, const void* p1, int planes, int h, int w, void* ws, size_t ws_bytes) {
    if (!p0 || !p1) {
        san_set_error("null tensor pointer");
        return SAN_E_ARG;
    }
    if (planes < 1 || h < 1 || w < 1) {
        san_set_error("bad dims planes=%d h=%d w=%d", planes, h, w);
        return SAN_E_ARG;
    }
    if (!ws || ws_bytes < san_fft_workspace_bytes(planes, h, w)) {
        san_set_error("workspace too small: need %zu bytes", san_fft_workspace_bytes(planes, h, w));
        return SAN_E_WORKSPACE;
    }
    return ensure_big_lds();
}

}  // namespace

extern "C" {

int san_fft_prepare(int h, int w) {
    Plan p;
    int r = get_plan(h, &p);
    if (r) return r;
    r = get_plan(w, &p);
    if (r) return r;
    return ensure_big_lds();
}

size_t san_fft_workspace_bytes(int planes, int h, int w) {
    return (size_t)planes * (size_t)h * (size_t)w * sizeof(float2);
}

int san_fft2(const float* in, float* out, int planes, int h, int w, int inverse, const float* colmask_in,
             const float* colmask_out, int planar_ctot, void* ws, size_t ws_bytes, void* stream) {
    int r = common_checks(in, out, planes, h, w, ws, ws_bytes);
    const int planar_out = planar_ctot != 0;
    if (planar_out && planar_ctot < 2) {
        san_set_error("planar_ctot must be 0 (interleaved) or >= 2");
        return SAN_E_ARG;
    }
    if (r) return r;
    Plan ph, pw;
    if ((r = get_plan(h, &ph))) return r;
    if ((r = get_plan(w, &pw))) return r;
    hipStream_t s = (hipStream_t)stream;
    const float sgn = inverse ? -1.f : 1.f;
    FftArgs c = base_args(1, h, w);
    fill_pass(c, ph);
    c.in = (const float2*)in;
    c.out = (float2*)ws;
    c.cm_in = colmask_in;
    c.sgn = sgn;
    c.epi = CE_STORE;
    if ((r = launch_cols(c, planes, s))) return r;
    FftArgs a = base_args(1, h, w);
    fill_pass(a, pw);
    a.in = (const float2*)ws;
    a.out = (float2*)out;
    a.out_real = out;
    a.cm_out = planar_out ? nullptr : colmask_out;
    a.sgn = sgn;
    a.scale = (float)(1.0 / std::sqrt((double)h * (double)w));
    a.pro = RP_NONE;
    a.epi = planar_out ? RE_STORE_PLANAR : RE_STORE;
    if (planar_out) a.out_ctot = planar_ctot;
    if (planar_out && colmask_out) {
        san_set_error("planar_out with colmask_out is not supported");
        return SAN_E_UNSUPPORTED;
    }
    return launch_rows(a, planes, s);
}

// cols_done != 0: `k` already holds the inverse column transform (written by san_sens_expand_dc_next)
static int sens_reduce_impl(const float* k, const float* sens, float* out_planar, int out_ctot, int n, int c, int h, int w,
                            void* ws, size_t ws_bytes, int cols_done, void* stream) {
    int r = common_checks(k, out_planar, n * c, h, w, ws, ws_bytes);
    if (r) return r;
    SAN_CHECK_ARG(sens != nullptr, "sens is null");
    SAN_CHECK_ARG(out_ctot >= 2, "out_ctot must be >= 2");
    Plan ph, pw;
    if ((r = get_plan(h, &ph))) return r;
    if ((r = get_plan(w, &pw))) return r;
    hipStream_t s = (hipStream_t)stream;
    if (!cols_done) {
        FftArgs cargs = base_args(c, h, w);
        fill_pass(cargs, ph);
        cargs.in = (const float2*)k;
        cargs.out = (float2*)ws;
        cargs.sgn = -1.f;
        cargs.epi = CE_STORE;
        if ((r = launch_cols(cargs, n * c, s))) return r;
    }
    FftArgs a = base_args(c, h, w);
    fill_pass(a, pw);
    a.in = cols_done ? (const float2*)k : (const float2*)ws;
    a.sens = (const float2*)sens;
    a.out_real = out_planar;
    a.out_ctot = out_ctot;
    a.sgn = -1.f;
    a.scale = (float)(1.0 / std::sqrt((double)h * (double)w));
    a.inner = c;
    a.pro = RP_NONE;
    a.epi = RE_CONJ_SENS_SUM_PLANAR;
    return launch_rows(a, n, s);
}

int san_sens_reduce(const float* k, const float* sens, float* out_planar, int out_ctot, int n, int c, int h, int w,
                    void* ws, size_t ws_bytes, void* stream) {
    return sens_reduce_impl(k, sens, out_planar, out_ctot, n, c, h, w, ws, ws_bytes, 0, stream);
}

int san_sens_reduce_from_cols(const float* k_cols, const float* sens, float* out_planar, int out_ctot, int n, int c,
                              int h, int w, void* ws, size_t ws_bytes, void* stream) {
    return sens_reduce_impl(k_cols, sens, out_planar, out_ctot, n, c, h, w, ws, ws_bytes, 1, stream);
}

static int sens_expand_dc_impl(const float* r_planar, const float* sens, const float* k, const float* k0,
                               const float* mask, const float* dc_w, float* k_out, float* next_cols, int n, int c, int h,
                               int w, void* ws, size_t ws_bytes, void* stream) {
    int r = common_checks(r_planar, k_out, n * c, h, w, ws, ws_bytes);
    if (r) return r;
    SAN_CHECK_ARG(sens && k && k0 && mask && dc_w, "null input");
    Plan ph, pw;
    if ((r = get_plan(h, &ph))) return r;
    if ((r = get_plan(w, &pw))) return r;
    hipStream_t s = (hipStream_t)stream;
    FftArgs a = base_args(c, h, w);
    fill_pass(a, pw);
    a.in_planar = r_planar;
    a.sens = (const float2*)sens;
    a.out = (float2*)ws;
    a.sgn = 1.f;
    a.pro = RP_PLANAR_MUL_SENS;
    a.epi = RE_STORE;
    if ((r = launch_rows(a, n * c, s))) return r;
    FftArgs cargs = base_args(c, h, w);
    fill_pass(cargs, ph);
    cargs.in = (const float2*)ws;
    cargs.out = (float2*)k_out;
    cargs.k = (const float2*)k;
    cargs.k0 = (const float2*)k0;
    cargs.mask = mask;
    cargs.dcw = dc_w;
    cargs.sgn = 1.f;
    cargs.scale = (float)(1.0 / std::sqrt((double)h * (double)w));
    const bool fuse = next_cols != nullptr && h == kN320;           // the register-resident column kernel
    cargs.epi = fuse ? CE_DC_NEXT : CE_DC;
    cargs.out2 = (float2*)next_cols;
    if ((r = launch_cols(cargs, n * c, s))) return r;
    if (next_cols && !fuse) {                                         // other sizes: the same result in a second launch
        FftArgs nx = base_args(c, h, w);
        fill_pass(nx, ph);
        nx.in = (const float2*)k_out;
        nx.out = (float2*)next_cols;
        nx.sgn = -1.f;
        nx.epi = CE_STORE;
        return launch_cols(nx, n * c, s);
    }
    return SAN_OK;
}

int san_sens_expand_dc(const float* r_planar, const float* sens, const float* k, const float* k0,
                       const float* mask, const float* dc_w, float* k_out, int n, int c, int h, int w, void* ws,
                       size_t ws_bytes, void* stream) {
    return sens_expand_dc_impl(r_planar, sens, k, k0, mask, dc_w, k_out, nullptr, n, c, h, w, ws, ws_bytes, stream);
}

int san_sens_expand_dc_next(const float* r_planar, const float* sens, const float* k, const float* k0,
                            const float* mask, const float* dc_w, float* k_out, float* next_cols, int n, int c, int h,
                            int w, void* ws, size_t ws_bytes, void* stream) {
    SAN_CHECK_ARG(next_cols != nullptr, "next_cols is null");
    return sens_expand_dc_impl(r_planar, sens, k, k0, mask, dc_w, k_out, next_cols, n, c, h, w, ws, ws_bytes, stream);
}

static int ifft2_rss_impl(const float* k, float* out, int n, int c, int h, int w, void* ws, size_t ws_bytes,
                          int cols_done, void* stream) {
    int r = common_checks(k, out, n * c, h, w, ws, ws_bytes);
    if (r) return r;
    Plan ph, pw;
    if ((r = get_plan(h, &ph))) return r;
    if ((r = get_plan(w, &pw))) return r;
    hipStream_t s = (hipStream_t)stream;
    if (!cols_done) {
        FftArgs cargs = base_args(c, h, w);
        fill_pass(cargs, ph);
        cargs.in = (const float2*)k;
        cargs.out = (float2*)ws;
        cargs.sgn = -1.f;
        cargs.epi = CE_STORE;
        if ((r = launch_cols(cargs, n * c, s))) return r;
    }
    FftArgs a = base_args(c, h, w);
    fill_pass(a, pw);
    a.in = cols_done ? (const float2*)k : (const float2*)ws;
    a.out_real = out;
    a.sgn = -1.f;
    a.scale = (float)(1.0 / std::sqrt((double)h * (double)w));
    a.inner = c;
    a.pro = RP_NONE;
    a.epi = RE_RSS;
    return launch_rows(a, n, s);
}

int san_ifft2_rss(const float* k, float* out, int n, int c, int h, int w, void* ws, size_t ws_bytes,
                  void* stream) {
    return ifft2_rss_impl(k, out, n, c, h, w, ws, ws_bytes, 0, stream);
}

int san_ifft2_rss_from_cols(const float* k_cols, float* out, int n, int c, int h, int w, void* ws, size_t ws_bytes,
                            void* stream) {
    return ifft2_rss_impl(k_cols, out, n, c, h, w, ws, ws_bytes, 1, stream);
}

// rows per workgroup and grid of the image-domain cascade kernels (also sizes the dc_weight partials)
// Lines per wave of dc_rows320_kernel.  Two at the benchmark shape (N = 8 single-coil slices: 1,280 waves); ONE where two would leave
// SIMDs without a wave (a 15-coil 320^2 slice: 160 -> 320 waves, 64.1 -> 52.8 us forward, 62.6 -> 49.4 backward).  Measured at N = 8,
// C = 1 (round 6, profiles/r06_ab_dc_rows_lines_per_wave.txt): one line per wave = 2.5 waves per SIMD changes nothing forward
// (10.9 vs 11.0 us) and costs the backward 1 us -- the launch is bound by each wave's load -> transform -> load -> transform -> store
// chain, not by how many chains a SIMD interleaves.  SAN_DC_L320 = 1 / 2 forces a form (tuning hook).
static int dc_l320(int n, int h) {
    static const int forced = getenv("SAN_DC_L320") ? atoi(getenv("SAN_DC_L320")) : 0;
    if (forced == 1 || forced == 2) return forced;
    return (long long)n * san_cdiv(h, kDcL320) < 1024 ? 1 : kDcL320;
}

static void dc_rows_geom(int n, int h, int w, int* B, int* gx) {
    if (w == kN320) {
        *B = dc_l320(n, h);
    } else if (w == kN368) {
        *B = kL368;
    } else {
        int b = 8;
        while (b > 1 && (size_t)b * w * sizeof(float2) * 2 > 60 * 1024) b >>= 1;
        while (b > 1 && b * w > 24 * kThreads) b >>= 1;
        *B = b;
    }
    *gx = san_cdiv(h, *B);
}

int san_dc_rows_partials(int n, int c, int h, int w) {
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return 0;
    int B, gx;
    dc_rows_geom(n, h, w, &B, &gx);
    return gx * n * (w == kN320 ? 1 : c);      // the general-length kernel runs one coil per workgroup when c > 1
}

int san_fft_cols(const float* in, float* out, int planes, int h, int w, int inverse, void* stream) {
    SAN_CHECK_ARG(in && out, "null pointer");
    SAN_CHECK_ARG(planes > 0 && h > 0 && w > 0, "bad dims");
    int r = ensure_big_lds();
    if (r) return r;
    Plan ph;
    if ((r = get_plan(h, &ph))) return r;
    FftArgs c = base_args(1, h, w);
    fill_pass(c, ph);
    c.in = (const float2*)in;
    c.out = (float2*)out;
    c.sgn = inverse ? -1.f : 1.f;
    c.scale = (float)(1.0 / std::sqrt((double)h));
    c.epi = CE_STORE;
    return launch_cols(c, planes, (hipStream_t)stream);
}

static int dc_rows_impl(const float* x, const float* sens, const float* k0x, const float* mask, const float* dc_w,
                        const float* r_planar, float* x_out, float* m_out, int m_ctot, float* dk_out, const float* dk_in,
                        float* dcw_part, int backward, int n, int c, int h, int w, float* m_stats, void* stream);

int san_dc_rows(const float* x, const float* sens, const float* k0x, const float* mask, const float* dc_w,
                const float* r_planar, float* x_out, float* m_out, int m_ctot, float* dk_out, const float* dk_in,
                float* dcw_part, int backward, int n, int c, int h, int w, void* stream) {
    return dc_rows_impl(x, sens, k0x, mask, dc_w, r_planar, x_out, m_out, m_ctot, dk_out, dk_in, dcw_part, backward, n, c, h, w, nullptr, stream);
}

// Statistics records per (sample, plane) that san_dc_rows_stats writes for this shape; 0: the shape's kernel does not emit them
// (the caller runs san_plane_stats on m_out as before).
int san_dc_rows_stat_tiles(int n, int c, int h, int w) {
    if (n <= 0 || c <= 0 || h <= 0 || w != kN320) return 0;
    int B, gx;
    dc_rows_geom(n, h, w, &B, &gx);
    return gx;
}

// The forward form of san_dc_rows that ALSO emits the (count, mean, M2) records of the two planes of m_out -- the statistics the next
// cascade's NormUnet normalises its input with (varnet.py:262-273, 311-314): m_stats [n][2][san_dc_rows_stat_tiles(n, c, h, w)][3],
// to be merged by san_norm_finalize exactly like san_plane_stats' records (round 6: one launch per cascade less).
int san_dc_rows_stats(const float* x, const float* sens, const float* k0x, const float* mask, const float* dc_w,
                      const float* r_planar, float* x_out, float* m_out, int m_ctot, float* dk_out, float* m_stats, int n, int c,
                      int h, int w, void* stream) {
    SAN_CHECK_ARG(m_stats && m_out, "m_stats and m_out are required");
    SAN_CHECK_ARG(san_dc_rows_stat_tiles(n, c, h, w) > 0, "this shape's kernel does not emit statistics (san_dc_rows_stat_tiles == 0)");
    return dc_rows_impl(x, sens, k0x, mask, dc_w, r_planar, x_out, m_out, m_ctot, dk_out, nullptr, nullptr, 0, n, c, h, w, m_stats, stream);
}

static int dc_rows_impl(const float* x, const float* sens, const float* k0x, const float* mask, const float* dc_w,
                        const float* r_planar, float* x_out, float* m_out, int m_ctot, float* dk_out, const float* dk_in,
                        float* dcw_part, int backward, int n, int c, int h, int w, float* m_stats, void* stream) {
    SAN_CHECK_ARG(x && sens && mask && dc_w, "null input");
    SAN_CHECK_ARG(n > 0 && c > 0 && h > 0 && w > 0, "bad dims");
    SAN_CHECK_ARG(!m_out || m_ctot >= 2, "m_ctot must be >= 2");
    SAN_CHECK_ARG(backward ? (!k0x && !r_planar && !dk_out) : (!dk_in && !dcw_part), "operand not used in this direction");
    int r = ensure_big_lds();
    if (r) return r;
    Plan pw;
    if ((r = get_plan(w, &pw))) return r;
    FftArgs a = base_args(c, h, w);
    fill_pass(a, pw);
    a.in = (const float2*)x;
    a.sens = (const float2*)sens;
    a.k0 = (const float2*)k0x;
    a.mask = mask;
    a.dcw = dc_w;
    a.in_planar = r_planar;
    a.out = (float2*)x_out;
    a.out_real = m_out;
    a.out_ctot = m_out ? m_ctot : 2;
    a.dk_out = (float2*)dk_out;
    a.dk_in = (const float2*)dk_in;
    a.dcw_part = dcw_part;
    a.m_stats = m_stats;
    a.m_scale = backward ? -1.f : 1.f;
    a.scale = (float)(1.0 / std::sqrt((double)w));
    hipStream_t s = (hipStream_t)stream;
    int B, gx;
    dc_rows_geom(n, h, w, &B, &gx);
    if (w == kN320) {
        const dim3 grid(gx, n);
        if (B == 1) {
            if (backward) hipLaunchKernelGGL((dc_rows320_kernel<1, 1>), grid, dim3(64), 0, s, a);
            else hipLaunchKernelGGL((dc_rows320_kernel<0, 1>), grid, dim3(64), 0, s, a);
        } else {
            if (backward) hipLaunchKernelGGL((dc_rows320_kernel<1, 2>), grid, dim3(64), 0, s, a);
            else hipLaunchKernelGGL((dc_rows320_kernel<0, 2>), grid, dim3(64), 0, s, a);
        }
        SAN_LAUNCH_CHECK();
        return SAN_OK;
    }
    static const bool no368 = getenv("SAN_DC_GENERIC") != nullptr;      // tuning / A-B hook: the general-width kernel at 368
    const bool w368 = w == kN368 && !no368;
    if (w368) {
        const dim3 grid(gx, n, c);
        if (backward) hipLaunchKernelGGL((dc_rows368_kernel<1>), grid, dim3(64), 0, s, a);
        else hipLaunchKernelGGL((dc_rows368_kernel<0>), grid, dim3(64), 0, s, a);
        SAN_LAUNCH_CHECK();
    } else {
    a.B = B;
    if (a.B * a.W > 24 * kThreads) {
        san_set_error("row length %d too long for the image-domain cascade kernel", w);
        return SAN_E_UNSUPPORTED;
    }
    a.logB = ilog2(a.B);
    a.pitch = a.len;
    const size_t lds = sizeof(float2) * (2 * (size_t)a.len + 2 * (size_t)a.B * a.pitch) + sizeof(float) * (size_t)w;
    if (lds > 160 * 1024) {
        san_set_error("fft row of %d does not fit LDS", a.len);
        return SAN_E_UNSUPPORTED;
    }
    // elements per thread: 12 covers 8 rows of up to 384; the 24 form (long rows) loads S and r after the inverse transform
    const bool small = a.B * a.W <= 12 * kThreads;
    const void* fn = small ? (backward ? (const void*)dc_rows_kernel<1, 12, kDcPre> : (const void*)dc_rows_kernel<0, 12, kDcPre>)
                           : (backward ? (const void*)dc_rows_kernel<1, 24, 0> : (const void*)dc_rows_kernel<0, 24, 0>);
    {
        static std::mutex mu;
        static std::map<const void*, hipError_t> done;
        std::lock_guard<std::mutex> g(mu);
        auto it = done.find(fn);
        if (it == done.end()) it = done.emplace(fn, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)).first;
        if (it->second != hipSuccess) return (int)it->second;
    }
    const dim3 grid(gx, n, c);
    void* params[] = {(void*)&a};
    hipError_t le = hipLaunchKernel(fn, grid, dim3(kThreads), params, lds, s);
    if (le != hipSuccess) return (int)le;
    SAN_LAUNCH_CHECK();
    }
    if (c > 1 && m_out) {
        // coil combination in its own pass: of the result (forward) or of the incoming gradient (backward)
        const float2* q = backward ? (const float2*)x : (const float2*)x_out;
        SAN_CHECK_ARG(q != nullptr, "the coil combination of a multi-coil forward launch needs x_out");
        int bx = san_cdiv(h * w, kThreads);
        if (bx > 1024) bx = 1024;
        hipLaunchKernelGGL(coil_combine_kernel, dim3(bx, n), dim3(kThreads), 0, s, q, (const float2*)sens, m_out, m_ctot,
                           a.m_scale, c, h * w);
        SAN_LAUNCH_CHECK();
    }
    return SAN_OK;
}

}  // extern "C"
