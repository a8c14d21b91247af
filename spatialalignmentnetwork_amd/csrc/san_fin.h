// InstanceNorm / BatchNorm finalisation INSIDE the kernel that produced the statistics records (round 6).
//
// Every convolution emits per-(sample, channel) statistics records (count, mean, M2) -- one per wave per tile -- and a tiny
// second launch (norm_finalize_kernel, san_norm.hip) merged them into the lazy affine (scale, shift) the next kernel applies.
// That launch does 2-5 KB of work per plane and costs what every launch costs (~5 us of ramp, drain and boundary); a cascade has
// 22 of them, a training step 299.  Here the LAST workgroup of a reduction domain to finish does the merge itself:
//
//   producer (every workgroup, after its records):   records stored WRITE-THROUGH (agent-scope stores: no L2 write-back fence),
//                                                    s_waitcnt vmcnt(0), barrier, ONE relaxed agent-scope add on the domain's ticket
//   last arriver (old + arrivals == expected):       resets the ticket, reads the domain's records with agent-scope loads and merges
//                                                    them in a FIXED order (a function of the record index only, never of the arrival
//                                                    order): Chan's formula in double, two passes like norm_finalize_kernel
//
// Domain: InstanceNorm -> one sample (all its channels): ticket[sample]; BatchNorm (training) -> the whole launch: ticket[0].
// Nothing waits for another workgroup (no spinning): the protocol is placement-independent.  The ticket array is zeroed once by
// the caller and left zero by every launch.
#pragma once
#include "san_common.h"


__device__ __forceinline__ void san_stat_store(float* o, float cnt, float mean, float m2, bool write_through) {
    if (write_through) {
        __hip_atomic_store(o, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 2, m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        o[0] = cnt;
        o[1] = mean;
        o[2] = m2;
    }
}

// The merge.  rec(ch, e) -> the e-th record (3 floats) of channel ch's statistic, e in [0, total), or null (no such record);
// n: samples (the BatchNorm affine is written for each), n_lo: the sample an InstanceNorm statistic belongs to.  T threads (all of
// the workgroup) call it; G = T / c threads share a channel (at least one), their partial sums meet in LDS in thread order.
// `red`: 3 T doubles of LDS scratch.
template <int T, typename Rec>
__device__ __forceinline__ void san_fin_merge(const SanFin& f, Rec rec_of, int total, int n, int c, int n_lo, double* red) {
    const int tid = threadIdx.x;
    // the records were stored write-through by other compute units: drop what this one may hold of those lines (agent-scope acquire)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int G = c >= T ? 1 : T / c;                   // threads per channel
    const int per_round = T / G;                        // channels per round
    constexpr int kKeep = 24;                           // records per thread and batch
    for (int c0 = 0; c0 < c; c0 += per_round) {
        const int ch = c0 + tid / G, g = tid - (tid / G) * G;
        const bool on = tid < per_round * G && ch < c;
        // A batch = kKeep records per thread, ALL requested before anything waits for one of them: no branch between the loads
        // (a record that does not exist is a clamped re-read of record 0, dropped afterwards).  Behind `if (exists)` the compiler
        // waited for every record's three words in turn: a dozen dependent trips to memory per thread, 25 us per layer.
        // ONE pass: a thread folds its records into (count, mean, M2) with Chan's update in double, in record order; the G
        // threads of a channel are then folded in thread order.  (One batch of registers live: the tail must not raise the
        // register count of the kernel it sits in.)
        const float* safe = rec_of(on ? ch : 0, 0);
        double cnt = 0.0, mean = 0.0, m2 = 0.0;
        for (int base = 0; base < total; base += G * kKeep) {                // (workgroup-uniform trip count)
            float bc[kKeep], bm[kKeep], b2[kKeep];
            bool ok[kKeep];
#pragma unroll
            for (int i = 0; i < kKeep; ++i) {
                const int e = base + g + G * i;
                ok[i] = on && e < total;
                const float* q = ok[i] ? rec_of(ch, e) : safe;
                // (plain loads behind the merge's agent-scope ACQUIRE fence: one 12-byte request per record, served by the caches;
                // as 3 agent-scope dword loads each, the 18,000 records of a 320 x 320 layer were 18,000 x 3 separate trips to the
                // fabric: +15-45 us per layer)
                bc[i] = q[0];
                bm[i] = q[1];
                b2[i] = q[2];
            }
#pragma unroll
            for (int i = 0; i < kKeep; ++i) {
                const double nb = ok[i] ? (double)bc[i] : 0.0;
                if (nb > 0.0) {
                    const double nab = cnt + nb, d = (double)bm[i] - mean;
                    mean += d * (nb / nab);
                    m2 += (double)b2[i] + d * d * (cnt * nb / nab);
                    cnt = nab;
                }
            }
        }
        // the channel's G partial triples, folded by its first thread in thread order
        __syncthreads();
        red[tid] = cnt;
        red[T + tid] = mean;
        red[2 * T + tid] = m2;
        __syncthreads();
        if (on && g == 0) {
            for (int k = 1; k < G; ++k) {
                const double nb = red[tid + k];
                if (nb > 0.0) {
                    const double nab = cnt + nb, d = red[T + tid + k] - mean;
                    mean += d * (nb / nab);
                    m2 += red[2 * T + tid + k] + d * d * (cnt * nb / nab);
                    cnt = nab;
                }
            }
            const double var_b = cnt > 0.0 ? m2 / cnt : 0.0;
            if (!f.batch) {
                // InstanceNorm2d: biased variance, eps inside the root (varnet.py:141) -- norm_finalize_kernel's arithmetic
                const float sc = (float)(1.0 / sqrt(var_b + (double)f.eps));
                f.scale[n_lo * f.sc_ctot + f.sc_coff + ch] = sc;
                f.shift[n_lo * f.sc_ctot + f.sc_coff + ch] = (float)(-mean) * sc;
            } else {
                // training BatchNorm2d (unet.py:125): batch statistics over (N, H, W); running statistics with the unbiased variance
                const float ga = f.gamma ? f.gamma[ch] : 1.f, bt = f.beta ? f.beta[ch] : 0.f;
                const float sc = ga * (float)(1.0 / sqrt(var_b + (double)f.eps));
                const float sh = bt - (float)mean * sc;
                for (int b = 0; b < n; ++b) {
                    f.scale[b * f.sc_ctot + f.sc_coff + ch] = sc;
                    f.shift[b * f.sc_ctot + f.sc_coff + ch] = sh;
                }
                const double var_u = cnt > 1.0 ? m2 / (cnt - 1.0) : 0.0;
                if (f.bmean) f.bmean[ch] = (float)mean;
                if (f.bvar) f.bvar[ch] = (float)var_u;
                if (f.rmean) {
                    f.rmean[ch] = f.rmean[ch] * (1.f - f.momentum) + f.momentum * (float)mean;
                    f.rvar[ch] = f.rvar[ch] * (1.f - f.momentum) + f.momentum * ((float)var_u * f.var_factor);
                    if (ch == 0 && f.nbt) *f.nbt += 1;
                }
            }
        }
        __syncthreads();
    }
}

// Every thread of the workgroup calls this (uniformly) after its statistics records of `sample` were stored with
// san_stat_store(..., true).  arrivals: how many of the domain's expected arrivals this call stands for.  True (workgroup-uniform)
// for the ONE workgroup that completed the domain: it then calls san_fin_merge.
template <int T>
__device__ __forceinline__ bool san_fin_arrive(const SanFin& f, int sample, unsigned arrivals) {
    __shared__ unsigned san_fin_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this thread's records are in memory before the arrival can be seen
    __syncthreads();
    unsigned* tk = f.ticket + (f.batch ? 0 : sample);
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(tk, arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = old + arrivals == f.expected ? 1u : 0u;
        if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // zero again for the next launch
        san_fin_last = last;
    }
    __syncthreads();
    // (through readfirstlane: the flag is workgroup-uniform, and the compiler must know it -- the callers branch on it inside loops
    // whose scalar state feeds inline-asm "s" operands)
    return __builtin_amdgcn_readfirstlane((int)san_fin_last) != 0;
}

// The common case: records [n][c][tiles][3], every one of them written by this launch.
template <int T>
__device__ __forceinline__ void san_fin_tail(const SanFin& f, const float* __restrict__ part, int n, int c, int tiles, int sample,
                                          unsigned arrivals) {
    __shared__ double san_fin_red[3 * T];
    if (!san_fin_arrive<T>(f, sample, arrivals)) return;
    const int n_lo = f.batch ? 0 : sample, n_hi = f.batch ? n : sample + 1;
    san_fin_merge<T>(f, [&](int ch, int e) -> const float* {
        const int b = n_lo + e / tiles, t = e - (e / tiles) * tiles;
        return part + ((size_t)(b * c + ch) * tiles + t) * 3;
    }, (n_hi - n_lo) * tiles, n, c, n_lo, san_fin_red);
}
