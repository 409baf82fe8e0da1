// Stable LSD radix sort of (key, value) pairs by ONE 1024-thread workgroup (device code, shared by dmt_kernels.hip and
// thin_kernels.hip).  PASSES passes of 8 bits from bit 0; PASSES even: the result ends in (k0, v0), (k1, v1) is scratch.
// Per pass: LDS histogram, wave scan, then chunks of 1024 pairs in order: every lane finds the lanes of its wave that
// hold the same digit with 8 ballots (peer mask), its rank among them is a popcount below its lane, per-wave digit
// counts are prefixed across the 16 waves in LDS.  Stability is part of the contract of both callers (the reference's
// tie-break is the position in the input).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmat {

constexpr int BRS_T = 1024, BRS_W = BRS_T / 64;

template <typename K, typename V, int PASSES>
__device__ __forceinline__ void block_radix_sort(K *k0, V *v0, K *k1, V *v1, int n)
{
    static_assert(PASSES % 2 == 0, "even number of passes: the result returns to the first buffer pair");
    __shared__ unsigned hist[256];           // digit counts of the pass, then the running output base of every digit
    __shared__ unsigned wcnt[BRS_W][256];    // per-wave digit counts of a chunk, then their exclusive prefix over the waves
    __shared__ unsigned tot[256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    K *kin = k0, *kout = k1;
    V *vin = v0, *vout = v1;
    for (int pass = 0; pass < PASSES; pass++) {
        const int sh = 8 * pass;
        if (t < 256) hist[t] = 0;
        __syncthreads();
        for (int i = t; i < n; i += BRS_T) atomicAdd(&hist[(unsigned)(kin[i] >> sh) & 255u], 1u);
        __syncthreads();
        if (wave == 0) {                     // exclusive scan of the 256 counts: 4 per lane + a wave scan
            unsigned c[4], s = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { c[j] = hist[lane * 4 + j]; s += c[j]; }
            unsigned incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o); if (lane >= o) incl += u; }
            unsigned run = incl - s;
#pragma unroll
            for (int j = 0; j < 4; j++) { hist[lane * 4 + j] = run; run += c[j]; }
        }
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += BRS_T) {
            for (int j = t; j < BRS_W * 256; j += BRS_T) (&wcnt[0][0])[j] = 0;
            __syncthreads();
            const int i = c0 + t;
            const bool valid = i < n;
            const K key = valid ? kin[i] : (K)0;
            const V val = valid ? vin[i] : (V)0;
            const unsigned d = (unsigned)(key >> sh) & 255u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const bool bit = (d >> b) & 1;
                const unsigned long long bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
            const unsigned rank = __popcll(peers & ((1ull << lane) - 1ull));
            if (valid && rank == 0) wcnt[wave][d] = __popcll(peers);
            __syncthreads();
            if (t < 256) {                   // exclusive prefix over the waves, per digit
                unsigned run = 0;
#pragma unroll
                for (int w = 0; w < BRS_W; w++) { const unsigned c = wcnt[w][t]; wcnt[w][t] = run; run += c; }
                tot[t] = run;
            }
            __syncthreads();
            if (valid) {
                const unsigned pos = hist[d] + wcnt[wave][d] + rank;
                kout[pos] = key;
                vout[pos] = val;
            }
            __syncthreads();
            if (t < 256) hist[t] += tot[t];
        }
        __syncthreads();
        __threadfence();                     // this pass's scattered stores -> the next pass's reads (same workgroup, through L2)
        __syncthreads();
        K *tk = kin; kin = kout; kout = tk;
        V *tv = vin; vin = vout; vout = tv;
    }
}

}  // namespace tmat
