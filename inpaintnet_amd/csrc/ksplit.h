// Register-streamed K-split contraction shared by the fused recurrent step kernels (GRU: gru.hip, LSTM: lstm.hip).
//
// A workgroup of 4 waves computes a (16*MS rows) x (NB groups of 16 weight rows) tile; each wave owns a contiguous
// quarter of K and streams its own MFMA fragments straight from L2 into registers (16 B per lane = 4 consecutive k of
// one row, which IS the operand layout of v_mfma_f32_16x16x4_f32 up to a k-permutation common to A and B).  No LDS and
// no barrier in the main loop; the partial accumulators are combined through LDS at the end (reduce_waves).
//
// Operand layouts.  ROW-MAJOR (PK=false): lane (i16,q) reads 16 B of row i16 at k = 16s+4q -- one wave instruction
// touches 16 rows x 64 B.  The CU's address path serialises on the 16 distinct rows: measured 36 GB/s per CU
// (tools/exp_load.hip, profiles/r01_f_load_patterns.txt).  FRAGMENT-MAJOR (PK=true): the operand is stored as
// [row/16][k/16][lane][4] so the same fragment is ONE contiguous KB per wave instruction: 126 GB/s per CU, 3.4x.
// The recurrent weights are re-packed once per call (pw_pack_frag), the hidden state / gate gradients are written in
// this layout by the previous step's epilogue (pk_offset), so the extra cost is a few KB of stores per workgroup.
#pragma once
#include "common.h"

namespace ksplit {

constexpr int TM_ROWS = 32;   // batch rows per workgroup of the base (MS = 2) geometry
constexpr int TH = 16;        // hidden units per workgroup

// Fragments of one k-step (16 k) for this wave: MS sub-tiles of 16 A rows, NB groups of 16 B rows.
template <int MS, int NB>
struct Frag {
    f32x4 a[MS];
    f32x4 b[NB];
};

// One k-step of fragments.  GUARD=false: every load is an unconditional 16-byte load (K multiple of 16).
// A rows past the batch are clamped, not zeroed: they only feed output rows that are never stored.
// (Per-lane "load or zero" guards make hipcc wrap each load in an exec-mask branch and wait vmcnt(0) per
// element -- the round trips serialise; the guarded form is kept only for K not a multiple of 512.)
// float offset of element (row, col) in a fragment-major buffer with S = K/16 k-steps per row block
__host__ __device__ __forceinline__ long pk_offset(int row, int col, int S) {
    return ((long)(row >> 4) * S + (col >> 4)) * 256 + ((((col & 15) >> 2) * 16 + (row & 15)) << 2) + (col & 3);
}

template <int MS, int NB, bool GUARD, bool PK = false>
__device__ __forceinline__ void load_step(Frag<MS, NB>& f, const float* __restrict__ A, long lda, int row0, int rowsA,
                                          const float* __restrict__ Bm, long ldb, const int (&brow)[NB], int K, int s,
                                          int i16, int q) {
    if (PK) {
        // fragment-major: block (row/16, s) is 256 floats in lane order; row blocks past the batch are clamped
        const int S = K >> 4, lane4 = (q * 16 + i16) * 4, last = (rowsA - 1) >> 4;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            f.a[ms] = *reinterpret_cast<const f32x4*>(A + ((long)min((row0 >> 4) + ms, last) * S + s) * 256 + lane4);
#pragma unroll
        for (int g = 0; g < NB; ++g)
            f.b[g] = *reinterpret_cast<const f32x4*>(Bm + ((long)(brow[g] >> 4) * S + s) * 256 + lane4);
        return;
    }
    const int k = 16 * s + 4 * q;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
        const int row = min(row0 + 16 * ms + i16, rowsA - 1);
        const float* p = A + (long)row * lda + k;
        if (!GUARD) f.a[ms] = ld4u(p);
        else {
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) if (k + e < K) x[e] = p[e];
            f.a[ms] = x;
        }
    }
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        const float* p = Bm + (long)(brow[g] + i16) * ldb + k;
        if (!GUARD) f.b[g] = ld4u(p);
        else {
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) if (k + e < K) x[e] = p[e];
            f.b[g] = x;
        }
    }
}

template <int MS, int NB>
__device__ __forceinline__ void mma_step(f32x4 (&acc)[MS][4], const int (&slot)[NB], const Frag<MS, NB>& f) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int g = 0; g < NB; ++g)
                acc[ms][slot[g]] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ms][e], f.b[g][e], acc[ms][slot[g]], 0, 0, 0);
}

// acc[ms][slot[g]] += A[16*MS rows, this wave's K quarter] * Bg[16 rows, same K]^T
//
// Issue order is the whole game here (profiles/r01_c): a wave issues in order, so a load placed behind an MFMA that
// waits on vmcnt is not even REQUESTED until that data is back.  The paths below therefore request as much of the
// wave's K range as the register file allows before the first MFMA: all of it when K = 512 (8 k-steps, 40 x 16-byte
// loads per lane for the forward step), otherwise two groups of GDEPTH k-steps kept in flight.
//
// `after_first_loads` is a hook the caller uses to request its epilogue operands: placed AFTER the first group of
// fragment loads (loads return in order, so cold epilogue operands requested first would hold the L2-warm fragments
// -- and the first MFMA -- behind them) and BEFORE the first MFMA (so their latency hides under the MFMA phase).
struct NoHook { __device__ __forceinline__ void operator()() const {} };

template <int MS, int NB, int GDEPTH, bool PK, class Hook>
__device__ __forceinline__ void ksplit_fast(f32x4 (&acc)[MS][4], const int (&slot)[NB], const float* __restrict__ A,
                                            long lda, int row0, int rowsA, const float* __restrict__ Bm, long ldb,
                                            const int (&brow)[NB], int K, int s_beg, int s_end, int i16, int q,
                                            Hook&& after_first_loads) {
    Frag<MS, NB> f0[GDEPTH], f1[GDEPTH];
#pragma unroll
    for (int d = 0; d < GDEPTH; ++d)
        load_step<MS, NB, false, PK>(f0[d], A, lda, row0, rowsA, Bm, ldb, brow, K, s_beg + d, i16, q);
    after_first_loads();
    for (int s = s_beg; s < s_end; s += 2 * GDEPTH) {
        const bool more1 = s + GDEPTH < s_end, more2 = s + 2 * GDEPTH < s_end;     // wave-uniform
        if (more1) {
#pragma unroll
            for (int d = 0; d < GDEPTH; ++d)
                load_step<MS, NB, false, PK>(f1[d], A, lda, row0, rowsA, Bm, ldb, brow, K, s + GDEPTH + d, i16, q);
        }
#pragma unroll
        for (int d = 0; d < GDEPTH; ++d) mma_step<MS, NB>(acc, slot, f0[d]);
        if (more2) {
#pragma unroll
            for (int d = 0; d < GDEPTH; ++d)
                load_step<MS, NB, false, PK>(f0[d], A, lda, row0, rowsA, Bm, ldb, brow, K, s + 2 * GDEPTH + d, i16, q);
        }
        if (more1) {
#pragma unroll
            for (int d = 0; d < GDEPTH; ++d) mma_step<MS, NB>(acc, slot, f1[d]);
        }
    }
}

// single group: the wave's whole K range (8 k-steps) is requested before the first MFMA
template <int MS, int NB, bool PK, class Hook>
__device__ __forceinline__ void ksplit_once8(f32x4 (&acc)[MS][4], const int (&slot)[NB], const float* __restrict__ A,
                                             long lda, int row0, int rowsA, const float* __restrict__ Bm, long ldb,
                                             const int (&brow)[NB], int K, int s_beg, int i16, int q,
                                             Hook&& after_first_loads) {
    Frag<MS, NB> f[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) load_step<MS, NB, false, PK>(f[d], A, lda, row0, rowsA, Bm, ldb, brow, K, s_beg + d, i16, q);
    after_first_loads();
#pragma unroll
    for (int d = 0; d < 8; ++d) mma_step<MS, NB>(acc, slot, f[d]);
}

// PK: both operands fragment-major (requires K % 256 == 0; the host only passes packed operands then).
template <int MS, int NB, bool PK = false, class Hook = NoHook>
__device__ __forceinline__ void ksplit_segment(f32x4 (&acc)[MS][4], const int (&slot)[NB],
                                               const float* __restrict__ A, long lda, int row0, int rowsA,
                                               const float* __restrict__ Bm, long ldb, const int (&brow)[NB],
                                               int K, int t, Hook&& after_first_loads = Hook()) {
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);      // provably wave-uniform -> scalar loop control
    const int i16 = lane & 15, q = lane >> 4;
    const int S = (K + 15) >> 4;                   // k-steps of 16
    constexpr int FR = MS + NB;                    // float4 fragment registers per k-step
    if (K == 512 && FR <= 7) {
        ksplit_once8<MS, NB, PK>(acc, slot, A, lda, row0, rowsA, Bm, ldb, brow, K, w * 8, i16, q, after_first_loads);
    } else if (PK) {
        // each wave owns S/4 = multiple of 4 steps
        const int Sq = S >> 2;
        ksplit_fast<MS, NB, 4, true>(acc, slot, A, lda, row0, rowsA, Bm, ldb, brow, K, w * Sq, w * Sq + Sq, i16, q,
                                     after_first_loads);
    } else if ((K & 511) == 0) {
        // each wave owns S/4 = multiple of 8 steps
        const int Sq = S >> 2;
        constexpr int GDEPTH = FR <= 3 ? 8 : 4;
        ksplit_fast<MS, NB, GDEPTH, false>(acc, slot, A, lda, row0, rowsA, Bm, ldb, brow, K, w * Sq, w * Sq + Sq, i16, q,
                                           after_first_loads);
    } else {
        // general path (small / odd K): guarded loads, one step at a time
        const int Sq = (S + 3) >> 2;
        const int s_beg = w * Sq;
        const int s_end = min(S, s_beg + Sq);
        after_first_loads();
        for (int s = s_beg; s < s_end; ++s) {
            Frag<MS, NB> f;
            load_step<MS, NB, true>(f, A, lda, row0, rowsA, Bm, ldb, brow, K, s, i16, q);
            mma_step<MS, NB>(acc, slot, f);
        }
    }
}

// Cross-wave reduction: every wave dumps its partial accumulators, then thread t owns output positions
// t + 256*p (p < MS) of the (16*MS) x 16 tile (pos = row*16 + col) for all NACC accumulators.
// C/D map of the 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + reg.
template <int MS, int NACC>
__device__ __forceinline__ void reduce_waves(const f32x4 (&acc)[MS][4], float* red, int t, float (&out)[MS][NACC]) {
    const int lane = t & 63, w = t >> 6;
    constexpr int TILE = MS * 256;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ms + 4 * (lane >> 4) + r;
                red[(w * NACC + a) * TILE + row * 16 + (lane & 15)] = acc[ms][a][r];
            }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < MS; ++p)
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) s += red[(ww * NACC + a) * TILE + t + 256 * p];
            out[p][a] = s;
        }
}

}  // namespace ksplit
