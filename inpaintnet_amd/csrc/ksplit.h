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
#include <type_traits>
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

// acc[ms][slot[g]] += A[16*MS rows, this wave's K quarter] * Bg[16 rows, same K]^T
//
// The caller's hook requests its epilogue operands from inside the contraction, so their latency hides under the MFMA
// phase.  It is called with a tag std::integral_constant<int, I>:
//   I = -1  right after the prologue loads are issued -- the place for kernarg_touch(): the wave is about to wait for
//           its first fragments anyway, the scalar round trip disappears in that shadow;
//   I >= 0  piece I, dealt out in front of the I-th MFMA chunk of the first k-step (MS + NB chunks per step), so the
//           requests are spread like the fragment loads instead of stalling the wave in one block.
struct NoHook {
    template <class Tag>
    __device__ __forceinline__ void operator()(Tag) const {}
};
template <int I>
using HookTag = std::integral_constant<int, I>;

template <int N, class Hook, int I = 0>
__device__ __forceinline__ void hook_pieces(Hook&& h) {
    if constexpr (I < N) {
        h(HookTag<I>{});
        hook_pieces<N, Hook, I + 1>(static_cast<Hook&&>(h));
    }
}

// One fragment of a k-step (unguarded forms only): I < MS -> a[I], else b[I - MS].
template <int MS, int NB, bool PK, int I>
__device__ __forceinline__ void load_frag(Frag<MS, NB>& f, const float* __restrict__ A, long lda, int row0, int rowsA,
                                          const float* __restrict__ Bm, long ldb, const int (&brow)[NB], int K, int s,
                                          int i16, int q) {
    if (PK) {
        const int S = K >> 4, lane4 = (q * 16 + i16) * 4;
        if (I < MS) {
            const int last = (rowsA - 1) >> 4;
            f.a[I < MS ? I : 0] = *reinterpret_cast<const f32x4*>(A + ((long)min((row0 >> 4) + I, last) * S + s) * 256 + lane4);
        } else {
            constexpr int g = I < MS ? 0 : I - MS;
            f.b[g] = *reinterpret_cast<const f32x4*>(Bm + ((long)(brow[g] >> 4) * S + s) * 256 + lane4);
        }
    } else {
        const int k = 16 * s + 4 * q;
        if (I < MS) {
            f.a[I < MS ? I : 0] = ld4u(A + (long)min(row0 + 16 * I + i16, rowsA - 1) * lda + k);
        } else {
            constexpr int g = I < MS ? 0 : I - MS;
            f.b[g] = ld4u(Bm + (long)(brow[g] + i16) * ldb + k);
        }
    }
}

// MFMAs [M0, M1) of a k-step in the order e (4) x ms x g
template <int MS, int NB, int M0, int M1>
__device__ __forceinline__ void mma_range(f32x4 (&acc)[MS][4], const int (&slot)[NB], const Frag<MS, NB>& f) {
#pragma unroll
    for (int m = M0; m < M1; ++m) {
        const int e = m / (MS * NB), ms = (m % (MS * NB)) / NB, g = m % NB;
        acc[ms][slot[g]] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ms][e], f.b[g][e], acc[ms][slot[g]], 0, 0, 0);
    }
}

// MFMAs of step `cur` with the loads of a later step `nxt` (k-step s_nxt) dealt in between them, one load per
// ceil(4*MS*NB / (MS+NB)) MFMAs.  Why: a wave issues in order and a vector-memory instruction waits in the issue
// stage until the CU's address path takes it (~100 cycles per 1 KB instruction with 4 waves streaming, measured:
// 56 back-to-back loads = 5,500 cycles during which the wave's MFMA pipe idles).  Spread out, the same loads need
// 18-40 B/clk of the CU's 64 and cost the MFMA stream nothing.  sched_barrier pins the interleave.
template <int MS, int NB, bool PK, bool LOAD, int I = 0, class Hook = NoHook>
__device__ __forceinline__ void mma_and_prefetch(f32x4 (&acc)[MS][4], const int (&slot)[NB], const Frag<MS, NB>& cur,
                                                 Frag<MS, NB>& nxt, const float* __restrict__ A, long lda, int row0,
                                                 int rowsA, const float* __restrict__ Bm, long ldb,
                                                 const int (&brow)[NB], int K, int s_nxt, int i16, int q,
                                                 Hook&& hook = Hook()) {
    constexpr int NM = 4 * MS * NB, FR = MS + NB, CH = (NM + FR - 1) / FR;
    if constexpr (I < FR) {
        hook(HookTag<I>{});
        if (LOAD) load_frag<MS, NB, PK, I>(nxt, A, lda, row0, rowsA, Bm, ldb, brow, K, s_nxt, i16, q);
        constexpr int M0 = I * CH < NM ? I * CH : NM, M1 = (I + 1) * CH < NM ? (I + 1) * CH : NM;
        mma_range<MS, NB, M0, M1>(acc, slot, cur);
        __builtin_amdgcn_sched_barrier(0);
        mma_and_prefetch<MS, NB, PK, LOAD, I + 1>(acc, slot, cur, nxt, A, lda, row0, rowsA, Bm, ldb, brow, K, s_nxt, i16, q,
                                                  static_cast<Hook&&>(hook));
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

// Streamed contraction over k-steps [s_beg, s_end) (count a multiple of R): ring of R step-fragments, R-1 steps of
// loads requested up front, then every step's MFMAs carry the loads of the step R-1 ahead.  The loop body (R steps,
// static ring slots) is branch-free; the last R-1 steps run without loads.
template <int MS, int NB, bool PK, int R, class Hook>
__device__ __forceinline__ void ksplit_stream(f32x4 (&acc)[MS][4], const int (&slot)[NB], const float* __restrict__ A,
                                              long lda, int row0, int rowsA, const float* __restrict__ Bm, long ldb,
                                              const int (&brow)[NB], int K, int s_beg, int s_end, int i16, int q,
                                              Hook&& after_first_loads) {
    Frag<MS, NB> f[R];
#pragma unroll
    for (int d = 0; d < R - 1; ++d) load_step<MS, NB, false, PK>(f[d], A, lda, row0, rowsA, Bm, ldb, brow, K, s_beg + d, i16, q);
    __builtin_amdgcn_sched_barrier(0);
    after_first_loads(HookTag<-1>{});
    __builtin_amdgcn_sched_barrier(0);
    // step j of a body that starts at k-step s: MFMAs of ring slot j, loads of step s+j+R-1 into the slot freed last;
    // the very first step also carries the caller's hook pieces
    auto body = [&](int s, auto with_hook) {
        if constexpr (decltype(with_hook)::value)
            mma_and_prefetch<MS, NB, PK, true>(acc, slot, f[0], f[R - 1], A, lda, row0, rowsA, Bm, ldb, brow, K, s + R - 1, i16,
                                               q, after_first_loads);
        else
            mma_and_prefetch<MS, NB, PK, true>(acc, slot, f[0], f[R - 1], A, lda, row0, rowsA, Bm, ldb, brow, K, s + R - 1, i16, q);
#pragma unroll
        for (int j = 1; j < R; ++j)
            mma_and_prefetch<MS, NB, PK, true>(acc, slot, f[j], f[(j + R - 1) % R], A, lda, row0, rowsA, Bm, ldb, brow, K,
                                               s + j + R - 1, i16, q);
    };
    auto drain = [&](int s, auto with_hook) {
        if constexpr (decltype(with_hook)::value)
            mma_and_prefetch<MS, NB, PK, true>(acc, slot, f[0], f[R - 1], A, lda, row0, rowsA, Bm, ldb, brow, K, s + R - 1, i16,
                                               q, after_first_loads);
        else
            mma_and_prefetch<MS, NB, PK, true>(acc, slot, f[0], f[R - 1], A, lda, row0, rowsA, Bm, ldb, brow, K, s + R - 1, i16, q);
#pragma unroll
        for (int j = 1; j < R; ++j)
            mma_and_prefetch<MS, NB, PK, false>(acc, slot, f[j], f[0], A, lda, row0, rowsA, Bm, ldb, brow, K, 0, i16, q);
    };
    using Yes = std::true_type;
    using No = std::false_type;
    int s = s_beg;
    if (s + R < s_end) {
        body(s, Yes{});
        for (s += R; s + R < s_end; s += R) body(s, No{});
        drain(s, No{});
    } else {
        drain(s, Yes{});
    }
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
    if ((K & 255) == 0) {
        // each wave owns S/4 steps, a multiple of 4: ring of 4, three steps requested up front.  (A ring of 8 where the
        // registers allow it measured slower: the 7-step prologue is a longer burst in front of the first MFMA.)
        const int Sq = S >> 2;
        ksplit_stream<MS, NB, PK, 4>(acc, slot, A, lda, row0, rowsA, Bm, ldb, brow, K, w * Sq, w * Sq + Sq, i16, q,
                                     after_first_loads);
    } else {
        // general path (small / odd K): guarded loads, one step at a time
        const int Sq = (S + 3) >> 2;
        const int s_beg = w * Sq;
        const int s_end = min(S, s_beg + Sq);
        after_first_loads(HookTag<-1>{});
        hook_pieces<MS + NB>(after_first_loads);
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
