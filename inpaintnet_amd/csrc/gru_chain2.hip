// Chain kernels, second generation (round 3): the same persistent-launch protocol as gru_chain.hip (chain.h), with
//   * one ROW BLOCK (16 batch rows) per wave over the whole K instead of one K quarter of four row blocks: no cross-wave
//     reduction, no __syncthreads inside a step, the gates come straight out of the accumulators (C layout: 4 rows x 1 unit
//     per lane); a "group" -- the unit of the hand-off -- is one row block x H/16 members, every wave polls the counter of
//     its own row block;
//   * the member's W_hh slice in LDS instead of registers, and
//   * the contraction on the bf16 matrix cores AT FP32 ACCURACY: every fp32 operand is split exactly into three bf16 pieces,
//     x = x0 + x1 + x2 (3 x 8 mantissa bits: nothing is lost), and a product a*b is the sum of the piece products a_i*b_j --
//     each exact in the f32 accumulation --, accumulated in f32 by v_mfma_f32_16x16x32_bf16.  NP = 9: all nine terms, i.e. the
//     products are exactly those of fp32 arithmetic and only the order of the f32 summation differs from the f32-input MFMA
//     (itself different from any CPU's); NP = 6 drops the three terms below 2^-24 |ab| (a1*b2, a2*b1, a2*b2: less than the
//     rounding of the f32 accumulation itself).  The bf16 MFMA issues 16x the MACs per cycle of the f32-input one, so the
//     matrix phase of a step shrinks from 5.7 us to 3.2 (NP = 9) / 2.1 us (NP = 6): 9.5 -> 7.7 / 6.6 us per step
//     (tools/exp_chain4.hip, profiles/r03_k_chain_bf16_split.txt).
// The W slice is split once per launch (into LDS: 3 pieces x 48 KB at H = 512), a state element once by the wave that
// produces it; the exchange carries the three pieces in the MFMA's A-fragment order (6 bytes per element instead of 4):
//   piece p of a [rows, K] state: [row/16][K/32][lane = (k%32)/8 * 16 + row%16][8 bf16 = k%8]   (1 KB per fragment)
// Semantics (operand sources, masks, saves, h0 / hlast / dh0, reverse, row chunks) are those of the first-generation kernels;
// tests run both against the oracle (INET_CHAIN2=0 selects the first generation).  The forward kernel is the default for chains of
// >= 6 steps; the second-generation BPTT kernel was removed in round 4: it was the faster kernel alone and the slower step, because a workgroup
// of it holds the CU's LDS and keeps the backward pass's leaf work out (gru_chain.hip gru_chain_bwd_is_v2).
#include <cstdio>
#include <cstdlib>
#include "chain.h"
#include "prof.h"
#include "gru_chain.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
using chain::u32x4;
#ifndef INET_CHAIN2_STAMPS
#define INET_CHAIN2_STAMPS 0      // 1: wave 0 of the first workgroup records six wall-clock stamps per step (tools/chain2_anatomy.py)
#endif
#define C2_STAMP(i) do { if (INET_CHAIN2_STAMPS && stamping && lane == 0 && step < 32) \
        reinterpret_cast<unsigned long long*>(status.gdev + kChainDiagWord)[step * 8 + (i)] = wall_clock64(); } while (0)

__device__ __forceinline__ void split3(float x, __bf16& a0, __bf16& a1, __bf16& a2) {
    a0 = (__bf16)x;                                  // round to nearest: |x - a0| <= 2^-9 |x|
    const float r1 = x - (float)a0;                  // exact
    a1 = (__bf16)r1;
    a2 = (__bf16)(r1 - (float)a1);                   // exact, and fits 8 bits
}

// 8 consecutive floats -> the three pieces; store them at byte offset `off` of each piece (piece stride `pb`)
__device__ __forceinline__ void publish8(__amdgpu_buffer_rsrc_t rs, int off, int pb, const float* src) {
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 a, b, c;
        split3(src[j], a, b, c);
        p0[j] = a; p1[j] = b; p2[j] = c;
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p0), rs, off, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p1), rs, off + pb, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p2), rs, off + 2 * pb, 0, 16);
}

__device__ __forceinline__ void pieces8(const float* src, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 a, b, c;
        split3(src[j], a, b, c);
        p0[j] = a; p1[j] = b; p2[j] = c;
    }
}
__device__ __forceinline__ void publish_pieces(__amdgpu_buffer_rsrc_t rs, int off, int pb, bf16x8 p0, bf16x8 p1, bf16x8 p2) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p0), rs, off, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p1), rs, off + pb, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p2), rs, off + 2 * pb, 0, 16);
}
// plain (cached) stores of the three pieces: outputs for kernels launched later (ChainEmit), not part of the hand-off
__device__ __forceinline__ void store_pieces(unsigned char* dst, long piece, bf16x8 p0, bf16x8 p1, bf16x8 p2) {
    *reinterpret_cast<bf16x8*>(dst) = p0;
    *reinterpret_cast<bf16x8*>(dst + piece) = p1;
    *reinterpret_cast<bf16x8*>(dst + 2 * piece) = p2;
}
// One 16 x 16 tile of a wave in the accumulator layout (lane (c, q): rows 4q .. 4q+3 of unit c) -> its half of the transposed
// fragment (unit block, 32-row m block): fragment lane (kg, unit c) holds rows 8 kg .. 8 kg + 7 of the m block; this wave's
// 16 rows are the m block's half `par`: dst_lane = the fragment + ((2 par + q / 2) * 16 + c) * 16 (meaningful for even q).
// Lanes of even q take the four rows of lane + 16 (q + 1) next to their own.
__device__ __forceinline__ void emit_cols(unsigned char* dst_lane, long piece, const float (&v)[4], int q) {
    float x[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[r] = v[r]; x[4 + r] = __shfl_down(v[r], 16, 64); }
    if (!(q & 1)) {
        bf16x8 p0, p1, p2;
        pieces8(x, p0, p1, p2);
        store_pieces(dst_lane, piece, p0, p1, p2);
    }
}

// acc[g] += A[row block, all K] x W_g^T for NG B-operands held as pieces in LDS: wl + ((p * NG + g) * S32 + s) * 1024
template <int NG, int S32, int NP>
__device__ __forceinline__ void contract2(f32x4 (&acc)[NG], const unsigned char* wl, __amdgpu_buffer_rsrc_t rs, int abase,
                                          int base, int pb, int lane) {
#ifndef INET_CHAIN2_RING
#define INET_CHAIN2_RING 3
#endif
    // k-steps of A fragments requested ahead of the MFMAs.  Build-time A/B (tools/ab_chain2_variants.sh, round 5, T24 launch in the
    // step): 2: 215-217 us, 3: 185-192, 4: 192-197 (rounds 3-4), 8: 203-206, 12: 210, 16: 220 -- the contraction is NOT short of loads
    // in flight; a deeper ring only lengthens the burst in front of the first MFMA.  Dealing the three loads of a k-step over its 27
    // MFMAs (INET_CHAIN2_DEAL=1) measures the same as the burst.
    constexpr int R = INET_CHAIN2_RING;
    bf16x8 Ar[R][3];
    auto ldA = [&](int s, int slot) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            Ar[slot][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, abase + s * 1024 + p * pb, base, 16));
    };
#pragma unroll
    for (int d = 0; d < R - 1; ++d)
        if (d < S32) ldA(d, d);
    __builtin_amdgcn_sched_barrier(0);
    // Consecutive MFMAs never share an accumulator: with NG >= 3 operands the piece products are issued operand-innermost
    // (distance NG between two updates of one accumulator); a single operand (the backward kernel) accumulates into three
    // partial sums, product k of a block into partial k % 3 (432 back-to-back dependent MFMAs measured 216 us per
    // 24-step launch against 187 for the forward kernel's three-gate form).
    constexpr int NPART = NG >= 3 ? 1 : 3;
    f32x4 part[NPART][NG];
#pragma unroll
    for (int k = 0; k < NPART; ++k)
#pragma unroll
        for (int g = 0; g < NG; ++g) part[k][g] = k == 0 ? acc[g] : f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int PI[9] = {0, 0, 1, 0, 1, 2, 1, 2, 2}, PJ[9] = {0, 1, 0, 2, 1, 0, 2, 1, 2};   // largest terms first; 6: the first six
#ifndef INET_CHAIN2_DEAL
#define INET_CHAIN2_DEAL 0
#endif
#ifndef INET_CHAIN2_BPF
#define INET_CHAIN2_BPF 0
#endif
    // B fragments (the member's W pieces, LDS): nine 16-byte reads per lane and k-step.  The sched_barrier that pins the A loads also
    // keeps hipcc from lifting the next k-step's LDS reads over this k-step's MFMAs, so every k-step started with an exposed LDS round
    // trip (16 per contraction).  INET_CHAIN2_BPF=1 (round 5, build-time A/B): the reads of k-step s + 1 are issued in front of the MFMAs of
    // k-step s (two register sets of nine fragments) -- measured NEUTRAL (188 vs 188 us per T24 launch, three alternations on one box):
    // the LDS round trip is not what the contraction waits for either.
    bf16x8 Bf[2][NG][3];
    auto ldB = [&](int s, int slot) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int p = 0; p < 3; ++p) Bf[slot][g][p] = *reinterpret_cast<const bf16x8*>(wl + ((p * NG + g) * S32 + s) * 1024 + lane * 16);
    };
    if (INET_CHAIN2_BPF) ldB(0, 0);
#pragma unroll
    for (int s = 0; s < S32; ++s) {
        if (!INET_CHAIN2_DEAL && s + R - 1 < S32) ldA(s + R - 1, (s + R - 1) % R);
        if (INET_CHAIN2_BPF) { if (s + 1 < S32) ldB(s + 1, (s + 1) & 1); }
        else ldB(s, s & 1);
        const bf16x8* a = Ar[s % R];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            // (build-time A/B: the three loads of the k-step R - 1 ahead dealt over this k-step's MFMAs, one in front of every third
            //  of them and pinned there, instead of as one burst in front of the first: measures the same)
            if (INET_CHAIN2_DEAL && k % (NP / 3) == 0 && k / (NP / 3) < 3 && s + R - 1 < S32) {
                const int p = k / (NP / 3), sn = s + R - 1;
                Ar[sn % R][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, abase + sn * 1024 + p * pb, base, 16));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g)
                part[k % NPART][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PI[k]], Bf[s & 1][g][PJ[k]], part[k % NPART][g], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        acc[g] = part[0][g];
#pragma unroll
        for (int k = 1; k < NPART; ++k) acc[g] += part[k][g];
    }
}

// bounded poll of one row block's counter by a whole wave (all lanes read the same word: one request)
__device__ __forceinline__ bool wait_rows(unsigned* counter, unsigned target, chain::Status status, int lane) {
    unsigned spins = 0;
    bool ok = true;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > chain::kSpinLimit ||
            ((spins & 63) == 0 && __hip_atomic_load(status.dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != chain::ST_OK)) {
            ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(INET_CHAIN_POLL_SLEEP);
    }
    if (spins >= chain::kSlowSpins && lane == 0) {              // (a wait that gave up has polled at least kSlowSpins times)
        if (!ok) chain::raise_timeout(status);
        chain::record_slow<chain::K_GRU2_FWD>(status, 2u, target, spins, !ok);
    }
    return ok;
}

__device__ __forceinline__ void arrive_rows(unsigned* counter, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's write-through stores are out
    if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward:  h_t = GRUCell(gi_t, h_{t-1})
// ---------------------------------------------------------------------------------------------------------------------------
// EM: the build that writes the ChainEmit piece outputs (four transpose tiles per wave: 144 + 16 KB = all of the LDS at H = 512);
// launches without piece outputs run the lean build (one tile, no descriptor in registers)
template <int WV, int S32, int NP, bool EM>          // WV waves (= row blocks) per workgroup; S32 = H / 32
__global__ __launch_bounds__(64 * WV) void gru_chain2_fwd_kernel(GruChainFwd A) {
    constexpr int H = 32 * S32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wl = smem;                             // [3 pieces][3 gates][S32][64][16 B]
    constexpr int NTILE = EM ? 4 : 1;                           // transpose tiles per wave
    float* const xt = reinterpret_cast<float*>(smem + 3 * 3 * S32 * 1024);   // [WV][NTILE][256]
    int group, member;
    chain::decode_block(blockIdx.x, A.members, group, member);
    if (group >= A.nprob * A.tiles_per_prob) return;
    if (A.fault && blockIdx.x == 0) return;                    // injected fault (test hook)
    if (A.prio) __builtin_amdgcn_s_setprio(3);
    const int prob = group / A.tiles_per_prob, tile = group % A.tiles_per_prob;
    const GruChainFwdProb& P = A.p[prob];
    const int B = A.B, T = A.T;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int j0 = member * 16, jc = j0 + c;
    const int nrb = (B + 15) >> 4;
    const int rb = tile * WV + w;
    // W slice -> three bf16 pieces in B-fragment order: fragment (g, s): lane (unit = lane % 16, k group = lane / 16) holds
    // W[g*H + j0 + unit][32 s + 8 (lane / 16) + 0..7]
    for (int i = t; i < 3 * S32 * 64; i += 64 * WV) {
        const int ln = i & 63, s = (i >> 6) % S32, g = i / (64 * S32);
        const float* src = P.W_hh + (long)(g * H + j0 + (ln & 15)) * H + 32 * s + 8 * (ln >> 4);
        const f32x4 x0 = ld4u(src), x1 = ld4u(src + 4);
        bf16x8 p0, p1, p2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 a, b, cc;
            split3(j < 4 ? x0[j & 3] : x1[j & 3], a, b, cc);
            p0[j] = a; p1[j] = b; p2[j] = cc;
        }
        *reinterpret_cast<bf16x8*>(wl + ((0 * 3 + g) * S32 + s) * 1024 + ln * 16) = p0;
        *reinterpret_cast<bf16x8*>(wl + ((1 * 3 + g) * S32 + s) * 1024 + ln * 16) = p1;
        *reinterpret_cast<bf16x8*>(wl + ((2 * 3 + g) * S32 + s) * 1024 + ln * 16) = p2;
    }
    __syncthreads();
    if (rb >= nrb) return;                                     // (a ragged last tile: this wave has no row block)
    const int pb = nrb * 16 * H * 2;                           // bytes of one piece of the [B,H] state
    const int slot_bytes = 3 * pb;
    const __amdgpu_buffer_rsrc_t rs = chain::make_rsrc(P.hx);
    unsigned* counter = A.counters + (prob * nrb + rb) * kChainCounterStride;
    const chain::Status status = A.status;
    float* myxt = xt + w * 256 * NTILE;
    const int abase = (rb * S32 * 64 + lane) * 16;
    // this wave's 16 columns inside a fragment: k block member / 2, k groups 2 (member % 2) + {0, 1}
    const int pub_off = ((rb * S32 + (member >> 1)) * 64 + (2 * (member & 1) + ((lane >> 4) & 1)) * 16 + (lane & 15)) * 16;
    int brow[4];
    float hp[4];
    const bool has_h0 = P.h0 != nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        brow[r] = min(rb * 16 + 4 * q + r, B - 1);
        hp[r] = has_h0 ? P.h0[(long)brow[r] * P.ld_h0 + jc] : 0.f;
    }
    const bool publish_h0 = has_h0 && !A.h0_packed;
    if (publish_h0) {                                          // the initial state enters the exchange like any later one
#pragma unroll
        for (int r = 0; r < 4; ++r) myxt[(4 * q + r) * 16 + c] = hp[r];
        __builtin_amdgcn_wave_barrier();
        if (lane < 32) publish8(rs, slot_bytes + pub_off, pb, myxt + (lane & 15) * 16 + 8 * (lane >> 4));
        __builtin_amdgcn_wave_barrier();
        arrive_rows(counter, lane);
    }
    const int arrivals0 = publish_h0 ? 1 : 0;
    float bh[3], bv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = P.b_hh[g * H + jc];
    if (P.gi_vec) {
#pragma unroll
        for (int g = 0; g < 3; ++g) bv[g] = P.gi_vec[g * H + jc];
    }
    // operand sources in registers, absent ones aimed at a zero word (gru_chain.hip)
    const float* const zf = reinterpret_cast<const float*>(A.counters + kChainZeroWord);
    const bool has_tab = P.gi_table != nullptr, has_den = P.gi_dense != nullptr, has_mask = P.outm && P.mask;
    const long long* const idxp = has_tab ? P.idx : reinterpret_cast<const long long*>(zf);
    const int idx_bs = has_tab ? (int)P.idx_bs : 0, idx_ts = has_tab ? (int)P.idx_ts : 0;
    const float* const tabp = has_tab ? P.gi_table : zf;
    const int tab_ld = has_tab ? (int)P.ld_table : 0, tab_g = has_tab ? H : 0, tab_j = has_tab ? jc : 0;
    const float* const denp = has_den ? P.gi_dense : zf;
    const int den_ld = has_den ? (int)P.ld_gi : 0, den_ts = has_den ? (int)P.ts_gi : 0, den_g = has_den ? H : 0, den_j = has_den ? jc : 0;
    const float* const mskp = has_mask ? P.mask : zf;
    const int msk_ld = has_mask ? (int)P.ld_mask : 0, msk_ts = has_mask ? (int)P.ts_mask : 0, msk_j = has_mask ? jc : 0;
    float* const outp = P.out; const int out_ld = (int)P.ld_out, out_ts = (int)P.ts_out;
    float* const outmp = P.outm; const int outm_ld = (int)P.ld_outm, outm_ts = (int)P.ts_outm;
    float* const svp = P.sv; const int sv_as = (int)P.sv_astride, sv_ts = P.sv_ts ? (int)P.sv_ts : B * H;
    float* const hlastp = P.hlast; const int hlast_ld = (int)P.ld_hlast;
    const int rev = P.reverse, members = A.members;
    // piece outputs (ChainEmit): the descriptor in SGPRs before the loop (kernarg reads sink to their first use otherwise)
    // The descriptor is folded into one destination pointer per lane and layout (time step 0) plus two byte strides per time
    // step: kept as 13 scalars it spilled 27 SGPRs in this kernel.
    const bool em_rows = EM && P.em.rows, em_colsA = EM && P.em.colsA, em_colsB = EM && P.em.colsB;   // (the lean build writes no piece outputs)
    const long em_rows_piece = P.em.rows_piece, em_colsA_piece = P.em.colsA_piece, em_colsB_piece = P.em.colsB_piece;
    const int em_b16 = P.em.B_full >> 4, em_kbm = T * (P.em.B_full >> 5);
    const int rbg = (P.em.r0 >> 4) + rb;                       // row block within the full batch
    const int em_rows_tstride = em_b16 * P.em.rows_kb * 1024, em_cols_tstride = (em_b16 >> 1) * 1024;
    unsigned char* const em_rows_lane = !em_rows ? nullptr :   // (lanes 0..31: row lane % 16, k half lane / 16 of this member's 16 columns)
        P.em.rows + (((long)rbg * P.em.rows_kb + P.em.rows_kb0 + (member >> 1)) * 64 + (2 * (member & 1) + ((lane >> 4) & 1)) * 16 + (lane & 15)) * 16;
    const int em_cl = ((2 * (rbg & 1) + (q >> 1)) * 16 + c) * 16;   // this lane inside a transposed fragment
    unsigned char* const em_colsA_lane = !em_colsA ? nullptr : P.em.colsA + ((long)(P.em.colsA_rb0 + member) * em_kbm + (rbg >> 1)) * 1024 + em_cl;
    unsigned char* const em_colsB_lane = !em_colsB ? nullptr : P.em.colsB + ((long)(P.em.colsB_rb0 + member) * em_kbm + (rbg >> 1)) * 1024 + em_cl;
    long tok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tok[r] = idxp[brow[r] * idx_bs + (rev ? T - 1 : 0) * idx_ts];
    const bool stamping = INET_CHAIN2_STAMPS && group == 0 && member == 0 && w == 0 && status.gdev != nullptr && T >= 12;
    for (int step = 0; step < T; ++step) {
        const int tt = rev ? T - 1 - step : step;
        const int tn = step + 1 < T ? (rev ? tt - 1 : tt + 1) : tt;
        C2_STAMP(0);
        float pgt[4][3], pgd[4][3], pm[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = brow[r];
#pragma unroll
            for (int g = 0; g < 3; ++g) pgt[r][g] = tabp[(int)tok[r] * tab_ld + g * tab_g + tab_j];
#pragma unroll
            for (int g = 0; g < 3; ++g) pgd[r][g] = denp[tt * den_ts + b * den_ld + g * den_g + den_j];
            pm[r] = mskp[tt * msk_ts + b * msk_ld + msk_j];
            tok[r] = idxp[b * idx_bs + tn * idx_ts];
        }
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (step > 0 || has_h0) {
            if ((step > 0 || publish_h0) && !wait_rows(counter, (unsigned)((step + arrivals0) * members), status, lane)) return;
            C2_STAMP(1);
            contract2<3, S32, NP>(acc, wl, rs, abase, ((step + 1) & 1) * slot_bytes, pb, lane);
            if (INET_CHAIN2_STAMPS && stamping) asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]));   // (the stamp behind the MFMAs)
            C2_STAMP(2);
        }
        float er[4], ez[4], en[4], eg[4], eh[4], ehp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ghn = acc[2][r] + bh[2];
            const float rr = sigmoid_f(acc[0][r] + pgd[r][0] + pgt[r][0] + bv[0] + bh[0]);
            const float z = sigmoid_f(acc[1][r] + pgd[r][1] + pgt[r][1] + bv[1] + bh[1]);
            const float n = tanh_f(pgd[r][2] + pgt[r][2] + bv[2] + rr * ghn);
            const float hprev = hp[r];
            const float hn = (1.f - z) * n + z * hprev;
            hp[r] = hn;
            myxt[(4 * q + r) * 16 + c] = hn;
            er[r] = rr; ez[r] = z; en[r] = n; eg[r] = ghn; eh[r] = hn; ehp[r] = hprev;
        }
        float em_v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) em_v[r] = has_mask ? eh[r] * pm[r] : eh[r];
        if (!EM) {
            C2_STAMP(3);
            if (step != T - 1) {                               // nobody reads the last state from the exchange
                __builtin_amdgcn_wave_barrier();               // (the tile is exchanged between lanes: see the backward kernel)
                if (lane < 32) publish8(rs, (step & 1) * slot_bytes + pub_off, pb, myxt + (lane & 15) * 16 + 8 * (lane >> 4));
                __builtin_amdgcn_wave_barrier();
                arrive_rows(counter, lane);
            }
            C2_STAMP(4);
        } else {
            // One pass through the wave's transpose tiles: tile 0 = the new state (the exchange's next operand), tile 1 = the
            // masked state (the row pieces of the layer's output).  The exchange stores go first and the arrival right behind
            // them (its vmcnt(0) then only waits for those); the piece outputs follow, re-read from the tiles.
            const bool rows_masked = em_rows && has_mask;
            C2_STAMP(3);
            if (step != T - 1) {
                __builtin_amdgcn_wave_barrier();
                if (lane < 32) publish8(rs, (step & 1) * slot_bytes + pub_off, pb, myxt + (lane & 15) * 16 + 8 * (lane >> 4));
                arrive_rows(counter, lane);
            }
            C2_STAMP(4);
            if (em_rows) {                                     // (everything below is behind the hand-off)
                if (rows_masked) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) myxt[256 + (4 * q + r) * 16 + c] = em_v[r];
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < 32) {
                    bf16x8 p0, p1, p2;
                    pieces8(myxt + (rows_masked ? 256 : 0) + (lane & 15) * 16 + 8 * (lane >> 4), p0, p1, p2);
                    store_pieces(em_rows_lane + (long)tt * em_rows_tstride, em_rows_piece, p0, p1, p2);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (em_colsA) emit_cols(em_colsA_lane + (long)tt * em_cols_tstride, em_colsA_piece, em_v, q);
        if (em_colsB) emit_cols(em_colsB_lane + (long)tt * em_cols_tstride, em_colsB_piece, ehp, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = rb * 16 + 4 * q + r;
            if (b < B) {
                outp[tt * out_ts + b * out_ld + jc] = eh[r];
                if (outmp) outmp[tt * outm_ts + b * outm_ld + jc] = has_mask ? eh[r] * pm[r] : eh[r];
                if (hlastp && step == T - 1) hlastp[b * hlast_ld + jc] = eh[r];
                if (svp) {
                    float* sp = svp + tt * sv_ts + b * H + jc;
                    sp[0] = er[r]; sp[sv_as] = ez[r]; sp[2 * sv_as] = en[r]; sp[3 * sv_as] = eg[r]; sp[4 * sv_as] = ehp[r];
                }
            }
        }
        C2_STAMP(5);
    }
}

int g_chain2 = -1;   // piece products per fp32 product: 0 = first-generation kernels, 9 = second generation

}  // namespace

int chain2_mode() {
    if (g_chain2 < 0) {
        const char* e = std::getenv("INET_CHAIN2");
        const int v = e ? std::atoi(e) : 9;
        g_chain2 = v == 9 ? 9 : 0;                           // (round 4: the six-product builds are gone; anything else = first generation)
    }
    return g_chain2;
}
void chain2_set_mode(int np) { g_chain2 = np == 9 ? 9 : 0; }

// waves per workgroup (4 or 8) with which the launch fits the chip and the sync area, or 0
static int chain2_waves(int H, int B, int T, int nprob) {
    // (T >= 6: a launch first splits its 96 KB W slice into 144 KB of bf16 pieces in LDS; over the beat GRU's four steps that
    //  costs more than the faster steps give back: 31 -> 43 us per launch)
    if (!chain_enabled() || chain2_mode() == 0 || (H != 256 && H != 512) || T < 6 || nprob < 1 || nprob > 4 || B < 1) return 0;
    if ((double)T * B * 6.0 * H >= 2.0e9) return 0;
    const int nrb = (B + 15) / 16;
    if (nprob * nrb > kChainMaxGroups) return 0;               // one counter per (problem, row block)
    // Four waves per workgroup (one per SIMD).  (Eight -- two row blocks per SIMD -- fit the decoder's four-beat tick BPTT into one
    // launch, but two waves of a SIMD do not overlap: 128 us against 127 for the first generation's two-tiles-per-workgroup form;
    // that build was removed in round 4, those shapes stay on the first generation.)
    return nprob * ((nrb + 3) / 4) * (H / 16) <= chain_capacity() ? 4 : 0;
}
bool gru_chain2_ok(int H, int B, int T, int nprob) { return chain2_waves(H, B, T, nprob) > 0; }
bool gru_chain2_emits(int H, int B, int T, int nprob) { return chain2_waves(H, B, T, nprob) == 4; }

int launch_gru_chain2_fwd(GruChainFwd a, hipStream_t s) {
    const int wv = chain2_waves(a.H, a.B, a.T, a.nprob);
    if (!wv) return -1;
    const int nrb = (a.B + 15) / 16;
    a.tiles_per_prob = (nrb + wv - 1) / wv;
    a.members = a.H / 16;
    const int groups = a.nprob * a.tiles_per_prob;
    a.prio = 1;
    a.fault = chain_take_fault();
    if (!a.prezeroed && hipMemsetAsync(a.counters, 0, kChainSyncWords * sizeof(unsigned), s) != hipSuccess) return -2;
    a.status = chain_status_for(a.counters + kChainStatusWord);
    const int np = chain2_mode();
    const double rows = (double)a.nprob * a.T * a.B;
    bool em = false;                                           // piece outputs wanted (four-wave build only): the EM build
    for (int i = 0; i < a.nprob; ++i) em = em || a.p[i].em.rows || a.p[i].em.colsA || a.p[i].em.colsB;
    if (wv != 4) { em = false; for (int i = 0; i < a.nprob; ++i) a.p[i].em = ChainEmit{}; }
    char label[72];                                            // "v2w4e": the build that writes piece outputs
    std::snprintf(label, sizeof label, "gru_chain_fwd v2w%d%s p%d np%d T%d B%d H%d", wv, em ? "e" : "", np, a.nprob, a.T, a.B, a.H);
    double em_bytes = 0.0;                                     // piece outputs: 6 bytes per element and layout
    for (int i = 0; i < a.nprob; ++i)
        em_bytes += 6.0 * a.T * a.B * a.H * ((a.p[i].em.rows ? 1 : 0) + (a.p[i].em.colsA ? 1 : 0) + (a.p[i].em.colsB ? 1 : 0));
    ProfScope prof(PROF_GRU_FWD, 2.0 * rows * 3.0 * a.H * a.H, s, label,
                   4.0 * (a.nprob * 3.0 * a.H * a.H + rows * a.H * (2 + 3 + (a.p[0].sv ? 5 : 0))) + em_bytes);
    const dim3 grid(chain::blocks_for(groups, a.members));
    const size_t lds = (size_t)3 * 3 * (a.H / 32) * 1024 + (size_t)wv * 256 * 4 * (em ? 4 : 1);
#define DISPATCH_C2F_(W, S, N, E)                                                                                           \
    do {                                                                                                                \
        static bool attr = false;                                                                                       \
        if (!attr) {                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_chain2_fwd_kernel<W, S, N, E>),                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                          \
            attr = true;                                                                                                \
        }                                                                                                               \
        hipLaunchKernelGGL((gru_chain2_fwd_kernel<W, S, N, E>), grid, dim3(64 * W), lds, s, a);                         \
    } while (0)
#define DISPATCH_C2F(W, S, N)                                                                                               \
    do {                                                                                                                \
        if (em) DISPATCH_C2F_(4, S, N, true);                                                                               \
        else DISPATCH_C2F_(4, S, N, false);                                                                                 \
    } while (0)
    if (np != 9 || wv != 4) return -1;
    if (a.H == 512) DISPATCH_C2F(4, 16, 9);
    else DISPATCH_C2F(4, 8, 9);
#undef DISPATCH_C2F
#undef DISPATCH_C2F_
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
