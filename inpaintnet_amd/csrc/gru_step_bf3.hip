// GRU forward steps for LARGE batches on the bf16 matrix cores at fp32 accuracy (round 4).
//
// A chain launch (gru_chain2.hip) holds at most one 256-row chunk per direction on the chip and every step of it is hand-off
// latency plus the MFMAs of one workgroup: 0.37 of the bf16 roof / 9.  LatentRNN's frozen encoder runs 2048 measures per
// step and the reference's default MeasureVAE batch is 4096: there the rows of ONE time step already fill the chip, so the
// step is a plain product  [B rows] x [3H gate columns] x [H]  with the GRU cell as its epilogue -- gemm_bf3.hip's main
// loop (both operands as three exact bf16 pieces per f32 value in MFMA fragment order, LDS-DMA operand stream, nine piece
// products accumulated in f32 = the products of fp32 arithmetic) and one launch per time step:
//   * A = the pieces of h_{t-1}, written by the previous step's epilogue straight into a two-slot piece ring (the wave
//     that produces a value splits it, as the chain kernels do): no f32 round trip, no split pass;
//   * B = the pieces of W_hh with its row blocks interleaved (unit block u, gate g) -> row block 3u + g, so that a wave's
//     three accumulator tiles are the r, z, n pre-activations of the SAME 16 rows x 16 units and the cell is computed by
//     the lane that holds them -- no exchange between waves;
//   * tile 128 rows x 64 units (x 3 gates): B = 2048, H = 512, two directions = 256 workgroups, one per CU;
//   * the epilogue also writes, where asked, the dropout-masked output, the five backward saves, the final state, and the
//     row pieces of the (masked) output for the layer-1 input product (ChainEmit.rows).
// Same arithmetic as gru_chain2_fwd_kernel (tests compare both with the oracle at B = 2048).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "prof.h"
#include "gemm_bf3.h"
#include "gru_step_bf3.h"
#include "seq.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int RN = 3;                                   // a wave's three accumulator tiles per row block: the gates r, z, n of 16 units
// tiling: WM x WN waves, RM row blocks per wave: 2 x 4 x 4 = 128 rows x 64 units, 8 waves, 120 KB of LDS, one workgroup per CU

struct StepProb {
    const unsigned char* A; unsigned char* An; long a_piece;   // pieces of h_{t-1} / h_t: [B/16][H/32] fragments per piece
    const unsigned char* W; long w_piece;                        // pieces of W_hh, row block 3u + g
    const float* b_hh;
    const float* gi; long gi_ld;                                 // dense input-side pre-activations at time t, or null
    const float* table; long table_ld; const long long* idx; long idx_bs;   // gathered rows (idx already at time t), or null
    const float* gvec;                                           // [3H] broadcast, or null
    const float* hprev; long hprev_ld;                           // f32 h_{t-1} (null: zeros)
    float* out; long out_ld;
    float* outm; long outm_ld; const float* mask; long mask_ld;
    float* hlast; long hlast_ld;
    float* sv; long sv_astride;                                  // at time t (row stride H), or null
    unsigned char* em; long em_piece; int em_kb, em_kb0; long em_rb0;   // row pieces of the (masked) output, or null
};
struct StepArgs { int H, B, nprob, zero_state; StepProb p[2]; };

__device__ __forceinline__ void pieces8_store(const float* src, unsigned char* dst, long piece) {
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = src[j];
        const __bf16 a = (__bf16)x;                    // |h| <= 1: no overflow case here (gemm_bf3.hip split3)
        const float r1 = x - (float)a;
        const __bf16 b = (__bf16)r1;
        p0[j] = a; p1[j] = b; p2[j] = (__bf16)(r1 - (float)b);
    }
    *reinterpret_cast<bf16x8*>(dst) = p0;
    *reinterpret_cast<bf16x8*>(dst + piece) = p1;
    *reinterpret_cast<bf16x8*>(dst + 2 * piece) = p2;
}

// Measured and not kept (one-box A/Bs, B = 2048, H = 512, 50 us per step either way): the epilogue's operand loads dealt over the
// unrolled k blocks instead of requested up front; 64 x 32 tiles with two workgroups per CU so that one's prologue / epilogue
// runs under the other's MFMAs (52.6 vs 50.6 us); an XCD owning 8 x 4 or 16 x 2 tiles.  The k loop sits on the MFMA issue
// ceiling (~17 clocks per v_mfma_f32_16x16x32_bf16 and SIMD at the 1.7 GHz the chip sustains under this load: 34.5 us for the
// nine piece products of a 128 x 192 x 512 tile); a launch without a k loop (step 0 of a zero initial state) takes 15 us.
template <bool TAB, bool DEN, bool SAVE, int WM, int WN, int RM>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN <= 4 ? 2 : 1)) void gru_step_bf3_kernel(StepArgs a) {
    constexpr int TMB = WM * RM, TNB = WN * RN, NW = WM * WN;
    constexpr int STAGE = (TMB + TNB) * 3 * 1024, CH = (TMB + TNB) * 3, CPW = (CH + NW - 1) / NW;
    constexpr int UT = 16 * WN;                                    // hidden units per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w / WN, wn = w % WN;
    const int H = a.H, KB = H >> 5;
    const int tiles_m = a.B / (TMB * 16), tiles_n = H / UT;
    // consecutive workgroup ids go round-robin over the 8 XCDs: each XCD takes a contiguous range of the (problem, tm, tn)
    // list -- 4 row tiles x all 8 unit tiles of one direction at B = 2048: 1.5 MB of h pieces + that direction's 4.7 MB of W
    // (an XCD owning 8 x 4 or 16 x 2 tiles instead measured the same, 50.2-50.4 us per step)
    const int nb = gridDim.x, id = blockIdx.x;
    const int tid = (nb % 8 == 0) ? (id % 8) * (nb / 8) + id / 8 : id;
    const int per_prob = tiles_m * tiles_n;
    const int prob = tid / per_prob, v = tid - prob * per_prob;
    const int tm = v / tiles_n, tn = v - tm * tiles_n;
    const StepProb& P = a.p[prob];
    const int c = lane & 15, q = lane >> 4;
    const int j0 = tn * UT + wn * 16, j = j0 + c;              // this wave's 16 hidden units; this lane's

    // ---- epilogue operands: branch-free sources (an absent one aims at a valid word with zero strides and is replaced by a
    // select -- a conditional load costs a branch and a full vmcnt(0) each) ----
    const bool has_hp = P.hprev != nullptr, has_mk = P.mask != nullptr;
    const float* const hpp = has_hp ? P.hprev + j : P.b_hh;
    const long hp_ld = has_hp ? P.hprev_ld : 0;
    const float* const mkp = has_mk ? P.mask + j : P.b_hh;
    const long mk_ld = has_mk ? P.mask_ld : 0;
    const long row00 = (long)(tm * TMB + wm * RM) * 16 + 4 * q;            // row of (i = 0, r = 0); (i, r) adds 16 i + r
    long tok[RM][4];
    if (TAB) {
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) tok[i][r] = P.idx[(row00 + 16 * i + r) * P.idx_bs];
    }
    float gr[RM][4], gz[RM][4], gn[RM][4], hp[RM][4], mk[RM][4];
    auto load_elem = [&](int i, int r) {
        const long row = row00 + 16 * i + r;
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        if (DEN) {
            const float* dp = P.gi + row * P.gi_ld + j;
            x0 = dp[0]; x1 = dp[H]; x2 = dp[2 * H];
        }
        if (TAB) {
            const float* tp = P.table + tok[i][r] * P.table_ld + j;
            x0 += tp[0]; x1 += tp[H]; x2 += tp[2 * H];
        }
        gr[i][r] = x0; gz[i][r] = x1; gn[i][r] = x2;
        const float hv = hpp[row * hp_ld], mv = mkp[row * mk_ld];
        hp[i][r] = has_hp ? hv : 0.f;
        mk[i][r] = has_mk ? mv : 1.f;
    };
    float bh[3], bv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = P.b_hh[g * H + j];
    if (P.gvec) {
#pragma unroll
        for (int g = 0; g < 3; ++g) bv[g] = P.gvec[g * H + j];
    }

    f32x4 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int g = 0; g < RN; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) load_elem(i, r);
    if (!a.zero_state) {
        // fragment c of a stage: c < 3 TMB: A piece c / TMB, row block c % TMB; then W alike (gemm_bf3_kernel's operand stream)
        const unsigned char* gsrc[CPW]; int loff[CPW];
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int cc0 = w + i * NW;
            const int cc = cc0 < CH ? cc0 : CH - 1;
            if (cc < TMB * 3) {
                const int p = cc / TMB, rbl = cc % TMB;
                gsrc[i] = P.A + p * P.a_piece + ((long)(tm * TMB + rbl) * KB) * 1024 + lane * 16;
            } else {
                const int c2 = cc - TMB * 3, p = c2 / TNB, rbl = c2 % TNB;
                gsrc[i] = P.W + p * P.w_piece + ((long)(tn * TNB + rbl) * KB) * 1024 + lane * 16;
            }
            loff[i] = cc * 1024;
        }
        auto fill = [&](int kb, unsigned char* stage) {
#pragma unroll
            for (int i = 0; i < CPW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[i] + (long)kb * 1024),
                                                 (__attribute__((address_space(3))) void*)(stage + loff[i]), 16, 0, 0);
        };
        auto block = [&](int kb) {
            const unsigned char* sa = smem + (kb & 1) * STAGE + lane * 16;
            const unsigned char* sb = sa + TMB * 3 * 1024;
            bf16x8 Af[RM][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < RM; ++i) Af[i][p] = *reinterpret_cast<const bf16x8*>(sa + (p * TMB + wm * RM + i) * 1024);
#pragma unroll
            for (int pj = 0; pj < 3; ++pj) {
                bf16x8 Bf[RN];
#pragma unroll
                for (int g = 0; g < RN; ++g) Bf[g] = *reinterpret_cast<const bf16x8*>(sb + (pj * TNB + wn * RN + g) * 1024);
#pragma unroll
                for (int pi = 0; pi < 3; ++pi)
#pragma unroll
                    for (int i = 0; i < RM; ++i)
#pragma unroll
                        for (int g = 0; g < RN; ++g)
                            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[i][pi], Bf[g], acc[i][g], 0, 0, 0);
            }
        };
        fill(0, smem);
        __syncthreads();
        for (int kb = 0; kb < KB; ++kb) {
            if (kb + 1 < KB) fill(kb + 1, smem + ((kb + 1) & 1) * STAGE);
            block(kb);
            __syncthreads();                                       // (carries the vmcnt(0) of this wave's fills)
        }
    }

    // ---- GRU cell: lane (c, q) holds rows 4q + r, unit c of every 16 x 16 tile; tiles g = 0, 1, 2 are the gates r, z, n ----
    // Every result leaves through wave-private 16 x 16 transpose tiles in LDS (the stages are free now), two row blocks at a
    // time: the f32 arrays as ONE 16-byte store per lane and tile (a whole 1 KB tile per instruction -- scalar stores of the
    // accumulator layout are store-issue-bound: 16 per lane and array), the pieces as 8 consecutive units per lane with the two
    // halves of the wave on the two row blocks.
    constexpr int NT = SAVE ? 7 : 2;                                   // tiles per row block: h, h * mask [, r, z, n, ghn, hprev]
    float* const xt = reinterpret_cast<float*>(smem) + w * (2 * NT * 256);
    const int kbj = j0 >> 5;                                          // k block of the piece layouts this wave's 16 units fall into
    const int prow = lane & 15, pgrp = (lane >> 4) & 1, phalf = lane >> 5;
    const int plane = ((2 * ((j0 >> 4) & 1) + pgrp) * 16 + prow) * 16; // byte offset of (row, 8-unit group) inside the fragment
    const int vrow = lane >> 2, vcol = (lane & 3) * 4;                // the lane's 4 consecutive units of one row (vector stores)
#pragma unroll
    for (int ip = 0; ip < RM; ip += 2) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = ip + ii < RM ? ip + ii : RM - 1;         // (odd RM -- the 96-row tile --: the last pair's second half is idle)
            if (ip + ii >= RM) break;
            float* const x = xt + ii * (NT * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ghn = acc[i][2][r] + bh[2];
                const float rg = sigmoid_f(acc[i][0][r] + gr[i][r] + bv[0] + bh[0]);
                const float zg = sigmoid_f(acc[i][1][r] + gz[i][r] + bv[1] + bh[1]);
                const float ng = tanh_f(gn[i][r] + bv[2] + rg * ghn);
                const float hprev = hp[i][r];
                const float hn = (1.f - zg) * ng + zg * hprev;
                const int o = (4 * q + r) * 16 + c;
                x[o] = hn; x[256 + o] = hn * mk[i][r];
                if (SAVE) { x[512 + o] = rg; x[768 + o] = zg; x[1024 + o] = ng; x[1280 + o] = ghn; x[1536 + o] = hprev; }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            if (ip + ii >= RM) break;
            const float* const x = xt + ii * (NT * 256);
            const long vr = (long)(tm * TMB + wm * RM + ip + ii) * 16 + vrow;
            const f32x4 hv4 = *reinterpret_cast<const f32x4*>(x + vrow * 16 + vcol);
            *reinterpret_cast<f32x4*>(P.out + vr * P.out_ld + j0 + vcol) = hv4;
            if (P.outm) *reinterpret_cast<f32x4*>(P.outm + vr * P.outm_ld + j0 + vcol) = *reinterpret_cast<const f32x4*>(x + 256 + vrow * 16 + vcol);
            if (P.hlast) *reinterpret_cast<f32x4*>(P.hlast + vr * P.hlast_ld + j0 + vcol) = hv4;
            if (SAVE) {
#pragma unroll
                for (int a5 = 0; a5 < 5; ++a5)
                    *reinterpret_cast<f32x4*>(P.sv + a5 * P.sv_astride + vr * H + j0 + vcol) =
                        *reinterpret_cast<const f32x4*>(x + (2 + a5) * 256 + vrow * 16 + vcol);
            }
        }
        if (ip + phalf < RM) {
            const float* const x = xt + phalf * (NT * 256);
            const long rb = tm * TMB + wm * RM + ip + phalf;
            if (P.An) pieces8_store(x + prow * 16 + 8 * pgrp, P.An + (rb * KB + kbj) * 1024 + plane, P.a_piece);
            if (P.em) pieces8_store(x + 256 + prow * 16 + 8 * pgrp,
                                    P.em + ((P.em_rb0 + rb) * P.em_kb + P.em_kb0 + kbj) * 1024 + plane, P.em_piece);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward steps of the same layers (round 4): dh_t = dgh_{t+1} W_hh + dh_{t+1} z_{t+1} + dout_t, then the gate derivatives.
// One launch per processed step: A = the pieces of dgh of the step processed before (K = 3H), B = the pieces of W_hh^T
// (k-major split of W_hh: no interleaving -- one accumulator tile per 16 units), tile 128 rows x 128 units.  The epilogue
// transposes the accumulators through LDS so that a lane owns four consecutive units of one row: the five saves, dout and
// dh z arrive as 16-byte loads, dgi / dgh / dh z leave as 16-byte stores, and the three gate tiles go back through LDS to
// become the next step's A pieces (32 units of one gate = exactly one k block: one 1 KB fragment per wave instruction).
// Bias gradients are column sums of dgi / dgh taken once after the last step (pw_colsum_multi), not per-step atomics.
// ---------------------------------------------------------------------------------------------------------------------------
struct BStepProb {
    const unsigned char* A; unsigned char* An; long a_piece;   // pieces of dgh (next / this step): [B/16][3H/32] fragments per piece
    const unsigned char* W; long w_piece;                        // pieces of W_hh^T: [H/16][3H/32]
    const float* sv; long sv_astride;                            // saves at this time step (row stride H)
    const float* dout; long dout_ld;                             // external gradient into h_t, or null
    const float* dhn; long dhn_ld;                               // gradient into the final state (first processed step), or null
    const float* dhz_in; float* dhz_out;                         // dh z of the step processed before / of this one, [B,H]
    float* dgi; long dgi_ld;                                     // input-side gate gradients at this time step
    float* dgh;                                                  // recurrent-side gate gradients at this time step, [B,3H]
    unsigned char* em; long em_piece; int em_kb, em_kb0; long em_rb0;   // row pieces of dgi (ChainEmit.rows), or null
    float* dh0; long dh0_ld; int dh0_acc;                        // tail launch: gradient into the initial state
};
struct BStepArgs { int H, B, nprob, first, tail; BStepProb p[2]; };

constexpr int BWM = 2, BWN = 4, BRM = 4, BRN = 2;               // 8 waves; wave tile 64 rows x 32 units; tile 128 x 128

__global__ __launch_bounds__(64 * BWM * BWN) void gru_step_bf3_bwd_kernel(BStepArgs a) {
    constexpr int TMB = BWM * BRM, TNB = BWN * BRN, NW = BWM * BWN;
    constexpr int STAGE = (TMB + TNB) * 3 * 1024, CH = (TMB + TNB) * 3, CPW = (CH + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w / BWN, wn = w % BWN;
    const int H = a.H, KB = (3 * H) >> 5;
    const int tiles_m = a.B / (TMB * 16), tiles_n = H / (TNB * 16);
    const int nb = gridDim.x, id = blockIdx.x;
    const int tid = (nb % 8 == 0) ? (id % 8) * (nb / 8) + id / 8 : id;
    const int per_prob = tiles_m * tiles_n;
    const int prob = tid / per_prob, v = tid - prob * per_prob;
    const int tm = v / tiles_n, tn = v - tm * tiles_n;
    const BStepProb& P = a.p[prob];
    const int c = lane & 15, q = lane >> 4;
    const int j0 = tn * (TNB * 16) + wn * 32;                   // this wave's 32 hidden units

    f32x4 acc[BRM][BRN];
#pragma unroll
    for (int i = 0; i < BRM; ++i)
#pragma unroll
        for (int u = 0; u < BRN; ++u) acc[i][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Epilogue operands of one row block (16 rows x 32 units): eight 16-byte loads per 16-unit half -- the five saves, dh z, dout,
    // dhn --, all issued back to back from branch-free sources (an absent one aims at the saves with a zero factor: a conditional
    // load costs a branch and a drained vmcnt each, 8 x 7 dependent round trips = 80 us of a 164 us launch in the first build).
    // Row block 0's are requested before the contraction, row block i + 1's while row block i is being computed.
    const int vrow = lane >> 2, vcol = (lane & 3) * 4;                 // vector phase: row vrow, units vcol .. +3 of each 16-unit half
    const float* const svp = a.tail ? P.dhz_in : P.sv;                 // (always valid memory of finite values)
    const long sv_as = a.tail ? 0 : P.sv_astride;
    const float f_dhz = P.dhz_in ? 1.f : 0.f, f_dout = (!a.tail && P.dout) ? 1.f : 0.f, f_dhn = (!a.tail && P.dhn) ? 1.f : 0.f;
    const float* const dhzp = P.dhz_in ? P.dhz_in : svp;
    const float* const doutp = f_dout != 0.f ? P.dout : svp;  const long dout_ld = f_dout != 0.f ? P.dout_ld : H;
    const float* const dhnp = f_dhn != 0.f ? P.dhn : svp;     const long dhn_ld = f_dhn != 0.f ? P.dhn_ld : H;
    auto issue = [&](int i, f32x4 (&L)[BRN][8]) {
        const long row = (long)(tm * TMB + wm * BRM + i) * 16 + vrow;
#pragma unroll
        for (int u = 0; u < BRN; ++u) {
            const int jj = j0 + 16 * u + vcol;
            const float* sp = svp + row * H + jj;
#pragma unroll
            for (int k = 0; k < 5; ++k) L[u][k] = *reinterpret_cast<const f32x4*>(sp + k * sv_as);
            L[u][5] = *reinterpret_cast<const f32x4*>(dhzp + row * H + jj);
            L[u][6] = *reinterpret_cast<const f32x4*>(doutp + row * dout_ld + jj);
            L[u][7] = *reinterpret_cast<const f32x4*>(dhnp + row * dhn_ld + jj);
        }
    };
    f32x4 ops[2][BRN][8];
    issue(0, ops[0]);

    if (!a.first) {
        const unsigned char* gsrc[CPW]; int loff[CPW];
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int cc0 = w + i * NW;
            const int cc = cc0 < CH ? cc0 : CH - 1;
            if (cc < TMB * 3) {
                const int p = cc / TMB, rbl = cc % TMB;
                gsrc[i] = P.A + p * P.a_piece + ((long)(tm * TMB + rbl) * KB) * 1024 + lane * 16;
            } else {
                const int c2 = cc - TMB * 3, p = c2 / TNB, rbl = c2 % TNB;
                gsrc[i] = P.W + p * P.w_piece + ((long)(tn * TNB + rbl) * KB) * 1024 + lane * 16;
            }
            loff[i] = cc * 1024;
        }
        auto fill = [&](int kb, unsigned char* stage) {
#pragma unroll
            for (int i = 0; i < CPW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[i] + (long)kb * 1024),
                                                 (__attribute__((address_space(3))) void*)(stage + loff[i]), 16, 0, 0);
        };
        // (two LDS stages.  A third -- fill kb + 2 in flight across the barrier, manual vmcnt -- measured the same 152 us: the launch
        //  is bound by its epilogue's memory traffic, ~390 MB per step at B = 4096, not by the fill latency)
        fill(0, smem);
        __syncthreads();
        for (int kb = 0; kb < KB; ++kb) {
            const unsigned char* sa = smem + (kb & 1) * STAGE + lane * 16;
            const unsigned char* sb = sa + TMB * 3 * 1024;
            if (kb + 1 < KB) fill(kb + 1, smem + ((kb + 1) & 1) * STAGE);
            bf16x8 Af[BRM][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < BRM; ++i) Af[i][p] = *reinterpret_cast<const bf16x8*>(sa + (p * TMB + wm * BRM + i) * 1024);
#pragma unroll
            for (int pj = 0; pj < 3; ++pj) {
                bf16x8 Bf[BRN];
#pragma unroll
                for (int u = 0; u < BRN; ++u) Bf[u] = *reinterpret_cast<const bf16x8*>(sb + (pj * TNB + wn * BRN + u) * 1024);
#pragma unroll
                for (int pi = 0; pi < 3; ++pi)
#pragma unroll
                    for (int i = 0; i < BRM; ++i)
#pragma unroll
                        for (int u = 0; u < BRN; ++u)
                            acc[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[i][pi], Bf[u], acc[i][u], 0, 0, 0);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: per row block, 16 rows x 32 units through wave-private LDS tiles ----
    float* const xt = reinterpret_cast<float*>(smem) + w * (5 * 512);  // dh, then the gate tiles dr, dz, dnr, dn: 16 x 32 floats each
    const int prow = lane & 15, pk8 = (lane >> 4) * 8;                 // piece phase: row prow, units pk8 .. +7 of the 32
#pragma unroll
    for (int i = 0; i < BRM; ++i) {
        const int rb = tm * TMB + wm * BRM + i;
        if (i + 1 < BRM) issue(i + 1, ops[(i + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < BRN; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) xt[(4 * q + r) * 32 + 16 * u + c] = acc[i][u][r];
        __builtin_amdgcn_wave_barrier();
        const long row = (long)rb * 16 + vrow;
        f32x4 (&L)[BRN][8] = ops[i & 1];
#pragma unroll
        for (int u = 0; u < BRN; ++u) {
            const int jj = j0 + 16 * u + vcol;                          // four consecutive units
            f32x4 dh = *reinterpret_cast<const f32x4*>(xt + vrow * 32 + 16 * u + vcol);
            dh += f_dhz * L[u][5];
            if (a.tail) {
                float* o = P.dh0 + row * P.dh0_ld + jj;
                if (P.dh0_acc) dh += *reinterpret_cast<const f32x4*>(o);
                *reinterpret_cast<f32x4*>(o) = dh;
                continue;
            }
            dh += f_dout * L[u][6] + f_dhn * L[u][7];
            const f32x4 rg = L[u][0], zg = L[u][1], ng = L[u][2], ghn = L[u][3], hpv = L[u][4];
            f32x4 dr, dz, dn, dnr, dhz;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dn_pre = dh[e] * (1.f - zg[e]) * (1.f - ng[e] * ng[e]);
                dz[e] = dh[e] * (hpv[e] - ng[e]) * zg[e] * (1.f - zg[e]);
                dr[e] = dn_pre * ghn[e] * rg[e] * (1.f - rg[e]);
                dn[e] = dn_pre; dnr[e] = dn_pre * rg[e];
                dhz[e] = dh[e] * zg[e];
            }
            float* gi = P.dgi + row * P.dgi_ld + jj;
            *reinterpret_cast<f32x4*>(gi) = dr; *reinterpret_cast<f32x4*>(gi + H) = dz; *reinterpret_cast<f32x4*>(gi + 2 * H) = dn;
            float* gh = P.dgh + row * 3 * H + jj;
            *reinterpret_cast<f32x4*>(gh) = dr; *reinterpret_cast<f32x4*>(gh + H) = dz; *reinterpret_cast<f32x4*>(gh + 2 * H) = dnr;
            *reinterpret_cast<f32x4*>(P.dhz_out + row * H + jj) = dhz;
            const int o = vrow * 32 + 16 * u + vcol;
            *reinterpret_cast<f32x4*>(xt + 512 + o) = dr; *reinterpret_cast<f32x4*>(xt + 1024 + o) = dz;
            *reinterpret_cast<f32x4*>(xt + 1536 + o) = dnr; *reinterpret_cast<f32x4*>(xt + 2048 + o) = dn;
        }
        if (a.tail) continue;
        __builtin_amdgcn_wave_barrier();
        // pieces: gate g's 32 units of these 16 rows = k block (g H + j0) / 32 of row block rb: one fragment per gate and layout
        const int lfrag = ((lane >> 4) * 16 + prow) * 16;               // byte offset of (row, 8-unit group) inside the fragment
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int kbg = (g * H + j0) >> 5;
            if (P.An) pieces8_store(xt + 512 * (g + 1) + prow * 32 + pk8, P.An + ((long)rb * KB + kbg) * 1024 + lfrag, P.a_piece);
            if (P.em) pieces8_store(xt + 512 * (g == 2 ? 4 : g + 1) + prow * 32 + pk8,
                                    P.em + ((P.em_rb0 + rb) * P.em_kb + P.em_kb0 + kbg) * 1024 + lfrag, P.em_piece);
        }
    }
}

template <bool TAB, bool DEN, bool SAVE, int RM>
int launch_one_rm(const StepArgs& a, hipStream_t s) {
    constexpr int WM = 2, WN = 4;                                  // 8 waves; wave tile 16 RM rows x (16 units x 3 gates)
    constexpr int TMB = WM * RM, TNB = WN * RN;
    auto kern = &gru_step_bf3_kernel<TAB, DEN, SAVE, WM, WN, RM>;
    const size_t lds = (size_t)2 * (TMB + TNB) * 3 * 1024;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int grid = a.nprob * (a.B / (TMB * 16)) * (a.H / (16 * WN));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// Tile height: 128 rows (RM = 4) or 96 (RM = 3, round 5): one workgroup per CU and launch, so what counts is rounds of 256 tiles x
// rows per tile -- 1536 rows (LatentRNN's frozen encoder without the target measures nobody reads) are 192 tiles of 128 rows = one
// round at 128 rows per CU, or 256 tiles of 96 rows = one round at 96.
int step_tile_rows(int H, int B, int nd) {
    auto cost = [&](int rows) { return B % rows ? 1 << 30 : ((nd * (B / rows) * (H / 64) + 255) / 256) * rows; };
    return cost(96) < cost(128) ? 96 : 128;
}
template <bool TAB, bool DEN, bool SAVE>
int launch_one(const StepArgs& a, hipStream_t s) {
    return step_tile_rows(a.H, a.B, a.nprob) == 96 ? launch_one_rm<TAB, DEN, SAVE, 3>(a, s) : launch_one_rm<TAB, DEN, SAVE, 4>(a, s);
}

}  // namespace

namespace {
int g_min_tiles = -1;
int min_tiles_now() {
    if (g_min_tiles < 0) g_min_tiles = 256;
    return g_min_tiles;
}
}  // namespace
void gru_step_bf3_set_min_tiles(int n) { g_min_tiles = n < 0 ? 0 : n; }
bool gru_step_bf3_ok(int H, int B, int T, int nd) {
    const int min_tiles = min_tiles_now();
    if (bf3_mode() == 0 || min_tiles <= 0) return false;
    if (H < 64 || H % 64 || B < 96 || (B % 128 && B % 96) || T < 1 || nd < 1 || nd > 2) return false;
    if ((double)T * B * 6.0 * H >= 2.0e9) return false;
    return nd * (B / step_tile_rows(H, B, nd)) * (H / 64) >= min_tiles;
}

size_t gru_step_bf3_w_bytes(int H) { return bf3_bytes(3L * H, H); }

int gru_step_bf3_split_w(int H, const float* const* W_hh, unsigned char* const* Wp, int nd, hipStream_t s) {
    // gate g's H rows (H / 16 row blocks) of direction d -> row blocks 3u + g of Wp[d]: all of them in one launch
    if (nd < 1 || nd > 2) return -1;
    Bf3SplitJob jobs[6];
    for (int d = 0; d < nd; ++d)
        for (int g = 0; g < 3; ++g) jobs[3 * d + g] = Bf3SplitJob{W_hh[d] + (long)g * H * H, Wp[d], g, 3};
    return bf3_split_strided_batch(jobs, 3 * nd, H, H, H, (long)bf3_piece_bytes(3L * H, H), H / 32, s);
}

int launch_gru_steps_bf3(const GruStepsBf3& L, hipStream_t s) {
    const int H = L.H, B = L.B, T = L.T, nd = L.nprob;
    if (!gru_step_bf3_ok(H, B, T, nd)) return -1;
    const long apiece = (long)bf3_piece_bytes(B, H);
    const long slot = 3 * apiece;
    for (int i = 0; i < nd; ++i) {
        const GruChainFwdProb& P = L.p[i];
        if (!L.Wp[i] || !P.hx || !P.out || !P.b_hh) return -1;
        if (P.h0)   // the initial state enters the ring like any later one: slot 1 is what step 0 reads
            INET_TRY(bf3_split(P.h0, P.ld_h0, 0, B, H, reinterpret_cast<unsigned char*>(P.hx) + slot, apiece, H / 32, 0, 0, s));
    }
    for (int i = 1; i < nd; ++i)
        if ((L.p[i].h0 != nullptr) != (L.p[0].h0 != nullptr)) return -1;      // (one zero_state flag per launch)
    const bool tab = L.p[0].gi_table != nullptr, den = L.p[0].gi_dense != nullptr, save = L.p[0].sv != nullptr;
    for (int i = 1; i < nd; ++i)
        if ((L.p[i].gi_table != nullptr) != tab || (L.p[i].gi_dense != nullptr) != den || (L.p[i].sv != nullptr) != save) return -1;
    if (tab == den) return -1;                               // one input-side source per layer (gather table: layer 0; dense: layer 1)
    char label[96];
    std::snprintf(label, sizeof label, "gru_step_bf3 p9 np%d B%d H%d%s", nd, B, H, save ? " sv" : "");
    for (int step = 0; step < T; ++step) {
        StepArgs a{};
        a.H = H; a.B = B; a.nprob = nd;
        bool zero = true;
        for (int i = 0; i < nd; ++i) {
            const GruChainFwdProb& P = L.p[i];
            StepProb& Q = a.p[i];
            const int tt = P.reverse ? T - 1 - step : step;
            const int tp = P.reverse ? tt + 1 : tt - 1;
            unsigned char* ring = reinterpret_cast<unsigned char*>(P.hx);
            Q.A = ring + (long)((step + 1) & 1) * slot; Q.a_piece = apiece;
            Q.An = step + 1 < T ? ring + (long)(step & 1) * slot : nullptr;
            Q.W = L.Wp[i]; Q.w_piece = (long)bf3_piece_bytes(3L * H, H);
            Q.b_hh = P.b_hh;
            if (P.gi_dense) { Q.gi = P.gi_dense + (long)tt * P.ts_gi; Q.gi_ld = P.ld_gi; }
            if (P.gi_table) { Q.table = P.gi_table; Q.table_ld = P.ld_table; Q.idx = P.idx + (long)tt * P.idx_ts; Q.idx_bs = P.idx_bs; }
            Q.gvec = P.gi_vec;
            if (step == 0) { Q.hprev = P.h0; Q.hprev_ld = P.ld_h0; if (P.h0) zero = false; }
            else { Q.hprev = P.out + (long)tp * P.ts_out; Q.hprev_ld = P.ld_out; zero = false; }
            Q.out = P.out + (long)tt * P.ts_out; Q.out_ld = P.ld_out;
            if (P.outm) { Q.outm = P.outm + (long)tt * P.ts_outm; Q.outm_ld = P.ld_outm; }
            if (P.mask) { Q.mask = P.mask + (long)tt * P.ts_mask; Q.mask_ld = P.ld_mask; }
            if (P.hlast && step == T - 1) { Q.hlast = P.hlast; Q.hlast_ld = P.ld_hlast; }
            if (P.sv) { Q.sv = P.sv + (long)tt * (P.sv_ts ? P.sv_ts : (long)B * H); Q.sv_astride = P.sv_astride; }
            if (P.em.rows) {
                Q.em = P.em.rows; Q.em_piece = P.em.rows_piece; Q.em_kb = P.em.rows_kb; Q.em_kb0 = P.em.rows_kb0;
                Q.em_rb0 = ((long)tt * P.em.B_full + P.em.r0) / 16;
            }
        }
        a.zero_state = zero ? 1 : 0;
        // algorithmic bytes of a step: W pieces once, state pieces in and out, gi, out (+ saves)
        ProfScope prof(PROF_GRU_FWD, zero ? 0.0 : 2.0 * nd * B * 3.0 * H * H, s, label,
                       nd * (6.0 * 3 * H * H + 12.0 * B * H + 4.0 * B * 3 * H + 4.0 * B * H * (save ? 7 : 2)));
        int rc;
        if (tab) rc = save ? launch_one<true, false, true>(a, s) : launch_one<true, false, false>(a, s);
        else rc = save ? launch_one<false, true, true>(a, s) : launch_one<false, true, false>(a, s);
        if (rc != 0) return rc;
    }
    return 0;
}

// ---- backward steps ----
bool gru_step_bf3_bwd_ok(int H, int B, int T, int nd) {
    const int min_tiles = min_tiles_now();
    if (bf3_mode() == 0 || min_tiles <= 0) return false;
    if (H < 128 || H % 128 || B < 128 || B % 128 || T < 1 || nd < 1 || nd > 2) return false;
    if ((double)T * B * 6.0 * H >= 2.0e9) return false;
    return nd * (B / 128) * (H / 128) >= (min_tiles + 1) / 2;       // (128 x 128 tiles: half as many as the forward kernel's for a shape)
}

int gru_step_bf3_split_wT(int H, const float* W_hh, unsigned char* WpT, hipStream_t s) {
    // B operand of dh = dgh W_hh: row n = hidden unit j, contraction over the 3H gate rows: B(j, k) = W_hh[k * H + j]
    return bf3_split(W_hh, H, 1, H, 3 * H, WpT, (long)bf3_piece_bytes(H, 3L * H), 3 * H / 32, 0, 0, s);
}

int launch_gru_steps_bf3_bwd(const GruStepsBf3Bwd& L, hipStream_t s) {
    const int H = L.H, B = L.B, T = L.T, nd = L.nprob;
    if (!gru_step_bf3_bwd_ok(H, B, T, nd)) return -1;
    const long apiece = (long)bf3_piece_bytes(B, 3L * H), slot = 3 * apiece, BH = (long)B * H;
    for (int i = 0; i < nd; ++i) {
        const GruChainBwdProb& P = L.p[i];
        if (!L.WpT[i] || !L.dhz[i] || !P.gx || !P.sv || !P.dgi || !P.dgh) return -1;
        if (P.ts_dgi != (long)B * P.ld_dgi || (P.dgh_ts && P.dgh_ts != 3 * BH)) return -1;    // (the column sums below walk [T*B] rows)
        if ((L.p[i].dh0 != nullptr) != (L.p[0].dh0 != nullptr) || P.dgi_sum) return -1;
    }
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_step_bf3_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int grid = nd * (B / 128) * (H / 128);
    const size_t lds = (size_t)2 * (BWM * BRM + BWN * BRN) * 3 * 1024;
    char label[96];
    std::snprintf(label, sizeof label, "gru_step_bf3_bwd p9 np%d B%d H%d", nd, B, H);
    const bool want_dh0 = L.p[0].dh0 != nullptr;
    for (int step = T - 1; step >= (want_dh0 ? -1 : 0); --step) {
        BStepArgs a{};
        a.H = H; a.B = B; a.nprob = nd; a.first = step == T - 1; a.tail = step < 0;
        for (int i = 0; i < nd; ++i) {
            const GruChainBwdProb& P = L.p[i];
            BStepProb& Q = a.p[i];
            unsigned char* ring = reinterpret_cast<unsigned char*>(P.gx);
            Q.A = ring + (long)((step + 1) & 1) * slot; Q.a_piece = apiece;
            Q.W = L.WpT[i]; Q.w_piece = (long)bf3_piece_bytes(H, 3L * H);
            Q.dhz_in = a.first ? nullptr : L.dhz[i] + (long)((step + 1) & 1) * BH;
            if (a.tail) { Q.dh0 = P.dh0; Q.dh0_ld = P.ld_dh0; Q.dh0_acc = P.dh0_accumulate; continue; }
            const int tt = P.reverse ? T - 1 - step : step;
            Q.An = (step > 0 || want_dh0) ? ring + (long)(step & 1) * slot : nullptr;
            Q.sv = P.sv + (long)tt * (P.sv_ts ? P.sv_ts : BH); Q.sv_astride = P.sv_astride;
            if (P.dout) { Q.dout = P.dout + (long)tt * P.ts_dout; Q.dout_ld = P.ld_dout; }
            if (P.dhn && a.first) { Q.dhn = P.dhn; Q.dhn_ld = P.ld_dhn; }
            Q.dhz_out = L.dhz[i] + (long)(step & 1) * BH;
            Q.dgi = P.dgi + (long)tt * P.ts_dgi; Q.dgi_ld = P.ld_dgi;
            Q.dgh = P.dgh + (long)tt * 3 * BH;
            if (P.em.rows) {
                Q.em = P.em.rows; Q.em_piece = P.em.rows_piece; Q.em_kb = P.em.rows_kb; Q.em_kb0 = P.em.rows_kb0;
                Q.em_rb0 = ((long)tt * P.em.B_full + P.em.r0) / 16;
            }
        }
        ProfScope prof(PROF_GRU_BWD, a.first ? 0.0 : 2.0 * nd * B * 3.0 * H * H, s, label,
                       nd * (6.0 * 3 * H * H + 2 * 18.0 * B * H + 4.0 * B * H * (5 + 2 + 7)));
        hipLaunchKernelGGL(gru_step_bf3_bwd_kernel, dim3(grid), dim3(64 * BWM * BWN), lds, s, a);
        if (hipGetLastError() != hipSuccess) return -2;
    }
    // bias gradients: db_ih += column sums of dgi, db_hh += column sums of dgh over all T*B rows
    PwColsumJob jobs[4];
    int nj = 0;
    for (int i = 0; i < nd; ++i) {
        const GruChainBwdProb& P = L.p[i];
        if (!P.db_ih || !P.db_hh) continue;
        jobs[nj++] = PwColsumJob{P.dgi, P.ld_dgi, T * B, 3 * H, P.db_ih};
        jobs[nj++] = PwColsumJob{P.dgh, 3L * H, T * B, 3 * H, P.db_hh};
    }
    if (nj) INET_TRY(pw_colsum_multi(jobs, nj, s));
    return 0;
}
