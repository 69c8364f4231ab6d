// Chain kernels for the GRU layers (chain.h): ALL time steps of one layer -- up to 4 independent problems (the two
// directions of a bi-GRU layer, the beats of the tick decoder) -- in ONE persistent launch.
//
//   forward : h_t = GRUCell(gi_t, h_{t-1}); gi_t comes from the caller as a dense [T,B,3H] tensor, a gather table indexed by
//             token, or a broadcast vector (the same three sources as gru_step_fwd_kernel).  W_hh slices (16 hidden units x
//             {r,z,n} x K/4 per wave = 96 VGPRs at H=512) are loaded once; h_{t-1} of the thread's own elements stays in
//             registers; per step only the group's hidden state [16*MS rows, H] moves, fragment-major, through L2.
//   backward: dh_t = dgh_{t+1} W_hh + dh_{t+1} z_{t+1} + dout_t; gate derivatives; the exchange carries dgh (K = 3H).
//             Bias gradients are summed in registers over the whole sequence and added once.
//
// Same arithmetic as gru_step_fwd_kernel / gru_step_bwd_kernel (gru.hip), which remain the path for shapes a chain does
// not cover (more row tiles than CUs, H other than 256/512, single steps): tests compare the two.
#include <cstdio>
#include <cstdlib>
#include "chain.h"
#include "ksplit.h"
#include "prof.h"
#include "gru_chain.h"
// Contraction schedule of the forward / backward kernels (chain.h): the backward kernels (K = 3H, 24 k-steps per wave)
// stream their A fragments with dealt, pinned loads -- 274 -> 246 us per encoder launch, 4.354 -> 4.340 ms per training
// step; the forward kernels (8 k-steps per wave) measure the same either way and keep the chunked double buffer.
#define GRU_CONTRACT contract
#define GRU_CONTRACT_B contract_stream

using namespace ksplit;

namespace {

// OCC = 2: built for two waves per SIMD (256 registers, a few spilled), so that two launches over independent row
// chunks share the chip and each one's hand-off latency (a third of a step) is filled by the other's MFMAs.
template <int MS, int SQ, int OCC>                 // SQ = H/64 k-steps per wave
__global__ __launch_bounds__(256, OCC) void gru_chain_fwd_kernel(GruChainFwd A) {
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * MS * 256];
    __shared__ __attribute__((aligned(16))) float xt[MS * 256];
    __shared__ unsigned flag[2];
    int group, member;
    chain::decode_block(blockIdx.x, A.members, group, member);
    if (group >= A.nprob * A.tiles_per_prob) return;
    if (A.fault && blockIdx.x == 0) return;        // injected fault (test hook)
    if (A.prio) __builtin_amdgcn_s_setprio(3);     // the chain's waves are mostly parked; when they have work they go first
    const GruChainFwdProb& P = A.p[group / A.tiles_per_prob];
    const int row0 = (group % A.tiles_per_prob) * 16 * MS;
    const int H = A.H, B = A.B, T = A.T;
    if (row0 >= B) return;
    const int S = H >> 4, t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int j0 = member * 16, jc = j0 + (t & 15);
    const int rb0 = row0 >> 4, rb_last = (B - 1) >> 4;
    const int slot_bytes = P.hx_slot_bytes ? P.hx_slot_bytes : ((B + 15) >> 4) * 16 * H * 4;
    float hp[MS];
    int brow[MS];
    const bool has_h0 = P.h0 != nullptr;          // null: an all-zero initial state (never exchanged, step 0 contracts nothing)
#pragma unroll
    for (int p = 0; p < MS; ++p) {
        brow[p] = min(row0 + ((t + 256 * p) >> 4), B - 1);
        hp[p] = has_h0 ? P.h0[(long)brow[p] * P.ld_h0 + jc] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t rs = chain::make_rsrc(P.hx);
    unsigned* counter = A.counters + group * kChainCounterStride;
    // The initial state enters the exchange like any later one: every member publishes its own 16 columns of h0 into
    // slot 1 and arrives (a pack launch in front of every chain used to do this: 8 launches per training step).
    const bool publish_h0 = has_h0 && !A.h0_packed;
    if (publish_h0) {
#pragma unroll
        for (int p = 0; p < MS; ++p) xt[((t + 256 * p) >> 4) * 16 + (t & 15)] = hp[p];
        __syncthreads();
        if (t < 64 * MS && rb0 + (t >> 6) <= rb_last)
            chain::publish_block(rs, slot_bytes, xt, t >> 6, lane, rb0 + (t >> 6), S, member);
        chain::arrive(counter);
    }
    const int arrivals0 = publish_h0 ? 1 : 0;
    // (the weights are requested AFTER the initial state is out: arrive() drains every outstanding load of the wave, and the
    // W slice is the one load here that comes from HBM -- the other members' wait should not include it)
    f32x4 Wr[3][SQ];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int si = 0; si < SQ; ++si)
            Wr[g][si] = ld4u(P.W_hh + (long)(g * H + j0 + i16) * H + 16 * (w * SQ + si) + 4 * q);
    float bh[3], bv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = P.b_hh[g * H + jc];
    if (P.gi_vec) {
#pragma unroll
        for (int g = 0; g < 3; ++g) bv[g] = P.gi_vec[g * H + jc];
    }
    // Operand sources as (pointer, strides) with every field in a register before the loop; an absent source points at a
    // zero word with zero strides, so the per-step requests are unconditional loads issued back to back (conditional
    // loads make hipcc wrap each in a branch with its own s_waitcnt vmcnt(0): one exposed round trip per operand).
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
    const chain::Status status = A.status;
    // the token of the NEXT step is requested one step ahead (token -> table row is a dependent load)
    long tok[MS];
#pragma unroll
    for (int p = 0; p < MS; ++p) tok[p] = idxp[brow[p] * idx_bs + (rev ? T - 1 : 0) * idx_ts];
    for (int step = 0; step < T; ++step) {
        const int tt = rev ? T - 1 - step : step;
        const int tn = step + 1 < T ? (rev ? tt - 1 : tt + 1) : tt;
        float pgt[MS][3], pgd[MS][3], pm[MS];
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int b = brow[p];
#pragma unroll
            for (int g = 0; g < 3; ++g) pgt[p][g] = tabp[(int)tok[p] * tab_ld + g * tab_g + tab_j];
#pragma unroll
            for (int g = 0; g < 3; ++g) pgd[p][g] = denp[tt * den_ts + b * den_ld + g * den_g + den_j];
            pm[p] = mskp[tt * msk_ts + b * msk_ld + msk_j];
            tok[p] = idxp[b * idx_bs + tn * idx_ts];
        }
        const bool recur = step > 0 || has_h0;
        if ((step > 0 || publish_h0) &&
            !chain::wait_group<chain::K_GRU_FWD>(counter, (unsigned)((step + arrivals0) * members), status, &flag[step & 1])) return;
        f32x4 acc[MS][4];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (recur) chain::GRU_CONTRACT<MS, 3, SQ>(acc, Wr, rs, ((step + 1) & 1) * slot_bytes, rb0, rb_last, S, w * SQ, lane);
        float v[MS][3];
        reduce_waves<MS, 3>(acc, red, t, v);
        // gates first, then the hand-off (what the other members wait for), then the stores nobody in the launch reads
        float er[MS], ez[MS], en[MS], eg[MS], eh[MS], ehp[MS];
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int rl = (t + 256 * p) >> 4;
            const float ghn = v[p][2] + bh[2];
            const float r = sigmoid_f(v[p][0] + pgd[p][0] + pgt[p][0] + bv[0] + bh[0]);
            const float z = sigmoid_f(v[p][1] + pgd[p][1] + pgt[p][1] + bv[1] + bh[1]);
            const float n = tanh_f(pgd[p][2] + pgt[p][2] + bv[2] + r * ghn);
            const float hprev = hp[p];
            const float hn = (1.f - z) * n + z * hprev;
            hp[p] = hn;
            xt[rl * 16 + (t & 15)] = hn;
            er[p] = r; ez[p] = z; en[p] = n; eg[p] = ghn; eh[p] = hn; ehp[p] = hprev;
        }
        if (step != T - 1) {                       // nobody reads the last state from the exchange
            __syncthreads();
            if (t < 64 * MS && rb0 + (t >> 6) <= rb_last)
                chain::publish_block(rs, (step & 1) * slot_bytes, xt, t >> 6, lane, rb0 + (t >> 6), S, member);
            chain::arrive(counter);
        }
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int b = row0 + ((t + 256 * p) >> 4);
            if (b < B) {
                outp[tt * out_ts + b * out_ld + jc] = eh[p];
                if (outmp) outmp[tt * outm_ts + b * outm_ld + jc] = has_mask ? eh[p] * pm[p] : eh[p];
                if (hlastp && step == T - 1) hlastp[b * hlast_ld + jc] = eh[p];
                if (svp) {
                    float* sp = svp + tt * sv_ts + b * H + jc;
                    sp[0] = er[p]; sp[sv_as] = ez[p]; sp[2 * sv_as] = en[p]; sp[3 * sv_as] = eg[p]; sp[4 * sv_as] = ehp[p];
                }
            }
        }
    }
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// x = x0 + x1 + x2 exactly (gru_chain2.hip / gemm_bf3.hip): 8 consecutive floats -> the three pieces, stored 16 bytes each
__device__ __forceinline__ void store_pieces8(const float* src, unsigned char* dst, long piece) {
    bf16x8_t p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = src[j];
        const __bf16 a = (__bf16)x;
        const float r1 = x - (float)a;
        const __bf16 b = (__bf16)r1;
        p0[j] = a; p1[j] = b; p2[j] = (__bf16)(r1 - (float)b);
    }
    *reinterpret_cast<bf16x8_t*>(dst) = p0;
    *reinterpret_cast<bf16x8_t*>(dst + piece) = p1;
    *reinterpret_cast<bf16x8_t*>(dst + 2 * piece) = p2;
}

// EMR: the build that also writes the ROW pieces of dgi (ChainEmit.rows: the A operand of the layer's data gradient on the bf16
// pipe, gemm_bf3.hip) -- from the transpose tiles it publishes from anyway, behind the hand-off.  (The transposed pieces of the
// weight gradients are left to split launches: they run beside this kernel, which keeps most of the CU's LDS free for them.)
template <int MS, int SQ, bool EMR = false>         // SQ = 3H/64 k-steps per wave over K = 3H
__global__ __launch_bounds__(256) void gru_chain_bwd_kernel(GruChainBwd A) {
    __shared__ __attribute__((aligned(16))) float red[4 * MS * 256];
    __shared__ __attribute__((aligned(16))) float xt[EMR ? 4 : 3][MS * 256];
    __shared__ unsigned flag[2];
    int group, member;
    chain::decode_block(blockIdx.x, A.members, group, member);
    if (group >= A.nprob * A.tiles_per_prob) return;
    if (A.prio) __builtin_amdgcn_s_setprio(3);
    const GruChainBwdProb& P = A.p[group / A.tiles_per_prob];
    const int row0 = (group % A.tiles_per_prob) * 16 * MS;
    const int H = A.H, B = A.B, T = A.T;
    if (row0 >= B) return;
    const int S3 = (3 * H) >> 4, t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int j0 = member * 16, jc = j0 + (t & 15);
    const int rb0 = row0 >> 4, rb_last = (B - 1) >> 4;
    const int slot_bytes = P.gx_slot_bytes ? P.gx_slot_bytes : ((B + 15) >> 4) * 16 * 3 * H * 4;
    // B operand = W_hh^T rows j0..j0+15, k over the 3H gate rows: element (j, k) = W_hh[k][j]; read once, strided
    f32x4 Wr[1][SQ];
#pragma unroll
    for (int si = 0; si < SQ; ++si) {
        const long k = 16 * (w * SQ + si) + 4 * q;
        const float* wp = P.W_hh + k * H + j0 + i16;
        Wr[0][si] = f32x4{wp[0], wp[H], wp[2 * H], wp[3 * (long)H]};
    }
    const __amdgpu_buffer_rsrc_t rs = chain::make_rsrc(P.gx);
    unsigned* counter = A.counters + group * kChainCounterStride;
    float dhz[MS], bs[4] = {0.f, 0.f, 0.f, 0.f}, gs[MS][3];
    int brow[MS];
#pragma unroll
    for (int p = 0; p < MS; ++p) {
        dhz[p] = 0.f; brow[p] = min(row0 + ((t + 256 * p) >> 4), B - 1);
        gs[p][0] = gs[p][1] = gs[p][2] = 0.f;
    }
    // operand sources in registers, absent ones aimed at a zero word (see the forward kernel)
    const float* const zf = reinterpret_cast<const float*>(A.counters + kChainZeroWord);
    const bool has_dout = P.dout != nullptr, has_dhn = P.dhn != nullptr;
    const float* const doutp = has_dout ? P.dout : zf;
    const int dout_ld = has_dout ? (int)P.ld_dout : 0, dout_ts = has_dout ? (int)P.ts_dout : 0, dout_j = has_dout ? jc : 0;
    const float* const dhnp = has_dhn ? P.dhn : zf;
    const int dhn_ld = has_dhn ? (int)P.ld_dhn : 0, dhn_j = has_dhn ? jc : 0;
    const float* const svp = P.sv; const int sv_as = (int)P.sv_astride, sv_ts = P.sv_ts ? (int)P.sv_ts : B * H;
    float* const dgip = P.dgi; const int dgi_ld = (int)P.ld_dgi, dgi_ts = (int)P.ts_dgi;
    float* const dghp = P.dgh; const int dgh_ts = P.dgh_ts ? (int)P.dgh_ts : B * 3 * H;
    float* const dh0p = P.dh0; const int dh0_ld = (int)P.ld_dh0; const int dh0_acc = P.dh0_accumulate;
    const int rev = P.reverse, members = A.members;
    const chain::Status status = A.status;
    // row pieces of dgi (EMR): row block (t * B_full / 16 + rbg), k block rows_kb0 + gate * H / 32 + member / 2, this member's half
    unsigned char* const em_rows = EMR ? P.em.rows : nullptr;
    const long em_rows_piece = P.em.rows_piece;
    const int em_rows_kb = P.em.rows_kb, em_rbg = (P.em.r0 >> 4) + rb0;
    const long em_tstride = (long)(P.em.B_full >> 4) * em_rows_kb * 1024;
    const int em_lane = ((P.em.rows_kb0 + (member >> 1)) * 64 + (2 * (member & 1) + ((lane >> 4) & 1)) * 16 + (lane & 15)) * 16;
    for (int step = T - 1; step >= -1; --step) {
        const bool tail = step < 0;                // dh0 = dgh(first step) W_hh + dhz
        if (tail && !dh0p) break;
        const int tt = tail ? 0 : (rev ? T - 1 - step : step);
        float pd[MS], psv[MS][5];
        if (!tail) {
#pragma unroll
            for (int p = 0; p < MS; ++p) {
                const int b = brow[p];
                const float d1 = doutp[tt * dout_ts + b * dout_ld + dout_j];
                const float d2 = dhnp[b * dhn_ld + dhn_j];
                const float* sp = svp + tt * sv_ts + b * H + jc;
#pragma unroll
                for (int a = 0; a < 5; ++a) psv[p][a] = sp[a * sv_as];
                pd[p] = step == T - 1 ? d1 + d2 : d1;
            }
        } else {
#pragma unroll
            for (int p = 0; p < MS; ++p) pd[p] = dh0_acc ? dh0p[brow[p] * dh0_ld + jc] : 0.f;
        }
        float v[MS][1];
#pragma unroll
        for (int p = 0; p < MS; ++p) v[p][0] = 0.f;
        if (step != T - 1) {
            if (!chain::wait_group<chain::K_GRU_BWD>(counter, (unsigned)((T - 1 - step) * members), status, &flag[step & 1])) return;
            f32x4 acc[MS][4];
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) acc[ms][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (MS == 8) {               // two 64-row tiles against the same register-resident W slice
                chain::GRU_CONTRACT_B<4, 1, SQ>(reinterpret_cast<f32x4(&)[4][4]>(acc[0]), Wr, rs, ((step + 1) & 1) * slot_bytes, rb0,
                                          rb_last, S3, w * SQ, lane);
                chain::GRU_CONTRACT_B<4, 1, SQ>(reinterpret_cast<f32x4(&)[4][4]>(acc[4]), Wr, rs, ((step + 1) & 1) * slot_bytes, rb0 + 4,
                                          rb_last, S3, w * SQ, lane);
            } else {
                chain::GRU_CONTRACT_B<MS, 1, SQ>(acc, Wr, rs, ((step + 1) & 1) * slot_bytes, rb0, rb_last, S3, w * SQ, lane);
            }
            reduce_waves<MS, 1>(acc, red, t, v);
        }
        if (tail) {
#pragma unroll
            for (int p = 0; p < MS; ++p) {
                const int b = row0 + ((t + 256 * p) >> 4);
                if (b < B) dh0p[b * dh0_ld + jc] = v[p][0] + dhz[p] + pd[p];
            }
            break;
        }
        float e_r[MS], e_z[MS], e_n[MS], e_nr[MS];
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int rl = (t + 256 * p) >> 4;
            const float dh = v[p][0] + dhz[p] + pd[p];
            const float r = psv[p][0], z = psv[p][1], n = psv[p][2], ghn = psv[p][3], hprev = psv[p][4];
            const float dn_pre = dh * (1.f - z) * (1.f - n * n);
            const float dz_pre = dh * (hprev - n) * z * (1.f - z);
            const float dr_pre = dn_pre * ghn * r * (1.f - r);
            const float dnr = dn_pre * r;
            dhz[p] = dh * z;
            const int xo = rl * 16 + (t & 15);
            xt[0][xo] = dr_pre; xt[1][xo] = dz_pre; xt[2][xo] = dnr;
            if (EMR) xt[3][xo] = dn_pre;
            e_r[p] = dr_pre; e_z[p] = dz_pre; e_n[p] = dn_pre; e_nr[p] = dnr;
        }
        if (step != 0 || dh0p) {                   // (nothing reads the last gate gradients unless dh0 is wanted)
            __syncthreads();
            for (int blk = t >> 6; blk < 3 * MS; blk += 4) {
                const int g = blk / MS, p = blk % MS;
                if (rb0 + p <= rb_last)
                    chain::publish_block(rs, (step & 1) * slot_bytes, xt[g], p, lane, rb0 + p, S3, g * (H >> 4) + member);
            }
            chain::arrive(counter);
        }
        if (EMR && em_rows) {                       // (behind the hand-off; the tiles stay valid until the next step's barrier)
            if (!(step != 0 || dh0p)) __syncthreads();             // (the publish above did not run: its barrier neither)
            for (int blk = t >> 6; blk < 3 * MS; blk += 4) {
                const int g = blk / MS, p = blk % MS;               // gate 0 r, 1 z, 2 n (tile 3); row block p of this workgroup's tile
                if (lane < 32 && rb0 + p <= rb_last) {
                    const float* src = xt[g == 2 ? 3 : g] + (p * 16 + (lane & 15)) * 16 + 8 * (lane >> 4);
                    store_pieces8(src, em_rows + ((long)tt * em_tstride + (long)(em_rbg + p) * em_rows_kb * 1024 + (long)g * (H >> 5) * 1024 +
                                                  em_lane), em_rows_piece);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int b = row0 + ((t + 256 * p) >> 4);
            if (b < B) {
                float* gi = dgip + tt * dgi_ts + b * dgi_ld;
                gi[jc] = e_r[p]; gi[H + jc] = e_z[p]; gi[2 * H + jc] = e_n[p];
                float* gh = dghp + tt * dgh_ts + b * 3 * H;
                gh[jc] = e_r[p]; gh[H + jc] = e_z[p]; gh[2 * H + jc] = e_nr[p];
                bs[0] += e_r[p]; bs[1] += e_z[p]; bs[2] += e_n[p]; bs[3] += e_nr[p];
                gs[p][0] += e_r[p]; gs[p][1] += e_z[p]; gs[p][2] += e_n[p];
            }
        }
    }
    if (P.dgi_sum) {
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int b = row0 + ((t + 256 * p) >> 4);
            if (b < B) {
                float* o = P.dgi_sum + (long)b * 3 * H + jc;
                o[0] = gs[p][0]; o[H] = gs[p][1]; o[2 * H] = gs[p][2];
            }
        }
    }
    if (P.db_ih) {
        __syncthreads();
        float* lb = &red[0];
#pragma unroll
        for (int a = 0; a < 4; ++a) lb[a * 256 + t] = bs[a];
        __syncthreads();
        if (t < 64) {
            const int a = t >> 4, cc = t & 15;
            float sum = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) sum += lb[a * 256 + rr * 16 + cc];
            const int j = j0 + cc;
            if (a == 0) { unsafeAtomicAdd(P.db_ih + j, sum); unsafeAtomicAdd(P.db_hh + j, sum); }
            else if (a == 1) { unsafeAtomicAdd(P.db_ih + H + j, sum); unsafeAtomicAdd(P.db_hh + H + j, sum); }
            else if (a == 2) unsafeAtomicAdd(P.db_ih + 2 * H + j, sum);
            else unsafeAtomicAdd(P.db_hh + 2 * H + j, sum);
        }
    }
}

// Rows per workgroup (16 * MS): the smallest tile whose launch still fits on the chip (every workgroup resident, at most
// kChainMaxGroups groups).  A chain step is hand-off latency plus the MFMAs of ONE workgroup, so more, smaller groups
// shorten every step: B = 128 with two directions runs 8 groups of 32 rows instead of 4 of 64.
int rows_ms(int B, int H, int nprob) {
    constexpr int force = 0;
    if (force > 0) return B <= 16 ? 1 : (B <= 32 ? 2 : 4);                 // the fixed rule of the first chain kernels
    for (int ms = 1; ms <= 4; ms *= 2) {
        const int groups = nprob * ((B + 16 * ms - 1) / (16 * ms));
        if (groups * (H / 16) <= chain_capacity() && groups <= kChainMaxGroups) return ms;
    }
    return 4;
}
int chain_prio() {
    static int v = -1;
    if (v < 0) v = 1;
    return v;
}

}  // namespace

// H = 1024 (LatentRNN's generator; round 4): first-generation kernels with 192 registers of W_hh per lane, a group's 64 members
// on two XCDs (chain.h decode_block); (rounds 1-3 ran that layer on the per-step kernels)
static bool h1024_on() { return true; }
bool gru_chain_ok(int H, int B, int T, int nprob) {
    if (H == 1024 && !h1024_on()) return false;
    if ((double)T * B * 6.0 * H >= 2.0e9) return false;   // the kernels index with 32-bit element offsets
    if (!chain_enabled() || (H != 256 && H != 512 && H != 1024) || T < 2 || nprob < 1 || nprob > 4 || B < 1) return false;
    const int ms = rows_ms(B, H, nprob), tiles = (B + 16 * ms - 1) / (16 * ms);
    return nprob * tiles * (H / 16) <= chain_capacity() && nprob * tiles <= kChainMaxGroups;   // every workgroup resident at once
}

// Backward chains may give a workgroup two 64-row tiles (MS = 8) when one per workgroup would need more than 256
// workgroups: the decoder's tick layers run their 4 beats as 4 problems x 256 rows.
int rows_ms_bwd(int H, int B, int nprob) {
    const int ms = rows_ms(B, H, nprob);
    constexpr bool wide = true;
    if (wide && H <= 512 && ms == 4 && B >= 128 && nprob * ((B + 63) / 64) * (H / 16) > chain_capacity()) return 8;
    return ms;
}
bool gru_chain_bwd_ok(int H, int B, int T, int nprob) {
    if (H == 1024 && !h1024_on()) return false;
    if ((double)T * B * 6.0 * H >= 2.0e9) return false;
    if (!chain_enabled() || (H != 256 && H != 512 && H != 1024) || T < 2 || nprob < 1 || nprob > 4 || B < 1) return false;
    const int ms = rows_ms_bwd(H, B, nprob), tiles = (B + 16 * ms - 1) / (16 * ms);
    return nprob * tiles * (H / 16) <= chain_capacity() && nprob * tiles <= kChainMaxGroups;
}

bool gru_chain_fwd_is_v2(int H, int B, int T, int nprob, int h0_packed) {
    constexpr bool v2f = true;
    return v2f && !h0_packed && gru_chain2_ok(H, B, T, nprob);
}
// The BPTT chains run on the FIRST generation: its kernel takes 28 KB of LDS and ~300 registers per lane, so the leaf work of the
// backward pass (weight-gradient products, column sums, the bf16-pipe products' split launches) shares the CUs with it.  A
// second-generation BPTT kernel existed in round 3 (the faster kernel alone, 220 vs 232 us per 24-step launch; the slower step, 3.73
// vs 3.62 ms: a workgroup held 148-160 KB of its CU's LDS for the length of the chain) and was removed in round 4 (HISTORY.md).
bool gru_chain_bwd_is_v2(int, int, int, int) { return false; }

int launch_gru_chain_fwd(GruChainFwd a, hipStream_t s) {
    if (gru_chain_fwd_is_v2(a.H, a.B, a.T, a.nprob, a.h0_packed)) return launch_gru_chain2_fwd(a, s);
    if (!gru_chain_ok(a.H, a.B, a.T, a.nprob)) return -1;
    const int ms = rows_ms(a.B, a.H, a.nprob);
    a.tiles_per_prob = (a.B + 16 * ms - 1) / (16 * ms);
    a.members = a.H / 16;
    const int groups = a.nprob * a.tiles_per_prob;
    if (groups > kChainMaxGroups) return -1;
    a.prio = chain_prio();
    a.fault = chain_take_fault();
    if (!a.prezeroed && hipMemsetAsync(a.counters, 0, kChainSyncWords * sizeof(unsigned), s) != hipSuccess) return -2;
    a.status = chain_status_for(a.counters + kChainStatusWord);
    char label[72];
    std::snprintf(label, sizeof label, "gru_chain_fwd ms%d%s np%d T%d B%d H%d", ms, a.shared_chip && ms == 4 ? "x2" : "", a.nprob, a.T, a.B, a.H);
    const double rows = (double)a.nprob * a.T * a.B;
    ProfScope prof(PROF_GRU_FWD, 2.0 * rows * 3.0 * a.H * a.H, s, label,
                   4.0 * (a.nprob * 3.0 * a.H * a.H + rows * a.H * (2 + 3 + (a.p[0].sv ? 5 : 0))));
    const dim3 grid(chain::blocks_for(groups, a.members));
#define DISPATCH_CF(M, Q, O) hipLaunchKernelGGL((gru_chain_fwd_kernel<M, Q, O>), grid, dim3(256), 0, s, a)
    if (a.shared_chip && ms == 4) { if (a.H == 512) DISPATCH_CF(4, 8, 2); else DISPATCH_CF(4, 4, 2); }
    else if (a.H == 1024) { if (ms == 1) DISPATCH_CF(1, 16, 1); else if (ms == 2) DISPATCH_CF(2, 16, 1); else DISPATCH_CF(4, 16, 1); }   // (LatentRNN's generator: 192 registers of W_hh per lane)
    else if (a.H == 512) { if (ms == 1) DISPATCH_CF(1, 8, 1); else if (ms == 2) DISPATCH_CF(2, 8, 1); else DISPATCH_CF(4, 8, 1); }
    else { if (ms == 1) DISPATCH_CF(1, 4, 1); else if (ms == 2) DISPATCH_CF(2, 4, 1); else DISPATCH_CF(4, 4, 1); }
#undef DISPATCH_CF
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// the first-generation BPTT launch for this shape writes ChainEmit.rows itself (H = 512, 64 rows per workgroup)
bool gru_chain_bwd_emits_rows(int H, int B, int T, int nprob) {
    return !gru_chain_bwd_is_v2(H, B, T, nprob) && H == 512 && gru_chain_bwd_ok(H, B, T, nprob) && rows_ms_bwd(H, B, nprob) == 4;
}
int launch_gru_chain_bwd(GruChainBwd a, hipStream_t s) {
    if (!gru_chain_bwd_ok(a.H, a.B, a.T, a.nprob)) return -1;
    const int ms = rows_ms_bwd(a.H, a.B, a.nprob);
    a.tiles_per_prob = (a.B + 16 * ms - 1) / (16 * ms);
    a.members = a.H / 16;
    const int groups = a.nprob * a.tiles_per_prob;
    if (groups > kChainMaxGroups) return -1;
    a.prio = chain_prio();
    if (!a.prezeroed && hipMemsetAsync(a.counters, 0, kChainSyncWords * sizeof(unsigned), s) != hipSuccess) return -2;
    a.status = chain_status_for(a.counters + kChainStatusWord);
    // row pieces of dgi (ChainEmit.rows): written by the H = 512, 64-rows-per-workgroup build; everything else of the descriptor is
    // the second generation's (the callers split what was not written)
    bool emr = a.H == 512 && ms == 4;
    for (int i = 0; i < a.nprob; ++i) {
        emr = emr && a.p[i].em.rows;
        a.p[i].em.colsA = a.p[i].em.colsB = nullptr; a.p[i].em.skip_dgi = a.p[i].em.skip_dgh = 0;
    }
    if (!emr) for (int i = 0; i < a.nprob; ++i) a.p[i].em.rows = nullptr;
    char label[72];
    std::snprintf(label, sizeof label, "gru_chain_bwd ms%d%s np%d T%d B%d H%d", ms, emr ? "e" : "", a.nprob, a.T, a.B, a.H);
    const double rows = (double)a.nprob * a.T * a.B;
    ProfScope prof(PROF_GRU_BWD, 2.0 * rows * 3.0 * a.H * a.H, s, label,
                   4.0 * (a.nprob * 3.0 * a.H * a.H + rows * a.H * (6 + 5 + 1)) + (emr ? 18.0 * rows * a.H : 0.0));
    const dim3 grid(chain::blocks_for(groups, a.members));
#define DISPATCH_CB(M, Q) hipLaunchKernelGGL((gru_chain_bwd_kernel<M, Q>), grid, dim3(256), 0, s, a)
    if (emr) hipLaunchKernelGGL((gru_chain_bwd_kernel<4, 24, true>), grid, dim3(256), 0, s, a);
    else
    if (a.H == 1024) { if (ms == 1) DISPATCH_CB(1, 48); else if (ms == 2) DISPATCH_CB(2, 48); else if (ms == 4) DISPATCH_CB(4, 48); else return -1; }
    else if (a.H == 512) { if (ms == 1) DISPATCH_CB(1, 24); else if (ms == 2) DISPATCH_CB(2, 24); else if (ms == 4) DISPATCH_CB(4, 24); else DISPATCH_CB(8, 24); }
    else { if (ms == 1) DISPATCH_CB(1, 12); else if (ms == 2) DISPATCH_CB(2, 12); else if (ms == 4) DISPATCH_CB(4, 12); else DISPATCH_CB(8, 12); }
#undef DISPATCH_CB
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
