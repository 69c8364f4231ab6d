// Fused free-running decode (SURVEY.md 2.3 "K7", north_star "fused GRU decode kernel"): ALL 24 ticks of
// HierarchicalDecoder.forward_tick_rnn (MeasureVAE/decoder.py:473-529, sampling = argmax, no teacher forcing) in ONE
// persistent launch -- tick layer 0, tick layer 1, the output projection, ReLU, argmax and the feedback of the sampled
// token, for small inference batches (B <= 32: LatentRNNTester.generate, VAETester.decode_mid_point, forward_test).
//
// Before: 72 dependent launches per call (24 x [layer-0 step, layer-1 step, logits+argmax]), 0.67 ms at b = 1, every
// launch re-reading its weights through a cold L2.  Here a group of H/16 workgroups owns one row tile; member m owns
// hidden units [16m, 16m+16) of BOTH layers for the whole call:
//   * W_hh(l0) and W_hh(l1) slices live in registers (2 x 96 VGPRs), W_ih(l1)'s slice in LDS (96 KB, fragment-major), the
//     16 x H slice of W_out of the members that also compute a logits tile in registers: weights are read ONCE per call;
//   * per tick three hand-offs inside the group (chain.h protocol): h0_t -> everybody (layer 1's input), h1_t ->
//     the logits tiles, (max, argmax) per 16-column block -> everybody (the token that selects the next gather row);
//   * the layer-1 recurrent contraction h1_{t-1} W_hh^T does not depend on h0_t and runs while the group is still
//     handing h0_t over.
// Arithmetic per element is that of gru_step_fwd_kernel / logits_argmax_kernel (same MFMA contraction order per wave,
// same gate formulas, lowest index wins ties), so results agree with the per-tick path to fp32 round-off.
#include <cstdio>
#include <cstdlib>
#include "chain.h"
#include "ksplit.h"
#include "prof.h"
#include "decode_chain.h"

using namespace ksplit;

namespace {

// acc[ms][slot[g]] += A x B_g^T with the B fragments supplied by `getB(g, si)` (registers or LDS); see chain::contract
template <int MS, int NG, int SQ, class GetB>
__device__ __forceinline__ void contract_b(f32x4 (&acc)[MS][4], const int (&slot)[NG], GetB&& getB,
                                           __amdgpu_buffer_rsrc_t r, int base, int rb0, int rb_last, int S, int s0, int lane) {
    constexpr int CH = SQ < 4 ? SQ : 4, NCH = SQ / CH;
    f32x4 A[2][MS][CH];
    int vo[MS];
    const int oz = chain::opaque_zero();           // offsets in vector registers (chain.h)
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) vo[ms] = (((min(rb0 + ms, rb_last) + oz) * S + s0) * 256 + lane * 4) * 4;
    auto load = [&](int c, int buf) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int i = 0; i < CH; ++i) A[buf][ms][i] = chain::ld16_sc1(r, vo[ms] + (c * CH + i) * 1024, base);
    };
    load(0, 0);
    __builtin_amdgcn_sched_barrier(0);             // keep the prefetch order (chain.h)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) load(c + 1, (c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            f32x4 Bf[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) Bf[g] = getB(g, c * CH + i);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ms = 0; ms < MS; ++ms)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        acc[ms][slot[g]] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c & 1][ms][i][e], Bf[g][e], acc[ms][slot[g]], 0, 0, 0);
        }
    }
}

// NV > 0 (inference, one row block per group): EVERY member computes all NV 16-column blocks of the logits of its rows and
// takes the argmax itself -- the same arithmetic in the same order everywhere, so all members agree on the token bit for
// bit -- instead of NV members computing one block each and handing (max, argmax) partials over: two hand-offs per tick
// instead of three (a hand-off costs more than the 64 extra MFMAs per wave at these batch sizes).
// The same contraction for at most VR <= 4 valid rows on the VALU (b = 1 inpainting: LatentRNNTester.generate).  An MFMA
// contracts a 16-row tile whether one row is valid or sixteen (96 x 32 cycles per gate triple and wave); here lane (unit i16, k
// group q) multiplies ITS OWN register-resident W fragments with the row's values -- fetched as the usual A fragment and
// broadcast from lane (q, row) with one shuffle per component -- and the four k groups are summed with two shuffles: ~130 issue
// slots per row instead of 3,072 cycles of matrix pipe.  The result lands in the C layout of the MFMA path (rows 0..3 = the
// registers of lanes 0..15), so everything downstream is shared.
template <int NG, int SQ, int VR, class GetB>
__device__ __forceinline__ void contract_valu(f32x4 (&acc)[1][4], const int (&slot)[NG], GetB&& getB, __amdgpu_buffer_rsrc_t r,
                                              int base, int rb0, int S, int s0, int lane) {
    const int vo = ((rb0 * S + s0) * 256 + lane * 4) * 4;
    f32x4 A[SQ];
#pragma unroll
    for (int si = 0; si < SQ; ++si) A[si] = chain::ld16_sc1(r, vo + si * 1024, base);
    float p[VR][NG];
#pragma unroll
    for (int rr = 0; rr < VR; ++rr)
#pragma unroll
        for (int g = 0; g < NG; ++g) p[rr][g] = 0.f;
    const int src0 = lane & 48;
#pragma unroll
    for (int si = 0; si < SQ; ++si) {
        f32x4 Bf[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) Bf[g] = getB(g, si);
#pragma unroll
        for (int rr = 0; rr < VR; ++rr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = __shfl(A[si][e], src0 | rr, 64);
#pragma unroll
                for (int g = 0; g < NG; ++g) p[rr][g] = fmaf(a, Bf[g][e], p[rr][g]);
            }
        }
    }
    const bool q0 = lane < 16;
#pragma unroll
    for (int rr = 0; rr < VR; ++rr)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float v = p[rr][g];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[0][slot[g]][rr] += q0 ? v : 0.f;
        }
}

// reduce_waves through a buffer sized for two row blocks: the 64-row build (MS = 4) reduces its halves one after the other
// (its [4 waves][4 acc][4 x 256] floats would not fit beside the 96 KB of W_ih(l1)).
template <int MS, int NACC>
__device__ __forceinline__ void reduce_tile(const f32x4 (&acc)[MS][4], float* red, int t, float (&out)[MS][NACC]) {
    if constexpr (MS <= 2) {
        reduce_waves<MS, NACC>(acc, red, t, out);
    } else {
        static_assert(MS == 4, "row blocks per group: 1, 2 or 4");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a2[2][4];
            float o2[2][NACC];
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int a = 0; a < 4; ++a) a2[ms][a] = acc[2 * h + ms][a];
            if (h) __syncthreads();                            // the first half has been read
            reduce_waves<2, NACC>(a2, red, t, o2);
            // reduce_waves hands thread t the elements t, t + 256 of its 32-row half: rows (t >> 4) and 16 + (t >> 4)
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int a = 0; a < NACC; ++a) out[2 * h + ms][a] = o2[ms][a];
        }
    }
}

template <int MS, int SQ, bool TRAIN, int NV = 0, int VR = 0>  // SQ = H/64; TRAIN: dropout mask + backward saves; VR: VALU rows
__global__ __launch_bounds__(256) void decode_chain_kernel(DecodeChainArgs P) {
    static_assert(NV == 0 || (MS == 1 && !TRAIN && NV <= 4), "redundant logits: small-batch inference only");
    static_assert(VR == 0 || (NV > 0 && VR <= 4), "VALU contraction: the smallest inference batches only");
    constexpr int S = 4 * SQ, H = 64 * SQ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wih = smem;                                   // [3][S][64][4]  W_ih(l1) slice, fragment-major
    float* const red = wih + 3 * S * 256;                      // [4 waves][4 acc][MS*256]
    float* const xt = red + 4 * 4 * (MS < 2 ? MS : 2) * 256;   // [MS*256]  (red holds two row blocks at a time: reduce_tile)
    float* const xm = xt + MS * 256;                           // [MS*256] masked h0 (TRAIN)
    unsigned* const flag = reinterpret_cast<unsigned*>(xm + MS * 256);   // [2]
    int group, member;
    chain::decode_block(blockIdx.x, P.members, group, member);
    const int row0 = group * 16 * MS, B = P.B, T = P.T, V = P.V;
    const int Bs = P.Bs ? P.Bs : B;                            // row stride of the time / beat-major buffers (a row chunk of a larger batch)
    if (row0 >= B) return;
    __builtin_amdgcn_s_setprio(3);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int j0 = member * 16, jc = j0 + (t & 15);
    const int rb0 = row0 >> 4, rb_last = (B - 1) >> 4;
    const int pkh = ((B + 15) >> 4) * 16 * H;                  // floats of one fragment-major [B,H] state
    const int NCB = (V + 15) >> 4;                             // 16-column blocks of the vocabulary; the last one may be ragged:
                                                               // its columns >= V carry zero weights, are never stored and
                                                               // never win the argmax (their logit is forced below ReLU's 0)
    // logits tile of this member (if any): row block lp of the group, column block lcb
    const bool has_tile = member < NCB * MS;
    const int lp = member / NCB, lcb = member % NCB;

    // ---- weights, once per call --------------------------------------------------------------------------------
    constexpr int NWO = NV > 0 ? NV : 1;
    f32x4 W0[3][SQ], W1[3][SQ], Wo[NWO][SQ];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int si = 0; si < SQ; ++si) {
            const long o = (long)(g * H + j0 + i16) * H + 16 * (w * SQ + si) + 4 * q;
            W0[g][si] = ld4u(P.W_hh0 + o);
            W1[g][si] = ld4u(P.W_hh1 + o);
        }
    if constexpr (NV > 0) {
#pragma unroll
        for (int cb = 0; cb < NV; ++cb)
#pragma unroll
            for (int si = 0; si < SQ; ++si)
                Wo[cb][si] = 16 * cb + i16 < V ? ld4u(P.W_out + (long)(16 * cb + i16) * H + 16 * (w * SQ + si) + 4 * q)
                                               : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int si = 0; si < SQ; ++si)
            Wo[0][si] = has_tile && 16 * lcb + i16 < V ? ld4u(P.W_out + (long)(16 * lcb + i16) * H + 16 * (w * SQ + si) + 4 * q)
                                                       : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = t; i < 3 * S * 64; i += 256) {                // W_ih(l1): slot (g, s, lane) <- 4 consecutive k of row g*H + j0 + lane%16
        const int ln = i & 63, s = (i >> 6) % S, g = i / (64 * S);
        *reinterpret_cast<f32x4*>(wih + (long)i * 4) = ld4u(P.W_ih1 + (long)(g * H + j0 + (ln & 15)) * H + 16 * s + 4 * (ln >> 4));
    }
    float bh0[3], bi1[3], bh1[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) { bh0[g] = P.b_hh0[g * H + jc]; bi1[g] = P.b_ih1[g * H + jc]; bh1[g] = P.b_hh1[g * H + jc]; }
    const bool col_ok = has_tile && 16 * lcb + (t & 15) < V;
    const float bo = col_ok ? P.b_out[16 * lcb + (t & 15)] : 0.f;
    float bov[NWO];                                            // (NV > 0) bias of column 16 cb + (t & 15)
#pragma unroll
    for (int cb = 0; cb < NWO; ++cb) bov[cb] = NV > 0 && 16 * cb + (t & 15) < V ? P.b_out[16 * cb + (t & 15)] : 0.f;
    long own_tok = 0;                                          // (NV > 0) token of row t >> 4 from the previous tick
    int brow[MS];
#pragma unroll
    for (int p = 0; p < MS; ++p) brow[p] = min(row0 + ((t + 256 * p) >> 4), B - 1);
    const __amdgpu_buffer_rsrc_t r_hx0 = chain::make_rsrc(P.hx0), r_hx1 = chain::make_rsrc(P.hx1),
                                 r_ht0 = chain::make_rsrc(P.ht0pk), r_am = chain::make_rsrc(P.amax);
    const bool masked = TRAIN && P.mask != nullptr;
    const __amdgpu_buffer_rsrc_t r_hxm = chain::make_rsrc(masked ? P.hx0m : P.hx0);
    unsigned* const counter = P.counters + group * kDecodeCounterStride;
    const chain::Status status = P.status;
    const int members = P.members, G = P.G, nb = T / G;
    const int am_slot = NCB * (((B + 15) >> 4) * 16);          // float2 entries per amax slot: [NCB][rows16]
    unsigned phase = 0;                                        // hand-offs completed so far by this member
    float hp0[MS], hp1[MS];
    __syncthreads();                                           // wih is complete

    auto token_of = [&](int tick, int p) -> long {             // argmax over the column-block partials of `tick`
        float best = -1.f;
        int bi = 0;
        for (int cb = 0; cb < NCB; ++cb) {
            const chain::u32x2 u2 = __builtin_amdgcn_raw_buffer_load_b64(r_am, (((tick & 1) * am_slot + cb * (am_slot / NCB) + brow[p]) * 8), 0, 16);
            const unsigned long long u = ((unsigned long long)u2.y << 32) | u2.x;
            const float m = __builtin_bit_cast(float, (unsigned)(u >> 32));
            if (m > best) { best = m; bi = (int)(unsigned)u; }
        }
        return bi;
    };

    for (int tick = 0; tick < T; ++tick) {
        const int beat = tick / G, j = tick % G;
        // ================= layer 0 =================
        float c0[MS][3], mk[MS];
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            mk[p] = masked ? P.mask[((long)tick * Bs + brow[p]) * H + jc] : 1.f;
#pragma unroll
            for (int g = 0; g < 3; ++g) c0[p][g] = P.cgi[((long)beat * Bs + brow[p]) * 3 * H + g * H + jc];
            if (j == 0) {                                      // hidden states restart from the beat embedding
                hp0[p] = P.ht0[((long)beat * Bs + brow[p]) * 2 * H + jc];
                hp1[p] = P.ht0[((long)beat * Bs + brow[p]) * 2 * H + H + jc];
            }
        }
        if (NV == 0 && tick > 0 && !chain::wait_group<chain::K_DECODE_CHAIN>(counter, phase * members, status, &flag[phase & 1])) return;   // tokens of tick-1
        long tok[MS];
#pragma unroll
        for (int p = 0; p < MS; ++p) tok[p] = tick == 0 ? V : (NV > 0 ? own_tok : token_of(tick - 1, p));
        if (NV == 0 && tick > 0 && member == 0 && (t & 15) == 0) {
#pragma unroll
            for (int p = 0; p < MS; ++p)
                if (row0 + ((t + 256 * p) >> 4) < B) P.samples[(long)brow[p] * T + tick - 1] = tok[p];
        }
        float g0[MS][3];
#pragma unroll
        for (int p = 0; p < MS; ++p)
#pragma unroll
            for (int g = 0; g < 3; ++g) g0[p][g] = P.table[tok[p] * 3 * H + g * H + jc];
        f32x4 acc[MS][4];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const int sl[3] = {0, 1, 2};
            auto gb = [&](int g, int si) { return W0[g][si]; };
            if constexpr (VR > 0) {
                if (j == 0) contract_valu<3, SQ, VR>(acc, sl, gb, r_ht0, beat * pkh * 4, rb0, S, w * SQ, lane);
                else contract_valu<3, SQ, VR>(acc, sl, gb, r_hx0, ((tick + 1) & 1) * pkh * 4, rb0, S, w * SQ, lane);
            } else {
                if (j == 0) contract_b<MS, 3, SQ>(acc, sl, gb, r_ht0, beat * pkh * 4, rb0, rb_last, S, w * SQ, lane);
                else contract_b<MS, 3, SQ>(acc, sl, gb, r_hx0, ((tick + 1) & 1) * pkh * 4, rb0, rb_last, S, w * SQ, lane);
            }
        }
        float v[MS][4];
        reduce_tile<MS, 4>(acc, red, t, v);
        float sv[MS][6];                                       // TRAIN: r, z, n, ghn, h_prev, h0 as layer 1 sees it
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const float ghn = v[p][2] + bh0[2];
            const float r = sigmoid_f(v[p][0] + c0[p][0] + g0[p][0] + bh0[0]);
            const float z = sigmoid_f(v[p][1] + c0[p][1] + g0[p][1] + bh0[1]);
            const float n = tanh_f(c0[p][2] + g0[p][2] + r * ghn);
            if (TRAIN) { sv[p][0] = r; sv[p][1] = z; sv[p][2] = n; sv[p][3] = ghn; sv[p][4] = hp0[p]; }
            hp0[p] = (1.f - z) * n + z * hp0[p];
            xt[((t + 256 * p) >> 4) * 16 + (t & 15)] = hp0[p];
            if (TRAIN) {
                sv[p][5] = hp0[p] * mk[p];
                if (masked) xm[((t + 256 * p) >> 4) * 16 + (t & 15)] = sv[p][5];
            }
        }
        __syncthreads();
        if (t < 64 * MS && rb0 + (t >> 6) <= rb_last) {
            chain::publish_block(r_hx0, (tick & 1) * pkh * 4, xt, t >> 6, lane, rb0 + (t >> 6), S, member);
            if (masked) chain::publish_block(r_hxm, beat * pkh * 4, xm, t >> 6, lane, rb0 + (t >> 6), S, member);
        }
        chain::arrive(counter);
        ++phase;
        if (TRAIN) {                                           // backward saves: after the hand-off, off the critical path
#pragma unroll
            for (int p = 0; p < MS; ++p) {
                if (row0 + ((t + 256 * p) >> 4) < B) {
                    const long o = ((long)tick * Bs + brow[p]) * H + jc;
                    if (P.sv0) {
#pragma unroll
                        for (int a = 0; a < 5; ++a) P.sv0[o + a * P.sv_stride] = sv[p][a];
                    }
                    if (P.h0out) P.h0out[o] = sv[p][5];
                }
            }
        }
        // ================= layer 1 =================
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        {   // recurrent half first: it needs h1 of the previous tick only, and hides the hand-off of h0
            const int sh[3] = {0, 1, 3};
            auto gb = [&](int g, int si) { return W1[g][si]; };
            if constexpr (VR > 0) {
                if (j == 0) contract_valu<3, SQ, VR>(acc, sh, gb, r_ht0, (nb + beat) * pkh * 4, rb0, S, w * SQ, lane);
                else contract_valu<3, SQ, VR>(acc, sh, gb, r_hx1, ((tick + 1) & 1) * pkh * 4, rb0, S, w * SQ, lane);
            } else {
                if (j == 0) contract_b<MS, 3, SQ>(acc, sh, gb, r_ht0, (nb + beat) * pkh * 4, rb0, rb_last, S, w * SQ, lane);
                else contract_b<MS, 3, SQ>(acc, sh, gb, r_hx1, ((tick + 1) & 1) * pkh * 4, rb0, rb_last, S, w * SQ, lane);
            }
        }
        if (!chain::wait_group<chain::K_DECODE_CHAIN>(counter, phase * members, status, &flag[phase & 1])) return;               // h0 of this tick
        {
            const int sx[3] = {0, 1, 2};
            auto gb = [&](int g, int si) { return *reinterpret_cast<const f32x4*>(wih + ((g * S + w * SQ + si) * 64 + lane) * 4); };
            if constexpr (VR > 0) contract_valu<3, SQ, VR>(acc, sx, gb, r_hxm, (tick & 1) * pkh * 4, rb0, S, w * SQ, lane);
            else contract_b<MS, 3, SQ>(acc, sx, gb, r_hxm, (masked ? beat : (tick & 1)) * pkh * 4, rb0, rb_last, S, w * SQ, lane);
        }
        reduce_tile<MS, 4>(acc, red, t, v);
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const float ghn = v[p][3] + bh1[2];
            const float r = sigmoid_f(v[p][0] + bi1[0] + bh1[0]);
            const float z = sigmoid_f(v[p][1] + bi1[1] + bh1[1]);
            const float n = tanh_f(v[p][2] + bi1[2] + r * ghn);
            if (TRAIN) { sv[p][0] = r; sv[p][1] = z; sv[p][2] = n; sv[p][3] = ghn; sv[p][4] = hp1[p]; }
            hp1[p] = (1.f - z) * n + z * hp1[p];
            xt[((t + 256 * p) >> 4) * 16 + (t & 15)] = hp1[p];
        }
        __syncthreads();
        if (t < 64 * MS && rb0 + (t >> 6) <= rb_last)
            chain::publish_block(r_hx1, (tick & 1) * pkh * 4, xt, t >> 6, lane, rb0 + (t >> 6), S, member);
        chain::arrive(counter);
        ++phase;
        if (TRAIN) {
#pragma unroll
            for (int p = 0; p < MS; ++p) {
                if (row0 + ((t + 256 * p) >> 4) < B) {
                    const long o = ((long)tick * Bs + brow[p]) * H + jc;
                    if (P.sv1) {
#pragma unroll
                        for (int a = 0; a < 5; ++a) P.sv1[o + a * P.sv_stride] = sv[p][a];
                    }
                    if (P.h1seq) P.h1seq[o] = hp1[p];
                }
            }
        }
        // ================= output projection + partial argmax (members that own a logits tile) =================
        // EVERY member waits for all of h1 before its third arrival, tile or not: the counter is a plain count, and a
        // member that skipped this wait could arrive a third time while a slow one has not arrived twice -- the count
        // then reaches 2 x members too early and a logits tile contracts a k-slice that is not published yet (seen as
        // one wrong 16 x 16 logits tile at tick 0 in ~5 % of the B = 256 calls before this wait was made unconditional).
        if (!chain::wait_group<chain::K_DECODE_CHAIN>(counter, phase * members, status, &flag[phase & 1])) return;               // h1 of this tick
        if constexpr (NV > 0) {
            f32x4 la[1][4];
#pragma unroll
            for (int a = 0; a < 4; ++a) la[0][a] = f32x4{0.f, 0.f, 0.f, 0.f};
            int sl[NV];
#pragma unroll
            for (int cb = 0; cb < NV; ++cb) sl[cb] = cb;
            if constexpr (VR > 0) contract_valu<NV, SQ, VR>(la, sl, [&](int g, int si) { return Wo[g][si]; }, r_hx1, (tick & 1) * pkh * 4, rb0,
                                                            S, w * SQ, lane);
            else contract_b<1, NV, SQ>(la, sl, [&](int g, int si) { return Wo[g][si]; }, r_hx1, (tick & 1) * pkh * 4, rb0, rb_last, S,
                                       w * SQ, lane);
            float lv[1][NV];
            reduce_waves<1, NV>(la, red, t, lv);
            const int rl = t >> 4, c = t & 15, b = row0 + rl;
            float m = -1.f;
            int am = 0;
#pragma unroll
            for (int cb = 0; cb < NV; ++cb) {
                const bool okc = 16 * cb + c < V;
                float x = lv[0][cb] + bov[cb];
                x = x > 0.f ? x : 0.f;
                if (member == cb % members && b < B && okc) P.weights[((long)b * T + tick) * V + 16 * cb + c] = x;
                if (!okc) x = -1.f;                            // a padded column never wins
                if (x > m) { m = x; am = 16 * cb + c; }        // (ascending cb: the lowest index wins ties)
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {                  // (max, lowest argmax) over the 16 lanes of the row
                const float m2 = __shfl_xor(m, o, 16);
                const int a2 = __shfl_xor(am, o, 16);
                if (m2 > m || (m2 == m && a2 < am)) { m = m2; am = a2; }
            }
            own_tok = am;
            if (member == 0 && c == 0 && b < B) P.samples[(long)b * T + tick] = am;
            continue;                                          // no third hand-off
        }
        if (has_tile) {
            f32x4 la[1][4];
            la[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int s1[1] = {0};
            contract_b<1, 1, SQ>(la, s1, [&](int g, int si) { return Wo[0][si]; }, r_hx1, (tick & 1) * pkh * 4, rb0 + lp, rb_last, S,
                                 w * SQ, lane);
            float lv[1][1];
            reduce_waves<1, 1>(la, red, t, lv);
            const int rl = t >> 4, c = t & 15, b = row0 + 16 * lp + rl;
            float x = lv[0][0] + bo;
            x = x > 0.f ? x : 0.f;
            if (b < B && col_ok) P.weights[((long)b * T + tick) * V + 16 * lcb + c] = x;
            if (!col_ok) x = -1.f;                             // a padded column: below every real (post-ReLU) logit
            // (max, lowest argmax) over the 16 columns of the row: butterfly inside each 16-lane group
            float m = x;
            int am = c;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const float m2 = __shfl_xor(m, o, 16);
                const int a2 = __shfl_xor(am, o, 16);
                if (m2 > m || (m2 == m && a2 < am)) { m = m2; am = a2; }
            }
            if (c == 0 && rb0 + lp <= rb_last) {
                const unsigned long long u = ((unsigned long long)__builtin_bit_cast(unsigned, m) << 32) | (unsigned)(16 * lcb + am);
                __builtin_amdgcn_raw_buffer_store_b64(chain::u32x2{(unsigned)u, (unsigned)(u >> 32)}, r_am,
                                                      ((tick & 1) * am_slot + lcb * (am_slot / NCB) + row0 + 16 * lp + rl) * 8, 0, 16);
            }
        }
        chain::arrive(counter);
        ++phase;
    }
    // the last tick's tokens
    if (NV == 0 && member == 0) {
        if (!chain::wait_group<chain::K_DECODE_CHAIN>(counter, phase * members, status, &flag[phase & 1])) return;
        if ((t & 15) == 0) {
#pragma unroll
            for (int p = 0; p < MS; ++p)
                if (row0 + ((t + 256 * p) >> 4) < B) P.samples[(long)brow[p] * T + T - 1] = token_of(T - 1, p);
        }
    }
}

int rows_ms(int B) {                                            // the smallest tile that fits (gru_chain.hip)
    return (B + 15) / 16 <= kDecodeMaxGroups ? 1 : (B + 31) / 32 <= kDecodeMaxGroups ? 2 : 4;
}
inline int groups_of(int B) { const int ms = rows_ms(B); return (B + 16 * ms - 1) / (16 * ms); }

}  // namespace

bool decode_chain_ok(int B, int H, int V, int T, int G) {
    if (!chain_enabled() || (H != 256 && H != 512) || B < 1 || V < 1 || V > 128 || T % G != 0) return false;
    if (groups_of(B) > kDecodeMaxGroups || groups_of(B) * (H / 16) > chain_capacity()) return false;   // every workgroup resident at once
    constexpr bool off = false;
    if (off) return false;
    const int ms = rows_ms(B);
    return ((V + 15) / 16) * ms <= H / 16;                   // one member per (row block, 16-column block) logits tile
}

size_t decode_chain_lds_bytes(int B, int H) {
    const int ms = rows_ms(B), S = H / 16;
    return (size_t)(3 * S * 256 + 4 * 4 * (ms < 2 ? ms : 2) * 256 + 2 * ms * 256 + 4) * sizeof(float);
}

int launch_decode_chain(DecodeChainArgs a, hipStream_t s) {
    if (!decode_chain_ok(a.B, a.H, a.V, a.T, a.G)) return -1;
    const int ms = rows_ms(a.B), groups = (a.B + 16 * ms - 1) / (16 * ms);
    a.members = a.H / 16;
    if (!a.prezeroed && hipMemsetAsync(a.counters, 0, kDecodeSyncWords * sizeof(unsigned), s) != hipSuccess) return -2;
    a.status = chain_status_for(a.counters + kDecodeStatusWord);
    if (decode_b1_ok(a)) return launch_decode_b1(a, s);        // one measure: weights in registers, two hand-offs per tick
    const size_t lds = decode_chain_lds_bytes(a.B, a.H);
    char label[72];
    const bool train = a.sv0 || a.sv1 || a.mask || a.h0out || a.h1seq;
    if (a.mask && !a.hx0m) return -1;
    std::snprintf(label, sizeof label, "decode_chain%s ms%d T%d B%d H%d V%d", train ? "_train" : (a.B == 1 ? "_valu" : ""), ms, a.T, a.B, a.H, a.V);
    // algorithmic bytes: the tick GRU + output weights once per call, logits out; training: the dropout mask in, the 2 x 5
    // backward saves and the two layer outputs out (13 arrays of [T,B,H])
    ProfScope prof(PROF_GRU_FWD, 2.0 * a.T * a.B * (9.0 * a.H * a.H + (double)a.V * a.H), s, label,
                   4.0 * (9.0 * a.H * a.H + (double)a.V * a.H + (double)a.B * a.T * a.V +
                          (train ? 13.0 * a.T * a.B * a.H : 0.0)));
    const dim3 grid(chain::blocks_for(groups, a.members));
#define DISPATCH_DC5(M, Q, TR, NVV, VRR)                                                                                    \
    do {                                                                                                                \
        static bool attr = false;                                                                                       \
        if (!attr) {                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_chain_kernel<M, Q, TR, NVV, VRR>),            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                          \
            attr = true;                                                                                                \
        }                                                                                                               \
        hipLaunchKernelGGL((decode_chain_kernel<M, Q, TR, NVV, VRR>), grid, dim3(256), lds, s, a);                      \
    } while (0)
#define DISPATCH_DC4(M, Q, TR, NVV) DISPATCH_DC5(M, Q, TR, NVV, 0)
#define DISPATCH_DC(M, Q, TR) DISPATCH_DC4(M, Q, TR, 0)
    if (train) {
        if (a.H == 512) { if (ms == 1) DISPATCH_DC(1, 8, true); else if (ms == 2) DISPATCH_DC(2, 8, true); else DISPATCH_DC(4, 8, true); }
        else { if (ms == 1) DISPATCH_DC(1, 4, true); else if (ms == 2) DISPATCH_DC(2, 4, true); else DISPATCH_DC(4, 4, true); }
    } else {
        // small-batch inference (one row block per group, V <= 64): every member computes the whole logits row itself
        constexpr bool fullv = true;
        const int nv = (fullv && ms == 1 && a.V <= 64) ? (a.V <= 48 ? 3 : 4) : 0;
        // one row (b = 1 inpainting): the contractions on the VALU instead of one-sixteenth-full MFMA tiles
        constexpr bool valu = true;
        // (one row only: the four-row build of the H = 512 kernel spills ~300 registers)
        const int vr = (valu && nv > 0 && a.B == 1) ? 1 : 0;
        if (nv == 3 && vr == 1) { if (a.H == 512) DISPATCH_DC5(1, 8, false, 3, 1); else DISPATCH_DC5(1, 4, false, 3, 1); }
        else if (nv == 4 && vr == 1) { if (a.H == 512) DISPATCH_DC5(1, 8, false, 4, 1); else DISPATCH_DC5(1, 4, false, 4, 1); }
        else if (nv == 3) { if (a.H == 512) DISPATCH_DC4(1, 8, false, 3); else DISPATCH_DC4(1, 4, false, 3); }
        else if (nv == 4) { if (a.H == 512) DISPATCH_DC4(1, 8, false, 4); else DISPATCH_DC4(1, 4, false, 4); }
        else if (a.H == 512) { if (ms == 1) DISPATCH_DC(1, 8, false); else if (ms == 2) DISPATCH_DC(2, 8, false); else DISPATCH_DC(4, 8, false); }
        else { if (ms == 1) DISPATCH_DC(1, 4, false); else if (ms == 2) DISPATCH_DC(2, 4, false); else DISPATCH_DC(4, 4, false); }
    }
#undef DISPATCH_DC
#undef DISPATCH_DC4
#undef DISPATCH_DC5
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
