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
#include "chain.h"
#include "ksplit.h"
#include "prof.h"
#include "gru_chain.h"

using namespace ksplit;

namespace {

template <int MS, int SQ>                          // SQ = H/64 k-steps per wave
__global__ __launch_bounds__(256) void gru_chain_fwd_kernel(GruChainFwd A) {
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * MS * 256];
    __shared__ __attribute__((aligned(16))) float xt[MS * 256];
    __shared__ unsigned flag;
    int group, member;
    chain::decode_block(blockIdx.x, A.members, group, member);
    if (group >= A.nprob * A.tiles_per_prob) return;
    const GruChainFwdProb& P = A.p[group / A.tiles_per_prob];
    const int row0 = (group % A.tiles_per_prob) * 16 * MS;
    const int H = A.H, B = A.B, T = A.T;
    if (row0 >= B) return;
    const int S = H >> 4, t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int j0 = member * 16, jc = j0 + (t & 15);
    const int rb0 = row0 >> 4, rb_last = (B - 1) >> 4;
    const int slot_bytes = ((B + 15) >> 4) * 16 * H * 4;
    f32x4 Wr[3][SQ];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int si = 0; si < SQ; ++si)
            Wr[g][si] = ld4u(P.W_hh + (long)(g * H + j0 + i16) * H + 16 * (w * SQ + si) + 4 * q);
    float bh[3], bv[3] = {0.f, 0.f, 0.f}, hp[MS];
    int brow[MS];
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = P.b_hh[g * H + jc];
    if (P.gi_vec) {
#pragma unroll
        for (int g = 0; g < 3; ++g) bv[g] = P.gi_vec[g * H + jc];
    }
#pragma unroll
    for (int p = 0; p < MS; ++p) {
        brow[p] = min(row0 + ((t + 256 * p) >> 4), B - 1);
        hp[p] = P.h0[(long)brow[p] * P.ld_h0 + jc];
    }
    const __amdgpu_buffer_rsrc_t rs = chain::make_rsrc(P.hx);
    unsigned* counter = A.counters + group;
    for (int step = 0; step < T; ++step) {
        const int tt = P.reverse ? T - 1 - step : step;
        // epilogue operands of this step: none depends on h, so they are requested before the group wait
        float pg[MS][3], pm[MS];
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int b = brow[p];
            if (P.gi_table) {
                const long tok = P.idx[(long)b * P.idx_bs + (long)tt * P.idx_ts];
#pragma unroll
                for (int g = 0; g < 3; ++g) pg[p][g] = P.gi_table[tok * P.ld_table + g * H + jc];
            } else if (P.gi_dense) {
#pragma unroll
                for (int g = 0; g < 3; ++g) pg[p][g] = P.gi_dense[(long)tt * P.ts_gi + (long)b * P.ld_gi + g * H + jc];
            } else {
#pragma unroll
                for (int g = 0; g < 3; ++g) pg[p][g] = 0.f;
            }
            pm[p] = (P.outm && P.mask) ? P.mask[(long)tt * P.ts_mask + (long)b * P.ld_mask + jc] : 1.f;
        }
        if (step > 0 && !chain::wait_group(counter, (unsigned)(step * A.members), A.status, &flag)) return;
        f32x4 acc[MS][4];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        chain::contract<MS, 3, SQ>(acc, Wr, rs, ((step + 1) & 1) * slot_bytes, rb0, rb_last, S, w * SQ, lane);
        float v[MS][3];
        reduce_waves<MS, 3>(acc, red, t, v);
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int rl = (t + 256 * p) >> 4;
            const int b = row0 + rl;
            const float ghn = v[p][2] + bh[2];
            const float r = sigmoid_f(v[p][0] + pg[p][0] + bv[0] + bh[0]);
            const float z = sigmoid_f(v[p][1] + pg[p][1] + bv[1] + bh[1]);
            const float n = tanh_f(pg[p][2] + bv[2] + r * ghn);
            const float hprev = hp[p];
            const float hn = (1.f - z) * n + z * hprev;
            hp[p] = hn;
            xt[rl * 16 + (t & 15)] = hn;
            if (b < B) {
                P.out[(long)tt * P.ts_out + (long)b * P.ld_out + jc] = hn;
                if (P.outm) P.outm[(long)tt * P.ts_outm + (long)b * P.ld_outm + jc] = hn * pm[p];
                if (P.hlast && step == T - 1) P.hlast[(long)b * P.ld_hlast + jc] = hn;
                if (P.sv) {
                    float* sp = P.sv + ((long)tt * B + b) * H + jc;
                    const long as = P.sv_astride;
                    sp[0] = r; sp[as] = z; sp[2 * as] = n; sp[3 * as] = ghn; sp[4 * as] = hprev;
                }
            }
        }
        if (step == T - 1) break;                  // nobody reads the last state from the exchange
        __syncthreads();
        if (t < 64 * MS && rb0 + (t >> 6) <= rb_last)
            chain::publish_block(rs, (step & 1) * slot_bytes, xt, t >> 6, lane, rb0 + (t >> 6), S, member);
        chain::arrive(counter);
    }
}

template <int MS, int SQ>                          // SQ = 3H/64 k-steps per wave over K = 3H
__global__ __launch_bounds__(256) void gru_chain_bwd_kernel(GruChainBwd A) {
    __shared__ __attribute__((aligned(16))) float red[4 * MS * 256];
    __shared__ __attribute__((aligned(16))) float xt[3][MS * 256];
    __shared__ unsigned flag;
    int group, member;
    chain::decode_block(blockIdx.x, A.members, group, member);
    if (group >= A.nprob * A.tiles_per_prob) return;
    const GruChainBwdProb& P = A.p[group / A.tiles_per_prob];
    const int row0 = (group % A.tiles_per_prob) * 16 * MS;
    const int H = A.H, B = A.B, T = A.T;
    if (row0 >= B) return;
    const int S3 = (3 * H) >> 4, t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int j0 = member * 16, jc = j0 + (t & 15);
    const int rb0 = row0 >> 4, rb_last = (B - 1) >> 4;
    const int slot_bytes = ((B + 15) >> 4) * 16 * 3 * H * 4;
    // B operand = W_hh^T rows j0..j0+15, k over the 3H gate rows: element (j, k) = W_hh[k][j]; read once, strided
    f32x4 Wr[1][SQ];
#pragma unroll
    for (int si = 0; si < SQ; ++si) {
        const long k = 16 * (w * SQ + si) + 4 * q;
        const float* wp = P.W_hh + k * H + j0 + i16;
        Wr[0][si] = f32x4{wp[0], wp[H], wp[2 * H], wp[3 * (long)H]};
    }
    const __amdgpu_buffer_rsrc_t rs = chain::make_rsrc(P.gx);
    unsigned* counter = A.counters + group;
    float dhz[MS], bs[4] = {0.f, 0.f, 0.f, 0.f};
    int brow[MS];
#pragma unroll
    for (int p = 0; p < MS; ++p) { dhz[p] = 0.f; brow[p] = min(row0 + ((t + 256 * p) >> 4), B - 1); }
    for (int step = T - 1; step >= -1; --step) {
        const bool tail = step < 0;                // dh0 = dgh(first step) W_hh + dhz
        if (tail && !P.dh0) break;
        const int tt = tail ? 0 : (P.reverse ? T - 1 - step : step);
        float pd[MS], psv[MS][5];
        if (!tail) {
#pragma unroll
            for (int p = 0; p < MS; ++p) {
                const int b = brow[p];
                float d = P.dout ? P.dout[(long)tt * P.ts_dout + (long)b * P.ld_dout + jc] : 0.f;
                if (step == T - 1 && P.dhn) d += P.dhn[(long)b * P.ld_dhn + jc];
                pd[p] = d;
                const float* sp = P.sv + ((long)tt * B + b) * H + jc;
#pragma unroll
                for (int a = 0; a < 5; ++a) psv[p][a] = sp[a * P.sv_astride];
            }
        } else {
#pragma unroll
            for (int p = 0; p < MS; ++p) pd[p] = P.dh0_accumulate ? P.dh0[(long)brow[p] * P.ld_dh0 + jc] : 0.f;
        }
        float v[MS][1];
#pragma unroll
        for (int p = 0; p < MS; ++p) v[p][0] = 0.f;
        if (step != T - 1) {
            if (!chain::wait_group(counter, (unsigned)((T - 1 - step) * A.members), A.status, &flag)) return;
            f32x4 acc[MS][4];
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) acc[ms][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            chain::contract<MS, 1, SQ>(acc, Wr, rs, ((step + 1) & 1) * slot_bytes, rb0, rb_last, S3, w * SQ, lane);
            reduce_waves<MS, 1>(acc, red, t, v);
        }
        if (tail) {
#pragma unroll
            for (int p = 0; p < MS; ++p) {
                const int b = row0 + ((t + 256 * p) >> 4);
                if (b < B) P.dh0[(long)b * P.ld_dh0 + jc] = v[p][0] + dhz[p] + pd[p];
            }
            break;
        }
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int rl = (t + 256 * p) >> 4;
            const int b = row0 + rl;
            const float dh = v[p][0] + dhz[p] + pd[p];
            const float r = psv[p][0], z = psv[p][1], n = psv[p][2], ghn = psv[p][3], hprev = psv[p][4];
            const float dn_pre = dh * (1.f - z) * (1.f - n * n);
            const float dz_pre = dh * (hprev - n) * z * (1.f - z);
            const float dr_pre = dn_pre * ghn * r * (1.f - r);
            const float dnr = dn_pre * r;
            dhz[p] = dh * z;
            const int xo = rl * 16 + (t & 15);
            xt[0][xo] = dr_pre; xt[1][xo] = dz_pre; xt[2][xo] = dnr;
            if (b < B) {
                float* gi = P.dgi + (long)tt * P.ts_dgi + (long)b * P.ld_dgi;
                gi[jc] = dr_pre; gi[H + jc] = dz_pre; gi[2 * H + jc] = dn_pre;
                float* gh = P.dgh + ((long)tt * B + b) * 3 * H;
                gh[jc] = dr_pre; gh[H + jc] = dz_pre; gh[2 * H + jc] = dnr;
                bs[0] += dr_pre; bs[1] += dz_pre; bs[2] += dn_pre; bs[3] += dnr;
            }
        }
        if (step == 0 && !P.dh0) break;            // nothing reads the last gate gradients from the exchange
        __syncthreads();
        for (int blk = t >> 6; blk < 3 * MS; blk += 4) {
            const int g = blk / MS, p = blk % MS;
            if (rb0 + p <= rb_last)
                chain::publish_block(rs, (step & 1) * slot_bytes, xt[g], p, lane, rb0 + p, S3, g * (H >> 4) + member);
        }
        chain::arrive(counter);
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

int rows_ms(int B) { return B <= 16 ? 1 : (B <= 32 ? 2 : 4); }

}  // namespace

bool gru_chain_ok(int H, int B, int T, int nprob) {
    if (!chain_enabled() || (H != 256 && H != 512) || T < 2 || nprob < 1 || nprob > 4 || B < 1) return false;
    const int ms = rows_ms(B), tiles = (B + 16 * ms - 1) / (16 * ms);
    return nprob * tiles * (H / 16) <= 256;        // every workgroup must be resident at once (one per CU)
}

int launch_gru_chain_fwd(GruChainFwd a, hipStream_t s) {
    if (!gru_chain_ok(a.H, a.B, a.T, a.nprob)) return -1;
    const int ms = rows_ms(a.B);
    a.tiles_per_prob = (a.B + 16 * ms - 1) / (16 * ms);
    a.members = a.H / 16;
    const int groups = a.nprob * a.tiles_per_prob;
    if (groups > kChainMaxGroups) return -1;
    a.status.host = chain_host_status();
    if (hipMemsetAsync(a.counters, 0, (kChainMaxGroups + 1) * sizeof(unsigned), s) != hipSuccess) return -2;
    a.status.dev = a.counters + kChainMaxGroups;
    char label[72];
    std::snprintf(label, sizeof label, "gru_chain_fwd ms%d np%d T%d B%d H%d", ms, a.nprob, a.T, a.B, a.H);
    const double rows = (double)a.nprob * a.T * a.B;
    ProfScope prof(PROF_GRU_FWD, 2.0 * rows * 3.0 * a.H * a.H, s, label,
                   4.0 * (a.nprob * 3.0 * a.H * a.H + rows * a.H * (2 + 3 + (a.p[0].sv ? 5 : 0))));
    const dim3 grid(chain::blocks_for(groups, a.members));
#define INET_CF(M, Q) hipLaunchKernelGGL((gru_chain_fwd_kernel<M, Q>), grid, dim3(256), 0, s, a)
    if (a.H == 512) { if (ms == 1) INET_CF(1, 8); else if (ms == 2) INET_CF(2, 8); else INET_CF(4, 8); }
    else { if (ms == 1) INET_CF(1, 4); else if (ms == 2) INET_CF(2, 4); else INET_CF(4, 4); }
#undef INET_CF
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_gru_chain_bwd(GruChainBwd a, hipStream_t s) {
    if (!gru_chain_ok(a.H, a.B, a.T, a.nprob)) return -1;
    const int ms = rows_ms(a.B);
    a.tiles_per_prob = (a.B + 16 * ms - 1) / (16 * ms);
    a.members = a.H / 16;
    const int groups = a.nprob * a.tiles_per_prob;
    if (groups > kChainMaxGroups) return -1;
    a.status.host = chain_host_status();
    if (hipMemsetAsync(a.counters, 0, (kChainMaxGroups + 1) * sizeof(unsigned), s) != hipSuccess) return -2;
    a.status.dev = a.counters + kChainMaxGroups;
    char label[72];
    std::snprintf(label, sizeof label, "gru_chain_bwd ms%d np%d T%d B%d H%d", ms, a.nprob, a.T, a.B, a.H);
    const double rows = (double)a.nprob * a.T * a.B;
    ProfScope prof(PROF_GRU_BWD, 2.0 * rows * 3.0 * a.H * a.H, s, label,
                   4.0 * (a.nprob * 3.0 * a.H * a.H + rows * a.H * (6 + 5 + 1)));
    const dim3 grid(chain::blocks_for(groups, a.members));
#define INET_CB(M, Q) hipLaunchKernelGGL((gru_chain_bwd_kernel<M, Q>), grid, dim3(256), 0, s, a)
    if (a.H == 512) { if (ms == 1) INET_CB(1, 24); else if (ms == 2) INET_CB(2, 24); else INET_CB(4, 24); }
    else { if (ms == 1) INET_CB(1, 12); else if (ms == 2) INET_CB(2, 12); else INET_CB(4, 12); }
#undef INET_CB
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
