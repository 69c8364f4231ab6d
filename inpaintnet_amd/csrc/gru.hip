// One GRU time step per launch, fused end to end: the recurrent contraction
// h_prev[B,H] x W_hh[3H,H]^T (and, for the tick decoder's second layer, the
// input contraction x[B,K2] x W_ih[3H,K2]^T in the same pass), the gather of
// embedding-derived gate pre-activations by token index, the sigmoid/tanh gate
// math, the state update, the dropout-masked copy for the next layer and the
// activations the backward pass needs -- no intermediate ever touches HBM.
// The backward step fuses dh = dgh_next x W_hh + (direct terms), the gate
// derivative math and the bias-gradient column sums.
//
// Why one launch per step and not one persistent kernel: every step needs the
// whole previous hidden state (an all-to-all over the 256 CUs).  On MI355X a
// dependent kernel boundary costs ~2 us (dispatch + L2 write-back), an in-kernel
// grid barrier 4-7 us (MI355X_MICROARCH.md price list), so the boundary IS the
// cheapest barrier.
//
// Geometry (gfx950): workgroup = 4 wavefronts, output tile = 32 or 64 batch rows
// x 16 hidden units x {r,z,n} gates -- all three gate columns of the same hidden
// units, so the gate math is register-local.  At B=256,H=512 one direction is
// 8 x 32 = 256 workgroups of 32 rows; bidirectional layers / the four beats run
// as several "problems" of the same launch with 64-row tiles, again ~256
// workgroups = one per CU.
//
// The four waves of a workgroup do NOT share a staged tile: each wave owns a
// contiguous quarter of K and streams its own MFMA fragments straight from L2
// into registers (ksplit.h: fragment-major operands, one contiguous KB per wave
// instruction; loads dealt between the MFMAs).  No LDS, no barrier in the main
// loop; the only LDS traffic is the final cross-wave reduction of the partial
// accumulators.  (Round-1 history in DESIGN.md section 8: LDS-staged 14.5 us,
// register-streamed row-major 12.4 us, fragment-major + streamed 10.8 us per
// two-direction step; 5.2 us of that is MFMA issue.)
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "ksplit.h"
#include "prof.h"

namespace {

using namespace ksplit;


// Optional in-kernel phase stamps (build with -DINET_STEP_TRACE; tools/trace_steps.py).  Not part of the product build.
#ifdef INET_STEP_TRACE
__device__ unsigned long long* g_trace = nullptr;   // [0] = slot counter, records of 8 x u64 from [8]
#define TRACE_DECL unsigned long long tr_[4] = {0, 0, 0, 0}, tc_[4] = {0, 0, 0, 0}, tx_[3] = {0, 0, 0}
#define TRACEX(i) do { if (threadIdx.x == 0) tx_[i] = clock64(); } while (0)
#define TRACE(i) do { if (threadIdx.x == 0) { tr_[i] = wall_clock64(); tc_[i] = clock64(); } } while (0)
#define TRACE_END(kind, ms)                                                                           \
    do {                                                                                              \
        if (threadIdx.x == 0 && g_trace) {                                                            \
            const unsigned long long sl = atomicAdd(g_trace, 1ull);                                   \
            if (sl < 400000ull) {                                                                     \
                unsigned long long* r = g_trace + 8 + sl * 8;                                         \
                r[0] = tr_[0]; r[1] = tr_[1]; r[2] = tr_[2]; r[3] = tr_[3];                           \
                r[4] = (unsigned long long)blockIdx.x | ((unsigned long long)blockIdx.y << 16) |      \
                       ((unsigned long long)(kind) << 32) | ((unsigned long long)(ms) << 40) |        \
                       ((unsigned long long)gridDim.y << 48);                                         \
                r[5] = ((tx_[0] - tc_[0]) & 0xfffffull) | (((tx_[1] - tc_[0]) & 0xfffffull) << 20) |      \
                       (((tx_[2] - tc_[0]) & 0xfffffull) << 40);                                      \
                r[6] = tc_[1] - tc_[0]; r[7] = tc_[3] - tc_[0];                                       \
            }                                                                                         \
        }                                                                                             \
    } while (0)
#else
#define TRACE_DECL
#define TRACEX(i)
#define TRACE(i)
#define TRACE_END(kind, ms)
#endif

// Row tiles are numbered over the concatenated problems: blockIdx.y = problem * tiles_per_prob + tile.
template <bool HAS_X, int MS, bool PK>
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(GruFwdBatch bt) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 4 * MS * 256];
    const int bcol = bt.rows_fastest ? blockIdx.y : blockIdx.x, brow_t = bt.rows_fastest ? blockIdx.x : blockIdx.y;
    const int prob = brow_t / bt.tiles_per_prob;
    const GruFwdProb& P = bt.p[prob];
    const int H = bt.H;
    const int t = threadIdx.x;
    const int j0 = bcol * TH;
    const int row0 = (brow_t % bt.tiles_per_prob) * (16 * MS);
    if (row0 >= P.B) return;
    TRACE_DECL;
    TRACE(0);

    f32x4 acc[MS][4];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Epilogue operands are requested inside the contraction, right after its first group of fragment loads (ksplit.h:
    // after_first_loads), so their latency hides under the MFMA phase.  Requested after the reduce they cost one
    // exposed round trip per row group (4.5 of 15.6 us at MS=4, profiles/r01_f).  Rows past the batch are clamped; their
    // results are never stored.  No arithmetic on the values up there: an add would pull a wait in front of the MFMAs.
    const int jc = j0 + (t & 15);
    long tok[MS];
    float pre_gd[MS][3], pre_gt[MS][3], pre_hp[MS], pre_mask[MS];
    float pb_hh[3], pb_ih[3] = {0.f, 0.f, 0.f}, pb_v[3] = {0.f, 0.f, 0.f};
    if (P.gi_table) {                    // the gather needs the token first: that one dependent load goes up front
#pragma unroll
        for (int p = 0; p < MS; ++p) tok[p] = P.idx[(long)min(row0 + ((t + 256 * p) >> 4), P.B - 1) * P.idx_stride];
    }
    auto prefetch = [&](auto tag) {
        constexpr int I = decltype(tag)::value;
        if constexpr (I == -1) {
            TRACEX(1);
            kernarg_touch(P.b_hh, P.b_ih, P.gi_vec, P.gi_dense, P.ld_gi, P.gi_table, P.ld_table, P.h_prev, P.ld_hprev,
                          P.hpk_prev, P.h_masked, P.mask, P.ld_mask);
        } else if constexpr (I == 0) {
#pragma unroll
            for (int g = 0; g < 3; ++g) pb_hh[g] = P.b_hh[g * H + jc];
            if (HAS_X && P.b_ih) {
#pragma unroll
                for (int g = 0; g < 3; ++g) pb_ih[g] = P.b_ih[g * H + jc];
            }
            if (P.gi_vec) {
#pragma unroll
                for (int g = 0; g < 3; ++g) pb_v[g] = P.gi_vec[g * H + jc];
            }
        } else if constexpr (I <= MS) {
            constexpr int p = I - 1;
            const int b = min(row0 + ((t + 256 * p) >> 4), P.B - 1);
#pragma unroll
            for (int g = 0; g < 3; ++g) pre_gd[p][g] = P.gi_dense ? P.gi_dense[(long)b * P.ld_gi + g * H + jc] : 0.f;
            // packed: the same lines the contraction streams (the row-major twin would be a second cold read)
            pre_hp[p] = PK ? P.hpk_prev[pk_offset(b, jc, H >> 4)] : P.h_prev[(long)b * P.ld_hprev + jc];
            pre_mask[p] = (P.h_masked && P.mask) ? P.mask[(long)b * P.ld_mask + jc] : 1.f;
#pragma unroll
            for (int g = 0; g < 3; ++g) pre_gt[p][g] = P.gi_table ? P.gi_table[tok[p] * P.ld_table + g * H + jc] : 0.f;
            if (I == MS) TRACEX(2);
        }
    };

    const int brow[3] = {j0, H + j0, 2 * H + j0};
    if (HAS_X) {
        const int slotx[3] = {0, 1, 2};          // r, z, gi_n
        if (PK) ksplit_segment<MS, 3, true>(acc, slotx, P.xpk, 0, row0, P.B, P.Wpk_ih, 0, brow, P.K2, t);
        else ksplit_segment<MS, 3>(acc, slotx, P.x, P.ldx, row0, P.B, P.W_ih, P.ld_wih, brow, P.K2, t);
    }
    const int sloth[3] = {0, 1, 3};              // r, z, gh_n
    TRACEX(0);
    if (PK) ksplit_segment<MS, 3, true>(acc, sloth, P.hpk_prev, 0, row0, P.B, P.Wpk_hh, 0, brow, H, t, prefetch);
    else ksplit_segment<MS, 3>(acc, sloth, P.h_prev, P.ld_hprev, row0, P.B, P.W_hh, (long)H, brow, H, t, prefetch);

    TRACE(1);
    float v[MS][4];
    reduce_waves<MS, 4>(acc, lds, t, v);
    TRACE(2);

#pragma unroll
    for (int p = 0; p < MS; ++p) {
        const int pos = t + 256 * p;
        const int b = row0 + (pos >> 4);
        const int j = jc;
        if (b >= P.B) continue;
        const float gr = v[p][0] + pb_ih[0] + pre_gd[p][0] + pre_gt[p][0] + pb_v[0] + pb_hh[0];
        const float gz = v[p][1] + pb_ih[1] + pre_gd[p][1] + pre_gt[p][1] + pb_v[1] + pb_hh[1];
        const float gn = v[p][2] + pb_ih[2] + pre_gd[p][2] + pre_gt[p][2] + pb_v[2];
        const float ghn = v[p][3] + pb_hh[2];
        const float r = sigmoid_f(gr);
        const float z = sigmoid_f(gz);
        const float n = tanh_f(gn + r * ghn);
        const float hp = pre_hp[p];
        const float hn = (1.f - z) * n + z * hp;
        P.h_new[(long)b * P.ld_hnew + j] = hn;
        if (P.h_copy) P.h_copy[(long)b * P.ld_hc + j] = hn;
        if (P.h_masked) P.h_masked[(long)b * P.ld_hm + j] = hn * pre_mask[p];
        if (P.hpk_new) P.hpk_new[pk_offset(b, j, H >> 4)] = hn;
        if (P.hmpk_new) P.hmpk_new[pk_offset(b, j, H >> 4)] = hn * pre_mask[p];
        if (P.sv_r) {
            const long o = (long)b * H + j;
            P.sv_r[o] = r; P.sv_z[o] = z; P.sv_n[o] = n; P.sv_ghn[o] = ghn; P.sv_hprev[o] = hp;
        }
    }
    TRACE(3);
    TRACE_END(HAS_X ? 1 : 0, MS);
}

// Backward of one step.
//   dh   = dgh_next * W_hh + dhz_next + dout + dout2          (gradient wrt this step's output h)
//   dn   = dh (1-z); dz = dh (hprev - n); dhz = dh z
//   dn_pre = dn (1-n^2); dz_pre = dz z(1-z); dr_pre = dn_pre ghn r(1-r)
//   dgi = [dr_pre, dz_pre, dn_pre]      dgh = [dr_pre, dz_pre, dn_pre r]
//   db_ih += colsum(dgi) ; db_hh += colsum(dgh)               (tile-reduced, one atomic per column per workgroup)
// Tile = 16*MS batch rows x 16*NC hidden columns.  The contraction runs over K = 3H, so the operand traffic per
// output is (rows + cols) * 3H: squarer tiles (NC = 2) move 20-33 % fewer bytes through the CU's address path than
// 16-column ones for the same number of outputs (64x16 -> 32x32: 480 -> 384 KB; 128x16 -> 64x32: 864 -> 576 KB).
template <int MS, int NC, bool PK>
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(GruBwdBatch bt) {
    __shared__ __attribute__((aligned(16))) float lds[(4 * MS * NC * 256 > 1024 * NC) ? 4 * MS * NC * 256 : 1024 * NC];
    const int bcol = bt.rows_fastest ? blockIdx.y : blockIdx.x, brow_t = bt.rows_fastest ? blockIdx.x : blockIdx.y;
    const int prob = brow_t / bt.tiles_per_prob;
    const GruBwdProb& P = bt.p[prob];
    const int H = bt.H;
    const int t = threadIdx.x;
    const int j0 = bcol * (TH * NC);
    const int row0 = (brow_t % bt.tiles_per_prob) * (16 * MS);
    if (row0 >= P.B) return;
    TRACE_DECL;
    TRACE(0);

    // Epilogue operands: direct gradient terms and the saved gates of this step, requested from inside the contraction
    // (ksplit.h: after_first_loads) like the forward kernel's.  Output (p, a) of this thread: row (t + 256p) >> 4,
    // column j0 + 16a + (t & 15).
    const int jc = j0 + (t & 15);
    float pd[MS][NC][3], psv[MS][NC][5];
    auto prefetch = [&](auto tag) {
        constexpr int I = decltype(tag)::value;
        if constexpr (I == -1) {
            kernarg_touch(P.dhz_next, P.dout, P.ld_dout, P.dout2, P.ld_dout2, P.sv_r, P.sv_z, P.sv_n, P.sv_ghn, P.sv_hprev,
                          P.dh_out, P.ld_dhout, P.dh_out_accumulate);
        } else if constexpr (I >= 1 && I <= MS) {
            constexpr int p = I - 1;
            const int b = min(row0 + ((t + 256 * p) >> 4), P.B - 1);
#pragma unroll
            for (int a = 0; a < NC; ++a) {
                const int j = jc + 16 * a;
                const long o = (long)b * H + j;
                pd[p][a][0] = P.dhz_next ? P.dhz_next[o] : 0.f;
                pd[p][a][1] = P.dout ? P.dout[(long)b * P.ld_dout + j] : 0.f;
                pd[p][a][2] = P.dout2 ? P.dout2[(long)b * P.ld_dout2 + j] : 0.f;
                psv[p][a][0] = P.sv_r ? P.sv_r[o] : (P.dh_out_accumulate ? P.dh_out[(long)b * P.ld_dhout + j] : 0.f);
                psv[p][a][1] = P.sv_r ? P.sv_z[o] : 0.f;
                psv[p][a][2] = P.sv_r ? P.sv_n[o] : 0.f;
                psv[p][a][3] = P.sv_r ? P.sv_ghn[o] : 0.f;
                psv[p][a][4] = P.sv_r ? P.sv_hprev[o] : 0.f;
            }
        }
    };

    float v[MS][NC];
#pragma unroll
    for (int p = 0; p < MS; ++p)
#pragma unroll
        for (int a = 0; a < NC; ++a) v[p][a] = 0.f;
    if (P.dgh_next) {
        f32x4 acc[MS][4];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        int brow[NC], slot[NC];
#pragma unroll
        for (int a = 0; a < NC; ++a) { brow[a] = j0 + 16 * a; slot[a] = a; }
        if (PK) ksplit_segment<MS, NC, true>(acc, slot, P.dghpk_next, 0, row0, P.B, P.Wpk_hhT, 0, brow, 3 * H, t, prefetch);
        else ksplit_segment<MS, NC>(acc, slot, P.dgh_next, P.ld_dgh, row0, P.B, P.W_hhT, (long)3 * H, brow, 3 * H, t, prefetch);
        TRACE(1);
        reduce_waves<MS, NC>(acc, lds, t, v);
    }
    else {
        prefetch(HookTag<-1>{});
        hook_pieces<MS + 1>(prefetch);
    }
    TRACE(2);
    float bs[NC][4];                               // this thread's column partials: dr, dz, dn, dn*r
#pragma unroll
    for (int a = 0; a < NC; ++a) bs[a][0] = bs[a][1] = bs[a][2] = bs[a][3] = 0.f;
#pragma unroll
    for (int p = 0; p < MS; ++p) {
        const int pos = t + 256 * p;
        const int b = row0 + (pos >> 4);
        if (b >= P.B) continue;
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            const int j = jc + 16 * a;
            const long o = (long)b * H + j;
            const float dh = v[p][a] + pd[p][a][0] + pd[p][a][1] + pd[p][a][2];
            if (!P.sv_r) {
                P.dh_out[(long)b * P.ld_dhout + j] = P.dh_out_accumulate ? psv[p][a][0] + dh : dh;
                continue;
            }
            const float r = psv[p][a][0], z = psv[p][a][1], n = psv[p][a][2], ghn = psv[p][a][3], hp = psv[p][a][4];
            const float dn_pre = dh * (1.f - z) * (1.f - n * n);
            const float dz_pre = dh * (hp - n) * z * (1.f - z);
            const float dr_pre = dn_pre * ghn * r * (1.f - r);
            P.dhz[o] = dh * z;
            float* gi = P.dgi + (long)b * P.ld_dgi;
            gi[j] = dr_pre; gi[H + j] = dz_pre; gi[2 * H + j] = dn_pre;
            float* gh = P.dgh + (long)b * P.ld_dghout;
            gh[j] = dr_pre; gh[H + j] = dz_pre; gh[2 * H + j] = dn_pre * r;
            if (P.dghpk) {
                const int S3 = (3 * H) >> 4;
                P.dghpk[pk_offset(b, j, S3)] = dr_pre;
                P.dghpk[pk_offset(b, H + j, S3)] = dz_pre;
                P.dghpk[pk_offset(b, 2 * H + j, S3)] = dn_pre * r;
            }
            bs[a][0] += dr_pre; bs[a][1] += dz_pre; bs[a][2] += dn_pre; bs[a][3] += dn_pre * r;
        }
    }
    if (P.sv_r && P.db_ih) {
        // thread t holds column (t & 15) of each group for rows (t >> 4) + 16p: reduce the 16 row-threads per column
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NC; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) lds[(a * 4 + e) * 256 + t] = bs[a][e];
        __syncthreads();
        if (t < 64 * NC) {
            const int a = t >> 6, e = (t >> 4) & 3, c = t & 15;
            float s = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) s += lds[(a * 4 + e) * 256 + rr * 16 + c];
            const int j = j0 + 16 * a + c;
            if (e == 0) { unsafeAtomicAdd(P.db_ih + j, s); unsafeAtomicAdd(P.db_hh + j, s); }
            else if (e == 1) { unsafeAtomicAdd(P.db_ih + H + j, s); unsafeAtomicAdd(P.db_hh + H + j, s); }
            else if (e == 2) unsafeAtomicAdd(P.db_ih + 2 * H + j, s);
            else unsafeAtomicAdd(P.db_hh + 2 * H + j, s);
        }
    }
    TRACE(3);
    TRACE_END(2, MS);
}

// Output projection of one decoder tick, fused: weights[:, t, :] = ReLU(h_top[B,H] x W_out[V,H]^T + b) and the
// next token = argmax (lowest index on ties).  Same register-streamed K-split as the step kernels; V = 16*NB <= 64.
// PK: h and W are the fragment-major twins (the tick step's packed h ring, W_out packed once per call).
template <int NB, bool PK>
__global__ __launch_bounds__(256) void logits_argmax_kernel(const float* __restrict__ h, long ldh, int B, int H,
                                                            const float* __restrict__ W, const float* __restrict__ bias,
                                                            float* __restrict__ out, long ldo,
                                                            long long* __restrict__ samples, long sstride) {
    __shared__ __attribute__((aligned(16))) float lds[4 * NB * 512 + 32 * 64];
    float* stash = lds + 4 * NB * 512;
    const int t = threadIdx.x;
    const int row0 = blockIdx.x * TM_ROWS;
    f32x4 acc[2][4];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    int brow[NB], slot[NB];
#pragma unroll
    for (int g = 0; g < NB; ++g) { brow[g] = 16 * g; slot[g] = g; }
    ksplit_segment<2, NB, PK>(acc, slot, h, ldh, row0, B, W, (long)H, brow, H, t);
    float v[2][NB];
    reduce_waves<2, NB>(acc, lds, t, v);
    constexpr int V = 16 * NB;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int pos = t + 256 * p;
        const int r = pos >> 4, c = pos & 15;
        const int b = row0 + r;
#pragma unroll
        for (int a = 0; a < NB; ++a) {
            float x = v[p][a] + bias[16 * a + c];
            x = x > 0.f ? x : 0.f;
            stash[r * 64 + 16 * a + c] = x;
            if (b < B) out[(long)b * ldo + 16 * a + c] = x;
        }
    }
    if (samples) {
        __syncthreads();
        if (t < TM_ROWS && row0 + t < B) {
            float m = stash[t * 64];
            int am = 0;
            for (int j = 1; j < V; ++j) {
                const float x = stash[t * 64 + j];
                if (x > m) { m = x; am = j; }
            }
            samples[(long)(row0 + t) * sstride] = am;
        }
    }
}

// Rows per workgroup = 16*MS.  One problem of B=256 fills the chip with MS=2 (256 workgroups).  With 2 or 4 problems
// per launch (directions, beats) the per-CU L2->L1 load path is the limit (~32 GB/s per CU measured), so bigger row
// tiles are used to keep ~256 workgroups while loading each W slice once per 64 / 128 rows instead of per 32.
// (Tried: 32-row tiles capped at 256 registers so that two workgroups share a CU and overlap each other's load and
// epilogue phases -- 6.14 vs 5.77 ms per teacher-forced step, the doubled W traffic and the longer tail lose.)
int pick_ms(int nprob, int maxB, int H, int ms_max) {
    int ms = 2;
    for (int cand = 4; cand <= ms_max; cand *= 2) {
        const long wgs = (long)nprob * ((maxB + 16 * cand - 1) / (16 * cand)) * (H / TH);
        if (wgs >= 256) ms = cand;
    }
    return ms;
}

}  // namespace

#ifdef INET_STEP_TRACE
extern "C" int inet_debug_trace_set(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

int launch_gru_fwd(const GruFwdBatch& bin, hipStream_t s) {
    GruFwdBatch b = bin;
    if (b.H % TH != 0 || b.nprob < 1 || b.nprob > 4) return -1;
    int maxB = 0;
    bool hasx = b.p[0].x != nullptr;
    for (int i = 0; i < b.nprob; ++i) {
        if (b.p[i].B > maxB) maxB = b.p[i].B;
        if ((b.p[i].x != nullptr) != hasx) return -1;
    }
    if (maxB <= 0) return 0;
    const int ms = pick_ms(b.nprob, maxB, b.H, 4);
    b.tiles_per_prob = (maxB + 16 * ms - 1) / (16 * ms);
    // Row tiles on the fastest grid axis: consecutive workgroups go to different XCDs, so each XCD then owns whole
    // row tiles (one direction's W stays in its 4 MB L2, it reads only its own rows of the fresh hidden state) instead
    // of column slices of every row.  ~2.5 % on these kernels; not with the extra W_ih (6 MB per XCD would not fit).
    b.rows_fastest = !hasx;
    dim3 grid(b.H / TH, b.tiles_per_prob * b.nprob, 1);
    if (b.rows_fastest) grid = dim3(b.tiles_per_prob * b.nprob, b.H / TH, 1);
    double fl = 0;
    for (int i = 0; i < b.nprob; ++i) fl += 2.0 * b.p[i].B * 3.0 * b.H * (b.H + (hasx ? b.p[i].K2 : 0));
    bool pk = b.H % 256 == 0;
    for (int i = 0; i < b.nprob; ++i) {
        const GruFwdProb& P = b.p[i];
        if (!P.hpk_prev || !P.Wpk_hh) pk = false;
        if (hasx && (!P.xpk || !P.Wpk_ih || P.K2 % 256 != 0)) pk = false;
    }
    char label[64];                               // names the template instantiation (tests assert which ones ran)
    std::snprintf(label, sizeof label, "gru_fwd x%d ms%d pk%d np%d B%d H%d", (int)hasx, ms, (int)pk, b.nprob, maxB, b.H);
    // algorithmic bytes: W_hh (+ W_ih) once per problem; per row h_prev in, h_new out, gi in, the 5 backward saves
    double by = 0;
    for (int i = 0; i < b.nprob; ++i)
        by += 4.0 * (3.0 * b.H * (b.H + (hasx ? b.p[i].K2 : 0)) +
                     (double)b.p[i].B * b.H * (2 + (b.p[i].gi_dense ? 3 : 0) + (b.p[i].sv_r ? 5 : 0) + (hasx ? 1 : 0)));
    ProfScope prof(PROF_GRU_FWD, fl, s, label, by);
#define DISPATCH_FWD(X, M)                                                                                       \
    do {                                                                                                     \
        if (pk) hipLaunchKernelGGL((gru_step_fwd_kernel<X, M, true>), grid, dim3(256), 0, s, b);             \
        else hipLaunchKernelGGL((gru_step_fwd_kernel<X, M, false>), grid, dim3(256), 0, s, b);               \
    } while (0)
    if (hasx) { if (ms == 2) DISPATCH_FWD(true, 2); else DISPATCH_FWD(true, 4); }
    else { if (ms == 2) DISPATCH_FWD(false, 2); else DISPATCH_FWD(false, 4); }
#undef DISPATCH_FWD
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_gru_bwd(const GruBwdBatch& bin, hipStream_t s) {
    GruBwdBatch b = bin;
    if (b.H % TH != 0 || b.nprob < 1 || b.nprob > 4) return -1;
    int maxB = 0;
    for (int i = 0; i < b.nprob; ++i) if (b.p[i].B > maxB) maxB = b.p[i].B;
    if (maxB <= 0) return 0;
    int ms = pick_ms(b.nprob, maxB, b.H, 8);
    // 128-row tiles (4 problems per launch) become 64 rows x 2 column groups (see the kernel's tile note): 22.5 -> 21.0 us.
    // 64-row tiles stay 64x16: as 32x32 they measured slower (13.5 vs 11.9 us) despite the smaller traffic.
    int nc = 1;
    if (ms >= 8 && b.H % (2 * TH) == 0) { ms /= 2; nc = 2; }
    b.tiles_per_prob = (maxB + 16 * ms - 1) / (16 * ms);
    b.rows_fastest = 1;
    dim3 grid(b.tiles_per_prob * b.nprob, b.H / (TH * nc), 1);
    double fl = 0;
    for (int i = 0; i < b.nprob; ++i) if (b.p[i].dgh_next) fl += 2.0 * b.p[i].B * 3.0 * b.H * b.H;
    bool pk = b.H % 256 == 0;
    for (int i = 0; i < b.nprob; ++i)
        if (b.p[i].dgh_next && (!b.p[i].dghpk_next || !b.p[i].Wpk_hhT)) pk = false;
    char label[64];
    std::snprintf(label, sizeof label, "gru_bwd ms%d nc%d pk%d np%d B%d H%d", ms, nc, (int)pk, b.nprob, maxB, b.H);
    // algorithmic bytes: W_hh^T once per problem; per row dgh_next (3H) in, dgi + dgh (6H) out, 5 saves + dhz in/out
    double by = 0;
    for (int i = 0; i < b.nprob; ++i)
        by += 4.0 * ((b.p[i].dgh_next ? 3.0 * b.H * b.H : 0.0) + (double)b.p[i].B * b.H * (3 + 6 + 5 + 2 + 1));
    ProfScope prof(PROF_GRU_BWD, fl, s, label, by);
#define DISPATCH_BWD(M, C)                                                                                       \
    do {                                                                                                     \
        if (pk) hipLaunchKernelGGL((gru_step_bwd_kernel<M, C, true>), grid, dim3(256), 0, s, b);             \
        else hipLaunchKernelGGL((gru_step_bwd_kernel<M, C, false>), grid, dim3(256), 0, s, b);               \
    } while (0)
    if (nc == 2) DISPATCH_BWD(4, 2);
    else if (ms == 2) DISPATCH_BWD(2, 1);
    else DISPATCH_BWD(4, 1);
#undef DISPATCH_BWD
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// returns 1 if the fused path does not apply (caller falls back to GEMM + argmax)
int launch_logits_argmax(const float* h, long ldh, int B, int H, const float* W, const float* bias, int V, float* out,
                         long ldo, long long* samples, long sstride, hipStream_t s, const float* hpk, const float* Wpk) {
    if (V % 16 != 0 || V > 64 || H % TH != 0) return 1;
    dim3 grid((B + TM_ROWS - 1) / TM_ROWS);
    ProfScope prof(PROF_GEMM, 2.0 * B * V * H, s, "logits_argmax", 4.0 * ((double)V * H + (double)B * (H + V)));
    const bool pk = hpk && Wpk && H % 256 == 0;
    const float* a = pk ? hpk : h;
    const float* w = pk ? Wpk : W;
#define DISPATCH_LOGITS(NBV)                                                                                                \
    do {                                                                                                                \
        if (pk) hipLaunchKernelGGL((logits_argmax_kernel<NBV, true>), grid, dim3(256), 0, s, a, ldh, B, H, w, bias, out, ldo, samples, sstride);  \
        else hipLaunchKernelGGL((logits_argmax_kernel<NBV, false>), grid, dim3(256), 0, s, a, ldh, B, H, w, bias, out, ldo, samples, sstride);    \
    } while (0)
    switch (V / 16) {
        case 1: DISPATCH_LOGITS(1); break;
        case 2: DISPATCH_LOGITS(2); break;
        case 3: DISPATCH_LOGITS(3); break;
        default: DISPATCH_LOGITS(4); break;
    }
#undef DISPATCH_LOGITS
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
