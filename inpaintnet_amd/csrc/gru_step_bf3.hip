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

constexpr int WM = 2, WN = 4, RM = 4, RN = 3;          // 8 waves: 2 x 4; wave tile 64 rows x (16 units x 3 gates)
constexpr int TMB = WM * RM, TNB = WN * RN;            // 8 row blocks of h, 12 (interleaved) row blocks of W
constexpr int NW = WM * WN;
constexpr int STAGE = (TMB + TNB) * 3 * 1024;
constexpr int CH = (TMB + TNB) * 3;
constexpr int CPW = (CH + NW - 1) / NW;

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
struct StepArgs { int H, B, nprob, zero_state, xm, xn; StepProb p[2]; };   // xm x xn: the XCDs of one problem over its (tm, tn) tiles; 0: contiguous ranges

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

template <bool TAB, bool DEN, bool SAVE>
__global__ __launch_bounds__(64 * NW) void gru_step_bf3_kernel(StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w / WN, wn = w % WN;
    const int H = a.H, KB = H >> 5;
    const int tiles_m = a.B / (TMB * 16), tiles_n = H / 64;
    // consecutive workgroup ids go round-robin over the 8 XCDs: each XCD takes a contiguous range of the (problem, tm, tn)
    // list -- 4 row tiles x all 8 unit tiles of one direction at B = 2048: 1.5 MB of h pieces + that direction's 4.7 MB of W
    const int nb = gridDim.x, id = blockIdx.x;
    const int tid = (nb % 8 == 0) ? (id % 8) * (nb / 8) + id / 8 : id;
    const int per_prob = tiles_m * tiles_n;
    int prob = tid / per_prob;
    const int v = tid - prob * per_prob;
    int tm = v / tiles_n, tn = v - tm * tiles_n;
    if (a.xm > 0) {
        const int xpp = 8 / a.nprob, x = id & 7, slot = id >> 3, xl = x % xpp;
        const int rm = tiles_m / a.xm, rn = tiles_n / a.xn;
        prob = x / xpp;
        tm = (xl / a.xn) * rm + slot / rn;
        tn = (xl % a.xn) * rn + slot % rn;
    }
    const StepProb& P = a.p[prob];
    const int c = lane & 15, q = lane >> 4;
    const int j = tn * 64 + wn * 16 + c;                       // this lane's hidden unit

    // ---- epilogue operands, requested before the contraction (their latency hides behind it) ----
    // (every batch of loads in a loop of its own: inside one loop hipcc waits for each token before it issues that row's table
    //  loads -- 16 dependent round trips in front of the contraction, 20 us of a 52 us step in the first build)
    float gr[RM][4], gz[RM][4], gn[RM][4], hp[RM][4], mk[RM][4];
    long rowv[RM][4];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) rowv[i][r] = (long)(tm * TMB + wm * RM + i) * 16 + 4 * q + r;
    long tok[RM][4];
    if (TAB) {
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) tok[i][r] = P.idx[rowv[i][r] * P.idx_bs];
    }
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { gr[i][r] = 0.f; gz[i][r] = 0.f; gn[i][r] = 0.f; hp[i][r] = 0.f; mk[i][r] = 1.f; }
    if (DEN) {
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* dp = P.gi + rowv[i][r] * P.gi_ld + j;
                gr[i][r] = dp[0]; gz[i][r] = dp[H]; gn[i][r] = dp[2 * H];
            }
    }
    if (P.hprev) {
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) hp[i][r] = P.hprev[rowv[i][r] * P.hprev_ld + j];
    }
    if (P.mask) {
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) mk[i][r] = P.mask[rowv[i][r] * P.mask_ld + j];
    }
    float tg[TAB ? RM : 1][4][3];
    if (TAB) {
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* tp = P.table + tok[i][r] * P.table_ld + j;
                tg[i][r][0] = tp[0]; tg[i][r][1] = tp[H]; tg[i][r][2] = tp[2 * H];
            }
    }
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
        fill(0, smem);
        __syncthreads();
        for (int kb = 0; kb < KB; ++kb) {
            const unsigned char* sa = smem + (kb & 1) * STAGE + lane * 16;
            const unsigned char* sb = sa + TMB * 3 * 1024;
            if (kb + 1 < KB) fill(kb + 1, smem + ((kb + 1) & 1) * STAGE);
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
            __syncthreads();
        }
    }

    // ---- GRU cell: lane (c, q) holds rows 4q + r, unit c of every 16 x 16 tile; tiles g = 0, 1, 2 are the gates r, z, n ----
    // Every result leaves through wave-private 16 x 16 transpose tiles in LDS (the stages are free now): the f32 arrays as ONE
    // 16-byte store per lane and tile (lane = row L / 4, units 4 (L % 4) ..: a whole 1 KB tile per instruction -- scalar stores of
    // the accumulator layout were store-issue-bound: 16 per lane and array), the pieces as 8 consecutive units per lane.
    constexpr int NT = SAVE ? 7 : 2;                                   // tiles per wave: h, h * mask [, r, z, n, ghn, hprev]
    float* const xt = reinterpret_cast<float*>(smem) + w * (NT * 256);
    const int kbj = tn * 2 + (wn >> 1);                               // k block of the piece layouts this wave's 16 units fall into
    const int prow = lane & 15, pgrp = (lane >> 4) & 1;
    const int plane = ((2 * (wn & 1) + pgrp) * 16 + prow) * 16;       // byte offset of (row, 8-unit group) inside the fragment
    const int vrow = lane >> 2, vcol = (lane & 3) * 4;                // the lane's 4 consecutive units of one row (vector stores)
    const int j0 = tn * 64 + wn * 16;
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        const int rb = tm * TMB + wm * RM + i;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ghn = acc[i][2][r] + bh[2];
            float xr = gr[i][r], xz = gz[i][r], xn = gn[i][r];
            if (TAB) { xr += tg[i][r][0]; xz += tg[i][r][1]; xn += tg[i][r][2]; }
            const float rg = sigmoid_f(acc[i][0][r] + xr + bv[0] + bh[0]);
            const float zg = sigmoid_f(acc[i][1][r] + xz + bv[1] + bh[1]);
            const float ng = tanh_f(xn + bv[2] + rg * ghn);
            const float hprev = hp[i][r];
            const float hn = (1.f - zg) * ng + zg * hprev;
            const int o = (4 * q + r) * 16 + c;
            xt[o] = hn; xt[256 + o] = hn * mk[i][r];
            if (SAVE) { xt[512 + o] = rg; xt[768 + o] = zg; xt[1024 + o] = ng; xt[1280 + o] = ghn; xt[1536 + o] = hprev; }
        }
        __builtin_amdgcn_wave_barrier();
        const long vr = (long)rb * 16 + vrow;
        const f32x4 hv4 = *reinterpret_cast<const f32x4*>(xt + vrow * 16 + vcol);
        *reinterpret_cast<f32x4*>(P.out + vr * P.out_ld + j0 + vcol) = hv4;
        if (P.outm) *reinterpret_cast<f32x4*>(P.outm + vr * P.outm_ld + j0 + vcol) = *reinterpret_cast<const f32x4*>(xt + 256 + vrow * 16 + vcol);
        if (P.hlast) *reinterpret_cast<f32x4*>(P.hlast + vr * P.hlast_ld + j0 + vcol) = hv4;
        if (SAVE) {
#pragma unroll
            for (int a5 = 0; a5 < 5; ++a5)
                *reinterpret_cast<f32x4*>(P.sv + a5 * P.sv_astride + vr * H + j0 + vcol) =
                    *reinterpret_cast<const f32x4*>(xt + (2 + a5) * 256 + vrow * 16 + vcol);
        }
        if (lane < 32) {
            if (P.An) pieces8_store(xt + prow * 16 + 8 * pgrp, P.An + ((long)rb * KB + kbj) * 1024 + plane, P.a_piece);
            if (P.em) pieces8_store(xt + 256 + prow * 16 + 8 * pgrp,
                                    P.em + ((P.em_rb0 + rb) * P.em_kb + P.em_kb0 + kbj) * 1024 + plane, P.em_piece);
        }
    }
}

template <bool TAB, bool DEN, bool SAVE>
int launch_one(const StepArgs& a, int grid, hipStream_t s) {
    auto kern = &gru_step_bf3_kernel<TAB, DEN, SAVE>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), (size_t)2 * STAGE, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

namespace { int g_min_tiles = -1; }
void gru_step_bf3_set_min_tiles(int n) { g_min_tiles = n < 0 ? 0 : n; }
bool gru_step_bf3_ok(int H, int B, int T, int nd) {
    if (g_min_tiles < 0) { const char* v = std::getenv("INET_STEP_BF3_MIN_TILES"); g_min_tiles = v ? std::atoi(v) : 256; if (g_min_tiles < 0) g_min_tiles = 0; }
    const int min_tiles = g_min_tiles;
    if (bf3_mode() == 0 || min_tiles <= 0) return false;
    if (H < 64 || H % 64 || B < 128 || B % 128 || T < 1 || nd < 1 || nd > 2) return false;
    if ((double)T * B * 6.0 * H >= 2.0e9) return false;
    return nd * (B / 128) * (H / 64) >= min_tiles;
}

size_t gru_step_bf3_w_bytes(int H) { return bf3_bytes(3L * H, H); }

int gru_step_bf3_split_w(int H, const float* W_hh, unsigned char* Wp, hipStream_t s) {
    // gate g's H rows (H / 16 row blocks) -> row blocks 3u + g
    for (int g = 0; g < 3; ++g)
        if (bf3_split_strided(W_hh + (long)g * H * H, H, H, H, Wp, (long)bf3_piece_bytes(3L * H, H), H / 32, g, 3, s) != 0) return -2;
    return 0;
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
    if (!tab && !den) return -1;
    const int grid = nd * (B / 128) * (H / 64);
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
            if (P.outm) {
                Q.outm = P.outm + (long)tt * P.ts_outm; Q.outm_ld = P.ld_outm;
                Q.mask = P.mask ? P.mask + (long)tt * P.ts_mask : nullptr; Q.mask_ld = P.ld_mask;
            }
            if (P.hlast && step == T - 1) { Q.hlast = P.hlast; Q.hlast_ld = P.ld_hlast; }
            if (P.sv) { Q.sv = P.sv + (long)tt * (P.sv_ts ? P.sv_ts : (long)B * H); Q.sv_astride = P.sv_astride; }
            if (P.em.rows) {
                Q.em = P.em.rows; Q.em_piece = P.em.rows_piece; Q.em_kb = P.em.rows_kb; Q.em_kb0 = P.em.rows_kb0;
                Q.em_rb0 = ((long)tt * P.em.B_full + P.em.r0) / 16;
            }
        }
        a.zero_state = zero ? 1 : 0;
        {   // XCD arrangement (A/B switch INET_STEP_BF3_MAP="xm,xn"; default: contiguous ranges)
            static const int mx = [] { const char* v = std::getenv("INET_STEP_BF3_MAP"); return v ? std::atoi(v) : 0; }();
            static const int mn = [] { const char* v = std::getenv("INET_STEP_BF3_MAP"); const char* c = v ? std::strchr(v, ',') : nullptr; return c ? std::atoi(c + 1) : 0; }();
            const int tiles_m = B / 128, tiles_n = H / 64, xpp = 8 / nd;
            if (mx > 0 && mn > 0 && 8 % nd == 0 && mx * mn == xpp && tiles_m % mx == 0 && tiles_n % mn == 0 && grid % 8 == 0) { a.xm = mx; a.xn = mn; }
        }
        // algorithmic bytes of a step: W pieces once, state pieces in and out, gi, out (+ saves)
        ProfScope prof(PROF_GRU_FWD, zero ? 0.0 : 2.0 * nd * B * 3.0 * H * H, s, label,
                       nd * (6.0 * 3 * H * H + 12.0 * B * H + 4.0 * B * 3 * H + 4.0 * B * H * (save ? 7 : 2)));
        int rc;
        if (tab && den) rc = save ? launch_one<true, true, true>(a, grid, s) : launch_one<true, true, false>(a, grid, s);
        else if (tab) rc = save ? launch_one<true, false, true>(a, grid, s) : launch_one<true, false, false>(a, grid, s);
        else rc = save ? launch_one<false, true, true>(a, grid, s) : launch_one<false, true, false>(a, grid, s);
        if (rc != 0) return rc;
    }
    return 0;
}
