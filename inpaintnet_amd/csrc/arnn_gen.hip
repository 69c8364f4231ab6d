// AnticipationRNN's free-running token pass as ONE persistent launch (round 5).
//
// The reference's free-running forward (AnticipationRNN/anticipation_rnn_gauss_reg_model.py:190-259) feeds the argmax of BATCH
// ELEMENT 0 back to the whole batch (:253-256): the L tokens depend on that one row, and producing them is strictly sequential --
// per tick  [embedding of the previous token | constraint output of the tick] -> LSTM 0 -> LSTM 1 -> ReLU(linear_1) -> note head ->
// argmax.  Round 4 queued that as four small launches per tick (lstm.hip arnn_generate): 14.3 us per tick = 4 x 3.6 us per dependent
// launch, 5.5 ms of the 14.4 ms free-running training step.  Here 13 workgroups of 512 threads stay resident for the whole
// sequence, every one with its weights in REGISTERS (a 256 x 256 f32 slice: 128 VGPRs per thread, see "Threads and products"
// below), and 1-KB vectors move between them as 8-byte {value, tick} granules (one relaxed agent-scope 64-bit store / polled load
// per element: MI355X_MICROARCH.md "handoff-1to1", form R2 -- the tag travels with the value, nothing to order):
//
//   C   (1)      W1 = linear_1, W2 = note head.  Per tick: L0's gates = pre[t] + T0[tok] + hh0 (all precomputed or received:
//                NO product on this edge), L0's cell, publish h0_t; wait for h1_t; u = ReLU(W1 h1 + b1); logits = W2 u + b2;
//                argmax (numpy order: NaN is the maximum, lowest index among equals) -> tok_t
//   A_k (4)      W_hh0 rows of units 64k..64k+63: hh0 for tick t+1 = W_hh0 h0_t           (off the critical path)
//   Bi_k (4)     W_ih1 rows of units 64k..: gates of L1 = W_ih1 h0_t + b_ih1 + hh1, L1's cell, publish h1_t
//   Bh_k (4)     W_hh1 rows: hh1 for tick t+1 = W_hh1 h1_t + b_hh1                        (off the critical path)
//
// so a tick is TWO hand-offs on its critical path (C -> Bi -> C) with one 256 x 256 product behind each, instead of four
// launches: the input-side product of layer 0 does not depend on the sequence (pre[t] = W_ih0[:, E:] oc_t + b_ih0 + b_hh0 for all
// ticks and T0[v] = W_ih0[:, :E] emb[v] for every token are made by one small launch in front), and both recurrent products are
// started the moment their state exists and arrive before they are needed (their granules are REQUESTED a phase early and looked at
// late: a granule read is a ~0.8 us round trip through the memory side).  The gate summands hh0 / hh1 have ONE reader per granule
// (the cell thread of the unit) and are single-buffered: their writer's next write needs the whole next state, which the reader
// publishes behind its read.  The state vectors h0 / h1 have FIVE readers (the four recurrent-side workgroups and the next stage)
// while their writer's next write waits only for the summands of its own unit, from ONE of the four: they alternate between two
// slots by tag parity -- the overwrite of h_t happens at h_{t+2}, which needs hh_{t+2} of the unit, which needs ALL of h_{t+1},
// whose every unit needed hh_{t+1} from its own recurrent-side workgroup: all four have read h_t by then.  (With one slot a
// recurrent-side workgroup a tick late found tag t + 2 where it looked for t + 1 and ran into its bound: ADVICE r05.)
// Measured (tools/arnn_token_pass.py, INET_ARNN_GEN_STAMPS=1; profiles/r05_arnn_token_pass.txt): 4.0 us per tick = 0.74 + 0.89 us
// for the two hand-offs, 0.4 us per 256 x 256 product, 0.25 each for the head's product and the argmax, the rest barriers and the
// cells: 384 ticks in 1.5 ms (round 4: 5.5).
// Shapes: H = U = 256 (the reference's configuration), V <= 128; anything else keeps the per-tick launches.  Every spin is bounded.
#include <cstdio>
#include <cstdlib>
#include "chain.h"
#define INET_GRANULE_KID chain::K_ARNN_GEN
#include "granule.h"
#include "prof.h"
#include "lstm.h"
#include "pointwise.h"

namespace {
using namespace granule;

constexpr int GH = 256, G4 = 4 * GH;
constexpr int kExXcc = 4 * GH + 2 * G4;          // granules behind the exchange: the workgroups' XCC ids (granule::same_xcd)

struct GenArgs {
    int L, V, E, K0, stride, near;                   // K0 = E + Hc: row stride of W_ih0; stride: block b works iff b % stride == 0, role b / stride
    const float* emb; const float* W_ih0;
    const float* W_hh0; const float* W_ih1; const float* b_ih1; const float* W_hh1; const float* b_hh1;
    const float* W1; const float* b1; const float* W2; const float* b2;
    const float* pre; const float* T0;           // [L][4H], [V][4H]
    const float* hc_init; const long long* first_tok; long long* tokens;
    unsigned long long* ex;                      // granules: h0 [2][256] | h1 [2][256] | hh0 [1024] | hh1 [1024]  (h: slot = tag & 1)
    unsigned long long* stamps;                  // diagnostics (INET_ARNN_GEN_STAMPS=1): [C, Bi_0][L][8] wall-clock ticks (10 ns), or null
    chain::Status status;
};
#define GEN_STAMP(who, t, i) do { if (a.stamps && tid == 0) a.stamps[((long)(who) * a.L + (t)) * 8 + (i)] = wall_clock64(); } while (0)

// Threads and products.  512 threads per workgroup; thread (q4, e) = (tid >> 3, tid & 7) holds the k eighth e (32 values) of FOUR
// rows in 128 VGPRs and reads its eighth of the vector from LDS: 8 x ds_read_b128 feed 128 FMAs.  (A broadcast LDS read still
// costs the LDS its 4 cycles per wave instruction: one row per thread -- 64 reads per 256 FMAs -- kept the LDS busy for twice the
// FMA time, 0.9 us per product; here the LDS is busy half of it.)  The eight partial sums of a row meet by three DPP adds inside
// the 8-lane group, no LDS, no barrier.  In the LSTM roles the four rows are the four gates of ONE unit, so the cell is computed
// where the sums land.  The vector's eighths sit 36 floats apart in LDS: eight distinct 16-byte reads of a wave instruction then
// fall into eight different bank quads.
constexpr int NT = 512, EK = GH / 8, XP = EK + 4, XS = 8 * XP;
__device__ __forceinline__ int xs_index(int k) { return (k >> 5) * XP + (k & 31); }

__device__ __forceinline__ void load_rows(float (&w)[4][EK], const float* __restrict__ W, const int (&row)[4], int e) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < EK; k += 4) {
            const f32x4 v = ld4u(W + (long)row[i] * GH + EK * e + k);
            w[i][k] = v[0]; w[i][k + 1] = v[1]; w[i][k + 2] = v[2]; w[i][k + 3] = v[3];
        }
}
// y[i] = row i . x for the thread's four rows: partial sums over its eighth, then the 8-lane group's total in every lane of the group
__device__ __forceinline__ void dot4(const float (&w)[4][EK], const float* xe, float (&y)[4]) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < EK; k += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(xe + k);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) fmac(a[i], w[i][k + kk], x[kk]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s = a[i];
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x141, 0xF, 0xF, true));   // row_half_mirror: the other quad
        y[i] = s;
    }
}

// a recurrent-side product off the critical path: y_t = W x_{t-1} (+ b) for the ticks 0 .. L-1, x_{-1} = the initial state
__device__ __forceinline__ void recurrent_role(const GenArgs& a, int k, const float* __restrict__ W, const float* __restrict__ bias,
                                               const float* x_init, const unsigned long long* xin, unsigned long long* yout,
                                               float (*xs)[XS], volatile int* bad, bool near) {
    const int tid = threadIdx.x, j = tid >> 3, e = tid & 7;
    const int row[4] = {64 * k + j, GH + 64 * k + j, 2 * GH + 64 * k + j, 3 * GH + 64 * k + j};      // the four gates of unit 64 k + j
    float w[4][EK];
    load_rows(w, W, row, e);
    const int myrow = row[e & 3];                              // lanes 0 .. 3 of the group publish one gate each
    const float b = bias ? bias[myrow] : 0.f;
    for (int t = 0; t < a.L; ++t) {
        float y[4] = {0.f, 0.f, 0.f, 0.f};
        if (t == 0) {
            if (x_init) {
                if (tid < GH) xs[0][xs_index(tid)] = x_init[tid];
                lds_barrier();
                dot4(w, xs[0] + XP * e, y);
            }
        } else {
            if (tid < GH) {
                float x;
                if (!get_1(xin + (t & 1) * GH + tid, (unsigned)t, a.status, x)) *bad = 1;
                xs[t & 1][xs_index(tid)] = x;
            }
            lds_barrier();
            if (*bad) break;
            dot4(w, xs[t & 1] + XP * e, y);
        }
        if (e < 4) put(yout + myrow, (e == 0 ? y[0] : e == 1 ? y[1] : e == 2 ? y[2] : y[3]) + b, (unsigned)t + 1u, near);
    }
}

template <int NV>
__global__ __launch_bounds__(NT) void arnn_token_pass_kernel(GenArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[2][XS];
    __shared__ __attribute__((aligned(16))) float us[GH];
    __shared__ float ps[8][64 * NV];
    __shared__ int bad_s, near_s;
    if (blockIdx.x % a.stride) return;
    const int role = blockIdx.x / a.stride;
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    unsigned long long* const e_h0 = a.ex;
    unsigned long long* const e_h1 = a.ex + 2 * GH;
    unsigned long long* const e_hh0 = a.ex + 4 * GH;
    unsigned long long* const e_hh1 = a.ex + 4 * GH + G4;
    volatile int* const bad = &bad_s;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    // all 13 workgroups on one XCD (mode 3 asks; the answer is the hardware's): granules as plain stores (granule.h)
    const bool near = a.near && same_xcd(a.ex + kExXcc, role, 13, a.status, &near_s);

    if (role >= 1 && role <= 4) {
        recurrent_role(a, role - 1, a.W_hh0, nullptr, a.hc_init, e_h0, e_hh0, xs, bad, near);  // (b_hh0 sits in pre)
    } else if (role >= 9) {
        recurrent_role(a, role - 9, a.W_hh1, a.b_hh1, a.hc_init ? a.hc_init + 2 * GH : nullptr, e_h1, e_hh1, xs, bad, near);
    } else if (role >= 5) {
        // ---- Bi_k: layer 1's input-side product and its cell: thread (unit j, eighth e) holds the four gate rows of its unit, the
        // sums land in every lane of the 8-lane group, lane e = 0 computes the cell -- no LDS, no barrier behind the product ----
        const int k = role - 5, j = tid >> 3, e = tid & 7;
        const int row[4] = {64 * k + j, GH + 64 * k + j, 2 * GH + 64 * k + j, 3 * GH + 64 * k + j};
        const bool cell = e == 0;
        float w[4][EK];
        load_rows(w, a.W_ih1, row, e);
        float bb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bb[i] = a.b_ih1[row[i]];
        float c1 = (cell && a.hc_init) ? a.hc_init[3 * GH + 64 * k + j] : 0.f;
        for (int t = 0; t < a.L; ++t) {
            // the recurrent summands of this tick were started a tick ago: they are here before h0_t is
            float hh[4] = {0.f, 0.f, 0.f, 0.f};
            unsigned long long hw[4];
            if (k == 0) GEN_STAMP(1, t, 0);
            if (cell && !get_n<4>(e_hh1 + 64 * k + j, GH, (unsigned)t + 1u, a.status, hh, hw)) *bad = 1;
            if (k == 0) GEN_STAMP(1, t, 1);
            if (tid < GH) {
                float x;
                if (!get_1(e_h0 + ((t + 1) & 1) * GH + tid, (unsigned)t + 1u, a.status, x)) *bad = 1;
                xs[t & 1][xs_index(tid)] = x;
            }
            if (k == 0) GEN_STAMP(1, t, 2);
            lds_barrier();
            if (*bad) break;
            if (k == 0) GEN_STAMP(1, t, 3);
            float y[4];
            dot4(w, xs[t & 1] + XP * e, y);
            if (k == 0) GEN_STAMP(1, t, 4);
            if (cell) {
                const float ig = sigmoid_f(y[0] + bb[0] + hh[0]), fg = sigmoid_f(y[1] + bb[1] + hh[1]);
                const float gv = tanh_f(y[2] + bb[2] + hh[2]), og = sigmoid_f(y[3] + bb[3] + hh[3]);
                c1 = fg * c1 + ig * gv;
                put(e_h1 + ((t + 1) & 1) * GH + 64 * k + j, og * tanh_f(c1), (unsigned)t + 1u, near);
            }
            if (k == 0) GEN_STAMP(1, t, 5);
        }
    } else {
        // ---- C: layer 0's cell (no product: its three summands arrive), linear_1, the note head, argmax ----
        const int rq = tid >> 3, e = tid & 7;
        const int row[4] = {4 * rq, 4 * rq + 1, 4 * rq + 2, 4 * rq + 3};
        float w1[4][EK];
        load_rows(w1, a.W1, row, e);
        float w2[NV][32];                                      // rows lane + 64 j of the head, k eighth q
        float b2[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = min(lane + 64 * j, a.V - 1);
            b2[j] = a.b2[v];
#pragma unroll
            for (int i = 0; i < 32; i += 4) {
                const f32x4 x = ld4u(a.W2 + (long)v * GH + 32 * q + i);
                w2[j][i] = x[0]; w2[j][i + 1] = x[1]; w2[j][i + 2] = x[2]; w2[j][i + 3] = x[3];
            }
        }
        const float b1 = a.b1[4 * rq + (e & 3)];               // lanes 0 .. 3 of the group write one row of u each
        const bool unit = tid < GH;                            // threads 0 .. 255 also own one unit of layer 0
        float c0 = (unit && a.hc_init) ? a.hc_init[GH + tid] : 0.f;
        long long tok = a.first_tok ? *a.first_tok : 0;
        float pr[4] = {0.f, 0.f, 0.f, 0.f}, hh[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned long long hw[4];
        if (unit) {
#pragma unroll
            for (int g = 0; g < 4; ++g) pr[g] = a.pre[g * GH + tid];
            if (!get_n<4>(e_hh0 + tid, GH, 1u, a.status, hh, hw)) *bad = 1;
        }
        for (int t = 0; t < a.L; ++t) {
            const bool more = t + 1 < a.L;
            GEN_STAMP(0, t, 0);
            if (unit) {
                float gate[4];
                if (t == 0) {                                  // the token in front of the first tick may lie outside the head's range
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float s = 0.f;
                        for (int e = 0; e < a.E; ++e) s = fmaf(a.emb[tok * a.E + e], a.W_ih0[(long)(g * GH + tid) * a.K0 + e], s);
                        gate[g] = pr[g] + s + hh[g];
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) gate[g] = pr[g] + a.T0[tok * G4 + g * GH + tid] + hh[g];
                }
                const float ig = sigmoid_f(gate[0]), fg = sigmoid_f(gate[1]), gv = tanh_f(gate[2]), og = sigmoid_f(gate[3]);
                c0 = fg * c0 + ig * gv;
                put(e_h0 + ((t + 1) & 1) * GH + tid, og * tanh_f(c0), (unsigned)t + 1u, near);
                GEN_STAMP(0, t, 1);
                if (more) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) pr[g] = a.pre[(long)(t + 1) * G4 + g * GH + tid];
                }
                float x;
                if (!get_1(e_h1 + ((t + 1) & 1) * GH + tid, (unsigned)t + 1u, a.status, x)) *bad = 1;
                GEN_STAMP(0, t, 2);
                xs[0][xs_index(tid)] = x;
            }
            lds_barrier();
            if (*bad) break;
            GEN_STAMP(0, t, 3);
            // the next tick's recurrent summands left A_k about when h1_t left Bi_k: request them now, look at them behind the head
            if (unit && more) {
#pragma unroll
                for (int g = 0; g < 4; ++g) hw[g] = peek(e_hh0 + g * GH + tid);
            }
            float y[4];
            dot4(w1, xs[0] + XP * e, y);
            if (e < 4) us[4 * rq + e] = fmaxf((e == 0 ? y[0] : e == 1 ? y[1] : e == 2 ? y[2] : y[3]) + b1, 0.f);
            lds_barrier();
            GEN_STAMP(0, t, 4);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float p0 = 0.f, p1 = 0.f;
#pragma unroll
                for (int i = 0; i < 32; i += 4) {
                    const f32x4 u4 = *reinterpret_cast<const f32x4*>(us + 32 * q + i);
                    fmac(p0, w2[j][i], u4[0]); fmac(p1, w2[j][i + 1], u4[1]); fmac(p0, w2[j][i + 2], u4[2]); fmac(p1, w2[j][i + 3], u4[3]);
                }
                ps[q][lane + 64 * j] = p0 + p1;
            }
            lds_barrier();
            GEN_STAMP(0, t, 5);
            // every wave takes the argmax for itself (wave-uniform result, no fourth barrier): the maximum by DPP, its lowest index by
            // ballot; a NaN is the maximum (np.argmax), the lowest NaN wins
            {
                float lg[NV];
                unsigned long long nanm[NV];
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int v = lane + 64 * j;
                    float x = b2[j];
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) x += ps[qq][v];
                    lg[j] = v < a.V ? x : -INFINITY;
                    nanm[j] = __ballot(lg[j] != lg[j]);
                    m = fmaxf(m, lg[j]);                        // (fmaxf skips a NaN)
                }
                m = wave_max_dpp(m);
                int bi = -1;
#pragma unroll
                for (int j = NV - 1; j >= 0; --j) {
                    const unsigned long long eq = __ballot(lg[j] == m);
                    if (eq) bi = 64 * j + __builtin_ctzll(eq);
                }
#pragma unroll
                for (int j = NV - 1; j >= 0; --j)
                    if (nanm[j]) bi = 64 * j + __builtin_ctzll(nanm[j]);
                // (NaN in the lower half beats one in the upper: the loops run downwards and the lower overwrites)
                bool anynan = false;
#pragma unroll
                for (int j = 0; j < NV; ++j) anynan |= nanm[j] != 0;
                if (!anynan && bi < 0) bi = 0;
                tok = (bi >= 0 && bi < a.V) ? bi : 0;
                if (tid == 0) a.tokens[t] = tok;
            }
            GEN_STAMP(0, t, 6);
            if (unit && more && !get_n<4>(e_hh0 + tid, GH, (unsigned)t + 2u, a.status, hh, hw, false)) *bad = 1;
            GEN_STAMP(0, t, 7);
        }
    }
    __syncthreads();
    if (bad_s && tid == 0) chain::raise_timeout(a.status);
}

// pre[t][r] = sum_k oc[t][k] W_ih0[r][E + k] + b_ih0[r] + b_hh0[r]  (tiles of 16 ticks x 64 gate rows), and
// T0[v][r] = sum_e emb[v][e] W_ih0[r][e]                              (the blocks behind them, 256 outputs each)
struct PrepArgs {
    int L, V, E, Hc, K0, nb_pre;
    const float* oc0; long oc_stride;
    const float* emb; const float* W_ih0; const float* b_ih0; const float* b_hh0;
    float* pre; float* T0;
};
__global__ __launch_bounds__(256) void arnn_gen_prep_kernel(PrepArgs a) {
    __shared__ float Ws[64 * 65];
    __shared__ float Os[16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, tg = tid >> 6;
    if ((int)blockIdx.x >= a.nb_pre) {
        const long o = (long)(blockIdx.x - a.nb_pre) * 256 + tid;
        if (o < (long)a.V * G4) {
            const int v = (int)(o / G4), r = (int)(o % G4);
            float s = 0.f;
            for (int e = 0; e < a.E; ++e) s = fmaf(a.emb[(long)v * a.E + e], a.W_ih0[(long)r * a.K0 + e], s);
            a.T0[o] = s;
        }
        return;
    }
    const int rb = blockIdx.x % (G4 / 64), tb = blockIdx.x / (G4 / 64);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < a.Hc; k0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = tid + 256 * i, kk = e & 63, row = e >> 6;
            Ws[kk * 65 + row] = k0 + kk < a.Hc ? a.W_ih0[(long)(rb * 64 + row) * a.K0 + a.E + k0 + kk] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, kk = e & 63, tk = e >> 6, t = tb * 16 + tk;
            Os[tk * 64 + kk] = (t < a.L && k0 + kk < a.Hc) ? a.oc0[(long)t * a.oc_stride + k0 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 64; ++kk) {
            const float w = Ws[kk * 65 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(w, Os[(4 * tg + i) * 64 + kk], acc[i]);
        }
    }
    const int r = rb * 64 + lane;
    const float b = a.b_ih0[r] + a.b_hh0[r];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = tb * 16 + 4 * tg + i;
        if (t < a.L) a.pre[(long)t * G4 + r] = acc[i] + b;
    }
}

int g_mode = -1;                                 // 0 = the per-tick launches, 1 = persistent kernel on 13 consecutive workgroups,
                                                 // 2 = on every 8th workgroup of 104: one XCD under the round-robin dispatch observed
                                                 // today (speed only, 4.0 vs 4.2 us per tick; correct under any placement),
                                                 // 3 (default) = 2 + plain granule stores once the workgroups have FOUND themselves on
                                                 // one XCD (granule.h: 3.6 -> 3.2 us per tick; agent-scope stores otherwise);
                                                 // 4 = test hook: 3's request on consecutive workgroup ids (the check must say no)
int mode() {
    if (g_mode < 0) {
        const char* v = std::getenv("INET_ARNN_GEN");
        g_mode = v ? std::atoi(v) : 3;
        if (g_mode < 0 || g_mode > 4) g_mode = 3;
    }
    return g_mode;
}
constexpr long kExGranules = kExXcc + 16;         // 8-byte granules of the exchange
constexpr long kExFloats = 2 * kExGranules + 64; // ... as floats, + the launch's status word (64 floats behind them)
}  // namespace

void arnn_gen_set_mode(int m) { g_mode = (m < 0 || m > 4) ? 3 : m; }

bool arnn_token_pass_ok(int H, int U, int V) { return mode() != 0 && chain_enabled() && H == GH && U == GH && V >= 1 && V <= 128; }

// tables | exchange + status | stamps (2 x L x 8 64-bit words, written only under INET_ARNN_GEN_STAMPS=1: tools/arnn_token_pass.py)
size_t arnn_token_pass_ws_floats(int L, int V) { return (size_t)L * G4 + (size_t)V * G4 + (size_t)kExFloats + 64 + (size_t)32 * L; }
long arnn_token_pass_stamps_offset(int L, int V) { return (long)L * G4 + (long)V * G4 + kExFloats + 64; }

int arnn_token_pass(int L, int E, int Hc, int V, const float* emb, const float* oc0, long oc_stride, const float* W_ih0,
                    const float* b_ih0, const float* W_hh0, const float* b_hh0, const float* W_ih1, const float* b_ih1,
                    const float* W_hh1, const float* b_hh1, const float* W1, const float* b1, const float* W2, const float* b2,
                    const float* hc_init, const long long* first_tok, long long* tokens, float* ws, hipStream_t s) {
    float* pre = ws;
    float* T0 = pre + (size_t)L * G4;
    // (the exchange starts on a 16-float boundary behind the tables: L * 4H and V * 4H are multiples of 1024)
    unsigned long long* ex = reinterpret_cast<unsigned long long*>(T0 + (size_t)V * G4);
    unsigned* status = reinterpret_cast<unsigned*>(ex + kExGranules);
    if (hipMemsetAsync(ex, 0, (size_t)kExFloats * sizeof(float), s) != hipSuccess) return -2;
    PrepArgs p{};
    p.L = L; p.V = V; p.E = E; p.Hc = Hc; p.K0 = E + Hc;
    p.nb_pre = (G4 / 64) * ((L + 15) / 16);
    p.oc0 = oc0; p.oc_stride = oc_stride; p.emb = emb; p.W_ih0 = W_ih0; p.b_ih0 = b_ih0; p.b_hh0 = b_hh0; p.pre = pre; p.T0 = T0;
    const int nb_t0 = (int)(((long)V * G4 + 255) / 256);
    hipLaunchKernelGGL(arnn_gen_prep_kernel, dim3(p.nb_pre + nb_t0), dim3(256), 0, s, p);
    GenArgs a{};
    a.L = L; a.V = V; a.E = E; a.K0 = E + Hc; a.stride = (mode() == 2 || mode() == 3) ? 8 : 1; a.near = mode() >= 3;   // (4: test hook -- XCD-local stores REQUESTED on
                                                                                // consecutive ids: the workgroups must find out that they do not share an XCD)
    a.emb = emb; a.W_ih0 = W_ih0; a.W_hh0 = W_hh0; a.W_ih1 = W_ih1; a.b_ih1 = b_ih1; a.W_hh1 = W_hh1; a.b_hh1 = b_hh1;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.pre = pre; a.T0 = T0;
    a.hc_init = hc_init; a.first_tok = first_tok; a.tokens = tokens; a.ex = ex;
    static const bool stamps = [] { const char* v = std::getenv("INET_ARNN_GEN_STAMPS"); return v && v[0] == '1'; }();
    a.stamps = stamps ? reinterpret_cast<unsigned long long*>(ws + arnn_token_pass_stamps_offset(L, V)) : nullptr;
    a.status = chain_status_for(status);
    char label[64];
    std::snprintf(label, sizeof label, "arnn_token_pass L%d V%d", L, V);
    // per tick: four 1024 x 256 products (one of them, the input side of layer 0, in the prep launch), linear_1, the head
    ProfScope prof(PROF_GRU_FWD, 2.0 * L * (3.0 * G4 * GH + (double)GH * GH + (double)V * GH), s, label,
                   4.0 * (3.0 * G4 * GH + (double)GH * GH + (double)V * GH + (double)(L + V) * G4));
    const dim3 grid(13 * a.stride);
    if (V <= 64) hipLaunchKernelGGL((arnn_token_pass_kernel<1>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((arnn_token_pass_kernel<2>), grid, dim3(NT), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
