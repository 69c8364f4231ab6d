// The free-running decode of ONE measure (b = 1 inference: LatentRNNTester.generate, VAETester.decode_mid_point; the call the
// north_star prices, MeasureVAE/decoder.py:412-529) as ONE register-resident persistent launch (round 5).
//
// decode_chain.hip runs the 24 ticks of a b = 1 call in 0.207 ms: 8.6 us per tick = three all-to-all exchanges among 32 members
// (layer 0, layer 1, projection + argmax) at ~2.5 us each, behind eight small launches of the beat path (0.06 ms).  The token
// pass of AnticipationRNN (arnn_gen.hip) showed the cheaper shape for a one-row recurrence: keep every weight matrix in REGISTERS
// of a few workgroups, move vectors as 8-byte {value, tag} granules (granule.h), and take everything that does not depend on the
// newest token off the critical path.  Here, H = 512, 512 threads per workgroup:
//
//  tick path (forward_tick_rnn, decoder.py:473-529), 24 ticks
//   C    (1)     layer 0's cell -- its input side is cgi[beat] + table[token], its recurrent side arrives from TA: NO product on
//                this edge --, publish h0_t; wait for h1_t; logits = ReLU(W_out h1 + b_out) -> weights[t]; argmax (lowest index
//                among equals) -> token_t
//   TA_k (8)     W_hh0 rows of units 64k .. 64k+63:  gh0 for tick t+1 = W_hh0 h0_t + b_hh0          (off the critical path)
//   TBi_k (8)    W_ih1 rows:  gi1 = W_ih1 h0_t + b_ih1, layer 1's cell with gh1 from TBh, publish h1_t
//   TBh_k (8)    W_hh1 rows:  gh1 for tick t+1 = W_hh1 h1_t + b_hh1                                  (off the critical path)
//  A tick is two hand-offs (C -> TBi -> C) with one 1536 x 512 product (spread over 8 workgroups) and the V x 512 head behind them.
//  The tick GRU's hidden state is re-initialised at every beat (decoder.py:485-490): at a beat's first tick the recurrent-side
//  workgroups multiply the beat's initial state instead of the previous tick's output.
//
//  beat path (forward_beat_rnn + the per-beat projections, decoder.py:455-471, 485-497) -- FOLDED INTO THE SAME LAUNCH (`fused`): it
//  used to be eight launches in front (z -> beat state, two 4-step chain launches, three projections: 0.06 of a 0.17 ms call)
//   Z2B_k (4)    hb0 = SELU(W_zb z + b): the beat GRU's initial state
//   BA_k (8)     beat layer 0, product AND cell (its input gates are the constant gvec0): h0b_i, i = 0 .. 3
//   BBi_k / BBh_k (8 + 8)  beat layer 1: input-side product + cell / recurrent-side product -> beat output i
//   PH_k (8), PI_k (4)     ht0_i = SELU(W_bh out_i + b) (initial tick state of beat i), c_i = SELU(W_bi out_i + b)
//   CG_k (8)     cgi_i = W_ih0(tick)[:, E:] c_i   (the beat-constant half of the tick GRU's input projection)
//  Every step of the beat path writes its own granules (nothing is overwritten: tag 1 = written), the tick workgroups pick ht0_i /
//  cgi_i up at beat i's first tick.  While the beat workgroups compute, the tick workgroups load their weights; beats 1 .. 3 are
//  ready long before tick 6 i needs them: the call's critical path is [weights in] + z2b + L0 + L1 + projection + cgi of BEAT 0,
//  then the 24 ticks.
//
// Thread (p, s) = (tid >> 4, tid & 15) of a product workgroup holds the k slice s of R rows (6 = the gate rows of units 2p, 2p + 1;
// 4 or 8 plain rows) in up to 192 VGPRs; the vector's slices sit SK + 4 floats apart in LDS (16 distinct 16-byte reads of a wave
// instruction fall into 16 different bank quads); the 16 partial sums of a row meet by four DPP adds inside the 16-lane row, and
// lanes 0 / 1 of the row compute the cells of the two units -- no LDS, no barrier behind a product.
// Shapes: H = 512, Z = 256, V <= 128, <= 4 beats, inference (no dropout mask, no backward saves); else decode_chain.hip.
#include <cstdio>
#include <cstdlib>
#include "chain.h"
#include "granule.h"
#include "prof.h"
#include "decode_chain.h"

namespace {
using namespace granule;

constexpr int DH = 512, D3 = 3 * DH, DZ = 256, NT = 512, NS = 16, XS = NS * (DH / NS + 4);
constexpr int kTickRoles = 1 + 3 * (DH / 64);                    // C + TA, TBi, TBh
// fused roles behind the tick roles
constexpr int R_Z2B = kTickRoles, R_BA = R_Z2B + 4, R_BBI = R_BA + 8, R_BBH = R_BBI + 8, R_PH = R_BBH + 8, R_PI = R_PH + 8,
              R_CG = R_PI + 4, kFusedRoles = R_CG + 8;
// granule map (8-byte units)
constexpr int G_H0 = 0, G_H1 = DH, G_GH0 = 2 * DH, G_GH1 = 2 * DH + D3, G_TICK_END = 2 * DH + 2 * D3;
constexpr int G_HB0 = G_TICK_END, G_H0B = G_HB0 + 2 * DH, G_H1B = G_H0B + 4 * DH, G_GH1B = G_H1B + 4 * DH, G_C = G_GH1B + 4 * D3,
              G_HT0 = G_C + 4 * DH, G_CGI = G_HT0 + 4 * 2 * DH, G_END = G_CGI + 4 * D3;
static_assert(2 * G_END <= kDecodeB1Words, "the workspace's granule area holds the map");

struct B1Args {
    int T, G, V, stride, fused;
    const float* W_hh0; const float* b_hh0; const float* cgi; const float* table;
    const float* W_ih1; const float* b_ih1; const float* W_hh1; const float* b_hh1;
    const float* W_out; const float* b_out; const float* ht0;
    float* weights; long long* samples;
    unsigned long long* ex;
    DecodeB1Beat bp;                             // the beat path's operands (fused)
    chain::Status status;
};

template <int SK> __device__ __forceinline__ int xs_index(int k) { return (k / SK) * (SK + 4) + (k % SK); }

__device__ __forceinline__ float row_sum16(float s) {
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x141, 0xF, 0xF, true));   // row_half_mirror
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x140, 0xF, 0xF, true));   // row_mirror
    return s;
}
// slice s (SK values from column SK s) of R rows of a row-major matrix with leading dimension ld
template <int R, int SK>
__device__ __forceinline__ void load_rows(float (&w)[R][SK], const float* __restrict__ W, long ld, const int (&row)[R], int s) {
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int k = 0; k < SK; k += 4) {
            const f32x4 v = ld4u(W + (long)row[i] * ld + SK * s + k);
            w[i][k] = v[0]; w[i][k + 1] = v[1]; w[i][k + 2] = v[2]; w[i][k + 3] = v[3];
        }
}
// y[i] = row i . x: partial sums over the thread's k slice, then the 16-lane row's total in every lane of the row
// (No y[lane-dependent index] anywhere below: hipcc turns a select chain over a private array back into an indexed access and parks
//  the array in scratch / LDS -- two-way selects and predicated copies with compile-time indices only.)
template <int R, int SK>
__device__ __forceinline__ void dot_rows(const float (&w)[R][SK], const float* xsl, float (&y)[R]) {
    float a[R];
#pragma unroll
    for (int i = 0; i < R; ++i) a[i] = 0.f;
#pragma unroll
    for (int k = 0; k < SK; k += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(xsl + k);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < R; ++i) fmac(a[i], w[i][k + kk], x[kk]);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) y[i] = row_sum16(a[i]);
}
__device__ __forceinline__ float gru_cell(float gir, float giz, float gin, float ghr, float ghz, float ghn, float hprev) {
    const float r = sigmoid_f(gir + ghr), z = sigmoid_f(giz + ghz);
    const float n = tanh_f(gin + r * ghn);
    return (1.f - z) * n + z * hprev;
}

struct Ctx {                                     // what every role needs
    const B1Args& a; unsigned long long* ex; float (*xs)[XS]; volatile int* bad; int tid;
};

// ---- tick path: recurrent side of a layer, off the critical path: gh for tick t = W_hh x + b_hh, x = the beat's initial state at
// a beat's first tick, else the layer's output of tick t - 1 ----
__device__ __forceinline__ void tick_recurrent_role(const Ctx& c, int k, const float* __restrict__ W, const float* __restrict__ bias,
                                                    int layer, const unsigned long long* xin, unsigned long long* yout) {
    const B1Args& a = c.a;
    const int tid = c.tid, p = tid >> 4, s = tid & 15, u0 = 64 * k + 2 * p;
    const int row[6] = {u0, DH + u0, 2 * DH + u0, u0 + 1, DH + u0 + 1, 2 * DH + u0 + 1};
    float w[6][32];
    load_rows<6, 32>(w, W, DH, row, s);
    const bool second = s & 1;                                 // lanes 0 / 1 of the row publish the three gate rows of unit u0 / u0 + 1
    const int u = u0 + (s & 1);
    float b[3] = {0.f, 0.f, 0.f};
    if (s < 2) {
#pragma unroll
        for (int g = 0; g < 3; ++g) b[g] = bias[g * DH + u];
    }
    for (int t = 0; t < a.T; ++t) {
        // (the previous tick's output is waited for at a beat's first tick too, although the beat's initial state is what gets
        //  multiplied: the single-buffered granules are safe only while every producer stays behind its consumers -- a workgroup
        //  that ran ahead here would overwrite gh of tick t - 1 before the cell that needs it has looked)
        float x = 0.f;
        if (t > 0 && !get_1(xin + tid, (unsigned)t, a.status, x)) *c.bad = 1;
        if (t % a.G == 0) {
            const int beat = t / a.G;
            if (a.fused) { if (!get_1(c.ex + G_HT0 + beat * 2 * DH + layer * DH + tid, 1u, a.status, x)) *c.bad = 1; }
            else x = a.ht0[(long)beat * 2 * DH + layer * DH + tid];
        }
        c.xs[t & 1][xs_index<32>(tid)] = x;
        lds_barrier();
        if (*c.bad) break;
        float y[6];
        dot_rows<6, 32>(w, c.xs[t & 1] + 36 * s, y);
        if (s < 2) {
#pragma unroll
            for (int g = 0; g < 3; ++g) put(yout + g * DH + u, (second ? y[3 + g] : y[g]) + b[g], (unsigned)t + 1u);
        }
    }
}

// ---- beat path: one 512-input product per beat (R plain rows per thread): out_i[row] = f(W x_i + b), x_i / out_i = granule arrays
// with one slot per beat ----
template <int R, bool SELU_OUT>
__device__ __forceinline__ void beat_product_role(const Ctx& c, int row0, const float* __restrict__ W, long ld, const float* __restrict__ bias,
                                                  int g_in, int in_stride, int g_out, int out_stride, int nb) {
    const B1Args& a = c.a;
    const int tid = c.tid, p = tid >> 4, s = tid & 15;
    int row[R];
#pragma unroll
    for (int i = 0; i < R; ++i) row[i] = row0 + R * p + i;
    float w[R][32];
    load_rows<R, 32>(w, W, ld, row, s);
    float b[R];
#pragma unroll
    for (int i = 0; i < R; ++i) b[i] = bias ? bias[row[i]] : 0.f;
    for (int i = 0; i < nb; ++i) {
        float x;
        if (!get_1(c.ex + g_in + i * in_stride + tid, 1u, a.status, x)) *c.bad = 1;
        c.xs[i & 1][xs_index<32>(tid)] = x;
        lds_barrier();
        if (*c.bad) break;
        float y[R];
        dot_rows<R, 32>(w, c.xs[i & 1] + 36 * s, y);
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (s == j) {
                const float v = y[j] + b[j];
                put(c.ex + g_out + i * out_stride + row[j], SELU_OUT ? selu_f(v) : v, 1u);
            }
    }
}

template <int NJ, bool FUSED>
__global__ __launch_bounds__(NT) void decode_b1_kernel(B1Args a) {
    __shared__ __attribute__((aligned(16))) float xs[2][XS];
    __shared__ float lgs[32 * NJ];
    __shared__ int bad_s;
    if (blockIdx.x % a.stride) return;
    const int role = blockIdx.x / a.stride;
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned long long* const ex = a.ex;
    volatile int* const bad = &bad_s;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    const Ctx c{a, ex, xs, bad, tid};
    const int nb = a.T / a.G;
    const DecodeB1Beat& bp = a.bp;

    if (role == 0) {
        // ---- C: layer 0's cell (its three summands are made elsewhere and arrive), the output projection, argmax ----
        const int rv = tid >> 4, s = tid & 15;
        int row[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) row[j] = min(rv + 32 * j, a.V - 1);
        float wo[NJ][32];
        load_rows<NJ, 32>(wo, a.W_out, DH, row, s);
        float bo[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bo[j] = a.b_out[row[j]];
        const int u = tid;                                     // every thread owns one unit of layer 0
        float h0 = 0.f, cg[3] = {0.f, 0.f, 0.f}, gh[3] = {0.f, 0.f, 0.f};
        unsigned long long hw[3];
        long long tok = a.V;                                   // row V of the table: the start symbol x_0
        if (!get_n<3>(ex + G_GH0 + u, DH, 1u, a.status, gh, hw)) *bad = 1;
        for (int t = 0; t < a.T; ++t) {
            const bool more = t + 1 < a.T;
            if (t % a.G == 0) {
                const long beat = t / a.G;
                if (FUSED) {
                    unsigned long long cw[3];
                    if (!get_1(ex + G_HT0 + beat * 2 * DH + u, 1u, a.status, h0)) *bad = 1;
                    if (!get_n<3>(ex + G_CGI + beat * D3 + u, DH, 1u, a.status, cg, cw)) *bad = 1;
                } else {
                    h0 = a.ht0[beat * 2 * DH + u];
#pragma unroll
                    for (int g = 0; g < 3; ++g) cg[g] = a.cgi[beat * D3 + g * DH + u];
                }
            }
            float gi[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[g] = cg[g] + a.table[tok * D3 + g * DH + u];
            h0 = gru_cell(gi[0], gi[1], gi[2], gh[0], gh[1], gh[2], h0);
            put(ex + G_H0 + u, h0, (unsigned)t + 1u);
            float x;
            if (!get_1(ex + G_H1 + u, (unsigned)t + 1u, a.status, x)) *bad = 1;
            xs[0][xs_index<32>(tid)] = x;
            lds_barrier();
            if (*bad) break;
            // the next tick's recurrent summands left TA_k about when h1_t left TBi_k: request them now, look at them behind the head
            if (more) {
#pragma unroll
                for (int g = 0; g < 3; ++g) hw[g] = peek(ex + G_GH0 + g * DH + u);
            }
            float y[NJ];
            dot_rows<NJ, 32>(wo, xs[0] + 36 * s, y);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {                     // lane j of the row: logit rv + 32 j
                const int v = rv + 32 * j;
                if (s == j && v < a.V) {
                    float lg = y[j] + bo[j];
                    lg = lg > 0.f ? lg : 0.f;                  // ReLU (decoder.py:372, 503)
                    lgs[v] = lg;
                    a.weights[(long)t * a.V + v] = lg;
                }
            }
            lds_barrier();
            // every wave takes the argmax for itself (wave-uniform, no further barrier): the maximum by DPP, its lowest index by ballot
            {
                constexpr int NVL = (32 * NJ + 63) / 64;
                float lg[NVL], m = -1.f;
#pragma unroll
                for (int j = 0; j < NVL; ++j) {
                    const int v = lane + 64 * j;
                    lg[j] = v < a.V ? lgs[v] : -1.f;            // (below every post-ReLU logit)
                    m = fmaxf(m, lg[j]);
                }
                m = wave_max_dpp(m);
                int bi = 0;
#pragma unroll
                for (int j = NVL - 1; j >= 0; --j) {
                    const unsigned long long eq = __ballot(lg[j] == m);
                    if (eq) bi = 64 * j + __builtin_ctzll(eq);
                }
                tok = bi < a.V ? bi : 0;
                if (tid == 0) a.samples[t] = tok;
            }
            lds_barrier();                                     // (lgs is rewritten next tick)
            if (more && !get_n<3>(ex + G_GH0 + u, DH, (unsigned)t + 2u, a.status, gh, hw, false)) *bad = 1;
        }
    } else if (role <= 8) {
        tick_recurrent_role(c, role - 1, a.W_hh0, a.b_hh0, 0, ex + G_H0, ex + G_GH0);
    } else if (role <= 16) {
        // ---- TBi_k: tick layer 1's input-side product and its cell ----
        const int k = role - 9, p = tid >> 4, s = tid & 15, u0 = 64 * k + 2 * p;
        const int row[6] = {u0, DH + u0, 2 * DH + u0, u0 + 1, DH + u0 + 1, 2 * DH + u0 + 1};
        float w[6][32];
        load_rows<6, 32>(w, a.W_ih1, DH, row, s);
        const bool cell = s < 2, second = s & 1;               // lane 0 / 1 of the 16-lane row: unit u0 / u0 + 1
        const int u = u0 + (s & 1);
        float bi[3] = {0.f, 0.f, 0.f};
        if (cell) {
#pragma unroll
            for (int g = 0; g < 3; ++g) bi[g] = a.b_ih1[g * DH + u];
        }
        float h1 = 0.f;
        for (int t = 0; t < a.T; ++t) {
            // the recurrent summands of this tick were started a tick ago (or at the launch, for a beat's first tick)
            float gh[3] = {0.f, 0.f, 0.f};
            unsigned long long hw[3];
            if (cell) {
                if (!get_n<3>(ex + G_GH1 + u, DH, (unsigned)t + 1u, a.status, gh, hw)) *bad = 1;
                if (t % a.G == 0) {
                    const long beat = t / a.G;
                    if (FUSED) { if (!get_1(ex + G_HT0 + beat * 2 * DH + DH + u, 1u, a.status, h1)) *bad = 1; }
                    else h1 = a.ht0[beat * 2 * DH + DH + u];
                }
            }
            float x;
            if (!get_1(ex + G_H0 + tid, (unsigned)t + 1u, a.status, x)) *bad = 1;
            xs[t & 1][xs_index<32>(tid)] = x;
            lds_barrier();
            if (*bad) break;
            float y[6];
            dot_rows<6, 32>(w, xs[t & 1] + 36 * s, y);
            if (cell) {
                h1 = gru_cell((second ? y[3] : y[0]) + bi[0], (second ? y[4] : y[1]) + bi[1], (second ? y[5] : y[2]) + bi[2],
                              gh[0], gh[1], gh[2], h1);
                put(ex + G_H1 + u, h1, (unsigned)t + 1u);
            }
        }
    } else if (role < kTickRoles) {
        tick_recurrent_role(c, role - 17, a.W_hh1, a.b_hh1, 1, ex + G_H1, ex + G_GH1);
    } else if (FUSED) {
        if (role < R_BA) {
            // ---- Z2B_k: hb0 = SELU(W_zb z + b_zb) (decoder.py:455-461), 256 rows per workgroup, K = 256: 8 rows x 16 values per thread ----
            const int k = role - R_Z2B, p = tid >> 4, s = tid & 15;
            int row[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) row[i] = 256 * k + 8 * p + i;
            float w[8][16];
            load_rows<8, 16>(w, bp.zb_w, DZ, row, s);
            if (tid < DZ) xs[0][xs_index<16>(tid)] = bp.z[tid];
            lds_barrier();
            float y[8];
            dot_rows<8, 16>(w, xs[0] + 20 * s, y);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s == j) put(ex + G_HB0 + row[j], selu_f(y[j] + bp.zb_b[row[j]]), 1u);
        } else if (role < R_BBI) {
            // ---- BA_k: beat layer 0, product and cell (the input gates are the constant gvec0 = b_0 W_ih[:, 0] + b_ih) ----
            const int k = role - R_BA, p = tid >> 4, s = tid & 15, u0 = 64 * k + 2 * p;
            const int row[6] = {u0, DH + u0, 2 * DH + u0, u0 + 1, DH + u0 + 1, 2 * DH + u0 + 1};
            float w[6][32];
            load_rows<6, 32>(w, bp.W_hh0, DH, row, s);
            const bool cell = s < 2, second = s & 1;
            const int u = u0 + (s & 1);
            float gv[3] = {0.f, 0.f, 0.f}, bh[3] = {0.f, 0.f, 0.f}, h = 0.f;
            if (cell) {
#pragma unroll
                for (int g = 0; g < 3; ++g) { gv[g] = bp.gvec0[g * DH + u]; bh[g] = bp.b_hh0[g * DH + u]; }
            }
            for (int i = 0; i < nb; ++i) {
                const unsigned long long* src = i == 0 ? ex + G_HB0 : ex + G_H0B + (i - 1) * DH;
                float x;
                if (!get_1(src + tid, 1u, a.status, x)) *bad = 1;
                xs[i & 1][xs_index<32>(tid)] = x;
                lds_barrier();
                if (*bad) break;
                float y[6];
                dot_rows<6, 32>(w, xs[i & 1] + 36 * s, y);
                if (cell) {
                    if (i == 0) h = xs[0][xs_index<32>(u)];
                    h = gru_cell(gv[0], gv[1], gv[2], (second ? y[3] : y[0]) + bh[0], (second ? y[4] : y[1]) + bh[1],
                                 (second ? y[5] : y[2]) + bh[2], h);
                    put(ex + G_H0B + i * DH + u, h, 1u);
                }
            }
        } else if (role < R_BBH) {
            // ---- BBi_k: beat layer 1's input-side product and its cell -> the beat outputs ----
            const int k = role - R_BBI, p = tid >> 4, s = tid & 15, u0 = 64 * k + 2 * p;
            const int row[6] = {u0, DH + u0, 2 * DH + u0, u0 + 1, DH + u0 + 1, 2 * DH + u0 + 1};
            float w[6][32];
            load_rows<6, 32>(w, bp.W_ih1, DH, row, s);
            const bool cell = s < 2, second = s & 1;
            const int u = u0 + (s & 1);
            float bi[3] = {0.f, 0.f, 0.f}, h = 0.f;
            if (cell) {
#pragma unroll
                for (int g = 0; g < 3; ++g) bi[g] = bp.b_ih1[g * DH + u];
                if (!get_1(ex + G_HB0 + DH + u, 1u, a.status, h)) *bad = 1;      // layer 1's initial state
            }
            for (int i = 0; i < nb; ++i) {
                float gh[3] = {0.f, 0.f, 0.f};
                unsigned long long hw[3];
                if (cell && !get_n<3>(ex + G_GH1B + i * D3 + u, DH, 1u, a.status, gh, hw)) *bad = 1;
                float x;
                if (!get_1(ex + G_H0B + i * DH + tid, 1u, a.status, x)) *bad = 1;
                xs[i & 1][xs_index<32>(tid)] = x;
                lds_barrier();
                if (*bad) break;
                float y[6];
                dot_rows<6, 32>(w, xs[i & 1] + 36 * s, y);
                if (cell) {
                    h = gru_cell((second ? y[3] : y[0]) + bi[0], (second ? y[4] : y[1]) + bi[1], (second ? y[5] : y[2]) + bi[2],
                                 gh[0], gh[1], gh[2], h);
                    put(ex + G_H1B + i * DH + u, h, 1u);
                }
            }
        } else if (role < R_PH) {
            // ---- BBh_k: beat layer 1's recurrent-side product for step i from the output of step i - 1 (the initial state at i = 0) ----
            const int k = role - R_BBH, p = tid >> 4, s = tid & 15, u0 = 64 * k + 2 * p;
            const int row[6] = {u0, DH + u0, 2 * DH + u0, u0 + 1, DH + u0 + 1, 2 * DH + u0 + 1};
            float w[6][32];
            load_rows<6, 32>(w, bp.W_hh1, DH, row, s);
            const bool second = s & 1;
            const int u = u0 + (s & 1);
            float b[3] = {0.f, 0.f, 0.f};
            if (s < 2) {
#pragma unroll
                for (int g = 0; g < 3; ++g) b[g] = bp.b_hh1[g * DH + u];
            }
            for (int i = 0; i < nb; ++i) {
                const unsigned long long* src = i == 0 ? ex + G_HB0 + DH : ex + G_H1B + (i - 1) * DH;
                float x;
                if (!get_1(src + tid, 1u, a.status, x)) *bad = 1;
                xs[i & 1][xs_index<32>(tid)] = x;
                lds_barrier();
                if (*bad) break;
                float y[6];
                dot_rows<6, 32>(w, xs[i & 1] + 36 * s, y);
                if (s < 2) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) put(ex + G_GH1B + i * D3 + g * DH + u, (second ? y[3 + g] : y[g]) + b[g], 1u);
                }
            }
        } else if (role < R_PI) {
            beat_product_role<4, true>(c, 128 * (role - R_PH), bp.bh_w, DH, bp.bh_b, G_H1B, DH, G_HT0, 2 * DH, nb);     // ht0_i
        } else if (role < R_CG) {
            beat_product_role<4, true>(c, 128 * (role - R_PI), bp.bi_w, DH, bp.bi_b, G_H1B, DH, G_C, DH, nb);           // c_i
        } else {
            beat_product_role<6, false>(c, 192 * (role - R_CG), bp.wih0_c, bp.wih0_ld, nullptr, G_C, DH, G_CGI, D3, nb); // cgi_i
        }
    }
    __syncthreads();
    if (bad_s && tid == 0) chain::raise_timeout(a.status);
}

int g_mode = -1;                                 // 0 = off (decode_chain.hip's b = 1 build); 1 = tick path only, consecutive workgroup ids;
                                                 // 2 = tick path only, every 8th id (one XCD); 3 (default) = beat path folded in
int mode() {
    if (g_mode < 0) {
        const char* v = std::getenv("INET_DECODE_B1");
        g_mode = v ? std::atoi(v) : 3;
        if (g_mode < 0 || g_mode > 3) g_mode = 3;
    }
    return g_mode;
}
}  // namespace

void decode_b1_set_mode(int m) { g_mode = (m < 0 || m > 3) ? 3 : m; }

bool decode_b1_shape_ok(int B, int H, int V, int T, int G) {
    return mode() != 0 && chain_enabled() && B == 1 && H == DH && V >= 1 && V <= 128 && T % G == 0 && T / G <= 4 &&
           kFusedRoles <= chain_capacity();
}
bool decode_b1_fused(int Z) { return mode() == 3 && Z == DZ; }
bool decode_b1_ok(const DecodeChainArgs& a) {
    const bool train = a.sv0 || a.sv1 || a.mask || a.h0out || a.h1seq;
    return decode_b1_shape_ok(a.B, a.H, a.V, a.T, a.G) && !train && a.b1ex;
}

int launch_decode_b1(const DecodeChainArgs& d, hipStream_t s) {
    B1Args a{};
    a.fused = d.beat.z != nullptr;
    a.T = d.T; a.G = d.G; a.V = d.V; a.stride = (!a.fused && mode() == 2) ? 8 : 1;
    a.W_hh0 = d.W_hh0; a.b_hh0 = d.b_hh0; a.cgi = d.cgi; a.table = d.table;
    a.W_ih1 = d.W_ih1; a.b_ih1 = d.b_ih1; a.W_hh1 = d.W_hh1; a.b_hh1 = d.b_hh1;
    a.W_out = d.W_out; a.b_out = d.b_out; a.ht0 = d.ht0;
    a.weights = d.weights; a.samples = d.samples; a.ex = d.b1ex;
    a.bp = d.beat;
    a.status = d.status;
    char label[64];
    std::snprintf(label, sizeof label, "decode_b1%s T%d H%d V%d", a.fused ? "_beats" : "", d.T, d.H, d.V);
    // algorithmic work: the tick GRU + head per tick; fused: + the beat path (z2b, two beat layers, three projections per beat)
    const double nbt = (double)d.T / d.G;
    const double beat_mac = a.fused ? 2.0 * DH * DZ + nbt * (3.0 * 3 * DH * DH + 2.0 * DH * DH + 1.0 * DH * DH + 3.0 * DH * DH) : 0.0;
    const double beat_w = a.fused ? 2.0 * DH * DZ + 9.0 * DH * DH + 3.0 * DH * DH + 3.0 * DH * DH : 0.0;
    ProfScope prof(PROF_GRU_FWD, 2.0 * (d.T * (9.0 * DH * DH + (double)d.V * DH) + beat_mac), s, label,
                   4.0 * (9.0 * DH * DH + (double)d.V * DH + (double)d.T * d.V + beat_w));
    const dim3 grid((a.fused ? kFusedRoles : kTickRoles) * a.stride);
    const int nj = (d.V + 31) / 32;
#define INET_B1(NJ)                                                                                        \
    do {                                                                                                   \
        if (a.fused) hipLaunchKernelGGL((decode_b1_kernel<NJ, true>), grid, dim3(NT), 0, s, a);            \
        else hipLaunchKernelGGL((decode_b1_kernel<NJ, false>), grid, dim3(NT), 0, s, a);                   \
    } while (0)
    if (nj <= 1) INET_B1(1); else if (nj == 2) INET_B1(2); else if (nj == 3) INET_B1(3); else INET_B1(4);
#undef INET_B1
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
