// The fused free-running decode for ONE measure (b = 1 inference: LatentRNNTester.generate, VAETester.decode_mid_point; the call
// the north_star prices, MeasureVAE/decoder.py:473-529) as a register-resident persistent launch (round 5).
//
// decode_chain.hip runs the 24 ticks of a b = 1 call in 0.207 ms: 8.6 us per tick = three all-to-all exchanges among 32 members
// (layer 0, layer 1, projection + argmax) at ~2.5 us each.  The token pass of AnticipationRNN (arnn_gen.hip) showed the cheaper
// shape for a one-row recurrence: keep every weight matrix in REGISTERS of a few workgroups, move 2-KB vectors as 8-byte {value,
// tick} granules (granule.h), and take everything that does not depend on the newest token off the critical path.  Here, H = 512:
//
//   C    (1)     layer 0's cell -- its input side is cgi[beat] + table[token] (both made before the launch), its recurrent side
//                arrives from A: NO product on this edge --, publish h0_t; wait for h1_t; logits = ReLU(W_out h1 + b_out) -> weights[t];
//                argmax (lowest index among equals) -> token_t
//   A_k  (8)     W_hh0 rows of units 64k .. 64k+63:  gh0 for tick t+1 = W_hh0 h0_t + b_hh0          (off the critical path)
//   Bi_k (8)     W_ih1 rows:  gi1 = W_ih1 h0_t + b_ih1, layer 1's cell with gh1 from Bh, publish h1_t
//   Bh_k (8)     W_hh1 rows:  gh1 for tick t+1 = W_hh1 h1_t + b_hh1                                  (off the critical path)
//
// A tick is two hand-offs (C -> Bi -> C) with one 1536 x 512 product (spread over 8 workgroups) and the V x 512 head behind them.
// The tick GRU's hidden state is re-initialised at every beat (decoder.py:485-490): at a beat's first tick the recurrent-side
// workgroups multiply the beat's initial state (known before the launch) instead of the previous tick's output.
// Thread (p, s) = (tid >> 4, tid & 15) of a product workgroup holds the k slice s (32 values) of the SIX gate rows of units 2p,
// 2p + 1 of its 64 units in 192 VGPRs; the vector's slices sit 36 floats apart in LDS (16 distinct 16-byte reads of a wave
// instruction fall into 16 different bank quads); 8 LDS reads feed 192 FMAs; the 16 partial sums of a row meet by four DPP adds
// inside the 16-lane row, and lanes 0 / 1 of the row compute the cells of the two units -- no LDS, no barrier behind a product.
// Shapes: H = 512, V <= 128, inference (no dropout mask, no backward saves); anything else stays on decode_chain.hip.
#include <cstdio>
#include <cstdlib>
#include "chain.h"
#include "granule.h"
#include "prof.h"
#include "decode_chain.h"

namespace {
using namespace granule;

constexpr int DH = 512, D3 = 3 * DH, NT = 512, SK = 32, NS = DH / SK, XP = SK + 4, XS = NS * XP;
constexpr int kRolesPerMatrix = DH / 64, kRoles = 1 + 3 * kRolesPerMatrix;
__device__ __forceinline__ int xs_index(int k) { return (k >> 5) * XP + (k & 31); }

struct B1Args {
    int T, G, V, stride;
    const float* W_hh0; const float* b_hh0; const float* cgi; const float* table;
    const float* W_ih1; const float* b_ih1; const float* W_hh1; const float* b_hh1;
    const float* W_out; const float* b_out; const float* ht0;
    float* weights; long long* samples;
    unsigned long long* ex;                      // granules: h0 [H] | h1 [H] | gh0 [3H] | gh1 [3H]
    chain::Status status;
};

__device__ __forceinline__ float row_sum16(float s) {
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x141, 0xF, 0xF, true));   // row_half_mirror
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x140, 0xF, 0xF, true));   // row_mirror
    return s;
}
template <int R>
__device__ __forceinline__ void load_rows(float (&w)[R][SK], const float* __restrict__ W, const int (&row)[R], int s) {
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int k = 0; k < SK; k += 4) {
            const f32x4 v = ld4u(W + (long)row[i] * DH + SK * s + k);
            w[i][k] = v[0]; w[i][k + 1] = v[1]; w[i][k + 2] = v[2]; w[i][k + 3] = v[3];
        }
}
// y[i] = row i . x: partial sums over the thread's k slice, then the 16-lane row's total in every lane of the row
template <int R>
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
// (No y[lane-dependent index]: hipcc turns a select chain over a private array back into an indexed access and parks the array in
//  scratch / LDS -- two-way selects and predicated copies with compile-time indices only.)

// recurrent side of a layer, off the critical path: gh for tick t = W_hh x + b_hh, x = the beat's initial state at a beat's first
// tick (init + beat * 2H), else the layer's output of tick t - 1
__device__ __forceinline__ void recurrent_role(const B1Args& a, int k, const float* __restrict__ W, const float* __restrict__ bias,
                                               const float* init, const unsigned long long* xin, unsigned long long* yout,
                                               float (*xs)[XS], volatile int* bad) {
    const int tid = threadIdx.x, p = tid >> 4, s = tid & 15, u0 = 64 * k + 2 * p;
    const int row[6] = {u0, DH + u0, 2 * DH + u0, u0 + 1, DH + u0 + 1, 2 * DH + u0 + 1};
    float w[6][SK];
    load_rows<6>(w, W, row, s);
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
        if (t > 0 && !get_1(xin + tid, (unsigned)t, a.status, x)) *bad = 1;
        if (t % a.G == 0) x = init[(long)(t / a.G) * 2 * DH + tid];
        xs[t & 1][xs_index(tid)] = x;
        lds_barrier();
        if (*bad) break;
        float y[6];
        dot_rows<6>(w, xs[t & 1] + XP * s, y);
        if (s < 2) {
#pragma unroll
            for (int g = 0; g < 3; ++g) put(yout + g * DH + u, (second ? y[3 + g] : y[g]) + b[g], (unsigned)t + 1u);
        }
    }
}

template <int NJ>
__global__ __launch_bounds__(NT) void decode_b1_kernel(B1Args a) {
    __shared__ __attribute__((aligned(16))) float xs[2][XS];
    __shared__ float lgs[32 * NJ];
    __shared__ int bad_s;
    if (blockIdx.x % a.stride) return;
    const int role = blockIdx.x / a.stride;
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned long long* const e_h0 = a.ex;
    unsigned long long* const e_h1 = a.ex + DH;
    unsigned long long* const e_gh0 = a.ex + 2 * DH;
    unsigned long long* const e_gh1 = a.ex + 2 * DH + D3;
    volatile int* const bad = &bad_s;
    if (tid == 0) bad_s = 0;
    __syncthreads();

    if (role >= 1 && role <= kRolesPerMatrix) {
        recurrent_role(a, role - 1, a.W_hh0, a.b_hh0, a.ht0, e_h0, e_gh0, xs, bad);
    } else if (role > 2 * kRolesPerMatrix) {
        recurrent_role(a, role - 1 - 2 * kRolesPerMatrix, a.W_hh1, a.b_hh1, a.ht0 + DH, e_h1, e_gh1, xs, bad);
    } else if (role > kRolesPerMatrix) {
        // ---- Bi_k: layer 1's input-side product and its cell ----
        const int k = role - 1 - kRolesPerMatrix, p = tid >> 4, s = tid & 15, u0 = 64 * k + 2 * p;
        const int row[6] = {u0, DH + u0, 2 * DH + u0, u0 + 1, DH + u0 + 1, 2 * DH + u0 + 1};
        float w[6][SK];
        load_rows<6>(w, a.W_ih1, row, s);
        const bool cell = s < 2;                               // lane 0 / 1 of the 16-lane row: unit u0 / u0 + 1
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
                if (!get_n<3>(e_gh1 + u, DH, (unsigned)t + 1u, a.status, gh, hw)) *bad = 1;
                if (t % a.G == 0) h1 = a.ht0[(long)(t / a.G) * 2 * DH + DH + u];
            }
            float x;
            if (!get_1(e_h0 + tid, (unsigned)t + 1u, a.status, x)) *bad = 1;
            xs[t & 1][xs_index(tid)] = x;
            lds_barrier();
            if (*bad) break;
            float y[6];
            dot_rows<6>(w, xs[t & 1] + XP * s, y);
            if (cell) {
                const bool second = s & 1;
                const float r = sigmoid_f((second ? y[3] : y[0]) + bi[0] + gh[0]), z = sigmoid_f((second ? y[4] : y[1]) + bi[1] + gh[1]);
                const float n = tanh_f((second ? y[5] : y[2]) + bi[2] + r * gh[2]);
                h1 = (1.f - z) * n + z * h1;
                put(e_h1 + u, h1, (unsigned)t + 1u);
            }
        }
    } else {
        // ---- C: layer 0's cell (its three summands are made before the launch or arrive), the output projection, argmax ----
        const int rv = tid >> 4, s = tid & 15;
        int row[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) row[j] = min(rv + 32 * j, a.V - 1);
        float wo[NJ][SK];
        load_rows<NJ>(wo, a.W_out, row, s);
        float bo[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bo[j] = a.b_out[row[j]];
        const int u = tid;                                     // every thread owns one unit of layer 0
        float h0 = 0.f, cg[3] = {0.f, 0.f, 0.f}, gh[3] = {0.f, 0.f, 0.f};
        unsigned long long hw[3];
        long long tok = a.V;                                   // row V of the table: the start symbol x_0
        if (!get_n<3>(e_gh0 + u, DH, 1u, a.status, gh, hw)) *bad = 1;
        for (int t = 0; t < a.T; ++t) {
            const bool more = t + 1 < a.T;
            if (t % a.G == 0) {
                const long beat = t / a.G;
                h0 = a.ht0[beat * 2 * DH + u];
#pragma unroll
                for (int g = 0; g < 3; ++g) cg[g] = a.cgi[beat * D3 + g * DH + u];
            }
            float gi[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[g] = cg[g] + a.table[tok * D3 + g * DH + u];
            const float r = sigmoid_f(gi[0] + gh[0]), z = sigmoid_f(gi[1] + gh[1]);
            const float n = tanh_f(gi[2] + r * gh[2]);
            h0 = (1.f - z) * n + z * h0;
            put(e_h0 + u, h0, (unsigned)t + 1u);
            float x;
            if (!get_1(e_h1 + u, (unsigned)t + 1u, a.status, x)) *bad = 1;
            xs[0][xs_index(tid)] = x;
            lds_barrier();
            if (*bad) break;
            // the next tick's recurrent summands left A_k about when h1_t left Bi_k: request them now, look at them behind the head
            if (more) {
#pragma unroll
                for (int g = 0; g < 3; ++g) hw[g] = peek(e_gh0 + g * DH + u);
            }
            float y[NJ];
            dot_rows<NJ>(wo, xs[0] + XP * s, y);
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
            if (more && !get_n<3>(e_gh0 + u, DH, (unsigned)t + 2u, a.status, gh, hw, false)) *bad = 1;
        }
    }
    __syncthreads();
    if (bad_s && tid == 0) chain::raise_timeout(a.status);
}

int g_mode = -1;                                 // 0 = off (decode_chain.hip's b = 1 build), 1 = consecutive workgroup ids, 2 = every 8th (one XCD)
int mode() {
    if (g_mode < 0) {
        const char* v = std::getenv("INET_DECODE_B1");
        g_mode = v ? std::atoi(v) : 2;
        if (g_mode < 0 || g_mode > 2) g_mode = 2;
    }
    return g_mode;
}
}  // namespace

void decode_b1_set_mode(int m) { g_mode = (m < 0 || m > 2) ? 2 : m; }

bool decode_b1_shape_ok(int B, int H, int V, int T, int G) {
    return mode() != 0 && chain_enabled() && B == 1 && H == DH && V >= 1 && V <= 128 && T % G == 0 && kRoles <= chain_capacity();
}
bool decode_b1_ok(const DecodeChainArgs& a) {
    const bool train = a.sv0 || a.sv1 || a.mask || a.h0out || a.h1seq;
    return decode_b1_shape_ok(a.B, a.H, a.V, a.T, a.G) && !train && a.b1ex;
}

int launch_decode_b1(const DecodeChainArgs& d, hipStream_t s) {
    B1Args a{};
    a.T = d.T; a.G = d.G; a.V = d.V; a.stride = mode() == 2 ? 8 : 1;
    a.W_hh0 = d.W_hh0; a.b_hh0 = d.b_hh0; a.cgi = d.cgi; a.table = d.table;
    a.W_ih1 = d.W_ih1; a.b_ih1 = d.b_ih1; a.W_hh1 = d.W_hh1; a.b_hh1 = d.b_hh1;
    a.W_out = d.W_out; a.b_out = d.b_out; a.ht0 = d.ht0;
    a.weights = d.weights; a.samples = d.samples; a.ex = d.b1ex;
    a.status = d.status;
    char label[64];
    std::snprintf(label, sizeof label, "decode_b1 T%d H%d V%d", d.T, d.H, d.V);
    ProfScope prof(PROF_GRU_FWD, 2.0 * d.T * (9.0 * DH * DH + (double)d.V * DH), s, label,
                   4.0 * (9.0 * DH * DH + (double)d.V * DH + (double)d.T * d.V));
    const dim3 grid(kRoles * a.stride);
    const int nj = (d.V + 31) / 32;
    if (nj <= 1) hipLaunchKernelGGL((decode_b1_kernel<1>), grid, dim3(NT), 0, s, a);
    else if (nj == 2) hipLaunchKernelGGL((decode_b1_kernel<2>), grid, dim3(NT), 0, s, a);
    else if (nj == 3) hipLaunchKernelGGL((decode_b1_kernel<3>), grid, dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((decode_b1_kernel<4>), grid, dim3(NT), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
