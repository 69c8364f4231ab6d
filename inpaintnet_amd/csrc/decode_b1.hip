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
//   TA_k (16)    W_hh0 rows of units 32k .. 32k+31:  gh0 for tick t+1 = W_hh0 h0_t + b_hh0          (off the critical path)
//   TBi_k (16)   W_ih1 rows:  gi1 = W_ih1 h0_t + b_ih1, layer 1's cell with gh1 from TBh, publish h1_t
//   TBh_k (16)   W_hh1 rows:  gh1 for tick t+1 = W_hh1 h1_t + b_hh1                                  (off the critical path)
//  A tick is two hand-offs (C -> TBi -> C) with one 1536 x 512 product (spread over 16 workgroups) and the V x 512 head behind them.
//  The tick GRU's hidden state is re-initialised at every beat (decoder.py:485-490): at a beat's first tick the recurrent-side
//  workgroups multiply the beat's initial state instead of the previous tick's output.
//
//  beat path (forward_beat_rnn + the per-beat projections, decoder.py:455-471, 485-497) -- FOLDED INTO THE SAME LAUNCH (`fused`): it
//  used to be eight launches in front (z -> beat state, two 4-step chain launches, three projections: 0.06 of a 0.17 ms call)
//   Z2B_k (4)    hb0 = SELU(W_zb z + b): the beat GRU's initial state
//   BA_k (16)    beat layer 0, product AND cell (its input gates are the constant gvec0): h0b_i, i = 0 .. 3
//   BBi_k / BBh_k (16 + 16)  beat layer 1: input-side product + cell / recurrent-side product -> beat output i
//   PH_k (8), PI_k (4)     ht0_i = SELU(W_bh out_i + b) (initial tick state of beat i), c_i = SELU(W_bi out_i + b)
//   CG_k (16)    cgi_i = W_ih0(tick)[:, E:] c_i   (the beat-constant half of the tick GRU's input projection)
//  Every step of the beat path writes its own granules (nothing is overwritten: tag 1 = written), the tick workgroups pick ht0_i /
//  cgi_i up at beat i's first tick.  While the beat workgroups compute, the tick workgroups load their weights; beats 1 .. 3 are
//  ready long before tick 6 i needs them: the call's critical path is [weights in] + z2b + L0 + L1 + projection + cgi of BEAT 0,
//  then the 24 ticks.
//
// Thread (p, s) = (tid >> 4, tid & 15) of a product workgroup holds the k slice s of R rows (3 = the gate rows of unit p of its 32
// units; 4 or 8 plain rows) in 96-128 VGPRs; the vector's slices sit SK + 4 floats apart in LDS (16 distinct 16-byte reads of a wave
// instruction fall into 16 different bank quads); the 16 partial sums of a row meet by four DPP adds inside the 16-lane row, and
// lane 0 of the row computes the unit's cell -- no LDS, no barrier behind a product.
//
// MORE THAN ONE MEASURE (B = 2 .. 4: the reference's non-auto-regressive inpainting decodes its n_target measures in one call,
// LatentRNN/latent_rnn.py:237-240): the same workgroups loop over the rows inside every phase -- the weights are in registers
// already, a row costs one more product per phase (~0.2 us), and ALL rows' granules of a phase are requested together, so the two
// hand-offs of a tick are paid once, not per row.  B is rounded up to NB = 1 / 2 / 4 (rows beyond B repeat row B - 1 and store
// nothing); every row has its own granule area.  (Eight rows do not fit: workgroup C keeps seven floats and six granule words per row
// and unit next to its slice of the head, 256 VGPRs are gone at NB = 8.)
// THREE TO SIXTEEN MEASURES: TEAMS of the tick path's 49 workgroups in one launch, every team with its own rows and granule areas: teams
// of TWO rows while they fit the chip (a two-row tick is 5.5 us, a four-row tick 8.4; five teams = 245 of 256 CUs: B <= 10), of four
// rows beyond (196 CUs at B = 16).  With two or three teams (B <= 6) the beat path's 80 workgroups still fit beside them (227 CUs)
// and serve all six rows of the call; beyond, the beat path runs as its own launches in front.  Per call
// (profiles/r05_u_decode_team_rows.txt): B = 4 0.194 ms (one four-row team: 0.256), B = 6 0.201, B = 8 0.224 (four-row teams 0.288;
// decode_chain.hip 0.348), B = 16 0.30 (0.355).
//
// THE MERGED BUILD (one row with V <= 64, two rows with V <= 32 -- what fits 256 registers without spills): workgroup C does not
// exist.  Its work -- layer 0's cell, which needs no product, the V x 512 head and the argmax -- is REPLICATED in every TBi_k: h0_t
// never leaves the workgroup, the token never travels, and a tick is ONE hand-off (the all-gather of h1_t among the 16 workgroups)
// instead of two.  The 16 copies run the same instructions on the same values and agree bit for bit.  3.95 -> 3.78 us per tick:
// less than the 0.8 us a hand-off costs, because an all-gather among 16 waits for the slowest of 16 (1.5 us, profiles/r05_decode_b1_*).
// Shapes: H = 512, Z = 256, V <= 128, <= 4 beats, B <= 16, inference (no dropout mask, no backward saves); else decode_chain.hip.
#include <cstdio>
#include <cstdlib>
#include "chain.h"
#define INET_GRANULE_KID chain::K_DECODE_B1
#include "granule.h"
#include "prof.h"
#include "decode_chain.h"

namespace {
using namespace granule;

constexpr int DH = 512, D3 = 3 * DH, DZ = 256, NT = 512, NS = 16, XS = NS * (DH / NS + 4), UW = 32;   // UW: units per GRU product workgroup
constexpr int NU = DH / UW;                                      // workgroups per 1536 x 512 matrix
constexpr int R_C = 0, R_TA = 1, R_TBI = R_TA + NU, R_TBH = R_TBI + NU, kTickRoles = R_TBH + NU;
// fused roles behind the tick roles
constexpr int R_Z2B = kTickRoles, R_BA = R_Z2B + 4, R_BBI = R_BA + NU, R_BBH = R_BBI + NU, R_PH = R_BBH + NU, R_PI = R_PH + 8,
              R_CG = R_PI + 4, kFusedRoles = R_CG + NU;
// granule map of ONE row (8-byte units); row r lives at r * G_END
constexpr int G_H0 = 0, G_H1 = DH, G_GH0 = 2 * DH, G_GH1 = 2 * DH + D3, G_H1X = 2 * DH + 2 * D3, G_GH0X = G_H1X + DH, G_H0X = G_GH0X + D3,
              G_TICK_END = G_H0X + DH;
// (G_H0X / G_H1X: the second slots of h0 and h1 -- every state vector alternates between two slots by its tag's parity, see "Two
//  slots" below; G_GH0X: the second slot of gh0 in the merged build, where gh0 has 16 readers as well)
// Two slots.  A state vector (h0_t, h1_t) is read by SEVENTEEN workgroups -- the 16 recurrent-side ones of its layer and the next
// stage -- while its writer's next write waits only for what ITS units need (gh of its own units, from ONE of those 16): with one
// slot a recurrent-side workgroup that is a tick late would find tag t + 2 where it looks for t + 1, exact-match tags never
// match again and the wait runs into its bound (a liveness margin of one tick, ~3 us).  With two slots by tag parity
// the overwrite of h_t happens at h_{t+2}, whose writer needs gh_{t+2} of its units, which needs ALL of h_{t+1}, whose every unit
// needed gh_{t+1} from its own recurrent-side workgroup -- so all 16 have read h_t by then, transitively.
__device__ __forceinline__ int slot_h0(unsigned tag) { return (tag & 1) ? G_H0X : G_H0; }
__device__ __forceinline__ int slot_h1(unsigned tag) { return (tag & 1) ? G_H1X : G_H1; }
constexpr int G_HB0 = G_TICK_END, G_H0B = G_HB0 + 2 * DH, G_H1B = G_H0B + 4 * DH, G_GH1B = G_H1B + 4 * DH, G_C = G_GH1B + 4 * D3,
              G_HT0 = G_C + 4 * DH, G_CGI = G_HT0 + 4 * 2 * DH, G_H0N = G_CGI + 4 * D3, G_H1N = G_H0N + 2 * DH, G_GH0N = G_H1N + 2 * DH,
              G_XCC = G_GH0N + 2 * D3, G_END = G_XCC + 64;
// (G_H0N / G_H1N / G_GH0N: XCD-LOCAL copies of h0 / h1 / gh0, two slots each, written with plain stores for the readers on the writer's
//  XCD -- "One XCD for the critical path" below; G_XCC: the critical workgroups' XCC ids, granule::same_xcd)
static_assert(2 * G_END == kDecodeB1WordsPerRow, "the workspace's granule area holds the map");
// One XCD for the critical path (round 6).  A tick's critical hand-offs run among C and the 16 TBi (h0 out, h1 back), or among
// the 16 CB of the merged build (the all-gather of h1: 1.54 of a 3.78 us tick, profiles/r05_n_decode_b1_merged_stamps.txt); the
// recurrent-side workgroups TA / TBh read the same vectors a tick early, off the critical path.  An agent-scope granule store
// drops the line from the XCD's L2 and sends every reader through the memory side; a PLAIN store stays in L2, where the sc1
// loads of readers ON THE SAME XCD find it (granule.h: 10 % of AnticipationRNN's token pass) -- and is never seen from another
// XCD.  So the launch maps workgroup ids to roles such that every team's critical workgroups are ids of one residue mod 8
// (`place`: one XCD under the round-robin dispatch observed today), the critical workgroups CHECK that they really share an XCD
// (same_xcd: XCC ids exchanged with agent-scope granules, one hand-off at the start), and if so every state vector is stored
// TWICE: a plain store into the XCD-local copy for the critical readers, the agent-scope store for TA / TBh wherever they run.
// If the check says no, everybody uses the agent-scope copies: correct under any placement.
__device__ __forceinline__ int slot_h0n(unsigned tag) { return G_H0N + (int)(tag & 1) * DH; }
__device__ __forceinline__ int slot_h1n(unsigned tag) { return G_H1N + (int)(tag & 1) * DH; }
__device__ __forceinline__ void put_local(unsigned long long* g, float v, unsigned tag) {
    const unsigned long long x = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(g), "v"(x) : "memory");
}
constexpr int kCrit = 17;                        // critical workgroups per team: C and the 16 TBi (merged build: C's slot idles)
// ... and kCritTA = 32 in the ONE-team merged build: the 16 TA join them (16 CB + 16 TA = the 32 CUs of an XCD).  There layer 0's
// recurrent summands were the last thing a tick waited for (0.27 us: TA gathers h0_t, multiplies, publishes gh0 -- two cross-XCD
// hand-offs that end ~0.3 us behind the argmax); inside the XCD both hand-offs use plain stores and gh0 is there when it is looked at.
constexpr int kCritTA = 2 * NU;                  // (no slot for C: it does not exist in the merged build)
// workgroup id -> (team, role) with every team's critical workgroups on ids of one residue mod 8; false: no role (the id leaves)
__host__ __device__ inline int crit_below(int b, int teams, int crit) {
    int n = 0;
    for (int x = 0; x < teams; ++x)
        if (b > x) { const int c = (b - x + 7) / 8; n += c < crit ? c : crit; }
    return n;
}
// (rteams: the groups of 2 NU recurrent-side workgroups -- one per team, or the shared groups' count; `team` of such a role = its group)
__host__ __device__ inline bool place_role(int b, int teams, int rteams, int beat_wgs, int crit, int& team, int& role) {
    const int x = b & 7, i = b >> 3;
    if (x < teams && i < crit) {
        team = x;
        role = crit == kCritTA ? (i < NU ? R_TBI + i : R_TA + (i - NU)) : crit == NU ? R_TBI + i : (i == 0 ? R_C : R_TBI + i - 1);
        return true;
    }
    const int n = b - crit_below(b, teams, crit);               // the id's rank among the non-critical ones
    const int per = crit == kCritTA ? NU : 2 * NU;              // recurrent-side workgroups per team that are NOT critical
    if (n < rteams * per) {
        team = n / per;
        const int r = n % per;
        role = crit == kCritTA ? R_TBH + r : (r < NU ? R_TA + r : R_TBH + (r - NU));
        return true;
    }
    if (n - rteams * per < beat_wgs) { team = 0; role = kTickRoles + (n - rteams * per); return true; }
    return false;
}

struct B1Args {
    int B, T, G, V, Z, stride, fused, teams;     // teams: groups of kTickRoles workgroups, NB rows each (tick path only beyond one)
    int rgroups;                                 // > 0: SHARED recurrent groups -- TA / TBh workgroups of NBR rows each serve several critical teams ("Shared recurrent groups")
    int crit;                                    // critical workgroups per team under place_role: kCrit, or kCritTA (one-team merged build)
    int place;                                   // 1: ids -> roles by place_role (critical workgroups of a team on one XCD), XCD-local copies requested
    const float* W_hh0; const float* b_hh0; const float* cgi; const float* table;
    const float* W_ih1; const float* b_ih1; const float* W_hh1; const float* b_hh1;
    const float* W_out; const float* b_out; const float* ht0;
    float* weights; long long* samples;
    unsigned long long* ex;
    unsigned long long* stamps;                  // diagnostics: [C, TBi_0][T][8] wall-clock ticks (10 ns), or null
    DecodeB1Beat bp;                             // the beat path's operands (fused)
    chain::Status status;
};

#define B1_STAMP(who, t, i) do { if (stamps && tid == 0) stamps[((who) * 32 + (t)) * 8 + (i)] = wall_clock64(); } while (0)

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
// NR x NG granules of one tag, `rs` apart between rows and `gs` apart between gates: all requested before the first is looked at
template <int NR, int NG>
__device__ __forceinline__ bool get_2d(const unsigned long long* g, int rs, int gs, unsigned tag, const chain::Status& st,
                                       float (&v)[NR][NG], unsigned long long (&w)[NR][NG], bool first = true) {
    unsigned spins = 0;
    for (;;) {
        if (first) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int i = 0; i < NG; ++i) w[r][i] = peek(g + (long)r * rs + (long)i * gs);
        }
        first = true;
        bool all = true;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int i = 0; i < NG; ++i) all &= (unsigned)(w[r][i] >> 32) == tag;
        if (all) {
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int i = 0; i < NG; ++i) v[r][i] = __uint_as_float((unsigned)w[r][i]);
            if (spins > chain::kGranuleSlowSpins) note_slow(st, tag, spins, false);
            return true;
        }
        if (++spins > kSpin ||
            ((spins & 1023) == 0 && __hip_atomic_load(st.dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != chain::ST_OK)) {
            note_slow(st, tag, spins, true);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// NB granules of one tag, one per row, where only the first `nact` rows have a producer: the others read row nact - 1 again
template <int NB>
__device__ __forceinline__ bool get_rows(const unsigned long long* g, int nact, unsigned tag, const chain::Status& st, float (&v)[NB],
                                         unsigned long long (&w)[NB]) {
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < NB; ++i) w[i] = peek(g + (long)min(i, nact - 1) * G_END);
        bool all = true;
#pragma unroll
        for (int i = 0; i < NB; ++i) all &= (unsigned)(w[i] >> 32) == tag;
        if (all) {
#pragma unroll
            for (int i = 0; i < NB; ++i) v[i] = __uint_as_float((unsigned)w[i]);
            if (spins > chain::kGranuleSlowSpins) note_slow(st, tag, spins, false);
            return true;
        }
        if (++spins > kSpin ||
            ((spins & 1023) == 0 && __hip_atomic_load(st.dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != chain::ST_OK)) {
            note_slow(st, tag, spins, true);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

template <int NB>
struct Ctx {                                     // what every role needs
    const B1Args& a; unsigned long long* ex; float (*xs)[2][XS]; volatile int* bad; int tid;
    int rb, nrow;                                // the team's first row, and how many of its NB rows exist
    int nact;                                    // rows that have a producer (NB for a team; a shared recurrent group's may be fewer)
    // every thread fetches granule `tid` (of the vector at `off`) of every row and files it as that row's x (buffer `buf`)
    __device__ __forceinline__ void gather(int off, unsigned tag, int buf) const {
        float v[NB];
        unsigned long long w[NB];
        if (nact == NB) {
            if (!get_n<NB>(ex + off + tid, G_END, tag, a.status, v, w)) *bad = 1;
        } else {
            if (!get_rows<NB>(ex + off + tid, nact, tag, a.status, v, w)) *bad = 1;
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) xs[r][buf][xs_index<32>(tid)] = v[r];
    }
};

// ---- tick path: recurrent side of a layer, off the critical path: gh for tick t = W_hh x + b_hh, x = the beat's initial state at
// a beat's first tick, else the layer's output of tick t - 1 ----
// `near` (TA of the one-team merged build, once the 32 critical workgroups have found themselves on one XCD): the input is read from
// its XCD-local copy (g_in_near + parity * DH) and the output is ALSO written there (g_out_near + parity * D3) with plain stores.
template <int NB>
__device__ __forceinline__ void tick_recurrent_role(const Ctx<NB>& c, int k, const float* __restrict__ W, const float* __restrict__ bias,
                                                    int layer, int g_in, int g_out, int g_in_odd, int g_out_odd, bool near = false,
                                                    int g_in_near = 0, int g_out_near = 0) {
    const B1Args& a = c.a;
    const int tid = c.tid, p = tid >> 4, s = tid & 15, u = UW * k + p;
    const int row[3] = {u, DH + u, 2 * DH + u};
    float w[3][32];
    load_rows<3, 32>(w, W, DH, row, s);
    float b[3] = {0.f, 0.f, 0.f};
    if (s == 0) {
#pragma unroll
        for (int g = 0; g < 3; ++g) b[g] = bias[g * DH + u];
    }
    for (int t = 0; t < a.T; ++t) {
        // (the previous tick's output is waited for at a beat's first tick too, although the beat's initial state is what gets
        //  multiplied: the single-buffered granules are safe only while every producer stays behind its consumers -- a workgroup
        //  that ran ahead here would overwrite gh of tick t - 1 before the cell that needs it has looked)
        if (t > 0) c.gather(near ? g_in_near + (t & 1) * DH : ((t & 1) ? g_in_odd : g_in), (unsigned)t, t & 1);
        if (t % a.G == 0) {
            const int beat = t / a.G;
            if (a.fused) c.gather(G_HT0 + beat * 2 * DH + layer * DH, 1u, t & 1);
            else {
#pragma unroll
                for (int r = 0; r < NB; ++r)
                    c.xs[r][t & 1][xs_index<32>(tid)] = a.ht0[((long)beat * a.B + c.rb + min(r, c.nrow - 1)) * 2 * DH + layer * DH + tid];
            }
        }
        lds_barrier();
        if (*c.bad) break;
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            if (r >= c.nact) continue;                         // (a shared recurrent group's rows without a team)
            float y[3];
            dot_rows<3, 32>(w, c.xs[r][t & 1] + 36 * s, y);
            if (s == 0) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    if (near) put_local(c.ex + (long)r * G_END + g_out_near + ((t + 1) & 1) * D3 + g * DH + u, y[g] + b[g], (unsigned)t + 1u);
                    put(c.ex + (long)r * G_END + (((t + 1) & 1) ? g_out_odd : g_out) + g * DH + u, y[g] + b[g], (unsigned)t + 1u);
                }
            }
        }
    }
}

// ---- beat path: one 512-input product per beat (R plain rows per thread): out_i[row] = f(W x_i + b), x_i / out_i = granule arrays
// with one slot per beat ----
template <int NB, int R, bool SELU_OUT>
__device__ __forceinline__ void beat_product_role(const Ctx<NB>& c, int row0, const float* __restrict__ W, long ld, const float* __restrict__ bias,
                                                  int g_in, int in_stride, int g_out, int out_stride, int nb) {
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
        c.gather(g_in + i * in_stride, 1u, i & 1);
        lds_barrier();
        if (*c.bad) break;
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            float y[R];
            dot_rows<R, 32>(w, c.xs[r][i & 1] + 36 * s, y);
#pragma unroll
            for (int j = 0; j < R; ++j)
                if (s == j) {
                    const float v = y[j] + b[j];
                    put(c.ex + (long)r * G_END + g_out + i * out_stride + row[j], SELU_OUT ? selu_f(v) : v, 1u);
                }
        }
    }
}

// ---- beat path (the launch's other 80 workgroups when it is folded in): NR rows, which may be MORE than a tick team's -- with two or
// three two-row teams (B = 3 .. 6) the beat path's workgroups serve all rows of the call ----
template <int NR>
__device__ __forceinline__ void beat_path_role(const Ctx<NR>& c, int role, int nb) {
    const B1Args& a = c.a;
    const DecodeB1Beat& bp = a.bp;
    const int tid = c.tid;
    if (role < R_BA) {
        // ---- Z2B_k: hb0 = SELU(W_zb z + b_zb) (decoder.py:455-461), 256 rows per workgroup, K = 256: 8 rows x 16 values per thread ----
        const int k = role - R_Z2B, p = tid >> 4, s = tid & 15;
        int row[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) row[i] = 256 * k + 8 * p + i;
        float w[8][16];
        load_rows<8, 16>(w, bp.zb_w, DZ, row, s);
        if (tid < DZ) {
#pragma unroll
            for (int r = 0; r < NR; ++r) c.xs[r][0][xs_index<16>(tid)] = bp.z[(long)(c.rb + min(r, c.nrow - 1)) * DZ + tid];
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float y[8];
            dot_rows<8, 16>(w, c.xs[r][0] + 20 * s, y);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s == j) put(c.ex + (long)r * G_END + G_HB0 + row[j], selu_f(y[j] + bp.zb_b[row[j]]), 1u);
        }
    } else if (role < R_BBI) {
        // ---- BA_k: beat layer 0, product and cell (the input gates are the constant gvec0 = b_0 W_ih[:, 0] + b_ih) ----
        const int k = role - R_BA, p = tid >> 4, s = tid & 15, u = UW * k + p;
        const int row[3] = {u, DH + u, 2 * DH + u};
        float w[3][32];
        load_rows<3, 32>(w, bp.W_hh0, DH, row, s);
        const bool cell = s == 0;
        float gv[3] = {0.f, 0.f, 0.f}, bh[3] = {0.f, 0.f, 0.f}, h[NR];
        if (cell) {
#pragma unroll
            for (int g = 0; g < 3; ++g) { gv[g] = bp.gvec0[g * DH + u]; bh[g] = bp.b_hh0[g * DH + u]; }
        }
        for (int i = 0; i < nb; ++i) {
            c.gather(i == 0 ? G_HB0 : G_H0B + (i - 1) * DH, 1u, i & 1);
            lds_barrier();
            if (*c.bad) break;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                float y[3];
                dot_rows<3, 32>(w, c.xs[r][i & 1] + 36 * s, y);
                if (cell) {
                    if (i == 0) h[r] = c.xs[r][0][xs_index<32>(u)];
                    h[r] = gru_cell(gv[0], gv[1], gv[2], y[0] + bh[0], y[1] + bh[1], y[2] + bh[2], h[r]);
                    put(c.ex + (long)r * G_END + G_H0B + i * DH + u, h[r], 1u);
                }
            }
        }
    } else if (role < R_BBH) {
        // ---- BBi_k: beat layer 1's input-side product and its cell -> the beat outputs ----
        const int k = role - R_BBI, p = tid >> 4, s = tid & 15, u = UW * k + p;
        const int row[3] = {u, DH + u, 2 * DH + u};
        float w[3][32];
        load_rows<3, 32>(w, bp.W_ih1, DH, row, s);
        const bool cell = s == 0;
        float bi[3] = {0.f, 0.f, 0.f}, h[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) h[r] = 0.f;
        if (cell) {
#pragma unroll
            for (int g = 0; g < 3; ++g) bi[g] = bp.b_ih1[g * DH + u];
            unsigned long long hw0[NR];
            if (!get_n<NR>(c.ex + G_HB0 + DH + u, G_END, 1u, a.status, h, hw0)) *c.bad = 1;      // layer 1's initial state
        }
        for (int i = 0; i < nb; ++i) {
            float gh[NR][3];
            unsigned long long hw[NR][3];
            if (cell && !get_2d<NR, 3>(c.ex + G_GH1B + i * D3 + u, G_END, DH, 1u, a.status, gh, hw)) *c.bad = 1;
            c.gather(G_H0B + i * DH, 1u, i & 1);
            lds_barrier();
            if (*c.bad) break;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                float y[3];
                dot_rows<3, 32>(w, c.xs[r][i & 1] + 36 * s, y);
                if (cell) {
                    h[r] = gru_cell(y[0] + bi[0], y[1] + bi[1], y[2] + bi[2], gh[r][0], gh[r][1], gh[r][2], h[r]);
                    put(c.ex + (long)r * G_END + G_H1B + i * DH + u, h[r], 1u);
                }
            }
        }
    } else if (role < R_PH) {
        // ---- BBh_k: beat layer 1's recurrent-side product for step i from the output of step i - 1 (the initial state at i = 0) ----
        const int k = role - R_BBH, p = tid >> 4, s = tid & 15, u = UW * k + p;
        const int row[3] = {u, DH + u, 2 * DH + u};
        float w[3][32];
        load_rows<3, 32>(w, bp.W_hh1, DH, row, s);
        float b[3] = {0.f, 0.f, 0.f};
        if (s == 0) {
#pragma unroll
            for (int g = 0; g < 3; ++g) b[g] = bp.b_hh1[g * DH + u];
        }
        for (int i = 0; i < nb; ++i) {
            c.gather(i == 0 ? G_HB0 + DH : G_H1B + (i - 1) * DH, 1u, i & 1);
            lds_barrier();
            if (*c.bad) break;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                float y[3];
                dot_rows<3, 32>(w, c.xs[r][i & 1] + 36 * s, y);
                if (s == 0) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) put(c.ex + (long)r * G_END + G_GH1B + i * D3 + g * DH + u, y[g] + b[g], 1u);
                }
            }
        }
    } else if (role < R_PI) {
        beat_product_role<NR, 4, true>(c, 128 * (role - R_PH), bp.bh_w, DH, bp.bh_b, G_H1B, DH, G_HT0, 2 * DH, nb);     // ht0_i
    } else if (role < R_CG) {
        beat_product_role<NR, 4, true>(c, 128 * (role - R_PI), bp.bi_w, DH, bp.bi_b, G_H1B, DH, G_C, DH, nb);           // c_i
    } else {
        beat_product_role<NR, 3, false>(c, 96 * (role - R_CG), bp.wih0_c, bp.wih0_ld, nullptr, G_C, DH, G_CGI, D3, nb); // cgi_i
    }
}

// Shared recurrent groups (round 6, NBR > NB): seven to sixteen measures used to run as teams of FOUR rows (49 workgroups each: four
// two-row teams are all the chip holds) at 8-9 us per tick against 4.7 for a two-row team.  But only 17 of a team's 49 workgroups
// are on the tick's critical path (C and the 16 TBi); TA and TBh produce the NEXT tick's recurrent summands and idle most of a
// tick.  So they are shared: two-row critical teams (17 workgroups each, on one XCD) for every pair of rows, and recurrent groups
// of 32 workgroups that serve NBR = 6 rows -- three teams -- each: 8 x 17 + 3 x 32 = 232 workgroups for sixteen measures, every
// row on a two-row tick.  A group's rows without a team (the last group of a call) are skipped (Ctx.nact).
template <int NJ, bool FUSED, int NB, int NBB, int NBR = NB>
__global__ __launch_bounds__(NT) void decode_b1_kernel(B1Args a) {
    constexpr int XROWS = NB > NBB ? (NB > NBR ? NB : NBR) : (NBB > NBR ? NBB : NBR);
    __shared__ __attribute__((aligned(16))) float xs[XROWS][2][XS];
    __shared__ float lgs[NB][32 * NJ];
    __shared__ int toks[NB];
    __shared__ int bad_s;
    __shared__ int near_s;
    if (blockIdx.x % a.stride) return;
    // more than four rows: `teams` teams of the tick path's workgroups, NB rows and one granule area per row each, nothing shared
    // (workgroup order with several teams: the teams' tick workgroups first, the beat path's 80 behind them when it is folded in;
    //  a.place: place_role's order instead)
    const int wg = blockIdx.x / a.stride, tick_wgs = a.teams * kTickRoles;
    int team = (a.teams > 1 && wg < tick_wgs) ? wg / kTickRoles : 0;
    int role = a.teams > 1 ? (wg < tick_wgs ? wg % kTickRoles : kTickRoles + (wg - tick_wgs)) : wg;
    if (a.place == 1 && !place_role((int)blockIdx.x, a.teams, a.rgroups ? a.rgroups : a.teams, a.fused ? kFusedRoles - kTickRoles : 0, a.crit, team, role)) return;
    const int rb = team * NB, nrow = min(NB, a.B - rb);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* const ex = a.ex + (long)rb * G_END;
    unsigned long long* const stamps = team == 0 ? a.stamps : nullptr;      // (diagnostics: the first team's workgroups only)
    volatile int* const bad = &bad_s;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    const Ctx<NB> c{a, ex, xs, bad, tid, rb, nrow, NB};
    const int nb = a.T / a.G;
    const DecodeB1Beat& bp = a.bp;
    // MG, the merged build: workgroup C does not exist, every TBi_k does C's work for itself next to its own (CB below)
    constexpr bool MG = NB * NJ <= 2;

    if (role == R_C) {
        if (MG) return;
        // ---- C: layer 0's cell (its three summands are made elsewhere and arrive), the output projection, argmax ----
        const bool near = a.place && same_xcd(ex + G_XCC, 0, kCrit, a.status, &near_s);
        const int rv = tid >> 4, s = tid & 15;
        int row[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) row[j] = min(rv + 32 * j, a.V - 1);
        float wo[NJ][32];
        load_rows<NJ, 32>(wo, a.W_out, DH, row, s);
        float bo[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bo[j] = a.b_out[row[j]];
        const int u = tid;                                     // every thread owns one unit of layer 0, for every row
        float h0[NB], cg[NB][3], gh[NB][3];
        unsigned long long hw[NB][3];
        int tok[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) { h0[r] = 0.f; tok[r] = a.V; }      // row V of the table: the start symbol x_0
        float tb[NB][3];                                       // the tokens' rows of the gather table, requested a phase ahead
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int g = 0; g < 3; ++g) tb[r][g] = a.table[(long)tok[r] * D3 + g * DH + u];
        if (!get_2d<NB, 3>(ex + G_GH0 + u, G_END, DH, 1u, a.status, gh, hw)) *bad = 1;
        for (int t = 0; t < a.T; ++t) {
            const bool more = t + 1 < a.T;
            if (t % a.G == 0) {
                const long beat = t / a.G;
                if (FUSED) {
                    unsigned long long cw[NB][3], h1w[NB];
                    if (!get_n<NB>(ex + G_HT0 + beat * 2 * DH + u, G_END, 1u, a.status, h0, h1w)) *bad = 1;
                    if (!get_2d<NB, 3>(ex + G_CGI + beat * D3 + u, G_END, DH, 1u, a.status, cg, cw)) *bad = 1;
                } else {
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        const long br = beat * a.B + rb + min(r, nrow - 1);
                        h0[r] = a.ht0[br * 2 * DH + u];
#pragma unroll
                        for (int g = 0; g < 3; ++g) cg[r][g] = a.cgi[br * D3 + g * DH + u];
                    }
                }
            }
            B1_STAMP(0, t, 0);
            {
                // (every row's token rows of the table were requested the moment the tokens were known, behind the previous tick's
                //  argmax and IN FRONT of the wait for this tick's gh0: two round trips side by side instead of one after the other)
#pragma unroll
                for (int r = 0; r < NB; ++r)
                    h0[r] = gru_cell(cg[r][0] + tb[r][0], cg[r][1] + tb[r][1], cg[r][2] + tb[r][2], gh[r][0], gh[r][1], gh[r][2], h0[r]);
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    if (near) put_local(ex + (long)r * G_END + slot_h0n((unsigned)t + 1u) + u, h0[r], (unsigned)t + 1u);
                    put(ex + (long)r * G_END + slot_h0((unsigned)t + 1u) + u, h0[r], (unsigned)t + 1u);
                }
            }
            B1_STAMP(0, t, 1);
            c.gather(near ? slot_h1n((unsigned)t + 1u) : slot_h1((unsigned)t + 1u), (unsigned)t + 1u, 0);
            B1_STAMP(0, t, 2);
            lds_barrier();
            if (*bad) break;
            B1_STAMP(0, t, 3);
            // the next tick's recurrent summands left TA_k about when h1_t left TBi_k: request them now, look at them behind the head
            // (measured and not kept: asking for them inside the polls for h1_t -- four requests per poll instead of one slow the
            //  hand-off itself down, 0.122 -> 0.140 ms per call)
            if (more) {
#pragma unroll
                for (int r = 0; r < NB; ++r)
#pragma unroll
                    for (int g = 0; g < 3; ++g) hw[r][g] = peek(ex + (long)r * G_END + G_GH0 + g * DH + u);
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                float y[NJ];
                dot_rows<NJ, 32>(wo, xs[r][0] + 36 * s, y);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {                 // lane j of the 16-lane row: logit rv + 32 j
                    const int v = rv + 32 * j;
                    if (s == j && v < a.V) {
                        float lg = y[j] + bo[j];
                        lg = lg > 0.f ? lg : 0.f;              // ReLU (decoder.py:372, 503)
                        lgs[r][v] = lg;
                        if (r < nrow) a.weights[((long)(rb + r) * a.T + t) * a.V + v] = lg;
                    }
                }
            }
            lds_barrier();
            B1_STAMP(0, t, 4);
            // wave r takes the argmax of row r (one row: every wave takes it for itself and the second barrier is not needed): the
            // maximum by DPP, its lowest index by ballot
            if (NB == 1 || wave < NB) {
                const int arow = NB == 1 ? 0 : wave;
                constexpr int NVL = (32 * NJ + 63) / 64;
                float lg[NVL], m = -1.f;
#pragma unroll
                for (int j = 0; j < NVL; ++j) {
                    const int v = lane + 64 * j;
                    lg[j] = v < a.V ? lgs[arow][v] : -1.f;      // (below every post-ReLU logit)
                    m = fmaxf(m, lg[j]);
                }
                m = wave_max_dpp(m);
                int bi = 0;
#pragma unroll
                for (int j = NVL - 1; j >= 0; --j) {
                    const unsigned long long eq = __ballot(lg[j] == m);
                    if (eq) bi = 64 * j + __builtin_ctzll(eq);
                }
                bi = bi < a.V ? bi : 0;
                if (NB == 1) {
                    tok[0] = bi;
                    if (tid == 0) a.samples[(long)rb * a.T + t] = bi;
                } else if (lane == 0) {
                    toks[arow] = bi;
                    if (arow < nrow) a.samples[(long)(rb + arow) * a.T + t] = bi;
                }
            }
            if (NB > 1) {
                lds_barrier();                                 // (toks; lgs is rewritten behind the next tick's first barrier)
#pragma unroll
                for (int r = 0; r < NB; ++r) tok[r] = toks[r];
            }
            B1_STAMP(0, t, 5);
            if (more) {
#pragma unroll
                for (int r = 0; r < NB; ++r)
#pragma unroll
                    for (int g = 0; g < 3; ++g) tb[r][g] = a.table[(long)tok[r] * D3 + g * DH + u];
            }
            if (more && !get_2d<NB, 3>(ex + G_GH0 + u, G_END, DH, (unsigned)t + 2u, a.status, gh, hw, false)) *bad = 1;
            B1_STAMP(0, t, 6);
        }
    } else if (role < R_TBI) {
        if (NBR != NB && a.crit != kCritTA) {                  // a shared group: rows [NBR team, NBR team + NBR) of the call
            const int rbr = team * NBR;
            const Ctx<NBR> cr{a, a.ex + (long)rbr * G_END, xs, bad, tid, rbr, min(NBR, a.B - rbr), min(NBR, a.teams * NB - rbr)};
            tick_recurrent_role<NBR>(cr, role - R_TA, a.W_hh0, a.b_hh0, 0, G_H0, G_GH0, G_H0X, MG ? G_GH0X : G_GH0);
        } else {
            // (merged build under place_role with kCritTA: the TA are critical workgroups, index 16 + k of the team's XCC-id check of 32)
            const bool ta_near = MG && a.place == 1 && a.crit == kCritTA && same_xcd(ex + G_XCC, NU + (role - R_TA), 2 * NU, a.status, &near_s);
            tick_recurrent_role<NB>(c, role - R_TA, a.W_hh0, a.b_hh0, 0, G_H0, G_GH0, G_H0X, MG ? G_GH0X : G_GH0, ta_near, G_H0N, G_GH0N);
        }
    } else if (role < R_TBH && MG) {
        // ---- CB_k (merged build): C's work REPLICATED in every TBi_k.  Layer 0's cell needs no product (its summands arrive), the
        // head is V x 512: cheap enough to compute 16 times over, and then h0_t never leaves the workgroup and the token never
        // travels -- of the two hand-offs of a tick (C -> TBi -> C) only the all-gather of h1_t among the 16 workgroups is left.
        // All of them run the same instructions on the same values, so they agree on every token bit for bit.  CB_0 alone
        // publishes h0_t (for TA) and writes the outputs.  h0, h1 and gh0 alternate between two granule slots by tag parity: a CB
        // workgroup does not wait for its 15 peers to have READ a value before it writes the next one (a peer's read of h1_t is
        // ordered before its own h1_t+1, which the writer of h1_t+2 has to have seen: two slots are enough; likewise gh0, h0).
        const int k = role - R_TBI, p = tid >> 4, s = tid & 15, uc = UW * k + p;
        const bool with_ta = a.place == 1 && a.crit == kCritTA;           // the 16 TA are part of the check (and of the XCD)
        const bool near = a.place && same_xcd(ex + G_XCC, k, with_ta ? 2 * NU : NU, a.status, &near_s);
        const bool gh0_near = near && with_ta;
        const int row[3] = {uc, DH + uc, 2 * DH + uc};
        float w[3][32];
        load_rows<3, 32>(w, a.W_ih1, DH, row, s);
        int orow[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) orow[j] = min(p + 32 * j, a.V - 1);
        float wo[NJ][32];
        load_rows<NJ, 32>(wo, a.W_out, DH, orow, s);
        float bo[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bo[j] = a.b_out[orow[j]];
        const bool cell = s == 0, out = k == 0;
        float bi[3] = {0.f, 0.f, 0.f};
        if (cell) {
#pragma unroll
            for (int g = 0; g < 3; ++g) bi[g] = a.b_ih1[g * DH + uc];
        }
        const int u = tid;                                     // layer 0: every thread owns one unit, for every row
        float h0[NB], h1[NB], cg[NB][3], gh[NB][3];
        unsigned long long hw[NB][3];
        int tok[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) { h0[r] = 0.f; h1[r] = 0.f; tok[r] = a.V; }
        float tb[NB][3];                                       // the tokens' rows of the gather table, requested a phase ahead
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int g = 0; g < 3; ++g) tb[r][g] = a.table[(long)tok[r] * D3 + g * DH + u];
        if (!get_2d<NB, 3>(ex + (gh0_near ? G_GH0N + D3 : G_GH0X) + u, G_END, DH, 1u, a.status, gh, hw)) *bad = 1;
        for (int t = 0; t < a.T; ++t) {
            const bool more = t + 1 < a.T;
            const unsigned tag = (unsigned)t + 1u;
            const int g_h1 = slot_h1(tag);
            float gh1[NB][3];
            unsigned long long hw1[NB][3];
            if (t % a.G == 0) {
                const long beat = t / a.G;
                if (FUSED) {
                    unsigned long long cw[NB][3], h1w[NB];
                    if (!get_n<NB>(ex + G_HT0 + beat * 2 * DH + u, G_END, 1u, a.status, h0, h1w)) *bad = 1;
                    if (!get_2d<NB, 3>(ex + G_CGI + beat * D3 + u, G_END, DH, 1u, a.status, cg, cw)) *bad = 1;
                    if (cell && !get_n<NB>(ex + G_HT0 + beat * 2 * DH + DH + uc, G_END, 1u, a.status, h1, h1w)) *bad = 1;
                } else {
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        const long br = beat * a.B + rb + min(r, nrow - 1);
                        h0[r] = a.ht0[br * 2 * DH + u];
                        if (cell) h1[r] = a.ht0[br * 2 * DH + DH + uc];
#pragma unroll
                        for (int g = 0; g < 3; ++g) cg[r][g] = a.cgi[br * D3 + g * DH + u];
                    }
                }
            }
            if (out) B1_STAMP(0, t, 0);
            {
                // (the token's rows of the table were requested behind the previous tick's argmax, in front of the wait for gh0)
                // layer 1's recurrent summands were started a tick ago: requested here
                if (cell) {
#pragma unroll
                    for (int r = 0; r < NB; ++r)
#pragma unroll
                        for (int g = 0; g < 3; ++g) hw1[r][g] = peek(ex + (long)r * G_END + G_GH1 + g * DH + uc);
                }
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    h0[r] = gru_cell(cg[r][0] + tb[r][0], cg[r][1] + tb[r][1], cg[r][2] + tb[r][2], gh[r][0], gh[r][1], gh[r][2], h0[r]);
                    xs[r][0][xs_index<32>(u)] = h0[r];
                }
                if (out) {
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        if (gh0_near) put_local(ex + (long)r * G_END + slot_h0n(tag) + u, h0[r], tag);
                        put(ex + (long)r * G_END + slot_h0(tag) + u, h0[r], tag);
                    }
                }
            }
            if (out) B1_STAMP(0, t, 1);
            lds_barrier();
            {
                float y[NB][3];
#pragma unroll
                for (int r = 0; r < NB; ++r) dot_rows<3, 32>(w, xs[r][0] + 36 * s, y[r]);
                if (cell) {
                    if (!get_2d<NB, 3>(ex + G_GH1 + uc, G_END, DH, tag, a.status, gh1, hw1, false)) *bad = 1;
#pragma unroll
                    for (int r = 0; r < NB; ++r)
                        h1[r] = gru_cell(y[r][0] + bi[0], y[r][1] + bi[1], y[r][2] + bi[2], gh1[r][0], gh1[r][1], gh1[r][2], h1[r]);
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        if (near) put_local(ex + (long)r * G_END + slot_h1n(tag) + uc, h1[r], tag);
                        put(ex + (long)r * G_END + g_h1 + uc, h1[r], tag);
                    }
                }
            }
            if (out) B1_STAMP(0, t, 2);
            c.gather(near ? slot_h1n(tag) : g_h1, tag, 1);
            if (out) B1_STAMP(0, t, 3);
            lds_barrier();
            if (*bad) break;
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                float y[NJ];
                dot_rows<NJ, 32>(wo, xs[r][1] + 36 * s, y);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int v = p + 32 * j;
                    if (s == j && v < a.V) {
                        float lg = y[j] + bo[j];
                        lg = lg > 0.f ? lg : 0.f;
                        lgs[r][v] = lg;
                        if (out && r < nrow) a.weights[((long)(rb + r) * a.T + t) * a.V + v] = lg;
                    }
                }
            }
            // the next tick's recurrent summands: TA publishes them ~2 us behind h0_t (gather + product + store), i.e. about HERE -- requested
            // behind the head product, looked at behind the argmax.  (Requested in front of the head product, 1.4 us behind h0_t, the
            // request came back stale most ticks and the look behind the argmax paid a second round trip.)
            if (more) {
                const int g_gh0 = gh0_near ? G_GH0N + (int)((tag + 1u) & 1) * D3 : (((tag + 1u) & 1) ? G_GH0X : G_GH0);
#pragma unroll
                for (int r = 0; r < NB; ++r)
#pragma unroll
                    for (int g = 0; g < 3; ++g) hw[r][g] = peek(ex + (long)r * G_END + g_gh0 + g * DH + u);
            }
            lds_barrier();
            if (out) B1_STAMP(0, t, 4);
            if (NB == 1 || wave < NB) {
                const int arow = NB == 1 ? 0 : wave;
                constexpr int NVL = (32 * NJ + 63) / 64;
                float lg[NVL], m = -1.f;
#pragma unroll
                for (int j = 0; j < NVL; ++j) {
                    const int v = lane + 64 * j;
                    lg[j] = v < a.V ? lgs[arow][v] : -1.f;
                    m = fmaxf(m, lg[j]);
                }
                m = wave_max_dpp(m);
                int best = 0;
#pragma unroll
                for (int j = NVL - 1; j >= 0; --j) {
                    const unsigned long long eq = __ballot(lg[j] == m);
                    if (eq) best = 64 * j + __builtin_ctzll(eq);
                }
                best = best < a.V ? best : 0;
                if (NB == 1) {
                    tok[0] = best;
                    if (out && tid == 0) a.samples[(long)rb * a.T + t] = best;
                } else if (lane == 0) {
                    toks[arow] = best;
                    if (out && arow < nrow) a.samples[(long)(rb + arow) * a.T + t] = best;
                }
            }
            if (NB > 1) {
                lds_barrier();
#pragma unroll
                for (int r = 0; r < NB; ++r) tok[r] = toks[r];
            }
            if (out) B1_STAMP(0, t, 5);
            if (more) {
#pragma unroll
                for (int r = 0; r < NB; ++r)
#pragma unroll
                    for (int g = 0; g < 3; ++g) tb[r][g] = a.table[(long)tok[r] * D3 + g * DH + u];
            }
            if (more && !get_2d<NB, 3>(ex + (gh0_near ? G_GH0N + (int)((tag + 1u) & 1) * D3 : (((tag + 1u) & 1) ? G_GH0X : G_GH0)) + u, G_END, DH, tag + 1u,
                                       a.status, gh, hw, false)) *bad = 1;
            if (out) B1_STAMP(0, t, 6);
        }
    } else if (role < R_TBH) {
        // ---- TBi_k: tick layer 1's input-side product and its cell ----
        const int k = role - R_TBI, p = tid >> 4, s = tid & 15, u = UW * k + p;
        const bool near = a.place && same_xcd(ex + G_XCC, 1 + k, kCrit, a.status, &near_s);
        const int row[3] = {u, DH + u, 2 * DH + u};
        float w[3][32];
        load_rows<3, 32>(w, a.W_ih1, DH, row, s);
        const bool cell = s == 0;                              // lane 0 of the 16-lane row owns the unit
        float bi[3] = {0.f, 0.f, 0.f};
        if (cell) {
#pragma unroll
            for (int g = 0; g < 3; ++g) bi[g] = a.b_ih1[g * DH + u];
        }
        float h1[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) h1[r] = 0.f;
        for (int t = 0; t < a.T; ++t) {
            // the recurrent summands of this tick were started a tick ago (or at the launch, for a beat's first tick)
            float gh[NB][3];
            unsigned long long hw[NB][3];
            if (k == 0) B1_STAMP(1, t, 0);
            if (cell) {
                if (!get_2d<NB, 3>(ex + G_GH1 + u, G_END, DH, (unsigned)t + 1u, a.status, gh, hw)) *bad = 1;
                if (t % a.G == 0) {
                    const long beat = t / a.G;
                    if (FUSED) {
                        unsigned long long h1w[NB];
                        if (!get_n<NB>(ex + G_HT0 + beat * 2 * DH + DH + u, G_END, 1u, a.status, h1, h1w)) *bad = 1;
                    } else {
#pragma unroll
                        for (int r = 0; r < NB; ++r) h1[r] = a.ht0[(beat * a.B + rb + min(r, nrow - 1)) * 2 * DH + DH + u];
                    }
                }
            }
            if (k == 0) B1_STAMP(1, t, 1);
            c.gather(near ? slot_h0n((unsigned)t + 1u) : slot_h0((unsigned)t + 1u), (unsigned)t + 1u, t & 1);
            if (k == 0) B1_STAMP(1, t, 2);
            lds_barrier();
            if (*bad) break;
            if (k == 0) B1_STAMP(1, t, 3);
            // (all rows' products first, then all rows' cells: the transcendental chains of the rows interleave)
            float y[NB][3];
#pragma unroll
            for (int r = 0; r < NB; ++r) dot_rows<3, 32>(w, xs[r][t & 1] + 36 * s, y[r]);
            if (cell) {
#pragma unroll
                for (int r = 0; r < NB; ++r)
                    h1[r] = gru_cell(y[r][0] + bi[0], y[r][1] + bi[1], y[r][2] + bi[2], gh[r][0], gh[r][1], gh[r][2], h1[r]);
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    if (near) put_local(ex + (long)r * G_END + slot_h1n((unsigned)t + 1u) + u, h1[r], (unsigned)t + 1u);
                    put(ex + (long)r * G_END + slot_h1((unsigned)t + 1u) + u, h1[r], (unsigned)t + 1u);
                }
            }
            if (k == 0) B1_STAMP(1, t, 4);
        }
    } else if (role < kTickRoles) {
        if (NBR != NB) {
            const int rbr = team * NBR;
            const Ctx<NBR> cr{a, a.ex + (long)rbr * G_END, xs, bad, tid, rbr, min(NBR, a.B - rbr), min(NBR, a.teams * NB - rbr)};
            tick_recurrent_role<NBR>(cr, role - R_TBH, a.W_hh1, a.b_hh1, 1, G_H1, G_GH1, G_H1X, G_GH1);
        } else
        tick_recurrent_role<NB>(c, role - R_TBH, a.W_hh1, a.b_hh1, 1, G_H1, G_GH1, G_H1X, G_GH1);
    } else if (FUSED) {
        // (one team: the beat path serves the team's NB rows; several teams: all NBB rows of the call, from the call's first granule area)
        const Ctx<NBB> cb{a, a.ex, xs, bad, tid, 0, min(NBB, a.B), NBB};
        beat_path_role<NBB>(cb, role, nb);
    }
    __syncthreads();
    if (bad_s && tid == 0) chain::raise_timeout(a.status);
}

int g_mode = -1;                                 // 0 = off (decode_chain.hip's small-batch builds); 1 = tick path only, consecutive workgroup ids;
                                                 // 2 = tick path only, every 4th id; 3 = beat path folded in; 4 (default) = 3 + every
                                                 // team's critical workgroups on one XCD with XCD-local copies of h0 / h1 (place_role);
                                                 // 5 = test hook: 4's REQUEST for XCD-local copies on mode 3's consecutive ids, where the
                                                 // critical workgroups do not share an XCD -- the XCC-id check has to refuse
int mode() {
    if (g_mode < 0) {
        const char* v = std::getenv("INET_DECODE_B1");
        g_mode = v ? std::atoi(v) : 4;
        if (g_mode < 0 || g_mode > 5) g_mode = 4;
    }
    return g_mode;
}
// workgroup ids a placed launch needs: the smallest count whose non-critical ids hold every non-critical role (and past the last
// critical id)
int placed_grid(int teams, int rteams, int beat_wgs, int crit = kCrit) {
    const int need = rteams * (crit == kCritTA ? NU : 2 * NU) + beat_wgs;
    int g = 8 * (crit - 1) + teams;
    while (g - crit_below(g, teams, crit) < need) ++g;
    return g;
}
}  // namespace

// The launch plan per call size under mode 4 (the default), in one place.  "critical" = the workgroups on a tick's critical path, which get
// workgroup ids of one residue mod 8 (one XCD) and XCD-local copies of what they exchange; M = merged build (rows x ceil(V / 32) <= 2).
//   B = 1        one team, beat path folded in.            M: 16 CB + 16 TA critical (kCritTA), 16 TBh, 80 beat = 129 workgroups
//   B = 2, 3     B one-row teams, beat path folded in (serves three rows).  M: 32 critical per team                  <= 224
//   B = 4 (M)    four one-row teams with CB + TA critical, TBh shared in pairs of rows, beat path folded in             240
//   B = 4 .. 6   B one-row critical teams (16 CB, or C + 16 TBi), TA + TBh shared by groups of three rows, beat folded <= 246
//   B = 7 .. 10  whole two-row teams (49 workgroups, 17 critical) behind the beat path's own launches                  <= 245
//   B = 11 .. 16 two-row critical teams (17) + TA / TBh shared by groups of six rows, behind the beat path's launches  <= 232
// Modes 1-3 keep round 5's plans (one team up to two rows, two-row teams to ten, four-row teams beyond); mode 5 = test hook.
constexpr int kSharedRows = 6;                   // rows a shared recurrent group serves beyond ten measures (8: the group's tick is longer than the two-row teams' and sets the pace)
constexpr int kSharedRowsSmall = 3;              // ... and for four to six measures, where the critical teams have ONE row
// Shared recurrent groups under mode 4 (the kernel's "Shared recurrent groups"): four to six measures = one-row critical teams (the
// 16 CB of the merged build, or C + 16 TBi) + groups of three rows + the beat path's 80 workgroups in the same launch (6 x 17 + 2 x 32
// + 80 = 246); eleven and more = two-row critical teams + groups of six rows.  (Seven to ten: five whole two-row teams fit the chip.)
static bool shared_groups(int B) { return mode() == 4 && (B > 10 || (B > kDecodeB1OneRowTeamsMax && B <= kDecodeB1BeatRowsMax)); }
static int shared_rows(int B) { return B <= kDecodeB1BeatRowsMax ? kSharedRowsSmall : kSharedRows; }
static int shared_group_count(int B) {
    const int rows = B <= kDecodeB1BeatRowsMax ? B : 2 * ((B + 1) / 2);         // rows that have a critical team
    return (rows + shared_rows(B) - 1) / shared_rows(B);
}
int decode_b1_team_rows(int B) {
    if (B <= 1) return 1;
    // two or three measures under mode 4: ONE row per team (round 6).  Two or three one-row teams and the beat path's 80 workgroups fit
    // the chip (3 x 48 + 80), a one-row tick is 3.2 us (merged build, V <= 64) against 4.6 for a two-row team, and each team's 32
    // critical workgroups get an XCD of their own -- the reference's non-auto-regressive inpainting call decodes its 2 .. 4 target
    // measures in one call (LatentRNN/latent_rnn.py:237-240)
    if (mode() == 4 && B <= kDecodeB1BeatRowsMax) return 1;          // (two, three: whole one-row teams; four to six: one-row CRITICAL teams + shared groups)
    if (B <= 2 || shared_groups(B)) return 2;
    return B <= 10 ? 2 : 4;                                    // (whole two-row teams while five of them fit the chip; modes 1-3 beyond: four rows)
}

void decode_b1_set_mode(int m) { g_mode = (m < 0 || m > 5) ? 4 : m; }

bool decode_b1_shape_ok(int B, int H, int V, int T, int G) {
    return mode() != 0 && chain_enabled() && B >= 1 && B <= kDecodeB1MaxRows && H == DH && V >= 1 && V <= 128 && T % G == 0 && T / G <= 4 &&
           kFusedRoles <= chain_capacity() &&
           (shared_groups(B) ? placed_grid(decode_b1_teams(B), shared_group_count(B), B <= kDecodeB1BeatRowsMax ? kFusedRoles - kTickRoles : 0)
                             : decode_b1_teams(B) * kTickRoles) <= chain_capacity();
}
// the beat path's 80 workgroups go into the same launch when they fit beside the teams: one team, or two / three two-row teams (B <= 6:
// 3 x 49 + 80 = 227 of 256 CUs); they then serve all (up to kBeatRowsMax) rows of the call
bool decode_b1_fused(int Z, int B) {
    const int teams = decode_b1_teams(B);
    if (mode() < 3 || Z != DZ) return false;
    if (shared_groups(B))                                      // four to six measures: one-row critical teams + shared groups + the beat path
        return B <= kDecodeB1BeatRowsMax && placed_grid(teams, shared_group_count(B), kFusedRoles - kTickRoles) <= chain_capacity();
    return (teams == 1 || (decode_b1_team_rows(B) == 2 && teams * 2 <= kDecodeB1BeatRowsMax) || (decode_b1_team_rows(B) == 1 && teams <= kDecodeB1OneRowTeamsMax)) &&
           teams * kTickRoles + (kFusedRoles - kTickRoles) <= chain_capacity();
}
bool decode_b1_ok(const DecodeChainArgs& a) {
    const bool train = a.sv0 || a.sv1 || a.mask || a.h0out || a.h1seq;
    return decode_b1_shape_ok(a.B, a.H, a.V, a.T, a.G) && !train && a.b1ex;
}

// What launch_decode_b1 decides for a call, as a value (also behind inet_decode_b1_plan: the planner is tested without a GPU).
struct B1Plan { int teams, nbr, nj, rgroups, crit, place, grid, beat_wgs; bool tbh_pairs; };
static B1Plan make_plan(int B, int V, bool fused, int stride) {
    B1Plan p{};
    p.teams = decode_b1_teams(B);
    p.rgroups = shared_groups(B) ? shared_group_count(B) : 0;
    p.nj = (V + 31) / 32; p.nbr = decode_b1_team_rows(B);
    // FOUR measures with the merged build (V <= 64): every team keeps its 16 TA next to its 16 CB on an XCD of its own (layer 0's
    // recurrent summands stay inside the XCD), only the TBh are shared, two rows per group: 4 x 32 + 2 x 16 + 80 = 240 workgroups.
    // (Five and six would need 288 / 336: they keep the groups of three rows for both recurrent sides.)
    p.tbh_pairs = fused && p.rgroups && B == 4 && p.nbr == 1 && p.nj <= 2;
    if (p.tbh_pairs) p.rgroups = 2;
    const int rteams = p.rgroups ? p.rgroups : p.teams;
    p.beat_wgs = fused ? kFusedRoles - kTickRoles : 0;
    // teams of the merged build (one row with V <= 64, two with V <= 32): the TA join the critical set -- 32 workgroups = one XCD per team
    p.crit = p.tbh_pairs ? kCritTA
             : (!p.rgroups && p.nbr * p.nj <= 2 && placed_grid(p.teams, p.teams, p.beat_wgs, kCritTA) <= chain_capacity()) ? kCritTA
             : (p.rgroups && p.nbr * p.nj <= 2) ? NU : kCrit;    // (merged build with shared groups: the 16 CB alone are a team's critical set)
    p.place = mode() == 4 && stride == 1 && placed_grid(p.teams, rteams, p.beat_wgs, p.crit) <= chain_capacity();
    if (mode() == 5 && stride == 1) { p.place = 2; p.crit = kCrit; }   // (test hook: the request without the placement)
    p.grid = p.place == 1 ? placed_grid(p.teams, rteams, p.beat_wgs, p.crit) : (p.teams * kTickRoles + p.beat_wgs) * stride;
    return p;
}

// Planner self-check without a GPU (tests/test_decode_plan.py): out[8] = {teams, rows per team, shared groups, critical workgroups per
// team, placed, grid, live workgroups, ok}.  ok = every (team, role) the kernel expects appears exactly once among the ids of a placed
// grid, every team's critical roles sit on ids of ONE residue mod 8, no residue carries more than 32 live workgroups (one XCD's CUs),
// and the grid fits the chip.  Returns 0, or -1 for a call the register-resident launch does not take.
int decode_b1_plan_check(int B, int V, int Z, int* out) {
    if (!out || !decode_b1_shape_ok(B, DH, V, 24, 6)) return -1;
    const bool fused = decode_b1_fused(Z, B);
    const B1Plan p = make_plan(B, V, fused, 1);
    int live = 0, ok = 1;
    if (p.place == 1) {
        const int rteams = p.rgroups ? p.rgroups : p.teams;
        const bool merged = p.nbr * p.nj <= 2;
        int per_residue[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // expected roles: per team the critical ones (+ its own recurrent side unless shared), per shared group its recurrent roles, the beat path
        static int seen[16][kFusedRoles], crit_res[16];
        for (int t = 0; t < 16; ++t) { crit_res[t] = -1; for (int r = 0; r < kFusedRoles; ++r) seen[t][r] = 0; }
        for (int b = 0; b < p.grid; ++b) {
            int team = 0, role = 0;
            if (!place_role(b, p.teams, rteams, p.beat_wgs, p.crit, team, role)) continue;
            if (team < 0 || team >= 16 || role < 0 || role >= kFusedRoles) { ok = 0; continue; }
            ++seen[team][role]; ++live; ++per_residue[b & 7];
            const bool critical = role == R_C || (role >= R_TBI && role < R_TBH) || (p.crit == kCritTA && role >= R_TA && role < R_TBI);
            if (critical && role < kTickRoles) {
                if (crit_res[team] < 0) crit_res[team] = b & 7;
                else if (crit_res[team] != (b & 7)) ok = 0;
            }
        }
        for (int t = 0; t < p.teams; ++t) {
            for (int r = R_TBI; r < R_TBH; ++r) if (seen[t][r] != 1) ok = 0;
            if (!merged && seen[t][R_C] != 1) ok = 0;
            if (merged && seen[t][R_C] != 0) ok = 0;
        }
        const bool ta_shared = p.rgroups && p.crit != kCritTA;
        for (int t = 0; t < (ta_shared ? p.rgroups : p.teams); ++t) for (int r = R_TA; r < R_TBI; ++r) if (seen[t][r] != 1) ok = 0;
        for (int t = 0; t < (p.rgroups ? p.rgroups : p.teams); ++t) for (int r = R_TBH; r < kTickRoles; ++r) if (seen[t][r] != 1) ok = 0;
        for (int r = kTickRoles; r < kFusedRoles; ++r) if (seen[0][r] != (fused ? 1 : 0)) ok = 0;
        for (int x = 0; x < 8; ++x) if (per_residue[x] > 32) ok = 0;
        if (p.grid > chain_capacity() || live > chain_capacity()) ok = 0;
    } else {
        live = p.grid;
        if (p.grid > chain_capacity()) ok = 0;
    }
    out[0] = p.teams; out[1] = p.nbr; out[2] = p.rgroups; out[3] = p.crit; out[4] = p.place; out[5] = p.grid; out[6] = live; out[7] = ok;
    return 0;
}

int launch_decode_b1(const DecodeChainArgs& d, hipStream_t s) {
    B1Args a{};
    a.fused = d.beat.z != nullptr;
    a.teams = decode_b1_teams(d.B);
    a.B = d.B; a.T = d.T; a.G = d.G; a.V = d.V; a.Z = DZ; a.stride = (!a.fused && mode() == 2 && a.teams == 1) ? 4 : 1;
    a.W_hh0 = d.W_hh0; a.b_hh0 = d.b_hh0; a.cgi = d.cgi; a.table = d.table;
    a.W_ih1 = d.W_ih1; a.b_ih1 = d.b_ih1; a.W_hh1 = d.W_hh1; a.b_hh1 = d.b_hh1;
    a.W_out = d.W_out; a.b_out = d.b_out; a.ht0 = d.ht0;
    a.weights = d.weights; a.samples = d.samples; a.ex = d.b1ex;
    a.bp = d.beat;
    a.stamps = d.b1stamps;
    a.status = d.status;
    char label[64];
    std::snprintf(label, sizeof label, "decode_b1%s T%d B%d H%d V%d", a.fused ? "_beats" : "", d.T, d.B, d.H, d.V);
    // algorithmic work: the tick GRU + head per tick and row; fused: + the beat path (z2b, two beat layers, three projections per beat)
    const double nbt = (double)d.T / d.G;
    const double beat_mac = a.fused ? 2.0 * DH * DZ + nbt * (3.0 * 3 * DH * DH + 2.0 * DH * DH + 1.0 * DH * DH + 3.0 * DH * DH) : 0.0;
    const double beat_w = a.fused ? 2.0 * DH * DZ + 9.0 * DH * DH + 3.0 * DH * DH + 3.0 * DH * DH : 0.0;
    ProfScope prof(PROF_GRU_FWD, 2.0 * d.B * (d.T * (9.0 * DH * DH + (double)d.V * DH) + beat_mac), s, label,
                   4.0 * (9.0 * DH * DH + (double)d.V * DH + (double)d.B * d.T * d.V + beat_w));
    const B1Plan pl = make_plan(d.B, d.V, a.fused != 0, a.stride);
    a.rgroups = pl.rgroups; a.crit = pl.crit; a.place = pl.place;
    const int nj = pl.nj, nbr = pl.nbr;
    const bool tbh_pairs = pl.tbh_pairs;
    if (pl.rgroups && pl.place != 1) return -1;                // (decode_b1_shape_ok has checked that the placed launch fits)
    const dim3 grid(pl.grid);
    if (a.fused && a.teams > 1 && (nbr > 2 || a.teams * nbr > ((nbr == 1 && !a.rgroups) ? kDecodeB1OneRowTeamsMax : kDecodeB1BeatRowsMax))) return -1;
#define DISPATCH_B1(NJ, NBR)                                                                                                    \
    do {                                                                                                                    \
        if (tbh_pairs && NJ <= 2) hipLaunchKernelGGL((decode_b1_kernel<(NJ <= 2 ? NJ : 1), true, 1, kDecodeB1BeatRowsMax, 2>), grid, dim3(NT), 0, s, a); \
        else if (a.fused && a.rgroups) hipLaunchKernelGGL((decode_b1_kernel<NJ, true, 1, kDecodeB1BeatRowsMax, kSharedRowsSmall>), grid, dim3(NT), 0, s, a); \
        else if (a.fused && a.teams > 1 && nbr == 1) hipLaunchKernelGGL((decode_b1_kernel<NJ, true, 1, kDecodeB1OneRowTeamsMax>), grid, dim3(NT), 0, s, a); \
        else if (a.fused && a.teams > 1) hipLaunchKernelGGL((decode_b1_kernel<NJ, true, 2, kDecodeB1BeatRowsMax>), grid, dim3(NT), 0, s, a); \
        else if (a.fused) hipLaunchKernelGGL((decode_b1_kernel<NJ, true, NBR, NBR>), grid, dim3(NT), 0, s, a);             \
        else if (a.rgroups) hipLaunchKernelGGL((decode_b1_kernel<NJ, false, 2, 2, kSharedRows>), grid, dim3(NT), 0, s, a);     \
        else hipLaunchKernelGGL((decode_b1_kernel<NJ, false, NBR, NBR>), grid, dim3(NT), 0, s, a);                        \
    } while (0)
#define DISPATCH_B1N(NJ) do { if (nbr == 1) DISPATCH_B1(NJ, 1); else if (nbr == 2) DISPATCH_B1(NJ, 2); else DISPATCH_B1(NJ, 4); } while (0)
    if (nj <= 1) DISPATCH_B1N(1); else if (nj == 2) DISPATCH_B1N(2); else if (nj == 3) DISPATCH_B1N(3); else DISPATCH_B1N(4);
#undef DISPATCH_B1N
#undef DISPATCH_B1
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
