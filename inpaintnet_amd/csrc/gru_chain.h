// Launch descriptors of the GRU chain kernels (gru_chain.hip; protocol in chain.h).
#pragma once
#include "chain.h"

// Sync area of one launch.  Every group counter sits alone in a 256-byte block: the counters are polled (L2-bypassing
// loads) by all members of their group and bumped with device-scope atomics, and with 8 of them in one line all 256
// pollers of the launch queued on the same memory channel -- 0.6 us of every step (profiles/r02_i_chain_counters.txt).
constexpr int kChainMaxGroups = 64;               // group counters per launch (second generation: one per problem and row block)
constexpr int kChainCounterStride = 64;           // words between two group counters
constexpr int kChainStatusWord = kChainMaxGroups * kChainCounterStride;   // the launch's status word (abort flag)
constexpr int kChainZeroWord = kChainStatusWord + 64; // words [+0, +1] stay zero: the target of absent operand pointers
constexpr int kChainSyncWords = kChainZeroWord + 4;
// A workspace holds kSyncAreas such areas: a library call zeroes all of them with ONE memset and gives each of its chain
// launches its own (`prezeroed`), instead of one 5 us fill kernel in front of every launch (11 per training step).
constexpr int kSyncAreas = 8;

// Piece outputs of a second-generation chain kernel (gemm_bf3.h layouts) for the bf16-matrix-core products that consume the
// layer's results: the wave that produces a value also splits it into its three bf16 pieces, so no separate pass reads the
// f32 array again.  Row index of (t, b) in every layout: m = t * B_full + r0 + b  (B_full % 32 == 0, r0 % 32 == 0).
//   forward:  rows = masked output [T*B, .] (this direction's H columns at k blocks rows_kb0 ..), colsA = masked output^T
//             (row blocks colsA_rb0 ..), colsB = previous hidden state^T (row blocks colsB_rb0 ..)
//   backward: rows = dgi [T*B, .] (gates r, z, n at k blocks rows_kb0 + g * H / 32 ..), colsA = (r, z[, n])^T (row blocks
//             colsA_rb0 + g * H / 16 ..; colsA_n: with the n block), colsB = (n * r)^T
// Null pointers: nothing is written.  First-generation kernels ignore the descriptor (callers check which kernel ran).
struct ChainEmit {
    unsigned char* rows; long rows_piece; int rows_kb, rows_kb0;
    unsigned char* colsA; long colsA_piece; int colsA_rb0, colsA_n;
    unsigned char* colsB; long colsB_piece; int colsB_rb0;
    int B_full, r0;
    int skip_dgi, skip_dgh;                       // backward: the f32 dgi / dgh arrays have no reader left, do not write them
};

struct GruChainFwdProb {
    const float* W_hh; const float* b_hh;         // [3H,H] row-major, [3H]
    const float* h0; long ld_h0;                  // [B,H] initial hidden (row-major); slot 1 of hx holds it fragment-major
    const float* gi_dense; long ld_gi, ts_gi;     // gi(t,b) = gi_dense[t*ts + b*ld + g*H + j], or null
    const float* gi_table; long ld_table;         // gi(t,b) = gi_table[tok*ld + g*H + j], tok = idx[b*idx_bs + t*idx_ts]
    const long long* idx; long idx_bs, idx_ts;
    const float* gi_vec;                          // [3H] broadcast, or null
    float* out; long ld_out, ts_out;              // h(t,b)
    float* outm; long ld_outm, ts_outm;           // h(t,b) * mask(t,b), or null
    const float* mask; long ld_mask, ts_mask;
    float* hlast; long ld_hlast;                  // copy of the final hidden, or null
    float* sv; long sv_astride;                   // r,z,n,ghn,hprev saves [T][B][H] each, or null
    long sv_ts;                                   // time stride of the saves in floats (0: B*H; a row chunk of a larger batch
                                                  // passes the full batch's)
    float* hx;                                    // exchange: [2][ceil16(B)][H] fragment-major
    int reverse;
    int hx_slot_bytes;                            // distance between the two slots of hx (0: adjacent)
    ChainEmit em;
};
struct GruChainFwd {
    int H, B, T, nprob, tiles_per_prob, members, prio;
    int h0_packed;                                // slot 1 of every hx already holds h0 (packed by the host: a removed A/B switch of round 2)
    int fault;                                    // test hook (inet_set_option key 6): workgroup 0 leaves at once, so its
                                                  // group runs into the bounded spin and the failure path can be tested
    int shared_chip;                              // this launch runs beside another chain launch (two workgroups per CU):
                                                  // the 256-register build of the kernel
    GruChainFwdProb p[4];
    unsigned* counters;                           // kChainSyncWords words owned by this launch (zeroed by the launcher
    int prezeroed;                                // unless the caller says they already are)
    chain::Status status;
};

struct GruChainBwdProb {
    const float* W_hh;                            // [3H,H] row-major (read transposed, once)
    const float* dout; long ld_dout, ts_dout;     // dLoss/dh(t,b), or null
    const float* dhn; long ld_dhn;                // dLoss/d final hidden, or null
    const float* sv; long sv_astride;
    long sv_ts;                                   // time stride of the saves in floats (0: B*H)
    float* dgi; long ld_dgi, ts_dgi;              // [.,3H] input-side gate gradients (strided)
    float* dgh;                                   // [T][B][3H] dense recurrent-side gate gradients
    long dgh_ts;                                  // time stride of dgh in floats (0: B*3H)
    int gx_slot_bytes;                            // distance between the two slots of gx (0: adjacent)
    float* db_ih; float* db_hh;                   // [3H], accumulated; or null
    float* dh0; long ld_dh0; int dh0_accumulate;  // dLoss/d initial hidden, or null
    float* gx;                                    // exchange: [2][ceil16(B)][3H] fragment-major
    int reverse;
    float* dgi_sum;                               // optional [B,3H]: sum over the T steps of the input-side gate gradients
    ChainEmit em;
};                                                // (the decoder's beat-constant input half: one value per beat)
struct GruChainBwd {
    int H, B, T, nprob, tiles_per_prob, members, prio;
    GruChainBwdProb p[4];
    unsigned* counters;
    int prezeroed;
    chain::Status status;
};

bool gru_chain_ok(int H, int B, int T, int nprob);
bool gru_chain_bwd_ok(int H, int B, int T, int nprob);      // as above, with two row tiles per workgroup when needed
int launch_gru_chain_fwd(GruChainFwd a, hipStream_t s);     // (dispatches to the second generation where it applies)
int launch_gru_chain_bwd(GruChainBwd a, hipStream_t s);
// which generation the two launchers above pick for a shape (second: the ChainEmit outputs are written)
bool gru_chain_fwd_is_v2(int H, int B, int T, int nprob, int h0_packed);
bool gru_chain_bwd_is_v2(int H, int B, int T, int nprob);
bool gru_chain_bwd_emits_rows(int H, int B, int T, int nprob);   // first-generation BPTT launch that writes ChainEmit.rows
// Second generation (gru_chain2.hip): one row block per wave, W in LDS, contraction on the bf16 matrix cores at fp32
// accuracy (three-way exact split, 9 or 6 piece products).  Its exchange holds three bf16 pieces per state: rings must be
// sized 3 * pk_floats(B, K) floats (seq.h chain_ring_floats) instead of 2 * pk_floats.
int chain2_mode();                                          // 0 = off (INET_CHAIN2=0), 9 piece products (default)
void chain2_set_mode(int np);
bool gru_chain2_ok(int H, int B, int T, int nprob);
bool gru_chain2_emits(int H, int B, int T, int nprob);      // ... and its build writes the ChainEmit outputs (four waves)
int launch_gru_chain2_fwd(GruChainFwd a, hipStream_t s);
