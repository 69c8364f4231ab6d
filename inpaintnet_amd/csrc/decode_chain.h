// Fused free-running decode kernel (decode_chain.hip; protocol in chain.h).
#pragma once
#include "chain.h"

constexpr int kDecodeMaxGroups = 8;               // row tiles of 16*MS rows: B <= 256 at MS = 2
constexpr int kDecodeCounterStride = 64;          // one 256-byte block per group counter (see gru_chain.h)
constexpr int kDecodeStatusWord = kDecodeMaxGroups * kDecodeCounterStride;
constexpr int kDecodeSyncWords = kDecodeStatusWord + 4;

// b = 1 (decode_b1.hip), beat path folded into the launch: the operands of forward_beat_rnn and the per-beat projections
// (decoder.py:455-471, 485-497).  z == null: the beat path ran as launches of its own and `cgi` / `ht0` below hold its results.
struct DecodeB1Beat {
    const float* z;                               // [Z]
    const float* zb_w; const float* zb_b;         // z_to_beat_rnn_input: [2H, Z], [2H]
    const float* gvec0;                           // [3H] beat layer 0's constant input gates b_0 W_ih[:, 0] + b_ih (prologue launch)
    const float* W_hh0; const float* b_hh0;       // beat layer 0
    const float* W_ih1; const float* b_ih1; const float* W_hh1; const float* b_hh1;   // beat layer 1
    const float* bh_w; const float* bh_b;         // beat_emb_to_tick_rnn_hidden: [2H, H]
    const float* bi_w; const float* bi_b;         // beat_emb_to_tick_rnn_input: [H, H]
    const float* wih0_c; long wih0_ld;            // tick layer 0's W_ih[:, E:] (row stride E + H)
};

struct DecodeChainArgs {
    int B, H, T, G, V, members;                   // G = ticks per beat; T = beats * G
    const float* W_hh0; const float* b_hh0;       // tick layer 0: recurrent weights [3H,H], [3H]
    const float* cgi;                             // [beats*B, 3H] beat-constant part of the layer-0 input projection
    const float* table;                           // [(V+1), 3H] token part incl. b_ih(l0); row V = x_0
    const float* W_ih1; const float* b_ih1; const float* W_hh1; const float* b_hh1;   // tick layer 1
    const float* W_out; const float* b_out;       // [V,H], [V]
    const float* ht0;                             // [beats*B, 2H] initial tick hiddens per beat (cols [0,H) layer 0)
    const float* ht0pk;                           // [2 layers][beats][pk(B,H)] the same, fragment-major
    float* hx0; float* hx1;                       // exchange rings [2][pk(B,H)]
    float* amax;                                  // [2][V/16][ceil16(B)] x {idx, max} (8 bytes each)
    unsigned long long* b1ex;                     // B <= 16 (decode_b1.hip): decode_b1_words(B) / 2 zeroed 8-byte granules, or null
    DecodeB1Beat beat;                            // ... with the beat path folded in (beat.z != null)
    unsigned long long* b1stamps;                 // diagnostics (INET_DECODE_B1_STAMPS=1): [C, TBi_0][T][8] wall-clock stamps, or null
    float* weights; long long* samples;           // outputs [B,T,V], [B,1,T]
    unsigned* counters; chain::Status status;
    int prezeroed;                                // the sync words are already zero (gru_chain.h kSyncAreas)
    int Bs;                                       // rows of the FULL batch when this launch covers a chunk of its rows (0: B):
                                                  // the stride of the [T,.,H] / [beats,.,.] buffers (cgi, ht0, mask, saves, outputs)
    // training (free-running forward with backward saves): everything below may be null for inference
    const float* mask;                            // [T,B,H] dropout mask of the layer-0 output (layer 1 sees h0 * mask)
    float* hx0m;                                  // [beats][pk(B,H)] exchange buffer of the masked h0 (needed iff mask)
    float* sv0; float* sv1; long sv_stride;       // saves r,z,n,W_hn h + b_hn,h_prev of layer 0 / 1: [5][T,B,H], stride
    float* h0out;                                 // [T,B,H] layer-0 output as layer 1 saw it (masked if mask)
    float* h1seq;                                 // [T,B,H] layer-1 output (input of the output projection's gradients)
};

bool decode_chain_ok(int B, int H, int V, int T, int G);
int launch_decode_chain(DecodeChainArgs a, hipStream_t s);
// one to sixteen measures, inference: the register-resident persistent launch of decode_b1.hip (launch_decode_chain takes it when it applies;
// INET_DECODE_B1 / inet_set_option key 15: 0 = never; 1 / 2 = the tick path only, behind the beat path's own launches, on consecutive
// workgroup ids / on every 8th id (one XCD); 3 = default: up to six measures with the beat path folded into the same launch; three to sixteen: teams of the tick path's workgroups)
constexpr int kDecodeB1WordsPerRow = 2 * 35392;   // 32-bit words of ONE row's granule area (decode_b1.hip's map: tick exchange + one slot per beat step)
constexpr int kDecodeB1StampWords = 2 * 2 * 32 * 8;   // 32-bit words of the stamp area (2 roles x <= 32 ticks x 8 stamps of 8 bytes)
constexpr int kDecodeB1MaxRows = 16;              // rows (measures) per call the register-resident launch takes: 1 / 2 / 4 per team of
                                                  // workgroups, up to five teams (tick path only beyond one team)
// rows per team and teams for a call of B rows.  One team up to B = 2; beyond, teams of TWO rows while they fit
// the chip (a two-row tick is 5.5 us, a four-row tick 8.4: 5 teams x 49 workgroups = 245 of 256 CUs -> B <= 10), else of four.
int decode_b1_team_rows(int B);                   // (decode_b1.hip: the rule above; mode 4 shares the recurrent groups beyond ten measures)
inline int decode_b1_teams(int B) { const int r = decode_b1_team_rows(B); return (B + r - 1) / r; }
constexpr int kDecodeB1BeatRowsMax = 6;           // rows the beat path's workgroups serve when they share the launch with several teams
constexpr int kDecodeB1OneRowTeamsMax = 3;        // ... with ONE-row teams (two or three measures under mode 4): the beat path serves three rows
// granule areas of a call: one per row of every team; three to six measures: at least kDecodeB1BeatRowsMax (the folded beat path
// computes that many rows, whatever B is -- rows beyond B repeat row B - 1)
inline int decode_b1_rows(int B) {
    const int r = decode_b1_teams(B) * decode_b1_team_rows(B);
    if (decode_b1_teams(B) > 1 && decode_b1_team_rows(B) == 1)          // (one-row teams: the folded beat path computes three rows, or six)
        return r <= kDecodeB1OneRowTeamsMax ? kDecodeB1OneRowTeamsMax : kDecodeB1BeatRowsMax;
    return (decode_b1_teams(B) > 1 && B <= kDecodeB1BeatRowsMax && r < kDecodeB1BeatRowsMax) ? kDecodeB1BeatRowsMax : r;
}
inline long decode_b1_words(int B) { return (long)decode_b1_rows(B) * kDecodeB1WordsPerRow; }
bool decode_b1_shape_ok(int B, int H, int V, int T, int G);
bool decode_b1_fused(int Z, int B);                    // ... and the beat path goes into the same launch
bool decode_b1_ok(const DecodeChainArgs& a);
int launch_decode_b1(const DecodeChainArgs& a, hipStream_t s);
void decode_b1_set_mode(int m);
// the launch plan of a call of B measures (V notes, latent size Z) and a host-side self-check of it: decode_b1.hip, no GPU needed
int decode_b1_plan_check(int B, int V, int Z, int* out8);
