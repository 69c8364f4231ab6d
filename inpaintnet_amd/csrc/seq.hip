#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include "seq.h"
#include "gemm_bf3.h"
#include "gru_step_bf3.h"

// Rows per launch when a batch is too large for one resident chain launch: the largest of 1024 / 512 / 256 / 128 / 64 that
// divides B and fits the chip (B itself when it fits; 0: no chain launch applies).
// Chunking pays only while the batch is a few launches' worth: a chain launch is latency-bound (about 57 % of the MFMA
// rate at 256 rows), a per-step launch over thousands of rows is not -- at the reference's default 4096 measures per step
// the per-step kernels are 1.5x faster than sixteen chunk launches per layer (profiles/r03_e_*).  INET_CHAIN_CHUNK_MAX
// = largest batch that is still chunked.
static int chunk_max_rows() {
    static const int v = [] { const char* e = std::getenv("INET_CHAIN_CHUNK_MAX"); return e ? std::atoi(e) : 1024; }();
    return v;
}
int chain_chunk_rows(int H, int B, int T, int nd, int save) {
    if (gru_chain_ok(H, B, T, nd)) return B;
    // (forward-only passes -- LatentRNN's frozen encoder, 2048 rows -- chunk at any size: two chunks run side by side there;
    //  no cap)
    constexpr int fwd_cap = 1 << 30;
    if (B > (save ? chunk_max_rows() : fwd_cap)) return 0;
    for (int ch = 1024; ch >= 64; ch >>= 1)
        if (ch < B && B % ch == 0 && gru_chain_ok(H, ch, T, nd)) return ch;
    return 0;
}
int chain_chunk_rows_bwd(int H, int B, int T, int nd) {
    if (gru_chain_bwd_ok(H, B, T, nd)) return B;
    if (B > chunk_max_rows()) return 0;
    for (int ch = 1024; ch >= 64; ch >>= 1)
        if (ch < B && B % ch == 0 && gru_chain_bwd_ok(H, ch, T, nd)) return ch;
    return 0;
}

// which piece outputs the chains write themselves (bit 0: the forward chains' rows, 1: their transposed pieces, 2: the first-generation
// BPTT kernel's dgi rows); what they do not write is split from the f32 arrays by bf3_split launches (the transposed ones on the
// side stream).  inet_set_option key 9: the tests run the same step both ways.
static int g_emit_mask = -1;
static int emit_mask() {
    if (g_emit_mask < 0) g_emit_mask = 7;
    return g_emit_mask;
}
void bf3_set_emit_mask(int m) { g_emit_mask = m & 7; }
// Layer 1's weight gradients run on the bf16 pipe, on a side stream; layer 0's stay on the LDS-free f32-input kernel: its product is
// the last kernel of the backward pass, and what the bf16 form saves (172 -> 118 us) is spent on making its operands (one-box A/Bs
// of round 3: 3.75 / 3.72 / 3.85 / 3.82 ms per step for both / layer 1 only / layer 0 only / neither; HISTORY.md).

bool gru_layer_fwd_emits(int H, int B, int T, int nd, bool save) {
    if (!pk_ok(H) || B % 32) return false;
    if (!gru_chain_ok(H, B, T, nd) && gru_step_bf3_ok(H, B, T, nd)) return false;   // (the step kernels write row pieces only)
    if (gru_chain_ok(H, B, T, nd)) return gru_chain_fwd_is_v2(H, B, T, nd, 0) && gru_chain2_emits(H, B, T, nd);
    const int CH = chain_chunk_rows(H, B, T, nd, save);
    return CH > 0 && CH < B && CH % 32 == 0 && gru_chain_fwd_is_v2(H, CH, T, nd, 0) && gru_chain2_emits(H, CH, T, nd);
}
int gru_layer_fwd(int H, int B, int T, int nd, const DirFwd* d, hipStream_t s) {
    for (int i = 0; i < nd; ++i) d[i].emitted = 0;
    const long BH = (long)B * H;
    const long pkh = (long)pk_floats(B, H);
    bool pk = pk_ok(H);
    for (int i = 0; i < nd; ++i) if (!d[i].Wpk_hh || !d[i].hpk) pk = false;
    // A/B switch: (round 2 had a switch that packed the initial state into slot 1 with a launch in front of the chain; removed in round 5)
    constexpr bool h0pack = false;
    bool all_h0 = true;
    for (int i = 0; i < nd; ++i) all_h0 = all_h0 && d[i].h0;
    auto pack_h0 = [&]() -> int {
        bool same_ld = true;
        const float* ins[4]; float* outs[4];
        for (int i = 0; i < nd; ++i) { ins[i] = d[i].h0; outs[i] = d[i].hpk + pkh; same_ld = same_ld && d[i].h0_ld == d[0].h0_ld; }
        if (same_ld) return pw_pack_frag_multi(ins, outs, nd, d[0].h0_ld, B, H, 0, s);
        for (int i = 0; i < nd; ++i) INET_TRY(pw_pack_frag(d[i].h0, d[i].h0_ld, B, H, d[i].hpk + pkh, 0, 1, 0, 0, s));
        return 0;
    };
    if (pk && d[0].sync && gru_chain_ok(H, B, T, nd)) {
        // one persistent launch for all T steps (gru_chain.hip); the exchange buffer is the hpk ring, slot 1 = h0 (published
        // by the kernel itself; a null h0 = zeros)
        GruChainFwd a{};
        a.H = H; a.B = B; a.T = T; a.nprob = nd;
        if (h0pack && all_h0 && !gru_chain2_ok(H, B, T, nd)) { INET_TRY(pack_h0()); a.h0_packed = 1; }
        for (int i = 0; i < nd; ++i) {
            const DirFwd& D = d[i];
            GruChainFwdProb& P = a.p[i];
            P.W_hh = D.W_hh; P.b_hh = D.b_hh;
            P.h0 = D.h0; P.ld_h0 = D.h0_ld;
            P.gi_dense = D.gi; P.ld_gi = D.gi_ld; P.ts_gi = D.gi_ts;
            P.gi_table = D.table; P.ld_table = D.table_ld; P.idx = D.idx; P.idx_bs = D.idx_bs; P.idx_ts = D.idx_ts;
            P.gi_vec = D.gvec;
            P.out = D.out; P.ld_out = D.out_ld; P.ts_out = D.out_ts;
            P.outm = D.outm; P.ld_outm = D.outm_ld; P.ts_outm = D.outm_ts;
            P.mask = D.mask; P.ld_mask = D.mask_ld; P.ts_mask = D.mask_ts;
            P.hlast = D.hlast; P.ld_hlast = D.hlast_ld;
            P.sv = D.sv; P.sv_astride = D.sv_astride;
            P.hx = D.hpk; P.reverse = D.reverse;
            if (B % 32 == 0 && gru_chain_fwd_is_v2(H, B, T, nd, a.h0_packed) && gru_chain2_emits(H, B, T, nd)) { P.em = D.em; P.em.B_full = B; P.em.r0 = 0; D.emitted = 3; }
        }
        a.counters = d[0].sync; a.prezeroed = d[0].sync_prezeroed;
        return launch_gru_chain_fwd(a, s);
    }
    // One time step of the layer fills the chip by itself (LatentRNN's frozen encoder: 2048 measures; the reference's default
    // MeasureVAE batch: 4096): a bf16-pipe product per step with the GRU cell as its epilogue (gru_step_bf3.hip) instead of
    // chunked chain launches at 0.37 of that pipe.
    bool stepbf3 = pk && nd <= 2 && gru_step_bf3_ok(H, B, T, nd);
    for (int i = 0; i < nd; ++i) stepbf3 = stepbf3 && d[i].wp3;
    if (stepbf3) {
        GruStepsBf3 L{};
        L.H = H; L.B = B; L.T = T; L.nprob = nd;
        for (int i = 0; i < nd; ++i) {
            const DirFwd& D = d[i];
            GruChainFwdProb& P = L.p[i];
            P.b_hh = D.b_hh;
            P.h0 = D.h0; P.ld_h0 = D.h0_ld;
            P.gi_dense = D.gi; P.ld_gi = D.gi_ld; P.ts_gi = D.gi_ts;
            P.gi_table = D.table; P.ld_table = D.table_ld; P.idx = D.idx; P.idx_bs = D.idx_bs; P.idx_ts = D.idx_ts;
            P.gi_vec = D.gvec;
            P.out = D.out; P.ld_out = D.out_ld; P.ts_out = D.out_ts;
            P.outm = D.outm; P.ld_outm = D.outm_ld; P.ts_outm = D.outm_ts;
            P.mask = D.mask; P.ld_mask = D.mask_ld; P.ts_mask = D.mask_ts;
            P.hlast = D.hlast; P.ld_hlast = D.hlast_ld;
            P.sv = D.sv; P.sv_astride = D.sv_astride;
            P.hx = D.hpk; P.reverse = D.reverse;
            if (D.em.rows) { P.em.rows = D.em.rows; P.em.rows_piece = D.em.rows_piece; P.em.rows_kb = D.em.rows_kb; P.em.rows_kb0 = D.em.rows_kb0;
                             P.em.B_full = B; P.em.r0 = 0; D.emitted = 1;
                             // forward-only: the masked output's only reader is the layer-1 product, which takes these pieces
                             if (!D.sv) P.outm = nullptr; }
            L.Wp[i] = D.wp3;
        }
        {
            const float* ws[2] = {d[0].W_hh, nd > 1 ? d[1].W_hh : nullptr};
            unsigned char* wp[2] = {d[0].wp3, nd > 1 ? d[1].wp3 : nullptr};
            INET_TRY(gru_step_bf3_split_w(H, ws, wp, nd, s));
        }
        return launch_gru_steps_bf3(L, s);
    }
    // More rows than one resident launch can take (the frozen encoder of LatentRNN runs 2048 measures at once, the
    // reference's default VAE batch is 4096 measures): the rows are independent, so the chain kernel runs over chunks of
    // rows, one launch after the other, each on a chunk-sized exchange ring inside the hpk buffer; backward saves keep the
    // full batch's time stride.
    bool any_sv = false;
    for (int i = 0; i < nd; ++i) any_sv = any_sv || d[i].sv;
    const int CH = chain_chunk_rows(H, B, T, nd, any_sv);
    const bool chunked = pk && d[0].sync && CH > 0 && CH < B;
    if (chunked) {
        // Two chunks at a time, on two streams: the kernel's 256-register build lets two launches share every CU, and one
        // chunk's hand-off latency (a third of each step) is filled by the other chunk's MFMAs.  Chunk c works on its rows
        // of the two slots of the full-batch ring (row blocks are the outermost index of the fragment-major layout); even /
        // odd chunks count on different sync areas (the caller's and the one behind it).
        constexpr bool twin = true;
        const long pkc = (long)pk_floats(CH, H);
        // (the second-generation kernel keeps its W slice in 144 KB of LDS: one workgroup per CU, nothing to gain from two
        // streams; every chunk gets its own contiguous ring of three-piece slots)
        const bool v2 = gru_chain2_ok(H, CH, T, nd);
        hipStream_t s2 = twin && !v2 ? twin_fork(s) : s;
        for (int c = 0; c < B / CH; ++c) {
            const long r0 = (long)c * CH;
            GruChainFwd a{};
            a.H = H; a.B = CH; a.T = T; a.nprob = nd;
            a.shared_chip = s2 != s;
            for (int i = 0; i < nd; ++i) {
                const DirFwd& D = d[i];
                GruChainFwdProb& P = a.p[i];
                P.W_hh = D.W_hh; P.b_hh = D.b_hh;
                P.h0 = D.h0 ? D.h0 + r0 * D.h0_ld : nullptr; P.ld_h0 = D.h0_ld;
                P.gi_dense = D.gi ? D.gi + r0 * D.gi_ld : nullptr; P.ld_gi = D.gi_ld; P.ts_gi = D.gi_ts;
                P.gi_table = D.table; P.ld_table = D.table_ld;
                P.idx = D.idx ? D.idx + r0 * D.idx_bs : nullptr; P.idx_bs = D.idx_bs; P.idx_ts = D.idx_ts;
                P.gi_vec = D.gvec;
                P.out = D.out + r0 * D.out_ld; P.ld_out = D.out_ld; P.ts_out = D.out_ts;
                P.outm = D.outm ? D.outm + r0 * D.outm_ld : nullptr; P.ld_outm = D.outm_ld; P.ts_outm = D.outm_ts;
                P.mask = D.mask ? D.mask + r0 * D.mask_ld : nullptr; P.ld_mask = D.mask_ld; P.ts_mask = D.mask_ts;
                P.hlast = D.hlast ? D.hlast + r0 * D.hlast_ld : nullptr; P.ld_hlast = D.hlast_ld;
                if (D.sv) { P.sv = D.sv + r0 * H; P.sv_astride = D.sv_astride; P.sv_ts = BH; }
                if (v2) P.hx = D.hpk + (long)c * 3 * pkc;
                else { P.hx = D.hpk + (long)c * pkc; P.hx_slot_bytes = (int)(pkh * sizeof(float)); }
                P.reverse = D.reverse;
                if (v2 && B % 32 == 0 && CH % 32 == 0 && gru_chain_fwd_is_v2(H, CH, T, nd, 0) && gru_chain2_emits(H, CH, T, nd)) {
                    P.em = D.em; P.em.B_full = B; P.em.r0 = (int)r0; D.emitted = 3;
                }
            }
            a.counters = d[0].sync + (c & 1) * kChainSyncWords;
            INET_TRY(launch_gru_chain_fwd(a, (c & 1) ? s2 : s));
        }
        return s2 != s ? twin_join(s) : 0;
    }
    // one launch per step (gru.hip): the fragment-major ring's slot 1 is packed from h0 here (step 0 reads it)
    for (int i = 0; i < nd; ++i) if (!d[i].h0) return -1;     // (a null h0 is the chain kernels' shorthand for zeros)
    if (pk) INET_TRY(pack_h0());
    for (int step = 0; step < T; ++step) {
        GruFwdBatch bt{};
        bt.H = H; bt.nprob = nd;
        for (int i = 0; i < nd; ++i) {
            const DirFwd& D = d[i];
            GruFwdProb& P = bt.p[i];
            const int t = D.reverse ? T - 1 - step : step;
            const int tp = D.reverse ? t + 1 : t - 1;
            P.B = B;
            if (step == 0) { P.h_prev = D.h0; P.ld_hprev = D.h0_ld; }
            else { P.h_prev = D.out + (long)tp * D.out_ts; P.ld_hprev = D.out_ld; }
            P.W_hh = D.W_hh; P.b_hh = D.b_hh;
            if (pk) {
                P.Wpk_hh = D.Wpk_hh;
                P.hpk_prev = D.hpk + (long)((step + 1) & 1) * pkh;
                if (step != T - 1) P.hpk_new = D.hpk + (long)(step & 1) * pkh;
            }
            if (D.gi) { P.gi_dense = D.gi + (long)t * D.gi_ts; P.ld_gi = D.gi_ld; }
            if (D.table) {
                P.gi_table = D.table; P.ld_table = D.table_ld;
                P.idx = D.idx + (long)t * D.idx_ts; P.idx_stride = D.idx_bs;
            }
            P.gi_vec = D.gvec;
            P.h_new = D.out + (long)t * D.out_ts; P.ld_hnew = D.out_ld;
            if (D.outm) {
                P.h_masked = D.outm + (long)t * D.outm_ts; P.ld_hm = D.outm_ld;
                P.mask = D.mask ? D.mask + (long)t * D.mask_ts : nullptr; P.ld_mask = D.mask_ld;
            }
            if (D.hlast && step == T - 1) { P.h_copy = D.hlast; P.ld_hc = D.hlast_ld; }
            if (D.sv) {
                float* b0 = D.sv + (long)t * BH;
                P.sv_r = b0; P.sv_z = b0 + D.sv_astride; P.sv_n = b0 + 2 * D.sv_astride;
                P.sv_ghn = b0 + 3 * D.sv_astride; P.sv_hprev = b0 + 4 * D.sv_astride;
            }
        }
        INET_TRY(launch_gru_fwd(bt, s));
    }
    return 0;
}

int gru_layer_bwd(int H, int B, int T, int nd, const DirBwd* d, hipStream_t s) {
    return gru_layer_bwd_range(H, B, T, nd, d, T - 1, 0, s);
}

// Steps step_hi .. step_lo (descending) of the BPTT chain; the gradient wrt the initial hidden follows step 0.
int gru_layer_bwd_range(int H, int B, int T, int nd, const DirBwd* d, int step_hi, int step_lo, hipStream_t s) {
    for (int i = 0; i < nd; ++i) d[i].emitted = 0;
    const long BH = (long)B * H, B3H = 3 * BH;
    const long pkg = (long)pk_floats(B, 3 * H);
    bool pk = pk_ok(H);
    for (int i = 0; i < nd; ++i) if (!d[i].Wpk_hhT || !d[i].dghpk) pk = false;
    {
        bool any0 = false, all0 = true, wok = true;
        for (int i = 0; i < nd; ++i) { if (d[i].dh0) any0 = true; else all0 = false; if (!d[i].W_hh) wok = false; }
        const int CHB = chain_chunk_rows_bwd(H, B, T, nd);
        if (pk && wok && d[0].sync && step_hi == T - 1 && step_lo == 0 && (!any0 || all0) && CHB > 0) {
            // one persistent launch -- or, for a batch beyond one resident launch, one launch per chunk of CHB rows (the rows
            // are independent; saves / dgh keep the full batch's time stride; bias gradients accumulate with atomics)
            const long pkc = (long)pk_floats(CHB, 3 * H);
            const bool v2b = gru_chain2_ok(H, CHB, T, nd);
            for (int c = 0; c < B / CHB; ++c) {
                const long r0 = (long)c * CHB;
                GruChainBwd a{};
                a.H = H; a.B = CHB; a.T = T; a.nprob = nd;
                for (int i = 0; i < nd; ++i) {
                    const DirBwd& D = d[i];
                    GruChainBwdProb& P = a.p[i];
                    P.W_hh = D.W_hh;
                    P.dout = D.dout ? D.dout + r0 * D.dout_ld : nullptr; P.ld_dout = D.dout_ld; P.ts_dout = D.dout_ts;
                    P.dhn = D.dhn ? D.dhn + r0 * D.dhn_ld : nullptr; P.ld_dhn = D.dhn_ld;
                    P.sv = D.sv + r0 * H; P.sv_astride = D.sv_astride; P.sv_ts = BH;
                    P.dgi = D.dgi + r0 * D.dgi_ld; P.ld_dgi = D.dgi_ld; P.ts_dgi = D.dgi_ts;
                    P.dgh = D.dgh + r0 * 3 * H; P.dgh_ts = B3H;
                    P.db_ih = D.db_ih; P.db_hh = D.db_hh;
                    P.dh0 = D.dh0 ? D.dh0 + r0 * D.dh0_ld : nullptr; P.ld_dh0 = D.dh0_ld; P.dh0_accumulate = D.dh0_acc;
                    if (v2b) P.gx = D.dghpk + (long)c * 3 * pkc;
                    else { P.gx = D.dghpk + (long)c * pkc; P.gx_slot_bytes = (int)(pkg * sizeof(float)); }
                    P.reverse = D.reverse;
                    P.dgi_sum = D.dgi_sum ? D.dgi_sum + r0 * 3 * H : nullptr;
                    if (B % 32 == 0 && CHB % 32 == 0 && gru_chain_bwd_is_v2(H, CHB, T, nd) && gru_chain2_emits(H, CHB, T, nd)) {
                        P.em = D.em; P.em.B_full = B; P.em.r0 = (int)r0; D.emitted = 3;
                    } else if (B % 32 == 0 && CHB % 32 == 0 && D.em.rows && gru_chain_bwd_emits_rows(H, CHB, T, nd)) {
                        P.em = D.em; P.em.B_full = B; P.em.r0 = (int)r0; D.emitted = 1;   // (the first generation: row pieces only)
                    }
                }
                a.counters = d[0].sync; a.prezeroed = (c == 0 && CHB == B) ? d[0].sync_prezeroed : 0;
                INET_TRY(launch_gru_chain_bwd(a, s));
            }
            for (int i = 0; i < nd; ++i)
                if (d[i].dgi_sum && d[i].dgi_sum_done) *d[i].dgi_sum_done = 1;
            return 0;
        }
    }
    {
        // More rows than the chain launches take (chunks stop at INET_CHAIN_CHUNK_MAX rows): one bf16-pipe product per time step with
        // the gate derivatives as its epilogue (gru_step_bf3.hip) instead of the f32-input per-step kernels
        bool any0 = false, all0 = true, stepb = pk && nd <= 2 && step_hi == T - 1 && step_lo == 0 && gru_step_bf3_bwd_ok(H, B, T, nd);
        for (int i = 0; i < nd; ++i) {
            if (d[i].dh0) any0 = true; else all0 = false;
            stepb = stepb && d[i].wp3T && d[i].W_hh && d[i].dhz && !d[i].dgi_sum;
        }
        if (stepb && (!any0 || all0)) {
            GruStepsBf3Bwd L{};
            L.H = H; L.B = B; L.T = T; L.nprob = nd;
            for (int i = 0; i < nd; ++i) {
                const DirBwd& D = d[i];
                GruChainBwdProb& P = L.p[i];
                P.dout = D.dout; P.ld_dout = D.dout_ld; P.ts_dout = D.dout_ts;
                P.dhn = D.dhn; P.ld_dhn = D.dhn_ld;
                P.sv = D.sv; P.sv_astride = D.sv_astride;
                P.dgi = D.dgi; P.ld_dgi = D.dgi_ld; P.ts_dgi = D.dgi_ts;
                P.dgh = D.dgh;
                P.db_ih = D.db_ih; P.db_hh = D.db_hh;
                P.dh0 = D.dh0; P.ld_dh0 = D.dh0_ld; P.dh0_accumulate = D.dh0_acc;
                P.gx = D.dghpk; P.reverse = D.reverse;
                if (D.em.rows && B % 32 == 0) { P.em.rows = D.em.rows; P.em.rows_piece = D.em.rows_piece; P.em.rows_kb = D.em.rows_kb;
                                                P.em.rows_kb0 = D.em.rows_kb0; P.em.B_full = B; P.em.r0 = 0; D.emitted = 1; }
                INET_TRY(gru_step_bf3_split_wT(H, D.W_hh, D.wp3T, s));
                L.WpT[i] = D.wp3T; L.dhz[i] = D.dhz;
            }
            return launch_gru_steps_bf3_bwd(L, s);
        }
    }
    for (int step = step_hi; step >= step_lo; --step) {
        GruBwdBatch bt{};
        bt.H = H; bt.nprob = nd;
        for (int i = 0; i < nd; ++i) {
            const DirBwd& D = d[i];
            GruBwdProb& P = bt.p[i];
            const int t = D.reverse ? T - 1 - step : step;
            const int tn = D.reverse ? t - 1 : t + 1;          // the step processed just before (later in time order)
            P.B = B;
            if (step != T - 1) {
                P.dgh_next = D.dgh + (long)tn * B3H; P.ld_dgh = 3L * H;
                P.W_hhT = D.W_hhT;
                if (pk) { P.Wpk_hhT = D.Wpk_hhT; P.dghpk_next = D.dghpk + (long)((step + 1) & 1) * pkg; }
                P.dhz_next = D.dhz + (long)((step + 1) & 1) * BH;
            } else if (D.dhn) {
                P.dout2 = D.dhn; P.ld_dout2 = D.dhn_ld;
            }
            if (D.dout) { P.dout = D.dout + (long)t * D.dout_ts; P.ld_dout = D.dout_ld; }
            const float* b0 = D.sv + (long)t * BH;
            P.sv_r = b0; P.sv_z = b0 + D.sv_astride; P.sv_n = b0 + 2 * D.sv_astride;
            P.sv_ghn = b0 + 3 * D.sv_astride; P.sv_hprev = b0 + 4 * D.sv_astride;
            P.dgi = D.dgi + (long)t * D.dgi_ts; P.ld_dgi = D.dgi_ld;
            P.dgh = D.dgh + (long)t * B3H; P.ld_dghout = 3L * H;
            P.dhz = D.dhz + (long)(step & 1) * BH;
            if (pk) P.dghpk = D.dghpk + (long)(step & 1) * pkg;
            P.db_ih = D.db_ih; P.db_hh = D.db_hh;
        }
        INET_TRY(launch_gru_bwd(bt, s));
    }
    if (step_lo > 0) return 0;
    bool any = false, all = true;
    for (int i = 0; i < nd; ++i) { if (d[i].dh0) any = true; else all = false; }
    if (any) {
        if (!all) return -1;
        GruBwdBatch bt{};
        bt.H = H; bt.nprob = nd;
        for (int i = 0; i < nd; ++i) {
            const DirBwd& D = d[i];
            GruBwdProb& P = bt.p[i];
            const int t0 = D.reverse ? T - 1 : 0;
            P.B = B;
            P.dgh_next = D.dgh + (long)t0 * B3H; P.ld_dgh = 3L * H;
            P.W_hhT = D.W_hhT;
            if (pk) { P.Wpk_hhT = D.Wpk_hhT; P.dghpk_next = D.dghpk; }
            P.dhz_next = D.dhz;                                 // step 0 wrote slot 0
            P.dh_out = D.dh0; P.ld_dhout = D.dh0_ld; P.dh_out_accumulate = D.dh0_acc;
        }
        INET_TRY(launch_gru_bwd(bt, s));
    }
    return 0;
}

int gru_dir_wgrad(int H, int B, int T, const float* dgh, const float* sv_hprev, float* dW_hh, hipStream_t s) {
    return linear_wgrad(dgh, 3L * H, sv_hprev, H, dW_hh, H, T * B, 3 * H, H, s);
}
int gru_dir_wgrad_range(int H, int B, int t_lo, int nt, const float* dgh, const float* sv_hprev, float* dW_hh, hipStream_t s) {
    return linear_wgrad(dgh + (long)t_lo * B * 3 * H, 3L * H, sv_hprev + (long)t_lo * B * H, H, dW_hh, H, nt * B, 3 * H, H, s);
}

size_t bigru2_carve(Carver& c, int B, int T, int H, int save, BiGru2Ws& w) {
    const size_t BH = (size_t)B * H, TBH = (size_t)T * BH;
    w.zeros = c.take<float>(BH);
    w.x1raw = c.take<float>(2 * TBH);
    w.x1m = c.take<float>(2 * TBH);
    w.gi1 = c.take<float>(6 * TBH);
    w.h1 = c.take<float>(2 * TBH);
    for (int i = 0; i < 4; ++i) w.sv[i] = save ? c.take<float>(5 * TBH) : nullptr;
    if (save) {
        for (int i = 0; i < 4; ++i) w.whhT[i] = c.take<float>((size_t)3 * H * H);
        w.dgi1 = c.take<float>(6 * TBH);
        for (int i = 0; i < 4; ++i) w.dgh[i] = c.take<float>(3 * TBH);
        w.dhz = c.take<float>(4 * BH);
        w.dx1 = c.take<float>(2 * TBH);
        w.dgi0 = c.take<float>(6 * TBH);
    } else {
        for (int i = 0; i < 4; ++i) { w.whhT[i] = nullptr; w.dgh[i] = nullptr; }
        w.dgi1 = w.dhz = w.dx1 = w.dgi0 = nullptr;
    }
    for (int i = 0; i < 4; ++i) {
        const bool pk = pk_ok(H);
        w.wpk[i] = pk ? c.take<float>((size_t)3 * H * H) : nullptr;
        w.hpk[i] = pk ? c.take<float>(chain_ring_floats(B, H)) : nullptr;
        w.wpkT[i] = pk && save ? c.take<float>((size_t)3 * H * H) : nullptr;
        w.dghpk[i] = pk && save ? c.take<float>(chain_ring_floats(B, 3 * H)) : nullptr;
        w.wp3[i] = pk && !gru_chain_ok(H, B, T, 2) && gru_step_bf3_ok(H, B, T, 2) ? c.take<unsigned char>(gru_step_bf3_w_bytes(H)) : nullptr;
        w.wp3T[i] = pk && save && chain_chunk_rows_bwd(H, B, T, 2) == 0 && gru_step_bf3_bwd_ok(H, B, T, 2) ? c.take<unsigned char>(gru_step_bf3_w_bytes(H)) : nullptr;
    }
    w.sync = c.take<unsigned>(kSyncAreas * kChainSyncWords);
    // The bf16-pipe products (gemm_bf3.hip) tile 192 rows x 192 / 128 columns: below ~3072 rows (T*B) a launch leaves most CUs idle
    // (LatentRNN's context GRUs, T*B = 768: 27-52 TFLOP/s against 65-105 on the f32-input kernels) -- those stay on gemm.hip.
    const bool bf3 = gemm_bf3_ok(T * B, 6 * H, 2 * H) && (long)T * B >= 3072;
    w.x1pk = bf3 ? c.take<unsigned char>(bf3_bytes((long)T * B, 2 * H)) : nullptr;
    w.wih1pk = bf3 ? c.take<unsigned char>(bf3_bytes(6 * H, 2 * H)) : nullptr;
    const bool bf3b = bf3 && save && gemm_bf3_ok(T * B, 2 * H, 6 * H);
    w.dgi1pk = bf3b ? c.take<unsigned char>(bf3_bytes((long)T * B, 6 * H)) : nullptr;
    w.wih1Tpk = bf3b ? c.take<unsigned char>(bf3_bytes(2 * H, 6 * H)) : nullptr;
    const bool bf3w = bf3b && gemm_bf3_ok(3 * H, H, T * B) && gemm_bf3_ok(3 * H, 2 * H, T * B) && (T * B) % 64 == 0;
    for (int l = 0; l < 2; ++l) w.gT[l] = bf3w ? c.take<unsigned char>(bf3_bytes(6 * H, (long)T * B)) : nullptr;
    w.x1T = bf3w ? c.take<unsigned char>(bf3_bytes(2 * H, (long)T * B)) : nullptr;
    for (int i = 0; i < 4; ++i) {
        w.nrT[i] = bf3w ? c.take<unsigned char>(bf3_bytes(H, (long)T * B)) : nullptr;
        w.hpT[i] = bf3w ? c.take<unsigned char>(bf3_bytes(H, (long)T * B)) : nullptr;
    }
    return c.bytes();
}

// The backward call re-derives from the library's options which kernels the forward call ran and which piece buffers it filled
// (keys 4, 7, 8, 9, 12).  Changing one of them in between used to be silent wrong weight gradients: the forward call
// now notes the options it ran under per workspace, and the backward call refuses (-3) a workspace written under other options.
namespace {
std::unordered_map<const void*, unsigned> g_ws_opts;      // workspace -> options of the forward call that wrote it
std::mutex g_ws_opts_mu;
void ws_opts_note(const void* key, unsigned opts) {
    std::lock_guard<std::mutex> lk(g_ws_opts_mu);
    if (g_ws_opts.size() > 4096) g_ws_opts.clear();       // (workspaces whose backward call never came: forget them all)
    g_ws_opts[key] = opts;
}
// 0 = fine (or unknown workspace), -3 = written under other options; `consume`: the last backward stage forgets the entry
int ws_opts_check(const void* key, unsigned opts, bool consume) {
    std::lock_guard<std::mutex> lk(g_ws_opts_mu);
    const auto it = g_ws_opts.find(key);
    if (it == g_ws_opts.end()) return 0;
    const bool same = it->second == opts;
    if (consume || !same) g_ws_opts.erase(it);
    return same ? 0 : -3;
}
unsigned opts_snapshot() {
    return (unsigned)chain_enabled() | ((unsigned)chain2_mode() << 1) | ((unsigned)bf3_mode() << 5) | ((unsigned)emit_mask() << 9) |
           ((unsigned)(gru_step_bf3_ok(512, 2048, 24, 2) ? 1 : 0) << 12) | ((unsigned)(gru_step_bf3_ok(512, 1 << 20, 2, 2) ? 1 : 0) << 13);
}
}  // namespace

int bigru2_core_fwd(int B, int T, int H, const GruDirPtr* P, const BiGru2In& in, const float* h0, const float* mask,
                    float* const* hn, long hn_ld, BiGru2Ws& w, int save, hipStream_t s, int sync_prezeroed) {
    const long BH = (long)B * H, TBH = (long)T * BH;
    if (save) ws_opts_note(w.sync, opts_snapshot());
    // fragment-major W_hh twins: only the per-step kernels read them (the chain kernels take W_hh as stored)
    // (one launch, row chunks, or -- big batches -- the bf16-pipe step kernels, which take the same null h0)
    const bool stepf = w.wp3[0] && w.hpk[0] && pk_ok(H) && !gru_chain_ok(H, B, T, 2) && gru_step_bf3_ok(H, B, T, 2);
    const bool chained = stepf || (w.wpk[0] && w.hpk[0] && w.sync && pk_ok(H) && chain_chunk_rows(H, B, T, 2, save) > 0);
    // zero initial state: the chain kernels take a null pointer (and skip step 0's contraction), the per-step kernels a buffer
    const float* const hzero = chained ? nullptr : w.zeros;
    if (!h0 && !chained && pw_zero(w.zeros, BH, s) != 0) return -2;
    const bool one_launch = chained && gru_chain_ok(H, B, T, 2);      // (not the chunked form: it reuses one area per launch)
    const long TBl = (long)T * B;
    const bool bf3f = w.x1pk && w.wih1pk && bf3_mode() != 0 && gemm_bf3_ok(T * B, 6 * H, 2 * H);
    if (one_launch && !sync_prezeroed &&
        hipMemsetAsync(w.sync, 0, (size_t)kSyncAreas * kChainSyncWords * sizeof(unsigned), s) != hipSuccess) return -2;
    if (w.wpk[0] && !chained)
    {
        const float* ins[4] = {P[0].w_hh, P[1].w_hh, P[2].w_hh, P[3].w_hh};
        INET_TRY(pw_pack_frag_multi(ins, w.wpk, 4, H, 3 * H, H, 0, s));
    }
    // The layer-1 input weights as bf16 pieces (both directions stacked: the forward products' B operand; with saves also
    // k-major, the data gradient's B operand of the backward call -- the weights do not change in between): on the side
    // stream, under the layer-0 chain.
    // (Two forks: the k-major split stages through LDS and cannot start beside a chain workgroup that holds all of it; only the
    //  row split gates the forward product, the other is joined at the end of this call.)
    hipStream_t wss = s, wss2 = s;
    if (bf3f) {
        wss = side_fork(s);
        const long wp = (long)bf3_piece_bytes(6 * H, 2 * H), wpT = (long)bf3_piece_bytes(2 * H, 6 * H);
        for (int dir = 0; dir < 2; ++dir)
            INET_TRY(bf3_split(P[2 + dir].w_ih, 2L * H, 0, 3 * H, 2 * H, w.wih1pk, wp, 2 * H / 32, dir * 3 * H / 16, 0, wss));
        if (save && w.wih1Tpk) {
            wss2 = side_fork(s);
            for (int dir = 0; dir < 2; ++dir)
                INET_TRY(bf3_split(P[2 + dir].w_ih, 2L * H, 1, 2 * H, 3 * H, w.wih1Tpk, wpT, 6 * H / 32, 0, dir * 3 * H / 32, wss2));
        }
    }
    DirFwd d[2];
    for (int dir = 0; dir < 2; ++dir) {
        DirFwd& D = d[dir];
        D = DirFwd{};
        D.W_hh = P[dir].w_hh; D.b_hh = P[dir].b_hh;
        if (in.gi0[dir]) { D.gi = in.gi0[dir]; D.gi_ld = in.gi0_ld; D.gi_ts = in.gi0_ts; }
        if (in.tab[dir]) { D.table = in.tab[dir]; D.table_ld = in.tab_ld; D.idx = in.idx; D.idx_bs = in.idx_bs; D.idx_ts = in.idx_ts; }
        D.gvec = in.gvec[dir];
        D.h0 = h0 ? h0 + dir * BH : hzero; D.h0_ld = H;
        D.out = w.x1raw + dir * H; D.out_ld = 2L * H; D.out_ts = 2 * BH;
        if (mask) {
            D.outm = w.x1m + dir * H; D.outm_ld = 2L * H; D.outm_ts = 2 * BH;
            D.mask = mask + dir * H; D.mask_ld = 2L * H; D.mask_ts = 2 * BH;
        }
        if (hn && hn[dir]) { D.hlast = hn[dir]; D.hlast_ld = hn_ld; }
        if (save) { D.sv = w.sv[dir]; D.sv_astride = TBH; }
        D.reverse = dir;
        D.Wpk_hh = w.wpk[dir]; D.hpk = w.hpk[dir]; D.wp3 = w.wp3[dir];
        D.sync = w.sync; D.sync_prezeroed = one_launch;
        if (bf3f && (emit_mask() & 1)) {                       // the layer-1 input products' A operand, written by the chain
            D.em.rows = w.x1pk; D.em.rows_piece = (long)bf3_piece_bytes(TBl, 2 * H); D.em.rows_kb = 2 * H / 32; D.em.rows_kb0 = dir * H / 32;
        }
        if (bf3f && save && w.x1T && w.hpT[dir] && (emit_mask() & 2)) {   // the weight gradients' B operands (read by the backward call)
            D.em.colsA = w.x1T; D.em.colsA_piece = (long)bf3_piece_bytes(2 * H, TBl); D.em.colsA_rb0 = dir * H / 16;   // (layer 1's dW_ih)
        }
    }
    INET_TRY(gru_layer_fwd(H, B, T, 2, d, s));
    const bool x1_emitted = bf3f && (d[0].emitted & 1) && (d[1].emitted & 1) && (emit_mask() & 1);
    // (the backward call relies on the predicate for the TRANSPOSED pieces the forward chains wrote)
    if (bf3f && (emit_mask() & 1) && ((d[0].emitted & 2) && (d[1].emitted & 2)) != gru_layer_fwd_emits(H, B, T, 2, save != 0)) return -4;   // (options changed between sizing the workspace and this call)
    const float* x1 = mask ? w.x1m : w.x1raw;
    if (bf3f) {
        // both directions' input products as ONE product on the bf16 matrix cores (gemm_bf3.hip): gi1 [TB, 6H] =
        // x1 [TB, 2H] . [W_ih_fwd; W_ih_bwd]^T + [b_fwd | b_bwd]
        const long xp = (long)bf3_piece_bytes((long)T * B, 2 * H), wp = (long)bf3_piece_bytes(6 * H, 2 * H);
        const int KB = 2 * H / 32;
        if (!x1_emitted) INET_TRY(bf3_split(x1, 2L * H, 0, T * B, 2 * H, w.x1pk, xp, KB, 0, 0, s));
        if (wss != s) INET_TRY(stream_wait(s, wss));           // the weight pieces
        Bf3Gemm g{};
        g.A = w.x1pk; g.a_piece = xp; g.a_kb = KB; g.B = w.wih1pk; g.b_piece = wp; g.b_kb = KB;
        g.C = w.gi1; g.ldc = 6L * H; g.M = T * B; g.N = 6 * H; g.K = 2 * H;
        g.bias = P[2].b_ih; g.bias2 = P[3].b_ih; g.bias2_from = 3 * H;
        g.epi = EPI_NONE; g.acc = ACC_STORE; g.ksplit = 1;
        INET_TRY(launch_gemm_bf3(g, s));
    } else
    for (int dir = 0; dir < 2; ++dir)
        INET_TRY(linear_fwd(x1, 2L * H, P[2 + dir].w_ih, 2L * H, P[2 + dir].b_ih, w.gi1 + dir * 3L * H, 6L * H,
                            T * B, 3 * H, 2 * H, EPI_NONE, s));
    for (int dir = 0; dir < 2; ++dir) {
        DirFwd& D = d[dir];
        D = DirFwd{};
        D.W_hh = P[2 + dir].w_hh; D.b_hh = P[2 + dir].b_hh;
        D.gi = w.gi1 + dir * 3L * H; D.gi_ld = 6L * H; D.gi_ts = 6 * BH;
        D.h0 = h0 ? h0 + (2 + dir) * BH : hzero; D.h0_ld = H;
        D.out = w.h1 + dir * H; D.out_ld = 2L * H; D.out_ts = 2 * BH;
        if (hn && hn[2 + dir]) { D.hlast = hn[2 + dir]; D.hlast_ld = hn_ld; }
        if (save) { D.sv = w.sv[2 + dir]; D.sv_astride = TBH; }
        D.reverse = dir;
        D.Wpk_hh = w.wpk[2 + dir]; D.hpk = w.hpk[2 + dir]; D.wp3 = w.wp3[2 + dir];
        D.sync = one_launch ? w.sync + kChainSyncWords : w.sync; D.sync_prezeroed = one_launch;
        if (bf3f && save && w.hpT[2 + dir] && (emit_mask() & 2)) {
            D.em.colsB = w.hpT[2 + dir]; D.em.colsB_piece = (long)bf3_piece_bytes(H, TBl); D.em.colsB_rb0 = 0;
        }
    }
    INET_TRY(gru_layer_fwd(H, B, T, 2, d, s));
    if (wss2 != s) INET_TRY(stream_wait(s, wss2));             // the k-major weight pieces: read by the backward call on this stream
    return 0;
}

// Weight gradients of one layer's recurrent weights, both directions in one launch on the bf16 matrix cores:
// dW_hh_d [3H, H] += dgh_d^T hprev_d.  The transposed gate gradients gT (r, z of both directions; with `n_too` the n block as
// well: the layer's dgi^T) and n*r (nrT) are split here from the chain's dgi / dgh arrays.
static int bigru2_wgrad_hh_bf3(int B, int T, int H, int layer, const GruDirPtr* P, BiGru2Ws& w, const float* dgi, bool n_too,
                               bool gates_emitted, bool hprev_emitted, hipStream_t ss) {
    const long TB = (long)T * B, TBH = TB * H;
    const long gp = (long)bf3_piece_bytes(6 * H, TB), hp = (long)bf3_piece_bytes(H, TB);
    const int KB = (int)(TB / 32);
    for (int dir = 0; dir < 2; ++dir) {
        const int i = 2 * layer + dir;
        // r, z (, n) of dgi_d [TB, 3H] at column dir * 3H of the [TB, 6H] array -> row blocks dir * 3H / 16 ..
        if (!gates_emitted) {
            INET_TRY(bf3_split(dgi + dir * 3L * H, 6L * H, 1, (n_too ? 3 : 2) * H, (int)TB, w.gT[layer], gp, KB, dir * 3 * H / 16, 0, ss));
            INET_TRY(bf3_split(w.dgh[i] + 2L * H, 3L * H, 1, H, (int)TB, w.nrT[i], hp, KB, 0, 0, ss));
        }
        if (!hprev_emitted) INET_TRY(bf3_split(w.sv[i] + 4 * TBH, H, 1, H, (int)TB, w.hpT[i], hp, KB, 0, 0, ss));
    }
    const int i0 = 2 * layer;
    Bf3Gemm g{};
    g.A = w.gT[layer]; g.A2 = w.gT[layer] + (long)(3 * H / 16) * KB * 1024; g.a_piece = gp; g.a_kb = KB;
    g.a_alt_from = 2 * H / 16; g.A_alt = w.nrT[i0]; g.A2_alt = w.nrT[i0 + 1]; g.a_alt_piece = hp;
    g.B = w.hpT[i0]; g.B2 = w.hpT[i0 + 1]; g.b_piece = hp; g.b_kb = KB;
    g.C = P[i0].dw_hh; g.C2 = P[i0 + 1].dw_hh; g.ldc = H;
    g.M = 3 * H; g.N = H; g.K = (int)TB; g.epi = EPI_NONE; g.acc = ACC_ADD; g.nbatch = 2;
    return launch_gemm_bf3(g, ss);
}

int bigru2_core_bwd(int B, int T, int H, const GruDirPtr* P, const float* mask, const float* dout1,
                    const float* const* dhn, long dhn_ld, float* dh0, BiGru2Ws& w, hipStream_t s, int stage) {
    const long BH = (long)B * H, TBH = (long)T * BH;
    if (ws_opts_check(w.sync, opts_snapshot(), stage != 1) != 0) return -3;      // options changed since the forward call
    const bool wg = P[0].dw_hh != nullptr;
    // both layers run as backward chains (they read W_hh as stored) iff the conditions of gru_layer_bwd_range hold:
    // the transposed fragment-major twins are then never read
    const bool chained = w.wpkT[0] && w.dghpk[0] && w.sync && pk_ok(H) && chain_chunk_rows_bwd(H, B, T, 2) > 0;
    const bool bf3d_pre = bf3_mode() != 0 && w.dgi1pk && w.wih1Tpk && gemm_bf3_ok(T * B, 2 * H, 6 * H);
    const bool bf3w_pre = bf3_mode() != 0 && w.gT[0] && w.hpT[0] && P[0].dw_hh != nullptr;
    // what the forward call's chains wrote (the same predicate it checked itself against)
    const bool fwd_emitted = bf3w_pre && w.x1pk && gemm_bf3_ok(T * B, 6 * H, 2 * H) && gru_layer_fwd_emits(H, B, T, 2, true) && (emit_mask() & 2);
    // (stage 2 continues on what stage 1 left in the workspace: zeroed sync areas, packed / transposed weights, dx1)
    if (stage != 2 && chained &&
        hipMemsetAsync(w.sync, 0, (size_t)kSyncAreas * kChainSyncWords * sizeof(unsigned), s) != hipSuccess) return -2;
    if (stage == 2 || (w.wpkT[0] && chained)) {
    } else if (w.wpkT[0]) {
        const float* ins[4] = {P[0].w_hh, P[1].w_hh, P[2].w_hh, P[3].w_hh};
        INET_TRY(pw_pack_frag_multi(ins, w.wpkT, 4, H, H, 3 * H, 1, s));
    } else {
        for (int i = 0; i < 4; ++i) INET_TRY(pw_transpose(P[i].w_hh, H, w.whhT[i], 3L * H, 3 * H, H, s));
    }
    // ---- layer 1 ----
    DirBwd d[2];
    for (int dir = 0; dir < 2; ++dir) {
        DirBwd& D = d[dir];
        D = DirBwd{};
        D.W_hhT = w.whhT[2 + dir];
        if (dout1) { D.dout = dout1 + dir * H; D.dout_ld = 2L * H; D.dout_ts = 2 * BH; }
        if (dhn && dhn[2 + dir]) { D.dhn = dhn[2 + dir]; D.dhn_ld = dhn_ld; }
        D.sv = w.sv[2 + dir]; D.sv_astride = TBH;
        D.dgi = w.dgi1 + dir * 3L * H; D.dgi_ld = 6L * H; D.dgi_ts = 6 * BH;
        D.dgh = w.dgh[2 + dir];
        D.dhz = w.dhz + dir * 2 * BH;
        D.db_ih = P[2 + dir].db_ih; D.db_hh = P[2 + dir].db_hh;
        if (dh0) { D.dh0 = dh0 + (2 + dir) * BH; D.dh0_ld = H; D.dh0_acc = 0; }
        D.reverse = dir;
        D.Wpk_hhT = w.wpkT[2 + dir]; D.dghpk = w.dghpk[2 + dir];
        D.W_hh = P[2 + dir].w_hh; D.sync = w.sync; D.sync_prezeroed = chained; D.wp3T = w.wp3T[2 + dir];
        if (bf3d_pre && (emit_mask() & 4)) {
            D.em.rows = w.dgi1pk; D.em.rows_piece = (long)bf3_piece_bytes((long)T * B, 6 * H); D.em.rows_kb = 6 * H / 32;
            D.em.rows_kb0 = dir * 3 * H / 32;
        }
    }
    // The chains can hand their weight-gradient products to the side stream a chunk of steps at a time (CH < T) instead
    // of a layer's whole K = T*B product at the end of its chain.  Measured at B=256 with 2, 3, 4 chunks per layer:
    // 5.12 / 5.17 / 5.21 ms per step against 5.12 for one -- during backward the two streams together already keep the
    // chip busy, and smaller-K products are less efficient -- so one chunk.
    const int CH = T;
    const float* x1 = mask ? w.x1m : w.x1raw;
    const bool l1_gem = false;                                 // (the gate gradients' transposed pieces always come from split launches)
    // layer 1's weight gradients on the bf16 pipe: dW_hh (both directions), then dW_ih_d [3H, 2H] += dgi1_d^T x1 on the transposed
    // gate gradients just made
    auto l1_wgrads_bf3 = [&](hipStream_t ss) -> int {
        INET_TRY(bigru2_wgrad_hh_bf3(B, T, H, 1, P, w, w.dgi1, true, l1_gem, fwd_emitted, ss));
        const long TBl = (long)T * B;
        const long gp = (long)bf3_piece_bytes(6 * H, TBl), xp = (long)bf3_piece_bytes(2 * H, TBl);
        const int KB = (int)(TBl / 32);
        if (!fwd_emitted) INET_TRY(bf3_split(x1, 2L * H, 1, 2 * H, (int)TBl, w.x1T, xp, KB, 0, 0, ss));
        Bf3Gemm g{};
        g.A = w.gT[1]; g.A2 = w.gT[1] + (long)(3 * H / 16) * KB * 1024; g.a_piece = gp; g.a_kb = KB;
        g.B = w.x1T; g.B2 = w.x1T; g.b_piece = xp; g.b_kb = KB;
        g.C = P[2].dw_ih; g.C2 = P[3].dw_ih; g.ldc = 2L * H;
        g.M = 3 * H; g.N = 2 * H; g.K = (int)TBl; g.epi = EPI_NONE; g.acc = ACC_ADD; g.nbatch = 2;
        return launch_gemm_bf3(g, ss);
    };
    const bool bf3d = bf3_mode() != 0 && w.dgi1pk && w.wih1Tpk && gemm_bf3_ok(T * B, 2 * H, 6 * H);
    const bool bf3w = bf3_mode() != 0 && w.gT[0] && w.hpT[0] && wg;
    for (int hi = T - 1; hi >= 0 && stage != 2; hi -= CH) {
        const int lo = hi - CH + 1 > 0 ? hi - CH + 1 : 0, nt = hi - lo + 1;
        INET_TRY(gru_layer_bwd_range(H, B, T, 2, d, hi, lo, s));
        if (wg) {
            hipStream_t ss = side_fork(s);                   // leaf work: overlaps the rest of the BPTT chains
            if (nt == T && bf3w) {                            // both directions of a product in one launch (gemm_bf3.hip)
                INET_TRY(l1_wgrads_bf3(ss));
            } else if (nt == T) {                            // both directions of a product in one launch
                INET_TRY(linear_wgrad2(w.dgh[2], w.dgh[3], 3L * H, w.sv[2] + 4 * TBH, w.sv[3] + 4 * TBH, H, P[2].dw_hh,
                                       P[3].dw_hh, H, T * B, 3 * H, H, ss));
                INET_TRY(linear_wgrad2(w.dgi1, w.dgi1 + 3L * H, 6L * H, x1, x1, 2L * H, P[2].dw_ih, P[3].dw_ih, 2L * H,
                                       T * B, 3 * H, 2 * H, ss));
            } else
            for (int dir = 0; dir < 2; ++dir) {
                const int t_lo = dir ? T - 1 - hi : lo;       // the reverse direction walks time forwards
                const float* dgi = w.dgi1 + dir * 3L * H + (long)t_lo * B * 6 * H;
                INET_TRY(gru_dir_wgrad_range(H, B, t_lo, nt, w.dgh[2 + dir], w.sv[2 + dir] + 4 * TBH, P[2 + dir].dw_hh, ss));
                INET_TRY(linear_wgrad(dgi, 6L * H, x1 + (long)t_lo * B * 2 * H, 2L * H, P[2 + dir].dw_ih, 2L * H, nt * B,
                                      3 * H, 2 * H, ss));
            }
        }
    }
    const bool l1_rows_emitted = stage != 2 && bf3d_pre && (d[0].emitted & 1) && (d[1].emitted & 1) && (emit_mask() & 4);
    if (stage != 2 && bf3d) {
        // dx1 [TB, 2H] = (dgi1 [TB, 6H] . [W_ih_fwd; W_ih_bwd] [6H, 2H]) * mask: both directions as one K = 6H product
        const long dp = (long)bf3_piece_bytes((long)T * B, 6 * H), wp = (long)bf3_piece_bytes(2 * H, 6 * H);
        const int KB = 6 * H / 32;
        if (!l1_rows_emitted) INET_TRY(bf3_split(w.dgi1, 6L * H, 0, T * B, 6 * H, w.dgi1pk, dp, KB, 0, 0, s));
        // (B operand: the k-major weight pieces the forward call left in the workspace)
        Bf3Gemm g{};
        g.A = w.dgi1pk; g.a_piece = dp; g.a_kb = KB; g.B = w.wih1Tpk; g.b_piece = wp; g.b_kb = KB;
        g.C = w.dx1; g.ldc = 2L * H; g.M = T * B; g.N = 2 * H; g.K = 6 * H;
        g.epi = mask ? EPI_MUL_AUX : EPI_NONE; g.aux = mask; g.ldaux = 2L * H; g.acc = ACC_STORE; g.ksplit = 1;
        INET_TRY(launch_gemm_bf3(g, s));
    } else
    for (int dir = 0; dir < 2 && stage != 2; ++dir) {
        const float* dgi = w.dgi1 + dir * 3L * H;
        // dx1 [TB,2H] (+)= dgi1_dir [TB,3H] . W_ih_l1_dir [3H,2H]
        // the dropout mask of the layer-0 output rides in both epilogues: (a + b) m = a m + b m, exactly for the 0 / 2 of
        // p = 0.5 and to an ulp otherwise -- one 75 MB elementwise pass less in front of the layer-0 BPTT chain
        INET_TRY(linear_dgrad(dgi, 6L * H, P[2 + dir].w_ih, 2L * H, w.dx1, 2L * H, T * B, 3 * H, 2 * H,
                              mask ? EPI_MUL_AUX : EPI_NONE, mask, 2L * H, dir == 0 ? ACC_STORE : ACC_ADD, s));
    }
    if (stage == 1) return 0;
    // ---- layer 0 ----
    for (int dir = 0; dir < 2; ++dir) {
        DirBwd& D = d[dir];
        D = DirBwd{};
        D.W_hhT = w.whhT[dir];
        D.dout = w.dx1 + dir * H; D.dout_ld = 2L * H; D.dout_ts = 2 * BH;
        if (dhn && dhn[dir]) { D.dhn = dhn[dir]; D.dhn_ld = dhn_ld; }
        D.sv = w.sv[dir]; D.sv_astride = TBH;
        D.dgi = w.dgi0 + dir * 3L * H; D.dgi_ld = 6L * H; D.dgi_ts = 6 * BH;
        D.dgh = w.dgh[dir];
        D.dhz = w.dhz + dir * 2 * BH;
        D.db_ih = P[dir].db_ih; D.db_hh = P[dir].db_hh;
        if (dh0) { D.dh0 = dh0 + dir * BH; D.dh0_ld = H; D.dh0_acc = 0; }
        D.reverse = dir;
        D.Wpk_hhT = w.wpkT[dir]; D.dghpk = w.dghpk[dir];
        D.W_hh = P[dir].w_hh; D.sync = chained ? w.sync + kChainSyncWords : w.sync; D.sync_prezeroed = chained; D.wp3T = w.wp3T[dir];
    }
    for (int hi = T - 1; hi >= 0; hi -= CH) {
        const int lo = hi - CH + 1 > 0 ? hi - CH + 1 : 0;
        INET_TRY(gru_layer_bwd_range(H, B, T, 2, d, hi, lo, s));
        if (wg) {
            hipStream_t ss = side_fork(s);
            if (hi - lo + 1 == T)
                INET_TRY(linear_wgrad2(w.dgh[0], w.dgh[1], 3L * H, w.sv[0] + 4 * TBH, w.sv[1] + 4 * TBH, H, P[0].dw_hh,
                                       P[1].dw_hh, H, T * B, 3 * H, H, ss));
            else
            for (int dir = 0; dir < 2; ++dir) {
                const int t_lo = dir ? T - 1 - hi : lo;         // the reverse direction walks time forwards
                INET_TRY(gru_dir_wgrad_range(H, B, t_lo, hi - lo + 1, w.dgh[dir], w.sv[dir] + 4 * TBH, P[dir].dw_hh, ss));
            }
        }
    }
    return 0;
}
