// MeasureVAE encoder / hierarchical decoder: host-side orchestration of the
// gfx950 kernels.  Follows MeasureVAE/encoder.py:104-134 and
// MeasureVAE/decoder.py:392-529 of the reference; see oracle/torch_ref.py for
// the CPU restatement these are tested against.
//
// Algebraic re-arrangements (exact in real arithmetic, fp32 round-off only):
//  * embedding -> GRU layer-0 input projection is folded into a gather table
//    Table[v] = E[v] . W_ih[:, :E]^T + b_ih  (V x 3H), rebuilt per call because
//    the weights train; the fused step kernel gathers Table[token].
//  * the beat-constant half of the tick GRU input, c_i . W_ih[:, E:]^T, is hoisted
//    out of the 6 ticks of a beat.
//  * embedding gradients are one-hot MFMA contractions (no contended atomics).
//  * in the backward pass the 4 beats are independent through the tick GRU's
//    hidden state (it is re-initialised per beat), so BPTT runs 6 steps over 4
//    problems instead of 24 steps.
#include "seq.h"
#include "layout.h"
#include "vae.h"
#include "decode_chain.h"
#include <cstdlib>
#include <string>

namespace {

struct EncWs {
    float *tabF, *tabR, *hcat, *a_mu, *a_ls;
    BiGru2Ws g;
    float *d_amu, *d_als, *dhcat, *onehot, *dtab;
};

size_t enc_carve(const inet_vae_config& c, int B, int save, void* base, EncWs& w) {
    Carver cv(base);
    const int T = c.beats * c.ticks_per_beat, H = c.enc_hidden, V = c.num_notes, E = c.emb_dim;
    w.tabF = cv.take<float>((size_t)V * 3 * H);
    w.tabR = cv.take<float>((size_t)V * 3 * H);
    w.hcat = cv.take<float>((size_t)B * 4 * H);
    w.a_mu = cv.take<float>((size_t)B * 2 * H);
    w.a_ls = cv.take<float>((size_t)B * 2 * H);
    bigru2_carve(cv, B, T, H, save, w.g);
    if (save) {
        w.d_amu = cv.take<float>((size_t)B * 2 * H);
        w.d_als = cv.take<float>((size_t)B * 2 * H);
        w.dhcat = cv.take<float>((size_t)B * 4 * H);
        const bool seg = V <= 128 && E <= 16;                // token_segsum / table_grad kernels (pointwise.hip); else one-hot products
        w.onehot = seg ? nullptr : cv.take<float>((size_t)T * B * V);
        w.dtab = cv.take<float>((size_t)V * 6 * H);
    } else {
        w.d_amu = w.d_als = w.dhcat = w.onehot = w.dtab = nullptr;
    }
    return cv.bytes();
}

void enc_ptrs(const VaeLayout& L, const float* p, float* g, GruDirPtr* P) {
    for (int i = 0; i < 4; ++i) {
        const GruDirOff& o = L.enc[i];
        P[i].w_ih = p + o.w_ih; P[i].w_hh = p + o.w_hh; P[i].b_ih = p + o.b_ih; P[i].b_hh = p + o.b_hh;
        P[i].dw_ih = g ? g + o.w_ih : nullptr; P[i].dw_hh = g ? g + o.w_hh : nullptr;
        P[i].db_ih = g ? g + o.b_ih : nullptr; P[i].db_hh = g ? g + o.b_hh : nullptr;
        P[i].K = o.K;
    }
}

}  // namespace

size_t vae_encoder_ws_bytes(const inet_vae_config& c, int B, int save) {
    EncWs w;
    return enc_carve(c, B, save, nullptr, w);
}

int vae_encoder_fwd(const inet_vae_config& c, int B, const long long* tokens, const float* p, const float* mask,
                    float* mu, float* logsigma, void* ws, int save, hipStream_t s) {
    const int T = c.beats * c.ticks_per_beat, H = c.enc_hidden, V = c.num_notes, E = c.emb_dim, Z = c.z_dim;
    VaeLayout L(c);
    EncWs w;
    enc_carve(c, B, save, ws, w);
    GruDirPtr P[4];
    enc_ptrs(L, p, nullptr, P);
    // gather tables: E_enc [V,E] x W_ih_l0[dir] [3H,E]^T + b_ih -- and the chain kernels' sync areas zeroed -- in one launch
    {
        PwPrologue pr{};
        for (int dir = 0; dir < 2; ++dir)
            pr.tab[dir] = PwTableJob{p + L.enc_emb, E, V, P[dir].w_ih, E, P[dir].b_ih, dir ? w.tabR : w.tabF, 3L * H, 3 * H, E};
        pr.ntab = 2;
        pr.zero_words = w.g.sync; pr.nzero = (long)kSyncAreas * kChainSyncWords;
        pr.tok_src = tokens; pr.tok_B = B; pr.tok_T = T; pr.tok_V = V;     // range check only (encoder.py:118 embeds them)
        INET_TRY(pw_prologue(pr, s));
    }
    BiGru2In in{};
    in.tab[0] = w.tabF; in.tab[1] = w.tabR; in.tab_ld = 3L * H;
    in.idx = tokens; in.idx_bs = T; in.idx_ts = 1;
    // final hiddens land directly in hcat = [l0f | l0b | l1f | l1b]   (encoder.py:126-127)
    float* hn[4] = {w.hcat, w.hcat + H, w.hcat + 2 * H, w.hcat + 3 * H};
    INET_TRY(bigru2_core_fwd(B, T, H, P, in, nullptr, mask, hn, 4L * H, w.g, save, s, 1));
    // the two heads (encoder.py:130-133) side by side: two launches of two products instead of four launches
    const GemmArgs l1[2] = {linear_fwd_args(w.hcat, 4L * H, p + L.mean_w0, 4L * H, p + L.mean_b0, w.a_mu, 2L * H, B, 2 * H, 4 * H, EPI_SELU),
                            linear_fwd_args(w.hcat, 4L * H, p + L.ls_w0, 4L * H, p + L.ls_b0, w.a_ls, 2L * H, B, 2 * H, 4 * H, EPI_SELU)};
    INET_TRY(launch_gemm_group(l1, 2, s));
    const GemmArgs l2[2] = {linear_fwd_args(w.a_mu, 2L * H, p + L.mean_w2, 2L * H, p + L.mean_b2, mu, Z, B, Z, 2 * H, EPI_NONE),
                            linear_fwd_args(w.a_ls, 2L * H, p + L.ls_w2, 2L * H, p + L.ls_b2, logsigma, Z, B, Z, 2 * H, EPI_NONE)};
    INET_TRY(launch_gemm_group(l2, 2, s));
    return 0;
}

int vae_encoder_bwd(const inet_vae_config& c, int B, const long long* tokens, const float* p, float* g,
                    const float* mask, const float* dmu, const float* dls, void* ws, hipStream_t s, int stage) {
    const int T = c.beats * c.ticks_per_beat, H = c.enc_hidden, V = c.num_notes, E = c.emb_dim, Z = c.z_dim;
    if (!g) return -1;
    VaeLayout L(c);
    EncWs w;
    enc_carve(c, B, 1, ws, w);
    GruDirPtr P[4];
    enc_ptrs(L, p, g, P);
    // heads  (stage 2 of a staged call starts at the layer-0 BPTT: see vae.h)
    if (stage != 2) {
    {
        const GemmArgs d2[2] = {linear_dgrad_args(dmu, Z, p + L.mean_w2, 2L * H, w.d_amu, 2L * H, B, Z, 2 * H, EPI_MUL_SELU_GRAD, w.a_mu, 2L * H, ACC_STORE),
                                linear_dgrad_args(dls, Z, p + L.ls_w2, 2L * H, w.d_als, 2L * H, B, Z, 2 * H, EPI_MUL_SELU_GRAD, w.a_ls, 2L * H, ACC_STORE)};
        INET_TRY(launch_gemm_group(d2, 2, s));
    }
    INET_TRY(linear_dgrad(w.d_amu, 2L * H, p + L.mean_w0, 4L * H, w.dhcat, 4L * H, B, 2 * H, 4 * H, EPI_NONE, nullptr, 0, ACC_STORE, s));
    INET_TRY(linear_dgrad(w.d_als, 2L * H, p + L.ls_w0, 4L * H, w.dhcat, 4L * H, B, 2 * H, 4 * H, EPI_NONE, nullptr, 0, ACC_ADD, s));
    {   // leaf work (weight / bias gradients of the heads) on the side stream
        hipStream_t ss = side_fork(s);
        const GemmArgs w2[2] = {linear_wgrad_args(dmu, Z, w.a_mu, 2L * H, g + L.mean_w2, 2L * H, B, Z, 2 * H),
                                linear_wgrad_args(dls, Z, w.a_ls, 2L * H, g + L.ls_w2, 2L * H, B, Z, 2 * H)};
        INET_TRY(launch_gemm_group(w2, 2, ss));
        const GemmArgs w0[2] = {linear_wgrad_args(w.d_amu, 2L * H, w.hcat, 4L * H, g + L.mean_w0, 4L * H, B, 2 * H, 4 * H),
                                linear_wgrad_args(w.d_als, 2L * H, w.hcat, 4L * H, g + L.ls_w0, 4L * H, B, 2 * H, 4 * H)};
        INET_TRY(launch_gemm_group(w0, 2, ss));
        const PwColsumJob cs[4] = {{dmu, Z, B, Z, g + L.mean_b2}, {dls, Z, B, Z, g + L.ls_b2},
                                   {w.d_amu, 2L * H, B, 2 * H, g + L.mean_b0}, {w.d_als, 2L * H, B, 2 * H, g + L.ls_b0}};
        INET_TRY(pw_colsum_multi(cs, 4, ss));
    }
    }
    // GRU stack
    const float* dhn[4] = {w.dhcat, w.dhcat + H, w.dhcat + 2 * H, w.dhcat + 3 * H};
    INET_TRY(bigru2_core_bwd(B, T, H, P, mask, nullptr, dhn, 4L * H, nullptr, w.g, s, stage));
    if (stage == 1) return side_join(s);
    // embedding / layer-0 input weights through the gather table.  On the MAIN stream: these are the last items of the
    // step's backward pass and the side stream is still busy with the layer-0 dW_hh products (r02 timeline: queued behind
    // them they delayed the optimizer by ~0.14 ms).
    {
        constexpr bool emb_main = true;
        hipStream_t ss = emb_main ? s : side_fork(s);
        // dTable [V, 6H] (both directions side by side, as dgi0 holds them): the rows of dgi0 summed by token -- one pass over
        // dgi0 at HBM rate (it shares the chip with the layer-0 dW_hh products, which own the MFMA pipes); then
        // dW_ih_l0[dir] [3H,E] += dTable_dir^T . E_enc and dE_enc [V,E] += dTable_dir . W_ih_l0[dir] in one small launch
        if (!w.onehot) {
            INET_TRY(pw_token_segsum(w.g.dgi0, 6L * H, tokens, B, 1, T, T * B, V, 6 * H, w.dtab, ss));   // row (t,b) -> tokens[b*T + t]
            const float* wih[2] = {P[0].w_ih, P[1].w_ih};
            float* dwih[2] = {P[0].dw_ih, P[1].dw_ih};
            INET_TRY(pw_table_grad(w.dtab, V, 3 * H, 2, E, p + L.enc_emb, E, wih, dwih, E, g + L.enc_emb, E, ss));
        } else {                                             // large vocabularies: dTable = onehot^T . dgi0 on the GEMM path
            INET_TRY(pw_onehot(tokens, B, 1, T, T * B, V, w.onehot, 1, ss));
            for (int dir = 0; dir < 2; ++dir) {
                const float* dgi = w.g.dgi0 + dir * 3L * H;
                INET_TRY(launch_gemm(gemm_args(w.onehot, V, 1, dgi, 6L * H, 1, w.dtab, 3L * H, V, 3 * H, T * B), ss));
                INET_TRY(linear_wgrad(w.dtab, 3L * H, p + L.enc_emb, E, P[dir].dw_ih, E, V, 3 * H, E, ss));
                INET_TRY(linear_dgrad(w.dtab, 3L * H, P[dir].w_ih, E, g + L.enc_emb, E, V, 3 * H, E, EPI_NONE, nullptr, 0, ACC_ADD, ss));
            }
        }
    }
    return side_join(s);
}

// =====================================================================================
// Decoder
// =====================================================================================
namespace {

struct DecWs {
    float *zsave, *hb0, *gvec0, *beat0, *beat0m, *svb0, *gi1b, *beat_out, *svb1;
    float *ht0, *c_all, *cgi, *table;
    long long* idxV;
    float *h0seq, *h0m, *svt0, *h1seq, *svt1, *wtm, *gi1t;
    long long* tokin;                                // teacher-forced input tokens [B,T] (start symbol, then target shifted)
    // backward
    float *whhT[4], *dlg, *dh1top, *dgi1t, *dgh1t, *dhz, *dht0, *dx1t, *dgi0t, *dgh0t, *dcgi, *dc_all, *onehot, *dtable;
    float *dbeat_out, *dgi1b, *dgh1b, *dxb, *dgi0b, *dgh0b, *dhb0, *tmp3h, *b0part;
    // fragment-major twins (ksplit.h), null unless pk_ok(H): packed recurrent / layer-1 input weights, packed initial
    // tick hiddens [layer][beat], ping-pong packed hiddens [beat][2], packed masked layer-0 output [beat]
    float *wpk_b[2], *wpk_t0, *wpk_t1hh, *wpk_t1ih, *wpk_out, *hpk_b, *ht0pk, *hpk_t0, *hpk_t1, *hm0pk;
    float *wpkT[4], *dghpk;
    float* amax; unsigned* sync;                     // fused decode kernel (decode_chain.h): partial argmax, counters
    unsigned* b1ex;                                  // b <= 4: the granule exchange of decode_b1.hip, right in front of `sync` (one zero range)
    unsigned* b1stamps;                              // ... and its diagnostic stamps (vae_ws_field which = 2, "b1stamps")
};

size_t dec_carve(const inet_vae_config& c, int B, int save, void* base, DecWs& w) {
    Carver cv(base);
    const size_t nb = c.beats, T = (size_t)c.beats * c.ticks_per_beat, H = c.dec_hidden, V = c.num_notes, Z = c.z_dim;
    const size_t BH = (size_t)B * H;
    w.zsave = cv.take<float>((size_t)B * Z);
    w.hb0 = cv.take<float>(2 * BH);
    w.gvec0 = cv.take<float>(3 * H);
    w.beat0 = cv.take<float>(nb * BH);
    w.beat0m = cv.take<float>(nb * BH);
    w.svb0 = save ? cv.take<float>(5 * nb * BH) : nullptr;
    w.gi1b = cv.take<float>(3 * nb * BH);
    w.beat_out = cv.take<float>(nb * BH);
    w.svb1 = save ? cv.take<float>(5 * nb * BH) : nullptr;
    w.ht0 = cv.take<float>(2 * nb * BH);
    w.c_all = cv.take<float>(nb * BH);
    w.cgi = cv.take<float>(3 * nb * BH);
    w.table = cv.take<float>((V + 1) * 3 * H);
    w.idxV = cv.take<long long>(B);
    w.h0seq = cv.take<float>(T * BH);
    w.h0m = cv.take<float>(T * BH);
    w.svt0 = save ? cv.take<float>(5 * T * BH) : nullptr;
    w.h1seq = cv.take<float>(T * BH);
    w.svt1 = save ? cv.take<float>(5 * T * BH) : nullptr;
    w.wtm = cv.take<float>(T * B * V);
    w.gi1t = cv.take<float>(3 * T * BH);
    w.tokin = cv.take<long long>((size_t)B * T);
    if (save) {
        for (int i = 0; i < 4; ++i) w.whhT[i] = cv.take<float>(3 * H * H);
        w.dlg = cv.take<float>(T * B * V);
        w.dh1top = cv.take<float>(T * BH);
        w.dgi1t = cv.take<float>(3 * T * BH);
        w.dgh1t = cv.take<float>(3 * T * BH);
        w.dhz = cv.take<float>(2 * nb * BH);
        w.dht0 = cv.take<float>(2 * nb * BH);
        w.dx1t = cv.take<float>(T * BH);
        w.dgi0t = cv.take<float>(3 * T * BH);
        w.dgh0t = cv.take<float>(3 * T * BH);
        w.dcgi = cv.take<float>(3 * nb * BH);
        w.dc_all = cv.take<float>(nb * BH);
        w.onehot = (V + 1 <= 128 && c.emb_dim <= 16) ? nullptr : cv.take<float>(T * B * (V + 1));   // (token_segsum path needs none)
        w.dtable = cv.take<float>((V + 1) * 3 * H);
        w.dbeat_out = cv.take<float>(nb * BH);
        w.dgi1b = cv.take<float>(3 * nb * BH);
        w.dgh1b = cv.take<float>(3 * nb * BH);
        w.dxb = cv.take<float>(nb * BH);
        w.dgi0b = cv.take<float>(3 * nb * BH);
        w.dgh0b = cv.take<float>(3 * nb * BH);
        w.dhb0 = cv.take<float>(2 * BH);
    }
    const bool pk = pk_ok((int)H);
    const size_t pkh = pk_floats(B, (int)H), W3 = 3 * H * H;
    w.wpk_b[0] = pk ? cv.take<float>(W3) : nullptr;
    w.wpk_b[1] = pk ? cv.take<float>(W3) : nullptr;
    w.wpk_t0 = pk ? cv.take<float>(W3) : nullptr;
    w.wpk_t1hh = pk ? cv.take<float>(W3) : nullptr;
    w.wpk_t1ih = pk ? cv.take<float>(W3) : nullptr;
    w.wpk_out = pk && V % 16 == 0 ? cv.take<float>(V * H) : nullptr;
    w.hpk_b = pk ? cv.take<float>(chain_ring_floats(B, (int)H)) : nullptr;
    w.ht0pk = pk ? cv.take<float>(2 * nb * pkh) : nullptr;
    w.hpk_t0 = pk ? cv.take<float>(nb * chain_ring_floats(B, (int)H)) : nullptr;   // per beat: a chain ring (3 pkh) or two fp32 slots
    w.hpk_t1 = pk ? cv.take<float>(nb * chain_ring_floats(B, (int)H)) : nullptr;
    w.hm0pk = pk ? cv.take<float>(nb * pkh) : nullptr;
    for (int i = 0; i < 4; ++i) w.wpkT[i] = pk && save ? cv.take<float>(W3) : nullptr;
    w.dghpk = pk && save ? cv.take<float>(nb * chain_ring_floats(B, 3 * (int)H)) : nullptr;
    w.amax = cv.take<float>(2 * 2 * ((V + 15) / 16) * ((B + 15) / 16) * 16);
    static_assert(kDecodeSyncWords <= kChainSyncWords, "one sync area serves either kind of chain launch");
    w.b1stamps = (B <= kDecodeB1MaxRows && H == 512 && !save) ? cv.take<unsigned>(kDecodeB1StampWords) : nullptr;
    w.b1ex = (B <= kDecodeB1MaxRows && H == 512 && !save) ? cv.take<unsigned>(decode_b1_words(B)) : nullptr;
    w.sync = cv.take<unsigned>(kSyncAreas * kChainSyncWords);
    w.tmp3h = save ? cv.take<float>(3 * H) : nullptr;          // right behind the sync areas: one memset zeroes both
    w.b0part = save ? cv.take<float>(64) : nullptr;            // fixed-order partial sums of the b_0 gradient (pw_beat_input_grad)
    return cv.bytes();
}

// fragment-major operands of tick j of beat i: layer 0 = P0, layer 1 = P1 (whose x is layer 0's masked output)
void tick_pk(const DecWs& w, int i, int j, int nb, long pkh, bool masked, GruFwdProb& P0, GruFwdProb& P1) {
    float* r0 = w.hpk_t0 + (long)i * 2 * pkh;
    float* r1 = w.hpk_t1 + (long)i * 2 * pkh;
    P0.Wpk_hh = w.wpk_t0;
    P0.hpk_prev = j == 0 ? w.ht0pk + (long)i * pkh : r0 + (long)((j + 1) & 1) * pkh;
    P0.hpk_new = r0 + (long)(j & 1) * pkh;
    if (masked) P0.hmpk_new = w.hm0pk + (long)i * pkh;
    P1.Wpk_hh = w.wpk_t1hh; P1.Wpk_ih = w.wpk_t1ih;
    P1.xpk = masked ? w.hm0pk + (long)i * pkh : P0.hpk_new;
    P1.hpk_prev = j == 0 ? w.ht0pk + (long)(nb + i) * pkh : r1 + (long)((j + 1) & 1) * pkh;
    P1.hpk_new = r1 + (long)(j & 1) * pkh;
}

}  // namespace

size_t vae_decoder_ws_bytes(const inet_vae_config& c, int B, int save) {
    DecWs w{};
    return dec_carve(c, B, save, nullptr, w);
}

int vae_decoder_fwd(const inet_vae_config& c, int B, const float* z, const long long* target, int teacher_forced,
                    const float* p, const float* mask_beat, const float* mask_tick, float* weights,
                    long long* samples, void* ws, int save, hipStream_t s, uint64_t multinomial_seed) {
    const int nb = c.beats, G = c.ticks_per_beat, T = nb * G, H = c.dec_hidden, V = c.num_notes, E = c.emb_dim, Z = c.z_dim;
    const long BH = (long)B * H;
    if (nb > 4) return -1;
    if (teacher_forced && !target) return -1;
    VaeLayout L(c);
    DecWs w{};
    dec_carve(c, B, save, ws, w);
    const bool pk = w.wpk_t0 != nullptr;
    constexpr bool tf_batch = true;
    const long pkh = (long)pk_floats(B, H);
    constexpr bool beat_chain = true;                         // (rounds 2-4 had an environment switch for the per-step beat path)
    constexpr bool train_chain = true;
    // Which kernels will run: the chain kernels read W_hh / W_ih as stored, only the per-step kernels want the
    // fragment-major twins -- each is packed only if its consumer runs.
    const bool beats_chained = pk && beat_chain && chain_chunk_rows(H, B, nb, 1, save) > 0;   // (one launch, or one per row chunk)
    const bool fused_shape = pk && !teacher_forced && !multinomial_seed && ((!save && !mask_tick) || train_chain);
    // batches beyond one resident launch (LatentRNN decodes 512 measures per step): the rows are independent, so the fused
    // kernel runs over chunks of 512 rows (the 64-row build: 27 us per tick instead of 2 x 20) or 256, one launch after the other
    const int kDecodeChunk = B % 512 == 0 ? 512 : 256;
    constexpr bool dec_chunks = true;
    const bool fused_whole = fused_shape && decode_chain_ok(B, H, V, T, G);
    const bool fused_chunked = fused_shape && !fused_whole && dec_chunks && B > kDecodeChunk && B % kDecodeChunk == 0 &&
                               decode_chain_ok(kDecodeChunk, H, V, T, G);
    const bool fused_decode = fused_whole || fused_chunked;
    // one measure, inference: decode_b1.hip's register-resident launch (reads the row-major initial hiddens: no packed twins)
    const bool b1_decode = fused_whole && !save && !mask_tick && w.b1ex && w.b1ex + decode_b1_words(B) == w.sync &&
                           decode_b1_shape_ok(B, H, V, T, G);
    const bool b1_fused = b1_decode && !mask_beat && decode_b1_fused((int)Z, B);   // ... with the beat path inside the same launch
    // teacher-forced ticks: every input token is known and the 4 beats are independent, so each tick layer is a chain of
    // G steps over the beats as problems -- `npl` beats per launch, as many as fit the chip at once (2 at B = 256)
    constexpr bool tf_chain = true;
    int npl = 0;
    if (pk && teacher_forced && tf_batch && tf_chain) {
        for (int n = nb; n >= 1 && !npl; --n)                  // whole batch in one launch per `n` beats ...
            if (nb % n == 0 && 3 + 2 * (nb / n) <= kSyncAreas && gru_chain_ok(H, B, G, n)) npl = n;
        for (int n = nb; n >= 1 && !npl; --n)                  // ... else row chunks
            if (nb % n == 0 && 3 + 2 * (nb / n) <= kSyncAreas && chain_chunk_rows(H, B, G, n, save) > 0) npl = n;
    }
    const bool ticks_chained = npl > 0;
    const float* wih0 = p + L.tick[0].w_ih;                   // [3H, E+H]
    const long ldw0 = E + H;
    {
        // One launch for the scattered little jobs (each used to be its own ~5 us launch): z saved for the backward pass, the
        // chain kernels' sync areas zeroed, the beat GRU's constant input gates gvec0 = b_0 W_ih[:,0] + b_ih, the tick
        // GRU's gather table (rows 0..V-1 = E_dec . W_ih[:, :E]^T + b_ih; row V = x_0 . W_ih[:, :E]^T + b_ih), the start
        // token index, and the teacher-forced token copy / shift.
        PwPrologue pr{};
        pr.tab[0] = PwTableJob{p + L.dec_emb, E, V, wih0, ldw0, p + L.tick[0].b_ih, w.table, 3L * H, 3 * H, E};
        pr.tab[1] = PwTableJob{p + L.x_0, E, 1, wih0, ldw0, p + L.tick[0].b_ih, w.table + (long)V * 3 * H, 3L * H, 3 * H, E};
        pr.ntab = 2;
        if (beats_chained || fused_decode || ticks_chained) { pr.zero_words = w.sync; pr.nzero = (long)kSyncAreas * kChainSyncWords; }
        if (b1_decode) {                                       // ... and the b = 1 kernel's granules in front of them
            pr.zero_words = w.b1ex; pr.nzero = decode_b1_words(B) + (long)kSyncAreas * kChainSyncWords;
        }
        if (save) { pr.copy_src = reinterpret_cast<const unsigned*>(z); pr.copy_dst = reinterpret_cast<unsigned*>(w.zsave); pr.ncopy = (long)B * Z; }
        pr.axpb_a = p + L.b_0; pr.axpb_x = p + L.beat[0].w_ih; pr.axpb_incx = 1; pr.axpb_b = p + L.beat[0].b_ih;
        pr.axpb_y = w.gvec0; pr.axpb_n = 3 * H;
        pr.fill_ptr = w.idxV; pr.nfill = B; pr.fill_val = V;
        if (teacher_forced) {
            pr.tok_src = target; pr.tok_copy = samples; pr.tok_B = B; pr.tok_T = T; pr.tok_first = V; pr.tok_V = V;
            if (ticks_chained) pr.tok_shift = w.tokin;
        }
        INET_TRY(pw_prologue(pr, s));
    }
    if (pk) {
        const float* ins[5]; float* outs[5];
        int n = 0;
        if (!beats_chained && !b1_fused) {
            ins[n] = p + L.beat[0].w_hh; outs[n++] = w.wpk_b[0];
            ins[n] = p + L.beat[1].w_hh; outs[n++] = w.wpk_b[1];
        }
        if (!fused_decode && !ticks_chained) {
            ins[n] = p + L.tick[0].w_hh; outs[n++] = w.wpk_t0;
            ins[n] = p + L.tick[1].w_hh; outs[n++] = w.wpk_t1hh;
            ins[n] = p + L.tick[1].w_ih; outs[n++] = w.wpk_t1ih;
        }
        if (n) INET_TRY(pw_pack_frag_multi(ins, outs, n, H, 3 * H, H, 0, s));
        if (w.wpk_out && !fused_decode && !(teacher_forced && tf_batch))
            INET_TRY(pw_pack_frag(p + L.out_w, H, V, H, w.wpk_out, 0, 1, 0, 0, s));
    }

    if (!b1_fused) {                                           // (one measure: these eight launches are roles of decode_b1.hip's launch)
        // ---- beat RNN (forward_beat_rnn, decoder.py:455-471) ----
        INET_TRY(linear_fwd(z, Z, p + L.zb_w, Z, p + L.zb_b, w.hb0, 2L * H, B, 2 * H, Z, EPI_SELU, s));
        DirFwd d{};
        d.W_hh = p + L.beat[0].w_hh; d.b_hh = p + L.beat[0].b_hh;
        d.gvec = w.gvec0;
        d.h0 = w.hb0; d.h0_ld = 2L * H;
        d.out = w.beat0; d.out_ld = H; d.out_ts = BH;
        if (mask_beat) { d.outm = w.beat0m; d.outm_ld = H; d.outm_ts = BH; d.mask = mask_beat; d.mask_ld = H; d.mask_ts = BH; }
        if (save) { d.sv = w.svb0; d.sv_astride = nb * BH; }
        d.Wpk_hh = w.wpk_b[0]; d.hpk = w.hpk_b;
        if (beats_chained) { d.sync = w.sync; d.sync_prezeroed = 1; }   // 4 steps in one launch (8 groups of 32 rows at B = 256)
        INET_TRY(gru_layer_fwd(H, B, nb, 1, &d, s));
        const float* xb = mask_beat ? w.beat0m : w.beat0;
        INET_TRY(linear_fwd(xb, H, p + L.beat[1].w_ih, H, p + L.beat[1].b_ih, w.gi1b, 3L * H, nb * B, 3 * H, H, EPI_NONE, s));
        d = DirFwd{};
        d.W_hh = p + L.beat[1].w_hh; d.b_hh = p + L.beat[1].b_hh;
        d.gi = w.gi1b; d.gi_ld = 3L * H; d.gi_ts = 3 * BH;
        d.h0 = w.hb0 + H; d.h0_ld = 2L * H;
        d.out = w.beat_out; d.out_ld = H; d.out_ts = BH;
        if (save) { d.sv = w.svb1; d.sv_astride = nb * BH; }
        d.Wpk_hh = w.wpk_b[1]; d.hpk = w.hpk_b;
        if (beats_chained) { d.sync = w.sync + kChainSyncWords; d.sync_prezeroed = 1; }
        INET_TRY(gru_layer_fwd(H, B, nb, 1, &d, s));

        // ---- per-beat constants for the tick RNN (decoder.py:494-495), all 4 beats at once ----
        {
            const GemmArgs bt[2] = {linear_fwd_args(w.beat_out, H, p + L.bh_w, H, p + L.bh_b, w.ht0, 2L * H, nb * B, 2 * H, H, EPI_SELU),
                                    linear_fwd_args(w.beat_out, H, p + L.bi_w, H, p + L.bi_b, w.c_all, H, nb * B, H, H, EPI_SELU)};
            INET_TRY(launch_gemm_group(bt, 2, s));
        }
        if (pk && !ticks_chained && !fused_chunked && !b1_decode)  // packed initial tick hiddens: [layer][beat]
            for (int l = 0; l < 2; ++l)
                INET_TRY(pw_pack_frag(w.ht0 + (long)l * H, 2L * H, B, H, w.ht0pk + (long)l * nb * pkh, 0, nb, (long)B * 2 * H, pkh, s));
        INET_TRY(linear_fwd(w.c_all, H, wih0 + E, ldw0, nullptr, w.cgi, 3L * H, nb * B, 3 * H, H, EPI_NONE, s));
    }

    // ---- tick RNN (forward_tick_rnn, decoder.py:473-529) ----
    if (ticks_chained) {
        const float* x1 = mask_tick ? w.h0m : w.h0seq;
        for (int layer = 0; layer < 2; ++layer) {
            if (layer == 1)                                    // layer 1's input-side pre-activations for all 24 ticks at once
                INET_TRY(linear_fwd(x1, H, p + L.tick[1].w_ih, H, p + L.tick[1].b_ih, w.gi1t, 3L * H, T * B, 3 * H, H, EPI_NONE, s));
            for (int i0 = 0; i0 < nb; i0 += npl) {
                DirFwd dd[4];
                for (int k = 0; k < npl; ++k) {
                    const int i = i0 + k;
                    DirFwd& D = dd[k];
                    D = DirFwd{};
                    D.W_hh = p + L.tick[layer].w_hh; D.b_hh = p + L.tick[layer].b_hh;
                    D.h0 = w.ht0 + (long)i * B * 2 * H + layer * H; D.h0_ld = 2L * H;
                    D.out_ld = H; D.out_ts = BH;
                    if (layer == 0) {
                        D.gi = w.cgi + (long)i * B * 3 * H; D.gi_ld = 3L * H; D.gi_ts = 0;   // the beat's constant half
                        D.table = w.table; D.table_ld = 3L * H;                              // the token half
                        D.idx = w.tokin + (long)i * G; D.idx_bs = T; D.idx_ts = 1;
                        D.out = w.h0seq + (long)i * G * BH;
                        if (mask_tick) {
                            D.outm = w.h0m + (long)i * G * BH; D.outm_ld = H; D.outm_ts = BH;
                            D.mask = mask_tick + (long)i * G * BH; D.mask_ld = H; D.mask_ts = BH;
                        }
                        if (save) { D.sv = w.svt0 + (long)i * G * BH; D.sv_astride = (long)T * BH; }
                        D.Wpk_hh = w.wpk_t0; D.hpk = w.hpk_t0 + (long)i * 3 * pkh;
                    } else {
                        D.gi = w.gi1t + (long)i * G * 3 * BH; D.gi_ld = 3L * H; D.gi_ts = 3 * BH;
                        D.out = w.h1seq + (long)i * G * BH;
                        if (save) { D.sv = w.svt1 + (long)i * G * BH; D.sv_astride = (long)T * BH; }
                        D.Wpk_hh = w.wpk_t1hh; D.hpk = w.hpk_t1 + (long)i * 3 * pkh;
                    }
                }
                dd[0].sync = w.sync + (long)(2 + layer * (nb / npl) + i0 / npl) * kChainSyncWords;
                dd[0].sync_prezeroed = 1;
                INET_TRY(gru_layer_fwd(H, B, G, npl, dd, s));
            }
        }
        INET_TRY(linear_fwd(w.h1seq, H, p + L.out_w, H, p + L.out_b, w.wtm, V, T * B, V, H, EPI_RELU, s));
        INET_TRY(pw_swap01(w.wtm, T, B, V, weights, s));             // [T,B,V] -> [B,T,V]
        return 0;
    }
    if (teacher_forced && tf_batch) {
        // Every input token is known, and the tick GRU's hidden state is re-initialised per beat, so the 4 beats are
        // independent: 6 steps x 4 problems per layer instead of 24 dependent steps, and ONE output projection.
        const long as = (long)T * BH;
        for (int j = 0; j < G; ++j) {
            GruFwdBatch b0{}, b1{};
            b0.H = b1.H = H; b0.nprob = b1.nprob = nb;
            for (int i = 0; i < nb; ++i) {
                const int t = i * G + j;
                GruFwdProb& P0 = b0.p[i];
                P0.B = B;
                if (j == 0) { P0.h_prev = w.ht0 + (long)i * B * 2 * H; P0.ld_hprev = 2L * H; }
                else { P0.h_prev = w.h0seq + (long)(t - 1) * BH; P0.ld_hprev = H; }
                P0.W_hh = p + L.tick[0].w_hh; P0.b_hh = p + L.tick[0].b_hh;
                P0.gi_dense = w.cgi + (long)i * B * 3 * H; P0.ld_gi = 3L * H;
                P0.gi_table = w.table; P0.ld_table = 3L * H;
                if (t == 0) { P0.idx = w.idxV; P0.idx_stride = 1; }
                else { P0.idx = samples + (t - 1); P0.idx_stride = T; }
                P0.h_new = w.h0seq + (long)t * BH; P0.ld_hnew = H;
                if (mask_tick) { P0.h_masked = w.h0m + (long)t * BH; P0.ld_hm = H; P0.mask = mask_tick + (long)t * BH; P0.ld_mask = H; }
                if (save) {
                    float* q = w.svt0 + (long)t * BH;
                    P0.sv_r = q; P0.sv_z = q + as; P0.sv_n = q + 2 * as; P0.sv_ghn = q + 3 * as; P0.sv_hprev = q + 4 * as;
                }
                GruFwdProb& P1 = b1.p[i];
                if (pk) tick_pk(w, i, j, nb, pkh, mask_tick != nullptr, P0, P1);
                P1.B = B;
                if (j == 0) { P1.h_prev = w.ht0 + (long)i * B * 2 * H + H; P1.ld_hprev = 2L * H; }
                else { P1.h_prev = w.h1seq + (long)(t - 1) * BH; P1.ld_hprev = H; }
                P1.W_hh = p + L.tick[1].w_hh; P1.b_hh = p + L.tick[1].b_hh;
                P1.x = (mask_tick ? w.h0m : w.h0seq) + (long)t * BH; P1.ldx = H; P1.K2 = H;
                P1.W_ih = p + L.tick[1].w_ih; P1.ld_wih = H; P1.b_ih = p + L.tick[1].b_ih;
                P1.h_new = w.h1seq + (long)t * BH; P1.ld_hnew = H;
                if (save) {
                    float* q = w.svt1 + (long)t * BH;
                    P1.sv_r = q; P1.sv_z = q + as; P1.sv_n = q + 2 * as; P1.sv_ghn = q + 3 * as; P1.sv_hprev = q + 4 * as;
                }
            }
            INET_TRY(launch_gru_fwd(b0, s));
            INET_TRY(launch_gru_fwd(b1, s));
        }
        INET_TRY(linear_fwd(w.h1seq, H, p + L.out_w, H, p + L.out_b, w.wtm, V, T * B, V, H, EPI_RELU, s));
        INET_TRY(pw_swap01(w.wtm, T, B, V, weights, s));             // [T,B,V] -> [B,T,V]
        return 0;
    }
    if (fused_decode) {
        // all 24 free-running ticks (layer 0, layer 1, projection, argmax, token feedback) in ONE launch: inference, and
        // the free-running half of the training steps (dropout mask between the layers, backward saves written on the way)
        const int Bc = fused_whole ? B : kDecodeChunk;
        const long pkc = (long)pk_floats(Bc, H);
        for (int r0 = 0; r0 < B; r0 += Bc) {
            if (fused_chunked)                                 // the chunk's initial tick hiddens, fragment-major [layer][beat]
                for (int l = 0; l < 2; ++l)
                    INET_TRY(pw_pack_frag(w.ht0 + (long)r0 * 2 * H + (long)l * H, 2L * H, Bc, H, w.ht0pk + (long)l * nb * pkc, 0, nb,
                                          (long)B * 2 * H, pkc, s));
            DecodeChainArgs a{};
            a.B = Bc; a.Bs = B; a.H = H; a.T = T; a.G = G; a.V = V;
            a.W_hh0 = p + L.tick[0].w_hh; a.b_hh0 = p + L.tick[0].b_hh;
            a.cgi = w.cgi + (long)r0 * 3 * H; a.table = w.table;
            a.W_ih1 = p + L.tick[1].w_ih; a.b_ih1 = p + L.tick[1].b_ih; a.W_hh1 = p + L.tick[1].w_hh; a.b_hh1 = p + L.tick[1].b_hh;
            a.W_out = p + L.out_w; a.b_out = p + L.out_b;
            a.ht0 = w.ht0 + (long)r0 * 2 * H; a.ht0pk = w.ht0pk;
            a.hx0 = w.hpk_t0; a.hx1 = w.hpk_t1; a.amax = w.amax;
            a.b1ex = b1_decode ? reinterpret_cast<unsigned long long*>(w.b1ex) : nullptr;
            static const bool b1st = [] { const char* v = std::getenv("INET_DECODE_B1_STAMPS"); return v && v[0] == '1'; }();
            a.b1stamps = (b1_decode && b1st && T <= 32) ? reinterpret_cast<unsigned long long*>(w.b1stamps) : nullptr;
            if (b1_fused) {
                DecodeB1Beat& bt = a.beat;
                bt.z = z; bt.zb_w = p + L.zb_w; bt.zb_b = p + L.zb_b; bt.gvec0 = w.gvec0;
                bt.W_hh0 = p + L.beat[0].w_hh; bt.b_hh0 = p + L.beat[0].b_hh;
                bt.W_ih1 = p + L.beat[1].w_ih; bt.b_ih1 = p + L.beat[1].b_ih; bt.W_hh1 = p + L.beat[1].w_hh; bt.b_hh1 = p + L.beat[1].b_hh;
                bt.bh_w = p + L.bh_w; bt.bh_b = p + L.bh_b; bt.bi_w = p + L.bi_w; bt.bi_b = p + L.bi_b;
                bt.wih0_c = wih0 + E; bt.wih0_ld = ldw0;
            }
            a.weights = weights + (long)r0 * T * V; a.samples = samples + (long)r0 * T;
            a.counters = w.sync + 2 * kChainSyncWords; a.prezeroed = r0 == 0;   // (later chunks: the launcher zeroes the area)
            if (mask_tick) { a.mask = mask_tick + (long)r0 * H; a.hx0m = w.hm0pk; }
            if (save) {
                a.sv0 = w.svt0 + (long)r0 * H; a.sv1 = w.svt1 + (long)r0 * H; a.sv_stride = (long)T * BH;
                a.h0out = (mask_tick ? w.h0m : w.h0seq) + (long)r0 * H; a.h1seq = w.h1seq + (long)r0 * H;
            }
            INET_TRY(launch_decode_chain(a, s));
        }
        return 0;
    }
    for (int t = 0; t < T; ++t) {
        const int i = t / G, j = t % G;
        GruFwdBatch b0{};
        b0.H = H; b0.nprob = 1;
        GruFwdProb& P0 = b0.p[0];
        P0.B = B;
        if (j == 0) { P0.h_prev = w.ht0 + (long)i * B * 2 * H; P0.ld_hprev = 2L * H; }
        else { P0.h_prev = w.h0seq + (long)(t - 1) * BH; P0.ld_hprev = H; }
        P0.W_hh = p + L.tick[0].w_hh; P0.b_hh = p + L.tick[0].b_hh;
        P0.gi_dense = w.cgi + (long)i * B * 3 * H; P0.ld_gi = 3L * H;
        P0.gi_table = w.table; P0.ld_table = 3L * H;
        if (t == 0) { P0.idx = w.idxV; P0.idx_stride = 1; }
        else { P0.idx = samples + (t - 1); P0.idx_stride = T; }
        P0.h_new = w.h0seq + (long)t * BH; P0.ld_hnew = H;
        if (mask_tick) { P0.h_masked = w.h0m + (long)t * BH; P0.ld_hm = H; P0.mask = mask_tick + (long)t * BH; P0.ld_mask = H; }
        if (save) {
            float* q = w.svt0 + (long)t * BH; const long as = (long)T * BH;
            P0.sv_r = q; P0.sv_z = q + as; P0.sv_n = q + 2 * as; P0.sv_ghn = q + 3 * as; P0.sv_hprev = q + 4 * as;
        }
        GruFwdBatch b1{};
        b1.H = H; b1.nprob = 1;
        GruFwdProb& P1 = b1.p[0];
        P1.B = B;
        if (j == 0) { P1.h_prev = w.ht0 + (long)i * B * 2 * H + H; P1.ld_hprev = 2L * H; }
        else { P1.h_prev = w.h1seq + (long)(t - 1) * BH; P1.ld_hprev = H; }
        P1.W_hh = p + L.tick[1].w_hh; P1.b_hh = p + L.tick[1].b_hh;
        P1.x = (mask_tick ? w.h0m : w.h0seq) + (long)t * BH; P1.ldx = H; P1.K2 = H;
        P1.W_ih = p + L.tick[1].w_ih; P1.ld_wih = H; P1.b_ih = p + L.tick[1].b_ih;
        P1.h_new = w.h1seq + (long)t * BH; P1.ld_hnew = H;
        if (save) {
            float* q = w.svt1 + (long)t * BH; const long as = (long)T * BH;
            P1.sv_r = q; P1.sv_z = q + as; P1.sv_n = q + 2 * as; P1.sv_ghn = q + 3 * as; P1.sv_hprev = q + 4 * as;
        }
        if (pk) tick_pk(w, i, j, nb, pkh, mask_tick != nullptr, P0, P1);
        INET_TRY(launch_gru_fwd(b0, s));
        INET_TRY(launch_gru_fwd(b1, s));

        // logits = ReLU(h_top . Wo^T + bo) straight into weights[:, t, :], fused with the argmax that feeds tick t+1
        const bool draw = multinomial_seed != 0 && !teacher_forced;       // decoder.py:506-509: sample the fed-back token
        int rc = draw ? 1 : launch_logits_argmax(w.h1seq + (long)t * BH, H, B, H, p + L.out_w, p + L.out_b, V,
                                                 weights + (long)t * V, (long)T * V, teacher_forced ? nullptr : samples + t,
                                                 T, s, pk ? P1.hpk_new : nullptr, w.wpk_out);
        if (rc < 0) return rc;
        if (rc == 1) {                                         // V not a multiple of 16 (or > 64), or sampling: two kernels
            INET_TRY(linear_fwd(w.h1seq + (long)t * BH, H, p + L.out_w, H, p + L.out_b, weights + (long)t * V,
                                (long)T * V, B, V, H, EPI_RELU, s));
            if (draw) INET_TRY(pw_sample_multinomial(weights + (long)t * V, (long)T * V, B, V, samples + t, T,
                                                     multinomial_seed, (uint64_t)t * B, s));
            else if (!teacher_forced) INET_TRY(pw_argmax(weights + (long)t * V, (long)T * V, B, V, samples + t, T, s));
        }
    }
    return 0;
}

int vae_decoder_bwd(const inet_vae_config& c, int B, const float* dweights, const float* weights,
                    const long long* tokens_in, const float* p, float* g, const float* mask_beat,
                    const float* mask_tick, float* dz, void* ws, hipStream_t s) {
    const int nb = c.beats, G = c.ticks_per_beat, T = nb * G, H = c.dec_hidden, V = c.num_notes, E = c.emb_dim, Z = c.z_dim;
    const long BH = (long)B * H, TBH = (long)T * BH;
    VaeLayout L(c);
    DecWs w{};
    dec_carve(c, B, 1, ws, w);
    const GruDirOff* gr[4] = {&L.beat[0], &L.beat[1], &L.tick[0], &L.tick[1]};
    constexpr bool beat_chain = true;                         // (rounds 2-4 had an environment switch for the per-step beat path)
    // the backward chain kernels read W_hh as stored (transposed on the fly, once): the fragment-major W_hh^T twins are
    // only packed for layers that fall back to one launch per step
    const bool beats_chained = w.wpkT[0] && w.dghpk && beat_chain && chain_chunk_rows_bwd(H, B, nb, 1) > 0;
    const bool ticks_chained = w.wpkT[0] && w.dghpk && chain_chunk_rows_bwd(H, B, G, nb) > 0;
    // one memset: the chain kernels' sync areas and, right behind them, the accumulator of the beat GRU's input-gate column sum
    if (hipMemsetAsync(w.sync, 0, (size_t)((char*)(w.tmp3h + 3 * H) - (char*)w.sync), s) != hipSuccess) return -2;
    if (w.wpkT[0]) {
        const float* ins[4]; float* outs[4];
        int n = 0;
        for (int i = 0; i < 4; ++i)
            if (!(i < 2 ? beats_chained : ticks_chained)) { ins[n] = p + gr[i]->w_hh; outs[n++] = w.wpkT[i]; }
        if (n) INET_TRY(pw_pack_frag_multi(ins, outs, n, H, H, 3 * H, 1, s));
    } else {
        for (int i = 0; i < 4; ++i) INET_TRY(pw_transpose(p + gr[i]->w_hh, H, w.whhT[i], 3L * H, 3 * H, H, s));
    }
    const long pkg = (long)pk_floats(B, 3 * H);

    // ---- output projection ----
    // Leaf work (weight / bias gradients: nothing downstream reads them before the optimizer) goes to the side streams in
    // TWO fork sessions -- after the layer-0 tick chain (both tick layers' products), at the end (the beat path's) -- instead of
    // one per module: every fork is an event on the main stream, and the small products of one session share a grouped launch.
    INET_TRY(pw_dlogits_relayout(dweights, weights, B, T, V, w.dlg, s));
    INET_TRY(linear_dgrad(w.dlg, V, p + L.out_w, H, w.dh1top, H, T * B, V, H, EPI_NONE, nullptr, 0, ACC_STORE, s));

    // ---- tick RNN layer 1: 6 steps x 4 beats ----
    DirBwd d[4];
    for (int i = 0; i < nb; ++i) {
        DirBwd& D = d[i];
        D = DirBwd{};
        D.W_hhT = w.whhT[3];
        D.dout = w.dh1top + (long)i * G * BH; D.dout_ld = H; D.dout_ts = BH;
        D.sv = w.svt1 + (long)i * G * BH; D.sv_astride = TBH;
        D.dgi = w.dgi1t + (long)i * G * 3 * BH; D.dgi_ld = 3L * H; D.dgi_ts = 3 * BH;
        D.dgh = w.dgh1t + (long)i * G * 3 * BH;
        D.dhz = w.dhz + (long)i * 2 * BH;
        if (g) { D.db_ih = g + L.tick[1].b_ih; D.db_hh = g + L.tick[1].b_hh; }
        D.dh0 = w.dht0 + (long)i * B * 2 * H + H; D.dh0_ld = 2L * H; D.dh0_acc = 0;
        D.Wpk_hhT = w.wpkT[3]; D.dghpk = w.dghpk ? w.dghpk + (long)i * 3 * pkg : nullptr;
        if (ticks_chained) { D.W_hh = p + L.tick[1].w_hh; D.sync = w.sync; D.sync_prezeroed = 1; }   // 4 beats = 4 problems, 2 row tiles per workgroup
    }
    INET_TRY(gru_layer_bwd(H, B, G, nb, d, s));
    const float* x1 = mask_tick ? w.h0m : w.h0seq;
    // Layer 1's leaf work is issued BEHIND the layer-0 chain, not beside it (round 5, profiles/r05_s_leaf_schedule.txt): beside the
    // chain its 6144-row products doubled the chain's time (231 us against 124 alone -- a persistent chain and a throughput product on
    // the same CUs overlap almost not at all), behind it they run beside the beat path's small latency-bound products, which lose
    // little: 3.60 -> 3.55 ms per step.  (leaf_when 0: beside the layer-0 chain, as rounds 2-4 had it; 2: with the beat path's
    // session at the end -- beside the encoder's layer-1 BPTT chain, 3.76 ms.)
    constexpr int leaf_when = 1;
    auto layer1_leaf = [&](hipStream_t ss) -> int {
        INET_TRY(linear_wgrad2(w.dgh1t, w.dgi1t, 3L * H, w.svt1 + 4 * TBH, x1, H, g + L.tick[1].w_hh, g + L.tick[1].w_ih, H,
                               T * B, 3 * H, H, ss));      // recurrent and input weights of layer 1 in one launch
        INET_TRY(linear_wgrad(w.dlg, V, w.h1seq, H, g + L.out_w, H, T * B, V, H, ss));
        INET_TRY(pw_colsum(w.dlg, V, T * B, V, g + L.out_b, ss));
        return 0;
    };
    if (g && leaf_when == 0) {
        hipStream_t ss = side_fork(s);
        INET_TRY(layer1_leaf(ss));
    }
    INET_TRY(linear_dgrad(w.dgi1t, 3L * H, p + L.tick[1].w_ih, H, w.dx1t, H, T * B, 3 * H, H,
                          mask_tick ? EPI_MUL_AUX : EPI_NONE, mask_tick, H, ACC_STORE, s));

    // ---- tick RNN layer 0 ----
    int dcgi_done = 0;
    for (int i = 0; i < nb; ++i) {
        DirBwd& D = d[i];
        D = DirBwd{};
        D.W_hhT = w.whhT[2];
        D.dout = w.dx1t + (long)i * G * BH; D.dout_ld = H; D.dout_ts = BH;
        D.sv = w.svt0 + (long)i * G * BH; D.sv_astride = TBH;
        D.dgi = w.dgi0t + (long)i * G * 3 * BH; D.dgi_ld = 3L * H; D.dgi_ts = 3 * BH;
        D.dgh = w.dgh0t + (long)i * G * 3 * BH;
        D.dhz = w.dhz + (long)i * 2 * BH;
        if (g) { D.db_ih = g + L.tick[0].b_ih; D.db_hh = g + L.tick[0].b_hh; }
        D.dh0 = w.dht0 + (long)i * B * 2 * H; D.dh0_ld = 2L * H; D.dh0_acc = 0;
        D.Wpk_hhT = w.wpkT[2]; D.dghpk = w.dghpk ? w.dghpk + (long)i * 3 * pkg : nullptr;
        if (ticks_chained) { D.W_hh = p + L.tick[0].w_hh; D.sync = w.sync + kChainSyncWords; D.sync_prezeroed = 1; }
        D.dgi_sum = w.dcgi + (long)i * 3 * BH; D.dgi_sum_done = &dcgi_done;     // beat-constant input half, see below
    }
    INET_TRY(gru_layer_bwd(H, B, G, nb, d, s));
    const float* wih0 = p + L.tick[0].w_ih;
    const long ldw0 = E + H;
    // beat-constant input half:  dcgi[i] = sum_j dgi0[6i+j]  (summed in registers by the chain kernel when it ran)
    if (!dcgi_done) INET_TRY(pw_group_sum(w.dgi0t, nb, G, 3 * BH, w.dcgi, s));
    INET_TRY(linear_dgrad(w.dcgi, 3L * H, wih0 + E, ldw0, w.dc_all, H, nb * B, 3 * H, H, EPI_MUL_SELU_GRAD, w.c_all, H,
                          ACC_STORE, s));
    if (g) {
        hipStream_t ss = side_fork(s);
        if (leaf_when == 1) INET_TRY(layer1_leaf(ss));
        INET_TRY(gru_dir_wgrad(H, B, T, w.dgh0t, w.svt0 + 4 * TBH, g + L.tick[0].w_hh, ss));
        INET_TRY(linear_wgrad(w.dcgi, 3L * H, w.c_all, H, g + L.tick[0].w_ih + E, ldw0, nb * B, 3 * H, H, ss));
        // token-embedding half through the gather table (rows 0..V-1 = the embeddings, row V = the start symbol x_0)
        if (!w.onehot) {
            // dTable [V+1, 3H] = the rows of dgi0 summed by the token that selected them -- one pass over dgi0 at HBM rate --
            // then dW_ih[:, :E] += dTable^T . [E_dec; x_0] and d[E_dec; x_0] += dTable . W_ih[:, :E] in one small launch
            INET_TRY(pw_shift_tokens(tokens_in, B, T, V, w.tokin, ss));                      // row (t,b) -> token fed at tick t
            INET_TRY(pw_token_segsum(w.dgi0t, 3L * H, w.tokin, B, 1, T, T * B, V + 1, 3 * H, w.dtable, ss));
            const float* wih[1] = {wih0};
            float* dwih[1] = {g + L.tick[0].w_ih};
            INET_TRY(pw_table_grad(w.dtable, V + 1, 3 * H, 1, E, p + L.dec_emb, E, wih, dwih, ldw0, g + L.dec_emb, E, ss,
                                   p + L.x_0, g + L.x_0));
        } else {
        INET_TRY(pw_zero(w.onehot, (long)T * B * (V + 1), ss));
        INET_TRY(pw_onehot(w.idxV, B, 0, 1, B, V + 1, w.onehot, 0, ss));                               // t = 0: x_0 row
        INET_TRY(pw_onehot(tokens_in, B, 1, T, (T - 1) * B, V + 1, w.onehot + (long)B * (V + 1), 0, ss));  // t >= 1: token t-1
        INET_TRY(launch_gemm(gemm_args(w.onehot, V + 1, 1, w.dgi0t, 3L * H, 1, w.dtable, 3L * H, V + 1, 3 * H, T * B), ss));
        INET_TRY(linear_wgrad(w.dtable, 3L * H, p + L.dec_emb, E, g + L.tick[0].w_ih, ldw0, V, 3 * H, E, ss));
        INET_TRY(linear_wgrad(w.dtable + (long)V * 3 * H, 3L * H, p + L.x_0, E, g + L.tick[0].w_ih, ldw0, 1, 3 * H, E, ss));
        INET_TRY(linear_dgrad(w.dtable, 3L * H, wih0, ldw0, g + L.dec_emb, E, V, 3 * H, E, EPI_NONE, nullptr, 0, ACC_ADD, ss));
        INET_TRY(linear_dgrad(w.dtable + (long)V * 3 * H, 3L * H, wih0, ldw0, g + L.x_0, E, 1, 3 * H, E, EPI_NONE, nullptr, 0, ACC_ADD, ss));
        }
    }

    // ---- beat -> tick linears (decoder.py:494-495) ----
    INET_TRY(pw_mul(w.dht0, w.ht0, 2L * nb * BH, 1, s));
    INET_TRY(linear_dgrad(w.dht0, 2L * H, p + L.bh_w, H, w.dbeat_out, H, nb * B, 2 * H, H, EPI_NONE, nullptr, 0, ACC_STORE, s));
    INET_TRY(linear_dgrad(w.dc_all, H, p + L.bi_w, H, w.dbeat_out, H, nb * B, H, H, EPI_NONE, nullptr, 0, ACC_ADD, s));

    // ---- beat RNN ----
    DirBwd b{};
    b.W_hhT = w.whhT[1];
    b.dout = w.dbeat_out; b.dout_ld = H; b.dout_ts = BH;
    b.sv = w.svb1; b.sv_astride = nb * BH;
    b.dgi = w.dgi1b; b.dgi_ld = 3L * H; b.dgi_ts = 3 * BH;
    b.dgh = w.dgh1b; b.dhz = w.dhz;
    if (g) { b.db_ih = g + L.beat[1].b_ih; b.db_hh = g + L.beat[1].b_hh; }
    b.dh0 = w.dhb0 + H; b.dh0_ld = 2L * H;
    b.Wpk_hhT = w.wpkT[1]; b.dghpk = w.dghpk;
    if (beats_chained) { b.W_hh = p + L.beat[1].w_hh; b.sync = w.sync + 2 * kChainSyncWords; b.sync_prezeroed = 1; }
    INET_TRY(gru_layer_bwd(H, B, nb, 1, &b, s));
    const float* xb = mask_beat ? w.beat0m : w.beat0;
    INET_TRY(linear_dgrad(w.dgi1b, 3L * H, p + L.beat[1].w_ih, H, w.dxb, H, nb * B, 3 * H, H,
                          mask_beat ? EPI_MUL_AUX : EPI_NONE, mask_beat, H, ACC_STORE, s));
    b = DirBwd{};
    b.W_hhT = w.whhT[0];
    b.dout = w.dxb; b.dout_ld = H; b.dout_ts = BH;
    b.sv = w.svb0; b.sv_astride = nb * BH;
    b.dgi = w.dgi0b; b.dgi_ld = 3L * H; b.dgi_ts = 3 * BH;
    b.dgh = w.dgh0b; b.dhz = w.dhz;
    if (g) { b.db_ih = g + L.beat[0].b_ih; b.db_hh = g + L.beat[0].b_hh; }
    b.dh0 = w.dhb0; b.dh0_ld = 2L * H;
    b.Wpk_hhT = w.wpkT[0]; b.dghpk = w.dghpk;
    if (beats_chained) { b.W_hh = p + L.beat[0].w_hh; b.sync = w.sync + 3 * kChainSyncWords; b.sync_prezeroed = 1; }
    INET_TRY(gru_layer_bwd(H, B, nb, 1, &b, s));

    // ---- z -> beat hidden ----
    INET_TRY(pw_mul(w.dhb0, w.hb0, 2 * BH, 1, s));
    if (dz) INET_TRY(linear_dgrad(w.dhb0, 2L * H, p + L.zb_w, Z, dz, Z, B, 2 * H, Z, EPI_NONE, nullptr, 0, ACC_STORE, s));
    if (g) {
        // the beat path's leaf work in one session: six weight gradients in two grouped launches, four column sums in one
        hipStream_t ss = side_fork(s);
        if (leaf_when == 2) INET_TRY(layer1_leaf(ss));
        const GemmArgs ga[4] = {linear_wgrad_args(w.dht0, 2L * H, w.beat_out, H, g + L.bh_w, H, nb * B, 2 * H, H),
                                linear_wgrad_args(w.dc_all, H, w.beat_out, H, g + L.bi_w, H, nb * B, H, H),
                                linear_wgrad_args(w.dgh1b, 3L * H, w.svb1 + 4 * nb * BH, H, g + L.beat[1].w_hh, H, nb * B, 3 * H, H),
                                linear_wgrad_args(w.dgi1b, 3L * H, xb, H, g + L.beat[1].w_ih, H, nb * B, 3 * H, H)};
        INET_TRY(launch_gemm_group(ga, 4, ss));
        const GemmArgs gb[2] = {linear_wgrad_args(w.dgh0b, 3L * H, w.svb0 + 4 * nb * BH, H, g + L.beat[0].w_hh, H, nb * B, 3 * H, H),
                                linear_wgrad_args(w.dhb0, 2L * H, w.zsave, Z, g + L.zb_w, Z, B, 2 * H, Z)};
        INET_TRY(launch_gemm_group(gb, 2, ss));
        // bias gradients; gi(beat layer 0) = b_0 * W_ih[:,0] + b_ih: its column sum lands in tmp3h (zeroed with the sync areas)
        const PwColsumJob cs[4] = {{w.dht0, 2L * H, nb * B, 2 * H, g + L.bh_b}, {w.dc_all, H, nb * B, H, g + L.bi_b},
                                   {w.dhb0, 2L * H, B, 2 * H, g + L.zb_b}, {w.dgi0b, 3L * H, nb * B, 3 * H, w.tmp3h}};
        INET_TRY(pw_colsum_multi(cs, 4, ss));
        INET_TRY(pw_beat_input_grad(w.tmp3h, p + L.beat[0].w_ih, 1, p + L.b_0, g + L.beat[0].w_ih, g + L.b_0, 3 * H,
                                    w.dgi0b, 3L * H, nb * B, w.b0part, ss));
    }
    return side_join(s);
}

// Test hook (inet_vae_ws_field): where the SELU outputs of the Linear+SELU heads live inside a workspace, so that a
// parity test can read the branch every element took (SELU's derivative jumps at 0: two fp32 implementations that
// differ by 1e-7 in a pre-activation next to 0 otherwise disagree by O(1) in that element's derivative).
int vae_ws_field(const inet_vae_config& c, int B, int which, const char* name, long long* offset_floats, long long* count) {
    char* const base = reinterpret_cast<char*>(static_cast<uintptr_t>(1) << 30);
    const long long nb = c.beats;
    const float* p = nullptr;
    long long n = 0;
    const std::string f(name ? name : "");
    if (which == 0) {
        EncWs w;
        enc_carve(c, B, 1, base, w);
        const long long H = c.enc_hidden;
        if (f == "a_mu") { p = w.a_mu; n = (long long)B * 2 * H; }
        else if (f == "a_ls") { p = w.a_ls; n = (long long)B * 2 * H; }
    } else if (which == 1) {
        DecWs w{};
        dec_carve(c, B, 1, base, w);
        const long long H = c.dec_hidden;
        if (f == "hb0") { p = w.hb0; n = (long long)B * 2 * H; }                 // [B, 2H]
        else if (f == "ht0") { p = w.ht0; n = nb * B * 2 * H; }                  // [beats, B, 2H]
        else if (f == "c_all") { p = w.c_all; n = nb * B * H; }                  // [beats, B, H]
    }
    else if (which == 2) {                                   // the inference workspace (save = 0)
        DecWs w{};
        dec_carve(c, B, 0, base, w);
        if (f == "b1stamps" && w.b1stamps) { p = reinterpret_cast<const float*>(w.b1stamps); n = kDecodeB1StampWords; }
    }
    if (!p) return -1;
    if (offset_floats) *offset_floats = (reinterpret_cast<const char*>(p) - base) / (long long)sizeof(float);
    if (count) *count = n;
    return 0;
}
