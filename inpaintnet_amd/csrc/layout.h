// Parameter-arena layout tables (host side).  Mirrors inpaintnet_amd/layout.py
// and the reference's state_dict() order; tests/test_layout.py
// holds the two against each other.
#pragma once
#include <string>
#include <vector>
#include "../../include/inpaintnet_hip.h"

struct ParamEntry {
    std::string name;
    int64_t offset;       // in floats, multiple of 4
    int64_t dims[4];
    int ndim;
    int64_t numel() const { int64_t n = 1; for (int i = 0; i < ndim; ++i) n *= dims[i]; return n; }
};

struct GruDirOff { int64_t w_ih, w_hh, b_ih, b_hh; int K; };

struct ArenaBuilder {
    std::vector<ParamEntry> entries;
    int64_t total = 0;
    int64_t add(const std::string& name, std::initializer_list<int64_t> dims) {
        ParamEntry e;
        e.name = name; e.offset = total; e.ndim = (int)dims.size();
        int i = 0; for (auto d : dims) e.dims[i++] = d;
        for (; i < 4; ++i) e.dims[i] = 1;
        entries.push_back(e);
        total += (e.numel() + 3) / 4 * 4;
        return e.offset;
    }
    // nn.GRU parameter block in state_dict order: for each layer, for each direction: w_ih, w_hh, b_ih, b_hh
    void add_gru(const std::string& prefix, int in0, int H, int layers, bool bidir, GruDirOff* out /*[layers][dirs]*/) {
        const int D = bidir ? 2 : 1;
        for (int l = 0; l < layers; ++l) {
            const int K = l == 0 ? in0 : H * D;
            for (int d = 0; d < D; ++d) {
                const std::string sfx = "_l" + std::to_string(l) + (d ? "_reverse" : "");
                GruDirOff& g = out[l * D + d];
                g.K = K;
                g.w_ih = add(prefix + ".weight_ih" + sfx, {3 * (int64_t)H, K});
                g.w_hh = add(prefix + ".weight_hh" + sfx, {3 * (int64_t)H, H});
                g.b_ih = add(prefix + ".bias_ih" + sfx, {3 * (int64_t)H});
                g.b_hh = add(prefix + ".bias_hh" + sfx, {3 * (int64_t)H});
            }
        }
    }
};

struct VaeLayout {
    ArenaBuilder ab;
    GruDirOff enc[4];                 // [layer*2 + dir]
    int64_t enc_emb;
    int64_t mean_w0, mean_b0, mean_w2, mean_b2, ls_w0, ls_b0, ls_w2, ls_b2;
    int64_t b_0, x_0, dec_emb, zb_w, zb_b;
    GruDirOff beat[2];
    int64_t bh_w, bh_b, bi_w, bi_b;
    GruDirOff tick[2];
    int64_t out_w, out_b;

    explicit VaeLayout(const inet_vae_config& c) {
        const int64_t V = c.num_notes, E = c.emb_dim, H = c.enc_hidden, Z = c.z_dim, Hd = c.dec_hidden;
        ab.add_gru("encoder.lstm", (int)E, (int)H, 2, true, enc);
        enc_emb = ab.add("encoder.note_embedding_layer.weight", {V, E});
        mean_w0 = ab.add("encoder.linear_mean.0.weight", {2 * H, 4 * H});
        mean_b0 = ab.add("encoder.linear_mean.0.bias", {2 * H});
        mean_w2 = ab.add("encoder.linear_mean.2.weight", {Z, 2 * H});
        mean_b2 = ab.add("encoder.linear_mean.2.bias", {Z});
        ls_w0 = ab.add("encoder.linear_log_std.0.weight", {2 * H, 4 * H});
        ls_b0 = ab.add("encoder.linear_log_std.0.bias", {2 * H});
        ls_w2 = ab.add("encoder.linear_log_std.2.weight", {Z, 2 * H});
        ls_b2 = ab.add("encoder.linear_log_std.2.bias", {Z});
        b_0 = ab.add("decoder.b_0", {1});
        x_0 = ab.add("decoder.x_0", {E});
        dec_emb = ab.add("decoder.note_embedding_layer.weight", {V, E});
        zb_w = ab.add("decoder.z_to_beat_rnn_input.0.weight", {2 * Hd, Z});
        zb_b = ab.add("decoder.z_to_beat_rnn_input.0.bias", {2 * Hd});
        ab.add_gru("decoder.rnn_beat", 1, (int)Hd, 2, false, beat);
        bh_w = ab.add("decoder.beat_emb_to_tick_rnn_hidden.0.weight", {2 * Hd, Hd});
        bh_b = ab.add("decoder.beat_emb_to_tick_rnn_hidden.0.bias", {2 * Hd});
        bi_w = ab.add("decoder.beat_emb_to_tick_rnn_input.0.weight", {Hd, Hd});
        bi_b = ab.add("decoder.beat_emb_to_tick_rnn_input.0.bias", {Hd});
        ab.add_gru("decoder.rnn_tick", (int)(E + Hd), (int)Hd, 2, false, tick);
        out_w = ab.add("decoder.tick_emb_to_note_emb.0.weight", {V, Hd});
        out_b = ab.add("decoder.tick_emb_to_note_emb.0.bias", {V});
    }
};

struct LatentLayout {
    ArenaBuilder ab;
    int64_t x_0 = -1;
    GruDirOff past[4], future[4], gen[4];
    int64_t lin_w, lin_b;
    explicit LatentLayout(const inet_latent_config& c) {
        const int64_t Z = c.z_dim, H = c.rnn_hidden;
        if (!c.auto_reg) x_0 = ab.add("x_0", {1, 1, 1});
        ab.add_gru("context_rnn_past", (int)Z, (int)H, 2, true, past);
        ab.add_gru("context_rnn_future", (int)Z, (int)H, 2, true, future);
        ab.add_gru("generation_rnn", c.auto_reg ? (int)Z : 1, (int)(2 * H), 2, true, gen);
        lin_w = ab.add("generation_linear.weight", {Z, 4 * H});
        lin_b = ab.add("generation_linear.bias", {Z});
    }
};
