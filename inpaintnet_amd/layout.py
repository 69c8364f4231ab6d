"""Flat fp32 parameter arena layout.

One contiguous fp32 buffer holds every parameter of a model, in the order and
with the names/shapes of the reference's ``state_dict()`` (SURVEY.md App. B;
MeasureVAE/encoder.py:28-52, MeasureVAE/decoder.py:335-372,
LatentRNN/latent_rnn.py:53-83) so reference checkpoints load key-for-key.
Gradients and the two Adam moments use arenas of the identical layout, which is
what makes the optimizer one kernel and the data-parallel exchange one
all-reduce.

The same table is compiled into the C-ABI library (inet_vae_param_* /
inet_latent_param_* in include/inpaintnet_hip.h); tests/test_layout.py checks
that both sides agree entry by entry.  Every tensor starts on a 16-byte
boundary (offsets are multiples of 4 floats).
"""
from collections import OrderedDict


def _gru(prefix, in0, hidden, layers, bidirectional):
    out = []
    dirs = ["", "_reverse"] if bidirectional else [""]
    D = len(dirs)
    for l in range(layers):
        k_in = in0 if l == 0 else hidden * D
        for d in dirs:
            sfx = f"_l{l}{d}"
            out.append((f"{prefix}.weight_ih{sfx}", (3 * hidden, k_in)))
            out.append((f"{prefix}.weight_hh{sfx}", (3 * hidden, hidden)))
            out.append((f"{prefix}.bias_ih{sfx}", (3 * hidden,)))
            out.append((f"{prefix}.bias_hh{sfx}", (3 * hidden,)))
    return out


def vae_param_shapes(num_notes, emb_dim=10, enc_hidden=512, z_dim=256, dec_hidden=512, prefix=""):
    V, E, H, Z, Hd = num_notes, emb_dim, enc_hidden, z_dim, dec_hidden
    p = []
    e = prefix + "encoder"
    p += _gru(e + ".lstm", E, H, 2, True)
    p.append((e + ".note_embedding_layer.weight", (V, E)))
    for head in ("linear_mean", "linear_log_std"):
        p.append((f"{e}.{head}.0.weight", (2 * H, 4 * H)))
        p.append((f"{e}.{head}.0.bias", (2 * H,)))
        p.append((f"{e}.{head}.2.weight", (Z, 2 * H)))
        p.append((f"{e}.{head}.2.bias", (Z,)))
    d = prefix + "decoder"
    p.append((d + ".b_0", (1,)))
    p.append((d + ".x_0", (E,)))
    p.append((d + ".note_embedding_layer.weight", (V, E)))
    p.append((d + ".z_to_beat_rnn_input.0.weight", (2 * Hd, Z)))
    p.append((d + ".z_to_beat_rnn_input.0.bias", (2 * Hd,)))
    p += _gru(d + ".rnn_beat", 1, Hd, 2, False)
    p.append((d + ".beat_emb_to_tick_rnn_hidden.0.weight", (2 * Hd, Hd)))
    p.append((d + ".beat_emb_to_tick_rnn_hidden.0.bias", (2 * Hd,)))
    p.append((d + ".beat_emb_to_tick_rnn_input.0.weight", (Hd, Hd)))
    p.append((d + ".beat_emb_to_tick_rnn_input.0.bias", (Hd,)))
    p += _gru(d + ".rnn_tick", E + Hd, Hd, 2, False)
    p.append((d + ".tick_emb_to_note_emb.0.weight", (V, Hd)))
    p.append((d + ".tick_emb_to_note_emb.0.bias", (V,)))
    return OrderedDict(p)


def latent_param_shapes(z_dim=256, rnn_hidden=512, auto_reg=False, gen_hidden=None):
    """Trainable LatentRNN parameters only (the frozen VAE lives in its own
    arena and appears in state_dict() under 'vae_model.').  latent_rnn.py:53-83.
    gen_hidden: hidden size of the generation GRU -- 2*rnn_hidden for the LatentRNN (its initial state is the
    concatenation of both contexts), rnn_hidden for the past-only / future-only ablations
    (latent_rnn_ablations.py:77-85)."""
    Z, H = z_dim, rnn_hidden
    G = 2 * H if gen_hidden is None else gen_hidden
    p = []
    if not auto_reg:
        p.append(("x_0", (1, 1, 1)))
    p += _gru("context_rnn_past", Z, H, 2, True)
    p += _gru("context_rnn_future", Z, H, 2, True)
    p += _gru("generation_rnn", Z if auto_reg else 1, G, 2, True)
    p.append(("generation_linear.weight", (Z, 2 * G)))
    p.append(("generation_linear.bias", (Z,)))
    return OrderedDict(p)


def numel(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


def arena_offsets(shapes):
    """-> (OrderedDict name -> (offset_floats, shape), total_floats). 16-byte aligned."""
    off = 0
    out = OrderedDict()
    for k, s in shapes.items():
        out[k] = (off, tuple(s))
        off += (numel(s) + 3) // 4 * 4
    return out, off


def arnn_param_shapes(num_notes, note_emb=10, meta_emb=2, hidden=256, linear_hidden=256, num_layers=2,
                      meta_values=(6, 6, 1)):
    """ConstraintModelGaussianReg state_dict (anticipation_rnn_gauss_reg_model.py:72-140), single voice,
    unary_constraint=True: the note embedding has one extra 'no constraint' row."""
    V, E, Em, H = num_notes, note_emb, meta_emb, hidden
    p = [("note_embeddings.0.weight", (V + 1, E))]
    for i, n in enumerate(meta_values):
        p.append((f"metadata_embeddings.{i}.weight", (n, Em)))

    def lstm(prefix, k_in):
        return [(f"{prefix}.weight_ih_l0", (4 * H, k_in)), (f"{prefix}.weight_hh_l0", (4 * H, H)),
                (f"{prefix}.bias_ih_l0", (4 * H,)), (f"{prefix}.bias_hh_l0", (4 * H,))]
    for l in range(num_layers):
        p += lstm(f"lstm_constraint.{l}", Em * len(meta_values) + E if l == 0 else H)
    for l in range(num_layers):
        p += lstm(f"lstm_generation.{l}", E + H if l == 0 else H)
    p.append(("linear_1.weight", (linear_hidden, H)))
    p.append(("linear_1.bias", (linear_hidden,)))
    p.append(("linear_ouput_notes.0.weight", (V, linear_hidden)))
    p.append(("linear_ouput_notes.0.bias", (V,)))
    return OrderedDict(p)
