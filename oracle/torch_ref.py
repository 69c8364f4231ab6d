"""CPU oracle: plain-PyTorch fp32 restatement of the InpaintNet hot path.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (inpaintnet_amd/) never
does, and fails loudly when its HIP library is missing.

Parity status: PINNED.  Every function below is checked against golden vectors
captured from the upstream reference itself (oracle/gen_golden.py imports
/root/reference in the build container and writes tests/golden/*.npz);
tests/test_oracle_golden.py is the check.

What is restated (reference file:line each function follows):
  gru_cell / gru_layer / gru_stack  torch.nn.GRU semantics as used at
                                    MeasureVAE/encoder.py:28-35,125; decoder.py:342-367,469,499;
                                    LatentRNN/latent_rnn.py:53-82,186-193,231
  encoder_forward                   MeasureVAE/encoder.py:104-134
  decoder_forward                   MeasureVAE/decoder.py:392-529
  vae_forward                       MeasureVAE/measure_vae.py:97-134
  vae_loss                          MeasureVAE/vae_trainer.py:16-40,128-139; utils/trainer.py:271-306
  adam_step                         torch.optim.Adam as built at utils/trainer.py:32-35 (torch 2.10 form)
  latent_forward                    LatentRNN/latent_rnn.py:110-263; context="past"/"future": LatentRNN/latent_rnn_ablations.py
  vae_forward_test / decode_mid_point  MeasureVAE/measure_vae.py:136-169, MeasureVAE/vae_tester.py:72-93
  arnn_forward_inpaint              AnticipationRNN/anticipation_rnn_gauss_reg_model.py:261-346
  latent_loss                       LatentRNN/latent_rnn_trainer.py:36-67; utils/trainer.py:344-376
  split_score                       LatentRNN/latent_rnn_trainer.py:134-176

Parameters are passed as a dict keyed by the reference's state_dict names
(SURVEY.md App. B), e.g. 'encoder.lstm.weight_ih_l0_reverse'.  Gradients come
from autograd over this explicit arithmetic.

Injection points for the reference's random draws: eps (rsample), the
teacher-forcing coin (an explicit bool), and dropout masks (dict of
pre-scaled {0, 1/(1-p)} tensors; None = no dropout).

Argmax tie rule: lowest index.  (torch-CPU topk used by the reference at
decoder.py:511 is implementation-defined on ties; fixtures only assert rows
with a unique maximum.)
"""
import math

import torch
import torch.nn.functional as F

SELU_ALPHA = 1.6732632423543772
SELU_SCALE = 1.0507009873554805


def selu(x):
    return SELU_SCALE * torch.where(x > 0, x, SELU_ALPHA * (torch.exp(x) - 1.0))


# Kink alignment (test hook).  SELU's derivative jumps at 0 (1.0507 -> 1.7581) and ReLU's from 0 to 1: two fp32
# implementations whose pre-activations differ by 1e-7 disagree by O(1) in the derivative of an element that sits
# within that distance of 0, and a batch of 256 measures has ~2.7 M such elements, so a handful always do.  The
# functions below take an optional boolean `pos` = the branch the implementation under test took (its output > 0) and
# follow it; with pos=None they are the plain functions.  Forward values change by O(|x|) <= 1e-6 at the affected
# elements only; what is gained is that gradients can then be compared at 1e-5 instead of "5e-4 with luck".
#
# The alignment is RESTRICTED to elements whose oracle pre-activation lies within KINK_TOL of 0: everywhere else the oracle
# keeps its own branch, so a kernel that mis-branches a clearly non-zero element is NOT followed and shows up in the
# comparison.  KINK_STATS counts what happened since kink_stats_reset(): `elements` seen, `flips` (elements inside the band
# whose branch was taken from the implementation under test and differs from the oracle's own), `max_abs_flip` (largest
# |x| among them) and `violations` (elements OUTSIDE the band where the implementation's branch differs from the oracle's:
# must be 0; tests and bench.py assert it).
KINK_TOL = 1e-5
KINK_STATS = {"elements": 0, "flips": 0, "max_abs_flip": 0.0, "violations": 0}


def kink_stats_reset():
    KINK_STATS.update(elements=0, flips=0, max_abs_flip=0.0, violations=0)


def _kink_branch(x, pos):
    own = x > 0
    with torch.no_grad():
        near = x.abs() < KINK_TOL
        differ = pos != own
        flips = differ & near
        nf = int(flips.sum())
        KINK_STATS["elements"] += x.numel()
        KINK_STATS["flips"] += nf
        KINK_STATS["violations"] += int((differ & ~near).sum())
        if nf:
            KINK_STATS["max_abs_flip"] = max(KINK_STATS["max_abs_flip"], float(x.detach().abs()[flips].max()))
    return torch.where(near, pos, own)


def selu_k(x, pos=None):
    if pos is None:
        return selu(x)
    return SELU_SCALE * torch.where(_kink_branch(x, pos), x, SELU_ALPHA * (torch.exp(x) - 1.0))


def relu_k(x, pos=None):
    if pos is None:
        return torch.relu(x)
    return torch.where(_kink_branch(x, pos), x, torch.zeros_like(x))


def argmax_first(w):
    """Lowest index among maxima, last dim."""
    m = w.max(dim=-1, keepdim=True).values
    V = w.shape[-1]
    idx = torch.arange(V, device=w.device).expand_as(w)
    return torch.where(w == m, idx, torch.full_like(idx, V)).min(dim=-1).values


# ----------------------------------------------------------------------------
# GRU
# ----------------------------------------------------------------------------
def gru_cell(gi, h, w_hh, b_hh):
    """gi = W_ih x + b_ih already formed. Gate rows ordered [r|z|n]."""
    H = h.shape[-1]
    gh = h @ w_hh.t() + b_hh
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1.0 - z) * n + z * h


def gru_layer(x, h0, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """x (B,T,K) -> out (B,T,H), h_T.  One direction of one layer."""
    B, T, _ = x.shape
    gi_all = x @ w_ih.t() + b_ih
    h = h0
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        h = gru_cell(gi_all[:, t], h, w_hh, b_hh)
        outs[t] = h
    return torch.stack(outs, 1), h


def gru_stack(x, h0, P, prefix, num_layers, bidirectional, masks=None):
    """Multi-layer (bi)GRU, batch_first.  h0: (L*D, B, H).  masks: list of
    length L-1 of pre-scaled dropout masks applied to layer l's output before
    layer l+1 (torch semantics: not after the last layer).  Returns
    (top-layer output (B,T,D*H), h_n (L*D,B,H) ordered [l0f,l0b,l1f,l1b])."""
    D = 2 if bidirectional else 1
    inp = x
    h_n = []
    for l in range(num_layers):
        outs = []
        for d in range(D):
            sfx = f"_l{l}" + ("_reverse" if d == 1 else "")
            o, hT = gru_layer(inp, h0[l * D + d],
                              P[f"{prefix}.weight_ih{sfx}"], P[f"{prefix}.weight_hh{sfx}"],
                              P[f"{prefix}.bias_ih{sfx}"], P[f"{prefix}.bias_hh{sfx}"],
                              reverse=(d == 1))
            outs.append(o)
            h_n.append(hT)
        inp = torch.cat(outs, 2) if D == 2 else outs[0]
        if l < num_layers - 1 and masks is not None and masks[l] is not None:
            inp = inp * masks[l]
    return inp, torch.stack(h_n, 0)


def gru_stack_fast(x, h0, P, prefix, num_layers, bidirectional):
    """Same contraction through torch's fused aten::gru (what nn.GRU calls on
    CPU) -- used only for the cpu_baseline timing leg, eval/no-dropout."""
    D = 2 if bidirectional else 1
    flat = []
    for l in range(num_layers):
        for d in range(D):
            sfx = f"_l{l}" + ("_reverse" if d == 1 else "")
            flat += [P[f"{prefix}.weight_ih{sfx}"], P[f"{prefix}.weight_hh{sfx}"],
                     P[f"{prefix}.bias_ih{sfx}"], P[f"{prefix}.bias_hh{sfx}"]]
    out, hn = torch._VF.gru(x, h0, flat, True, num_layers, 0.0, False, bidirectional, True)
    return out, hn


# ----------------------------------------------------------------------------
# MeasureVAE
# ----------------------------------------------------------------------------
def encoder_forward(P, tokens, masks=None, prefix="encoder", fast=False, kinks=None):
    """tokens (B,T) int64 -> (mu, logsigma) each (B,Z).  encoder.py:104-134.
    kinks: optional {'a_mu','a_ls': (B,2H) bool} branch of the two heads' SELUs (see selu_k)."""
    kinks = kinks or {}
    B = tokens.shape[0]
    emb = P[f"{prefix}.note_embedding_layer.weight"][tokens]
    H = P[f"{prefix}.lstm.weight_hh_l0"].shape[1]
    h0 = torch.zeros(4, B, H, dtype=emb.dtype)
    if fast:
        _, hn = gru_stack_fast(emb, h0, P, f"{prefix}.lstm", 2, True)
    else:
        _, hn = gru_stack(emb, h0, P, f"{prefix}.lstm", 2, True, masks)
    hcat = hn.transpose(0, 1).contiguous().view(B, -1)

    def head(name, kink):
        a = selu_k(hcat @ P[f"{prefix}.{name}.0.weight"].t() + P[f"{prefix}.{name}.0.bias"], kinks.get(kink))
        return a @ P[f"{prefix}.{name}.2.weight"].t() + P[f"{prefix}.{name}.2.bias"]
    return head("linear_mean", "a_mu"), head("linear_log_std", "a_ls")


def decoder_forward(P, z, target, teacher_forced, masks=None, prefix="decoder",
                    beats=4, ticks_per_beat=6, feed_tokens=None, kinks=None):
    """z (B,Z), target (B,T) int64 (used iff teacher_forced) ->
    weights (B,T,V) post-ReLU logits, samples (B,1,T) int64.  decoder.py:412-529.
    masks: {'beat': (B,beats,H) or None, 'tick': (B,T,H) or None}, pre-scaled.
    feed_tokens (B,T): test hook for free-running parity at large batch -- the token fed back after tick t is
    feed_tokens[:, t] (e.g. the samples of the implementation under test) while `samples` still reports this
    function's own argmax, so that a near-tie flip in one row cannot de-synchronise the two trajectories.
    kinks: optional {'hb0': (B,2H), 'ht0': (beats,B,2H), 'c_all': (beats,B,H), 'relu': (B,T,V)} bool branches of the
    three SELU heads and of the output ReLU (see selu_k)."""
    B = z.shape[0]
    H = P[f"{prefix}.rnn_beat.weight_hh_l0"].shape[1]
    masks = masks or {}
    kinks = kinks or {}

    def kink(name, *idx):
        k = kinks.get(name)
        return None if k is None else k[idx]
    # beat rnn (forward_beat_rnn, decoder.py:455-471)
    hb0 = selu_k(z @ P[f"{prefix}.z_to_beat_rnn_input.0.weight"].t() + P[f"{prefix}.z_to_beat_rnn_input.0.bias"],
                 kinks.get("hb0"))
    h_beat = hb0.view(B, 2, H).transpose(0, 1).contiguous()
    beat_in = P[f"{prefix}.b_0"].view(1, 1, 1).expand(B, beats, 1)
    bm = masks.get("beat")
    beat_out, _ = gru_stack(beat_in, h_beat, P, f"{prefix}.rnn_beat", 2, False,
                            [bm] if bm is not None else None)
    # tick rnn (forward_tick_rnn, decoder.py:473-529)
    E = P[f"{prefix}.note_embedding_layer.weight"]
    prev = P[f"{prefix}.x_0"].view(1, -1).expand(B, -1)
    tm = masks.get("tick")
    pf = f"{prefix}.rnn_tick"
    weights, samples = [], []
    for i in range(beats):
        o_i = beat_out[:, i]
        ht0 = selu_k(o_i @ P[f"{prefix}.beat_emb_to_tick_rnn_hidden.0.weight"].t()
                     + P[f"{prefix}.beat_emb_to_tick_rnn_hidden.0.bias"], kink("ht0", i))
        hid = ht0.view(B, 2, H).transpose(0, 1)
        h_l0, h_l1 = hid[0], hid[1]
        c_i = selu_k(o_i @ P[f"{prefix}.beat_emb_to_tick_rnn_input.0.weight"].t()
                     + P[f"{prefix}.beat_emb_to_tick_rnn_input.0.bias"], kink("c_all", i))
        for j in range(ticks_per_beat):
            t = i * ticks_per_beat + j
            u = torch.cat((prev, c_i), 1)
            gi0 = u @ P[f"{pf}.weight_ih_l0"].t() + P[f"{pf}.bias_ih_l0"]
            h_l0 = gru_cell(gi0, h_l0, P[f"{pf}.weight_hh_l0"], P[f"{pf}.bias_hh_l0"])
            x1 = h_l0 if tm is None else h_l0 * tm[:, t]
            gi1 = x1 @ P[f"{pf}.weight_ih_l1"].t() + P[f"{pf}.bias_ih_l1"]
            h_l1 = gru_cell(gi1, h_l1, P[f"{pf}.weight_hh_l1"], P[f"{pf}.bias_hh_l1"])
            w_t = relu_k(h_l1 @ P[f"{prefix}.tick_emb_to_note_emb.0.weight"].t()
                         + P[f"{prefix}.tick_emb_to_note_emb.0.bias"], kink("relu", slice(None), t))
            tok = target[:, t] if teacher_forced else argmax_first(w_t.detach())
            prev = E[tok if feed_tokens is None else feed_tokens[:, t]]
            weights.append(w_t)
            samples.append(tok)
    return torch.stack(weights, 1), torch.stack(samples, 1).unsqueeze(1)


def vae_forward(P, tokens, eps, teacher_forced, masks=None, feed_tokens=None, kinks=None):
    """measure_vae.py:97-134 with eps injected.  Returns weights, samples, mu, logsigma, z."""
    masks = masks or {}
    mu, ls = encoder_forward(P, tokens, [masks.get("enc")] if masks.get("enc") is not None else None, kinks=kinks)
    z = mu + eps * torch.exp(ls)
    w, s = decoder_forward(P, z, tokens, teacher_forced, masks, feed_tokens=feed_tokens, kinks=kinks)
    return w, s, mu, ls, z


def vae_forward_test(P, measures, eps_list):
    """MeasureVAE.forward_test (measure_vae.py:136-169): measures (B,M,T); one rsample per measure (eps_list[i] (B,Z)),
    eval-mode decode of every measure.  -> weights (B,M,T,V), samples (B,1,M*T)."""
    ws, ss = [], []
    for i in range(measures.shape[1]):
        mu, ls = encoder_forward(P, measures[:, i])
        w, smp = decoder_forward(P, mu + eps_list[i] * torch.exp(ls), None, False)
        ws.append(w.unsqueeze(1))
        ss.append(smp)
    return torch.cat(ws, 1), torch.cat(ss, 2)


def decode_mid_point(P, z1, z2, n):
    """VAETester.decode_mid_point (vae_tester.py:72-93): tokens of z1, n interpolated points, z2 -> (1, (n+2)*T);
    also returns the stacked logits (n+2, T, V)."""
    toks, ws = [], []
    for i in range(n + 2):
        w, smp = decoder_forward(P, z1 + (z2 - z1) * i / (n + 1), None, False)
        toks.append(smp[:, 0])
        ws.append(w)
    return torch.cat(toks, 1).view(1, -1), torch.cat(ws, 0)


def cross_entropy_mean(weights, targets):
    V = weights.shape[-1]
    return F.cross_entropy(weights.reshape(-1, V), targets.reshape(-1), reduction="mean")


def accuracy_mean(weights, targets):
    return (argmax_first(weights.reshape(-1, weights.shape[-1])) == targets.reshape(-1)).float().mean()


def kld(mu, ls, beta=1e-3):
    """beta * mean_b sum_d KL(N(mu,sigma)||N(0,1)) = 0.5(s^2+mu^2-1) - ls.  vae_trainer.py:128-139."""
    k = 0.5 * (torch.exp(2.0 * ls) + mu * mu - 1.0) - ls
    return beta * k.sum(1).mean()


def vae_loss(weights, tokens, mu, ls):
    ce = cross_entropy_mean(weights, tokens)
    kl = kld(mu, ls)
    return ce + kl, ce, kl, accuracy_mean(weights.detach(), tokens)


def adam_step(params, grads, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """In-place, torch 2.10 single-tensor form. t is 1-based."""
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    for k in params:
        g = grads[k]
        m[k].mul_(b1).add_(g, alpha=1.0 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-lr / bc1)


# ----------------------------------------------------------------------------
# LatentRNN
# ----------------------------------------------------------------------------
def split_score(score, n_past, n_future, n_target, measure_len=24):
    """(B,1,L) -> past (B,np,24), future (B,nf,24), target (B,nt,24) int64."""
    B = score.shape[0]
    m = score.reshape(B, -1, measure_len)
    n = m.shape[1]
    assert n == n_past + n_future + n_target
    return (m[:, :n_past].long().contiguous(), m[:, n - n_future:].long().contiguous(),
            m[:, n_past:n - n_future].long().contiguous())


def latent_get_z(P, measures, eps, enc_mask=None):
    """get_z_seq, latent_rnn.py:161-174: z SAMPLES from the frozen encoder.  enc_mask: (B*n, T, 2H) pre-scaled
    layer-0 -> layer-1 dropout mask (LatentRNN.train() also puts the frozen VAE in training mode, SURVEY App. C iv)."""
    B, n, T = measures.shape
    mu, ls = encoder_forward(P, measures.reshape(-1, T), [enc_mask] if enc_mask is not None else None,
                             prefix="vae_model.encoder")
    z = mu + eps * torch.exp(ls)
    return z.view(B, n, -1)


def latent_forward(P, past, future, target, eps_p, eps_f, eps_t, auto_reg=False,
                   teacher_forcing=False, eps_ar=None, masks=None, feed_tokens=None, dec_kinks=None, context="both"):
    """latent_rnn.py:110-263.  Returns weights (B,nt,T,V), samples (B,1,nt*T), gen_z (B,nt,Z).
    masks: {'ctx_past','ctx_future','gen': layer0->1 masks; 'dec': list of per-measure decoder masks;
    'enc_past','enc_future','enc_target': encoder masks of the three get_z_seq calls; free-running auto-regressive path:
    'gen' may be a list (one (B,1,4H) mask per generated measure) and 'enc_ar' a list of the re-encoding passes' masks}.
    feed_tokens (B,nt,T): see decoder_forward; on the free-running auto-regressive path they are also what is re-encoded
    (so that a near-tie argmax cannot de-synchronise the trajectory from the implementation under test).  dec_kinks: list (one per generated measure) of decoder_forward `kinks`.
    context: "both" (LatentRNN) | "past" | "future" (LatentRNNAblations, latent_rnn_ablations.py:143-146: the generator
    starts from one context only and has hidden size H instead of 2H)."""
    masks = masks or {}
    B, nt, T = target.shape
    with torch.no_grad():
        zp = latent_get_z(P, past, eps_p, masks.get("enc_past"))
        zf = latent_get_z(P, future, eps_f, masks.get("enc_future"))
        zt = latent_get_z(P, target, eps_t, masks.get("enc_target"))
    H = P["context_rnn_past.weight_hh_l0"].shape[1]
    h0 = torch.zeros(4, B, H)

    def m1(k):
        return [masks[k]] if masks.get(k) is not None else None
    _, cp = gru_stack(zp, h0, P, "context_rnn_past", 2, True, m1("ctx_past"))
    _, cf = gru_stack(zf, h0, P, "context_rnn_future", 2, True, m1("ctx_future"))
    ctx = torch.cat((cp, cf), 2) if context == "both" else (cp if context == "past" else cf)
    dec_masks = masks.get("dec") or [None] * nt

    def decode(zi, i):
        return decoder_forward(P, zi, None, False, dec_masks[i], prefix="vae_model.decoder",
                               feed_tokens=None if feed_tokens is None else feed_tokens[:, i],
                               kinks=None if dec_kinks is None else dec_kinks[i])

    Wg, bg = P["generation_linear.weight"], P["generation_linear.bias"]
    weights, samples = [], []
    if teacher_forcing or not auto_reg:
        if auto_reg:
            gen_in = torch.cat((zp[:, -1:].contiguous(), zt[:, :-1]), 1)
        else:
            gen_in = P["x_0"].expand(B, nt, -1)
        out, _ = gru_stack(gen_in, ctx, P, "generation_rnn", 2, True, m1("gen"))
        gen_z = out.reshape(B * nt, -1) @ Wg.t() + bg
        gen_z = gen_z.view(B, nt, -1)
        for i in range(nt):
            w, s = decode(gen_z[:, i], i)
            weights.append(w)
            samples.append(s)
    else:
        hidden = ctx
        gen_in = zp[:, -1:].contiguous()
        zs = []
        gen_masks, enc_ar = masks.get("gen"), masks.get("enc_ar") or [None] * nt
        for i in range(nt):
            gm = [gen_masks[i]] if isinstance(gen_masks, (list, tuple)) else m1("gen")
            out, hidden = gru_stack(gen_in, hidden, P, "generation_rnn", 2, True, gm)
            gz = out.reshape(B, -1) @ Wg.t() + bg
            zs.append(gz)
            w, s = decode(gz, i)
            weights.append(w)
            samples.append(s)
            with torch.no_grad():
                s_in = s if feed_tokens is None else feed_tokens[:, i].reshape(B, 1, T)
                gen_in = latent_get_z(P, s_in, eps_ar[i], enc_ar[i])
        gen_z = torch.stack(zs, 1)
    return torch.stack(weights, 1), torch.cat(samples, 2), gen_z


def latent_loss(weights, target):
    V = weights.shape[-1]
    ce = F.cross_entropy(weights.reshape(-1, V), target.reshape(-1), reduction="mean")
    return ce, accuracy_mean(weights.detach(), target)


# ----------------------------------------------------------------------------
# CPU baseline leg (bench.py cpu_baseline, kind "port"): the same training step
# expressed the way the reference runs it on CPU -- fused aten::gru calls (what
# nn.GRU dispatches to), one call per tick in the decoder, dropout active,
# autograd backward, torch.optim.Adam.  Checked against the explicit restatement
# above in tests/test_oracle_golden.py (dropout off).
# ----------------------------------------------------------------------------
def _flat_gru(P, prefix, num_layers, bidirectional):
    D = 2 if bidirectional else 1
    flat = []
    for l in range(num_layers):
        for d in range(D):
            sfx = f"_l{l}" + ("_reverse" if d == 1 else "")
            flat += [P[f"{prefix}.weight_ih{sfx}"], P[f"{prefix}.weight_hh{sfx}"],
                     P[f"{prefix}.bias_ih{sfx}"], P[f"{prefix}.bias_hh{sfx}"]]
    return flat


def vae_forward_fast(P, tokens, eps, teacher_forced, dropout=0.0, train=False):
    B, T = tokens.shape
    He = P["encoder.lstm.weight_hh_l0"].shape[1]
    emb = P["encoder.note_embedding_layer.weight"][tokens]
    _, hn = torch._VF.gru(emb, torch.zeros(4, B, He), _flat_gru(P, "encoder.lstm", 2, True), True, 2, dropout, train,
                          True, True)
    hcat = hn.transpose(0, 1).contiguous().view(B, -1)

    def head(name):
        a = F.selu(F.linear(hcat, P[f"encoder.{name}.0.weight"], P[f"encoder.{name}.0.bias"]))
        return F.linear(a, P[f"encoder.{name}.2.weight"], P[f"encoder.{name}.2.bias"])
    mu, ls = head("linear_mean"), head("linear_log_std")
    z = mu + eps * torch.exp(ls)
    H = P["decoder.rnn_beat.weight_hh_l0"].shape[1]
    hb0 = F.selu(F.linear(z, P["decoder.z_to_beat_rnn_input.0.weight"], P["decoder.z_to_beat_rnn_input.0.bias"]))
    h_beat = hb0.view(B, 2, H).transpose(0, 1).contiguous()
    beat_in = P["decoder.b_0"].view(1, 1, 1).expand(B, 4, 1)
    beat_out, _ = torch._VF.gru(beat_in, h_beat, _flat_gru(P, "decoder.rnn_beat", 2, False), True, 2, dropout, train,
                                False, True)
    tick_w = _flat_gru(P, "decoder.rnn_tick", 2, False)
    E = P["decoder.note_embedding_layer.weight"]
    prev = P["decoder.x_0"].view(1, 1, -1).expand(B, 1, -1)
    weights = []
    for i in range(4):
        o_i = beat_out[:, i]
        hid = F.selu(F.linear(o_i, P["decoder.beat_emb_to_tick_rnn_hidden.0.weight"],
                              P["decoder.beat_emb_to_tick_rnn_hidden.0.bias"])).view(B, 2, H).transpose(0, 1).contiguous()
        c_i = F.selu(F.linear(o_i, P["decoder.beat_emb_to_tick_rnn_input.0.weight"],
                              P["decoder.beat_emb_to_tick_rnn_input.0.bias"])).unsqueeze(1)
        for j in range(6):
            out, hid = torch._VF.gru(torch.cat((prev, c_i), 2), hid, tick_w, True, 2, dropout, train, False, True)
            w_t = torch.relu(F.linear(out[:, 0], P["decoder.tick_emb_to_note_emb.0.weight"],
                                      P["decoder.tick_emb_to_note_emb.0.bias"]))
            tok = tokens[:, i * 6 + j] if teacher_forced else w_t.detach().argmax(1)
            prev = E[tok].unsqueeze(1)
            weights.append(w_t)
    return torch.stack(weights, 1), mu, ls


class CpuVaeTrainStep:
    """fwd + CE + KL + bwd + Adam of the MeasureVAE on host cores (baseline only)."""

    def __init__(self, P, lr=1e-4, dropout=0.5):
        self.P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        self.opt = torch.optim.Adam(list(self.P.values()), lr=lr)
        self.dropout = dropout

    def step(self, tokens, teacher_forced):
        self.opt.zero_grad()
        eps = torch.randn(tokens.shape[0], self.P["encoder.linear_mean.2.bias"].shape[0])
        w, mu, ls = vae_forward_fast(self.P, tokens, eps, teacher_forced, self.dropout, True)
        loss, ce, kl, acc = vae_loss(w, tokens, mu, ls)
        loss.backward()
        self.opt.step()
        return float(loss.detach())


def decoder_forward_fast(P, z, tokens, teacher_forced, dropout=0.0, train=False, prefix="decoder"):
    """HierarchicalDecoder.forward through fused aten::gru calls (the decoder half of vae_forward_fast)."""
    B = z.shape[0]
    H = P[f"{prefix}.rnn_beat.weight_hh_l0"].shape[1]
    hb0 = F.selu(F.linear(z, P[f"{prefix}.z_to_beat_rnn_input.0.weight"], P[f"{prefix}.z_to_beat_rnn_input.0.bias"]))
    h_beat = hb0.view(B, 2, H).transpose(0, 1).contiguous()
    beat_in = P[f"{prefix}.b_0"].view(1, 1, 1).expand(B, 4, 1)
    beat_out, _ = torch._VF.gru(beat_in, h_beat, _flat_gru(P, f"{prefix}.rnn_beat", 2, False), True, 2, dropout, train,
                                False, True)
    tick_w = _flat_gru(P, f"{prefix}.rnn_tick", 2, False)
    E = P[f"{prefix}.note_embedding_layer.weight"]
    prev = P[f"{prefix}.x_0"].view(1, 1, -1).expand(B, 1, -1)
    weights = []
    for i in range(4):
        o_i = beat_out[:, i]
        hid = F.selu(F.linear(o_i, P[f"{prefix}.beat_emb_to_tick_rnn_hidden.0.weight"],
                              P[f"{prefix}.beat_emb_to_tick_rnn_hidden.0.bias"])).view(B, 2, H).transpose(0, 1).contiguous()
        c_i = F.selu(F.linear(o_i, P[f"{prefix}.beat_emb_to_tick_rnn_input.0.weight"],
                              P[f"{prefix}.beat_emb_to_tick_rnn_input.0.bias"])).unsqueeze(1)
        for j in range(6):
            out, hid = torch._VF.gru(torch.cat((prev, c_i), 2), hid, tick_w, True, 2, dropout, train, False, True)
            w_t = torch.relu(F.linear(out[:, 0], P[f"{prefix}.tick_emb_to_note_emb.0.weight"],
                                      P[f"{prefix}.tick_emb_to_note_emb.0.bias"]))
            tok = tokens[:, i * 6 + j] if teacher_forced else w_t.detach().argmax(1)
            prev = E[tok].unsqueeze(1)
            weights.append(w_t)
    return torch.stack(weights, 1)


class CpuLatentTrainStep:
    """LatentRNN (non-AR) training step with the frozen MeasureVAE on host cores -- BASELINE.json configs[2], baseline
    only: fused aten::gru everywhere (what nn.GRU dispatches to), 16 measures encoded per sequence, split 6/4/6."""

    def __init__(self, num_notes, dropout=0.5, lr=1e-4):
        from inpaintnet_amd import layout, synthetic
        self.V = num_notes
        shapes = layout.vae_param_shapes(num_notes, prefix="vae_model.")
        self.Pv = {k: torch.from_numpy(synthetic.det_param(k, s)) for k, s in shapes.items()}
        self.P = {k: torch.from_numpy(synthetic.det_param(k, s)).requires_grad_(True)
                  for k, s in layout.latent_param_shapes(256, 512, False).items()}
        self.opt = torch.optim.Adam(list(self.P.values()), lr=lr)
        self.dropout = dropout

    def step(self, n_seq):
        from inpaintnet_amd import synthetic
        P, Pv, dr = self.P, self.Pv, self.dropout
        score = torch.from_numpy(synthetic.folk_score(n_seq, self.V, seed=9))
        past, future, target = split_score(score, 6, 6, 4)
        self.opt.zero_grad()
        with torch.no_grad():
            allm = torch.cat((past, target, future), 1).reshape(-1, 24)
            emb = Pv["vae_model.encoder.note_embedding_layer.weight"][allm]
            _, hn = torch._VF.gru(emb, torch.zeros(4, allm.shape[0], 512), _flat_gru(Pv, "vae_model.encoder.lstm", 2, True),
                                  True, 2, dr, True, True, True)
            hcat = hn.transpose(0, 1).contiguous().view(allm.shape[0], -1)
            head = lambda n: F.linear(F.selu(F.linear(hcat, Pv[f"vae_model.encoder.{n}.0.weight"], Pv[f"vae_model.encoder.{n}.0.bias"])),
                                      Pv[f"vae_model.encoder.{n}.2.weight"], Pv[f"vae_model.encoder.{n}.2.bias"])
            mu, ls = head("linear_mean"), head("linear_log_std")
            z = (mu + torch.randn_like(mu) * torch.exp(ls)).view(n_seq, 16, -1)
        zp, zf = z[:, :6], z[:, 10:]
        h0 = torch.zeros(4, n_seq, 512)
        _, cp = torch._VF.gru(zp, h0, _flat_gru(P, "context_rnn_past", 2, True), True, 2, dr, True, True, True)
        _, cf = torch._VF.gru(zf, h0, _flat_gru(P, "context_rnn_future", 2, True), True, 2, dr, True, True, True)
        out, _ = torch._VF.gru(P["x_0"].expand(n_seq, 4, -1), torch.cat((cp, cf), 2), _flat_gru(P, "generation_rnn", 2, True),
                               True, 2, dr, True, True, True)
        gen_z = F.linear(out.reshape(n_seq * 4, -1), P["generation_linear.weight"], P["generation_linear.bias"])
        w = decoder_forward_fast(Pv, gen_z, None, False, dr, True, prefix="vae_model.decoder")
        loss = F.cross_entropy(w.reshape(-1, self.V), target.reshape(-1))
        loss.backward()
        self.opt.step()
        return float(loss.detach())


class CpuArnnTrainStep:
    """AnticipationRNN teacher-forced training step on host cores -- BASELINE.json configs[4], baseline only (fused
    aten::lstm, script defaults of train_arnn_reg.py: 2+2 layers, H = 256)."""

    def __init__(self, num_notes, lr=1e-4):
        from inpaintnet_amd import layout, synthetic
        self.V = num_notes
        shapes = layout.arnn_param_shapes(num_notes, 10, 2, 256, 256, 2)
        self.P = {k: torch.from_numpy(synthetic.det_param(k, s)).requires_grad_(True) for k, s in shapes.items()}
        self.opt = torch.optim.Adam(list(self.P.values()), lr=lr)

    def _lstm(self, x, prefix, reverse=False):
        P = self.P
        w = [P[f"{prefix}.weight_ih_l0"], P[f"{prefix}.weight_hh_l0"], P[f"{prefix}.bias_ih_l0"], P[f"{prefix}.bias_hh_l0"]]
        z = torch.zeros(1, x.shape[0], 256)
        if reverse:
            x = x.flip(1)
        out, _, _ = torch._VF.lstm(x, (z, z), w, True, 1, 0.0, True, False, True)
        return out.flip(1) if reverse else out

    def step(self, n_seq):
        from inpaintnet_amd import synthetic
        P = self.P
        score = torch.from_numpy(synthetic.folk_score(n_seq, self.V, seed=21)).long()
        md = torch.from_numpy(synthetic.folk_metadata(n_seq)).long()
        loc = torch.zeros_like(score)
        loc[:, :, :7 * 24] = 1
        loc[:, :, 11 * 24:] = 1
        self.opt.zero_grad()
        x, m = arnn_embed(P, score, md, loc)
        oc = m
        for l in range(2):
            oc = self._lstm(oc, f"lstm_constraint.{l}", reverse=True)
        h = torch.cat((torch.cat((torch.zeros(n_seq, 1, x.shape[2]), x[:, :-1]), 1), oc), 2)
        for l in range(2):
            h = self._lstm(h, f"lstm_generation.{l}")
        w = _arnn_head(P, h)[:, 7 * 24:11 * 24]
        loss = F.cross_entropy(w.reshape(-1, self.V), score[:, 0, 7 * 24:11 * 24].reshape(-1))
        loss.backward()
        self.opt.step()
        return float(loss.detach())


# ----------------------------------------------------------------------------
# AnticipationRNN (config 5): AnticipationRNN/anticipation_rnn_gauss_reg_model.py
#   lstm_cell / lstm_layer   torch.nn.LSTM(num_layers=1, batch_first) as stacked by lstm_with_activations (:14-39)
#   arnn_embed               embed_tensor_score (:448-457), embed_metadata (:478-510), mask_tensor_score (:512-532)
#   arnn_forward             _forward_tf (:348-404), _forward_no_tf (:190-259), output_lstm_constraints (:459-476)
#   arnn_loss                anticipation_rnn_trainer.py:21-49,154-182 (single voice)
# Single-voice datasets only (FolkDataset: num_voices = 1), as everything the reference trains on.
# ----------------------------------------------------------------------------
def lstm_cell(gi, h, c, w_hh, b_hh):
    """gi = W_ih x + b_ih.  Gate rows ordered [i|f|g|o] (torch)."""
    H = h.shape[-1]
    g = gi + h @ w_hh.t() + b_hh
    i = torch.sigmoid(g[..., :H])
    f = torch.sigmoid(g[..., H:2 * H])
    gg = torch.tanh(g[..., 2 * H:3 * H])
    o = torch.sigmoid(g[..., 3 * H:])
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


def lstm_layer(x, h0, c0, P, prefix, reverse=False):
    """x (B,T,K) -> out (B,T,H), (h_T, c_T).  `reverse` processes t = T-1..0 and returns outputs in original order
    (= index_select(reversed) -> LSTM -> index_select(reversed), output_lstm_constraints :467-474)."""
    B, T, _ = x.shape
    gi = x @ P[f"{prefix}.weight_ih_l0"].t() + P[f"{prefix}.bias_ih_l0"]
    h, c = h0, c0
    outs = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        h, c = lstm_cell(gi[:, t], h, c, P[f"{prefix}.weight_hh_l0"], P[f"{prefix}.bias_hh_l0"])
        outs[t] = h
    return torch.stack(outs, 1), (h, c)


def arnn_embed(P, score, metadata, constraints_loc):
    """score (B,1,L), metadata (B,1,L,3), constraints_loc (B,1,L) in {0,1} ->
    x (B,L,E) note embeddings, m (B,L,3*Em+E) constraint-LSTM input."""
    V = P["note_embeddings.0.weight"].shape[0] - 1
    tok = score[:, 0]
    x = P["note_embeddings.0.weight"][tok]
    md = metadata[:, 0]
    parts = [P[f"metadata_embeddings.{i}.weight"][md[..., i]] for i in range(md.shape[-1])]
    loc = constraints_loc[:, 0]
    masked = tok * loc + V * (1 - loc)
    parts.append(P["note_embeddings.0.weight"][masked])
    return x, torch.cat(parts, 2)


def _arnn_head(P, h):
    a = torch.relu(h @ P["linear_1.weight"].t() + P["linear_1.bias"])
    return a @ P["linear_ouput_notes.0.weight"].t() + P["linear_ouput_notes.0.bias"]


def arnn_forward(P, score, metadata, constraints_loc, teacher_forcing, num_layers=2, input_mask=None):
    """-> weights (B,L,V) for ALL ticks (the caller slices the unconstrained ones, forward :434), gen (B,L) tokens
    (no-teacher-forcing path only).  input_mask: (B,L,1) pre-scaled Dropout2d mask on the shifted note embeddings."""
    B, _, L = score.shape
    x, m = arnn_embed(P, score, metadata, constraints_loc)
    Hc = P["lstm_constraint.0.weight_hh_l0"].shape[1]
    z = torch.zeros(B, Hc)
    oc = m
    for l in range(num_layers):
        oc, _ = lstm_layer(oc, z, z, P, f"lstm_constraint.{l}", reverse=True)
    Hg = P["lstm_generation.0.weight_hh_l0"].shape[1]
    if teacher_forcing:
        off = torch.cat((torch.zeros(B, 1, x.shape[2]), x[:, :L - 1]), 1)
        if input_mask is not None:
            off = off * input_mask
        h = torch.cat((off, oc), 2)
        zg = torch.zeros(B, Hg)
        for l in range(num_layers):
            h, _ = lstm_layer(h, zg, zg, P, f"lstm_generation.{l}")
        return _arnn_head(P, h), None
    # free running: the argmax of BATCH ELEMENT 0 is written to the whole batch (:253-256)
    hs = [torch.zeros(B, Hg) for _ in range(num_layers)]
    cs = [torch.zeros(B, Hg) for _ in range(num_layers)]
    E = P["note_embeddings.0.weight"]
    prev = torch.zeros(B, dtype=torch.long)            # start symbol 0 (:218-223)
    ws, gen = [], []
    for t in range(L):
        inp = torch.cat((E[prev], oc[:, t]), 1)
        for l in range(num_layers):
            pf = f"lstm_generation.{l}"
            gi = inp @ P[f"{pf}.weight_ih_l0"].t() + P[f"{pf}.bias_ih_l0"]
            hs[l], cs[l] = lstm_cell(gi, hs[l], cs[l], P[f"{pf}.weight_hh_l0"], P[f"{pf}.bias_hh_l0"])
            inp = hs[l]
        w = _arnn_head(P, inp)
        ws.append(w)
        tok = argmax_first(w[0].detach())
        prev = tok.expand(B).clone()
        gen.append(prev)
    return torch.stack(ws, 1), torch.stack(gen, 1)


def arnn_forward_inpaint(P, score, metadata, constraints_loc, start_tick, end_tick, num_layers=2):
    """forward_inpaint (anticipation_rnn_gauss_reg_model.py:261-346), eval mode: the generation LSTMs consume the ground
    truth up to start_tick in one pass, then ticks start_tick..end_tick-1 are generated one by one from the argmax of batch
    element 0.  -> weights (B, end-start, V), gen (B,1,L)."""
    B, _, L = score.shape
    x, m = arnn_embed(P, score, metadata, constraints_loc)
    Hc = P["lstm_constraint.0.weight_hh_l0"].shape[1]
    z = torch.zeros(B, Hc)
    oc = m
    for l in range(num_layers):
        oc, _ = lstm_layer(oc, z, z, P, f"lstm_constraint.{l}", reverse=True)
    Hg = P["lstm_generation.0.weight_hh_l0"].shape[1]
    hs = [torch.zeros(B, Hg) for _ in range(num_layers)]
    cs = [torch.zeros(B, Hg) for _ in range(num_layers)]
    gen = torch.zeros_like(score)
    gen[:, :, :start_tick] = score[:, :, :start_tick]
    gen[:, :, end_tick:] = score[:, :, end_tick:]
    if start_tick > 0:
        off = torch.cat((torch.zeros(B, 1, x.shape[2]), x[:, :L - 1]), 1)
        h = torch.cat((off, oc), 2)[:, :start_tick]
        for l in range(num_layers):
            h, (hs[l], cs[l]) = lstm_layer(h, hs[l], cs[l], P, f"lstm_generation.{l}")
    E = P["note_embeddings.0.weight"]
    ws = []
    for tick in range(start_tick - 1, end_tick - 1):
        prev = gen[:, 0, tick] if tick >= 0 else torch.zeros(B, dtype=torch.long)
        inp = torch.cat((E[prev], oc[:, tick + 1]), 1)
        for l in range(num_layers):
            pf = f"lstm_generation.{l}"
            gi = inp @ P[f"{pf}.weight_ih_l0"].t() + P[f"{pf}.bias_ih_l0"]
            hs[l], cs[l] = lstm_cell(gi, hs[l], cs[l], P[f"{pf}.weight_hh_l0"], P[f"{pf}.bias_hh_l0"])
            inp = hs[l]
        w = _arnn_head(P, inp)
        ws.append(w)
        gen[:, 0, tick + 1] = argmax_first(w[0].detach())
    return torch.stack(ws, 1), gen


def arnn_loss(weights_free, targets_free):
    """mean CE / accuracy over (B, n_free) rows (anticipation_rnn_trainer.py:154-182, one voice)."""
    V = weights_free.shape[-1]
    ce = F.cross_entropy(weights_free.reshape(-1, V), targets_free.reshape(-1), reduction="mean")
    return ce, accuracy_mean(weights_free.detach(), targets_free)
