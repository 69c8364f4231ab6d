"""Parity at the sizes bench.py runs ("What's weak" 1): the kernel instantiations the headline number
spends its time in -- gru_step_fwd_kernel<*,4,true>, gru_step_bwd_kernel<4,1,true> / <4,2,true>, the 192-row GEMM
tiles -- are only selected at B >= 256 (H=512) / big M,K, so they get their own oracle comparisons here:

  (i)   one MeasureVAE training step at B=256, H=512, V=48 with mask-in dropout, teacher-forced and free-running,
        against oracle/torch_ref.py (loss 1e-4 rel, logits 2e-5, every gradient tensor 1e-4 of its max);
  (ii)  LatentRNN non-auto-regressive step at 128 sequences x 6/4/6 measures (encoder batch 2048, generator H=1024,
        decoder batch 512) with dropout on, masks recorded from the product's own stream and replayed in the oracle;
  (iii) every GEMM tile configuration x split-K forced over the big shapes of the step;
  (iv)  the per-launch profile labels prove that the MS=4 / NC=2 / 192-tile instantiations really ran in (i).

Kink alignment: SELU's derivative jumps at 0 and ReLU's from 0 to 1, and a batch this size always has a few of its
~2.7 M pre-activations within fp32 noise of 0, where two correct implementations pick different branches and then
disagree by up to 1e-2 of a small gradient tensor's max (measured: tools/parity_diag.py, profiles/r02_parity_diag.txt).
The tests therefore read the branch each element took on the GPU (logits > 0; SELU outputs through the
inet_vae_ws_field test hook) and make the oracle follow it (oracle `kinks`), which changes its forward values by
<= 1e-6 at those elements and nothing else; the gradient comparison is then asserted at 1e-4, not 5e-4.

Free-running decodes feed the oracle the product's sampled tokens (oracle `feed_tokens`), so a near-tie argmax in one
of the 6144 rows cannot de-synchronise the two trajectories; the oracle's own argmax is then compared with the
product's samples on every row whose top-2 margin exceeds 1e-4 (bit-exact there).
"""
import csv

import numpy as np
import pytest
import torch

from oracle import torch_ref as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from inpaintnet_amd import ops
    from tests.test_gpu_kernels import _assert_kinks, _vae_step_with_kinks, pack, relmax, unpack

DEV = "cuda:0"


def _margin(w):
    top2 = torch.topk(w, 2, dim=-1).values
    return (top2[..., 0] - top2[..., 1]).numpy()


def _comparable_rows(w, what):
    """Rows whose sampled token must agree bit for bit: top-2 margin above 1e-4 OF THE ROW'S OWN MAXIMUM (the logits agree
    within 2e-5 of the tensor's maximum, so a row is only excluded when its two best logits are closer than the arithmetic can
    tell apart -- in particular the rows whose two best post-ReLU logits are both exactly 0).  Prints the fraction compared
    (it lands in the committed GPU test log) and asserts it is most of the tensor."""
    top2 = torch.topk(w, 2, dim=-1).values
    ok = ((top2[..., 0] - top2[..., 1]) > 1e-4 * top2[..., 0].clamp_min(1e-30)).numpy()
    print(f"token rows compared ({what}): {int(ok.sum())} of {ok.size} = {ok.mean():.4f}; "
          f"excluded with a zero row maximum: {int((top2[..., 0] <= 0).sum())}")
    return ok


def _labels(path):
    with open(path) as f:
        return [row["label"] for row in csv.DictReader(f)]


@pytest.mark.parametrize("tf,B", [(True, 256), (False, 256), (True, 512), (False, 512), (True, 2048), (True, 4096), (False, 4096)],
                         ids=["tf-256", "fr-256", "tf-512", "fr-512", "tf-2048", "tf-4096", "fr-4096"])
def test_vae_step_at_bench_batch_vs_oracle(tf, B, tmp_path):
    """B = 256: the bench batch, every recurrent layer one resident chain launch.  B = 512: more rows than one launch holds
    (the reference's default VAE batch is 4096 measures, train_measure_vae.py:33): every layer runs as chain launches over
    256-row chunks WITH backward saves, forward and backward -- no per-step launches for the H = 512 layers.  B = 2048: beyond
    INET_CHAIN_CHUNK_MAX (1024 rows) the per-step kernels take over -- at that size one launch per step over all rows is
    the faster form (profiles/r03_e_batch_crossover.txt) -- and the same parity bar holds.  B = 4096: the reference's default step
    (256 sequences x 16 bars, train_measure_vae.py:33, vae_trainer.py:49-52), both sides of the teacher-forcing coin: what
    bench.py's `vae4096_measures_per_s` times."""
    T = 24
    c = G.CFGS["full"]
    H = c["H"]
    cfg = ops.vae_config(c["V"], c["E"], c["H"], c["Z"], c["H"])
    table, total = ops.vae_param_table(cfg)
    P = G.vae_params("full")
    params = pack(table, total, P)
    g = torch.Generator().manual_seed(2024)
    tok = torch.randint(0, c["V"], (B, T), generator=g)
    eps = torch.randn(B, c["Z"], generator=g)
    m_enc = ops.dropout_mask((T, B, 2 * H), 0.5, 77, 0, DEV)
    m_beat = ops.dropout_mask((4, B, H), 0.5, 77, 10 ** 8, DEV)
    m_tick = ops.dropout_mask((T, B, H), 0.5, 77, 2 * 10 ** 8, DEV)
    masks = {"enc": m_enc, "beat": m_beat, "tick": m_tick}
    grads = torch.zeros_like(params)
    ops.prof_enable(True)
    hl, hce, hkl, hacc, hw, hs, hz, kinks = _vae_step_with_kinks(cfg, params, grads, tok.to(DEV), eps.to(DEV), tf, masks)
    torch.cuda.synchronize()
    ops.prof_dump(tmp_path / "launches.csv")
    ops.prof_enable(False)
    labels = _labels(tmp_path / "launches.csv")
    if B >= 2048:                     # the encoder's layers: bf16-pipe step kernels, forward (with backward saves) and backward
        assert sum(l == f"gru_step_bf3 p9 np2 B{B} H512 sv" for l in labels) == 48, sorted(set(l for l in labels if l.startswith("gru")))
        assert sum(l == f"gru_step_bf3_bwd p9 np2 B{B} H512" for l in labels) == 48, sorted(set(l for l in labels if l.startswith("gru")))
    if B == 512:
        gl = sorted(set(l for l in labels if l.startswith("gru") or l.startswith("dec")))
        assert not any(l.startswith("gru_fwd") or l.startswith("gru_bwd") for l in labels), gl      # no per-step launches
        # every H = 512 layer ran as chain launches: row chunks (two 256-row chunks side by side in the encoder's forward pass,
        # 128-row chunks of all four beats in the teacher-forced tick layers) or one launch with two row tiles per workgroup
        assert sum(G.is_chain(l, "fwd", 2, 24, 256) for l in labels) == 4, gl
        assert sum(l.startswith("gru_chain_bwd") and " T24 " in l for l in labels) in (2, 4), gl
        assert sum(l.startswith("gru_chain_bwd") and " T6 " in l for l in labels) in (2, 4), gl
        if not tf:                                               # 512 free-running rows: ONE launch of the 64-row build
            assert sum(l == "decode_chain_train ms4 T24 B512 H512 V48" for l in labels) == 1, gl
    if B >= 2048:
        assert any(l.startswith("gru_fwd") for l in labels) and any(l.startswith("gru_bwd") for l in labels)
        assert not any(l.startswith("gru_chain") for l in labels), sorted(set(l for l in labels if l.startswith("gru")))
    # (iv) the instantiations the bench spends its time in were the ones that ran
    if B == 256:        # encoder layers, two directions, one launch per layer; encoder BPTT
        for kind in ("fwd", "bwd"):
            assert sum(G.is_chain(l, kind, 2, 24, 256) for l in labels) == 2, sorted(set(l for l in labels if l.startswith("gru")))
    assert B >= 2048 or any(G.is_chain(l, "bwd", 4, 6, 256) for l in labels)   # decoder tick layers: 4 beats x 256 rows in one launch
    if tf and B == 256:   # teacher-forced ticks: each tick layer = two chain launches of two beats (6 steps), no per-tick launches
        assert sum(G.is_chain(l, "fwd", 2, 6, 256) for l in labels) == 4, sorted(set(l for l in labels if l.startswith("gru")))
        assert not any(l.startswith("gru_fwd") for l in labels)
    # the big products run on the LDS-free direct kernels (forward NT 192x192, data-gradient NN 192x128, weight-gradient
    # TN 192x128 split over the XCDs); the 192-row LDS-tiled instantiations are covered by the forced-tile test below
    big = sorted(set(l for l in labels if l.startswith("M")))
    # the encoder's big products run on the bf16 matrix cores through exact three-piece splits (csrc/gemm_bf3.hip): both
    # directions' layer-1 input products as one N = 6H product, their data gradient as one K = 6H product (e4: dropout mask
    # epilogue), the weight gradients two directions per launch; the decoder's on the LDS-free f32-input direct kernels
    for wl in () if B != 256 else ("M6144 N3072 K1024 bf3p9 t192x192 s1 e0", "M6144 N1024 K3072 bf3p9 t192x128 s1 e4",
               "M1536 N512 K6144 bf3p9 t192x128 s4 e0 x2", "M1536 N1024 K6144 bf3p9 t192x128 s2 e0 x2",      # layer 1's
               "M1536 N512 K6144 TN d192x128 s8 e0", "M1536 N512 K6144 TN d192x128 s4 e0 x2",                 # layer 0's, the decoder's
               "group2 M256 N1024 K2048 NT k64x32 e1"):                                        # both SELU heads in one grouped split-K launch
        assert wl in labels, (wl, big)
    if B == 256:     # the forward chains wrote their pieces themselves (layer-1 input rows, x1^T, the previous states^T); the BPTT chains
        # run on the first generation (28 KB of LDS: the leaf work shares the CUs with them), so the gate gradients' pieces come from split
        # launches that run beside them -- (r, z, n)^T and (n*r)^T of layer 1 for its weight gradients -- next to the layer-1 input
        # weights' (rows for the forward product, k-major for the data gradient); the ROW pieces of dgi1 (the data gradient's A
        # operand, on the caller's stream) the layer-1 BPTT chain writes itself ("ms4e")
        assert sorted(set(l for l in labels if l.startswith("bf3_split"))) == [
            "bf3_split cols R1024 K1536", "bf3_split cols R1536 K6144", "bf3_split cols R512 K6144", "bf3_split rows R1536 K1024"], big
        assert sum(l == "gru_chain_bwd ms4e np2 T24 B256 H512" for l in labels) == 1 and \
            sum(l == "gru_chain_bwd ms4 np2 T24 B256 H512" for l in labels) == 1, big
    if not tf and B == 256:      # the 24 free-running ticks with dropout and backward saves: one launch of the fused decode kernel
        assert "decode_chain_train ms2 T24 B256 H512 V48" in labels, sorted(set(l for l in labels if l.startswith("dec")))
    print(sorted(set(l for l in labels if l.startswith("gru"))))

    om = {"enc": m_enc.cpu().permute(1, 0, 2), "beat": m_beat.cpu().permute(1, 0, 2), "tick": m_tick.cpu().permute(1, 0, 2)}
    for p in P.values():
        p.requires_grad_(True)
    feed = None if tf else hs.cpu()[:, 0]
    O.kink_stats_reset()
    w, s, mu, ls, z = O.vae_forward(P, tok, eps, tf, om, feed_tokens=feed, kinks=kinks)
    _assert_kinks("vae step")
    loss, ce, kl, acc = O.vae_loss(w, tok, mu, ls)
    loss.backward()
    assert relmax(hz, z) < 2e-5
    assert relmax(hw, w) < 2e-5
    ok = _comparable_rows(w.detach(), f"vae B={B} tf={tf}")
    assert ok.mean() > 0.99                     # (measured 0.9990-0.9998: near-ties only; no row of trained logits is all zeros)
    assert np.array_equal(hs.cpu().numpy()[:, 0][ok], s.numpy()[:, 0][ok])
    assert abs(hl - loss.item()) <= 1e-4 * abs(loss.item()), (hl, loss.item())
    assert abs(hce - ce.item()) <= 1e-4 * abs(ce.item())
    assert abs(hkl - kl.item()) <= 1e-4 * abs(kl.item())
    assert abs(hacc - acc.item()) <= 2.0 / (B * T)          # a near-tie row may count differently
    errs = {}
    for pname, off, shape in table:
        gg = unpack(table, grads, pname).cpu()
        errs[pname] = float((gg - P[pname].grad).abs().max() / (P[pname].grad.abs().max() + 1e-12))
    worst = max(errs, key=errs.get)
    print(f"worst gradient tensor {worst}: {errs[worst]:.2e} of its max")
    # with the SELU / ReLU branches aligned (module docstring) every gradient tensor agrees far inside the 5e-4 bar.
    # decoder.b_0 is ONE scalar: a 1536-term sum of both signs (sum_n colsum(dgi0_beat)[n] * W_ih[n]) whose terms are
    # themselves atomically accumulated column sums -- its "max" is the cancelled result, so run-to-run summation order
    # alone moves it by up to 2e-4 of itself; it keeps the north_star's 5e-4 bound, every tensor
    # with more than one element is held to 1e-4
    # (round 4: that scalar is now a fixed-order two-stage sum, rowdot_partials_kernel -> beat_input_grad_kernel; it is held to
    # the same bound as everything else)
    bound = {k: 1e-4 for k in errs}
    bad = {k: v for k, v in errs.items() if v >= bound[k]}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:5]


@pytest.mark.parametrize("variant", ["nar", "ar_tf", "ar_fr"])
def test_latent_step_at_bench_batch_vs_oracle(variant, tmp_path, monkeypatch):
    """BASELINE.json configs[2] (and the per-rank workload of configs[3]): 128 sequences x 16 measures, split 6/4/6.
    `nar` = the benched non-auto-regressive model; `ar_tf` / `ar_fr` = auto_reg=True (the reference script's default,
    train_inpaintnet.py:53) teacher-forced and free-running -- the free-running path re-encodes its own samples after
    every generated measure (latent_rnn.py:246-259), with one generator / decoder / encoder dropout mask per measure."""
    from inpaintnet_amd import synthetic
    from inpaintnet_amd import measure_vae as MV
    from inpaintnet_amd.latent_rnn import LatentRNN
    from inpaintnet_amd.latent_rnn_trainer import LatentRNNTrainer
    from inpaintnet_amd.measure_vae import MeasureVAE
    B, n_past, n_target, n_future = 128, 6, 4, 6
    auto_reg, tf = variant != "nar", variant == "ar_tf"
    free_ar = variant == "ar_fr"
    c = G.CFGS["full"]
    H, Z, V = c["H"], c["Z"], c["V"]
    ds = synthetic.SyntheticFolkDataset(num_notes=V)
    vae = MeasureVAE(ds)                                            # reference defaults: dropout 0.5 everywhere
    model = LatentRNN(ds, vae, num_rnn_layers=2, rnn_hidden_size=H, dropout=0.5, rnn_class=torch.nn.GRU,
                      auto_reg=auto_reg, teacher_forcing=True)
    P = G.latent_params("full", auto_reg)
    model.load_state_dict(P)
    model.encode_unused_target = True              # the reference's work measure for measure: all 16 measures are encoded (what bench.py times)
    trainer = LatentRNNTrainer(ds, model, lr=1e-4)
    model.train()
    MV.set_dropout_seed(99)
    score = torch.from_numpy(synthetic.folk_score(B, V, seed=31))
    past, future, target = LatentRNNTrainer.split_score(score, n_past, n_future, n_target, 24)
    g = torch.Generator().manual_seed(5)
    eps = tuple(torch.randn(B, n, Z, generator=g) for n in (n_past, n_future, n_target))
    eps_ar = [torch.randn(B, Z, generator=g) for _ in range(n_target)] if free_ar else None

    rec = []
    real_mask = ops.dropout_mask

    def recording_mask(shape, p, seed, offset, device):
        m = real_mask(shape, p, seed, offset, device)
        rec.append((tuple(shape), m))
        return m
    monkeypatch.setattr(ops, "dropout_mask", recording_mask)

    # branch of every SELU / ReLU element of the (frozen) decoder: one workspace per decoder call
    wss = []
    real_dec = ops.decoder_fwd

    def recording_dec(*a, **k):
        out = real_dec(*a, **k)
        wss.append(out[2])
        return out
    monkeypatch.setattr(ops, "decoder_fwd", recording_dec)

    trainer.zero_grad()
    ops.prof_enable(True)
    w, s, gz = model(past, future, target, n_target, train=True, eps=tuple(e.cuda() for e in eps), teacher_forcing=tf,
                     eps_ar=[e.cuda() for e in eps_ar] if free_ar else None)
    k_relu = w.detach().cpu() > 0
    if not free_ar:                                                 # one decoder call, rows ordered (sequence, measure)
        assert len(wss) == 1
        dws, R = wss[0], B * n_target
        k_hb0 = (ops.ws_field(vae.cfg, dws, R, 1, "hb0").view(B, n_target, 2 * H).cpu() > 0)
        k_ht0 = (ops.ws_field(vae.cfg, dws, R, 1, "ht0").view(4, B, n_target, 2 * H).cpu() > 0)
        k_c = (ops.ws_field(vae.cfg, dws, R, 1, "c_all").view(4, B, n_target, H).cpu() > 0)
        dec_kinks = [{"hb0": k_hb0[:, i], "ht0": k_ht0[:, :, i], "c_all": k_c[:, :, i], "relu": k_relu[:, i]}
                     for i in range(n_target)]
    else:                                                           # one decoder call of B rows per generated measure
        assert len(wss) == n_target
        dec_kinks = [{"hb0": ops.ws_field(vae.cfg, wss[i], B, 1, "hb0").view(B, 2 * H).cpu() > 0,
                      "ht0": ops.ws_field(vae.cfg, wss[i], B, 1, "ht0").view(4, B, 2 * H).cpu() > 0,
                      "c_all": ops.ws_field(vae.cfg, wss[i], B, 1, "c_all").view(4, B, H).cpu() > 0,
                      "relu": k_relu[:, i]} for i in range(n_target)]
    loss, acc = trainer.mean_crossentropy_loss_and_accuracy(w, target)
    loss.backward()
    ops.side_join()
    torch.cuda.synchronize()
    ops.prof_dump(tmp_path / "launches.csv")
    ops.prof_enable(False)
    assert ops.chain_status() == 0
    labels = _labels(tmp_path / "launches.csv")
    # frozen encoder over all 128 x 16 measures at once: one time step of 2048 rows fills the chip, so each layer runs 24 bf16-pipe
    # products with the GRU cell as their epilogue (csrc/gru_step_bf3.hip; round 3: the chain kernel over eight 256-row chunks)
    assert sum(l == "gru_step_bf3 p9 np2 B2048 H512" for l in labels) == 48, sorted(set(l for l in labels if l.startswith("gru")))
    assert any(G.is_chain(l, "fwd", 2, 6, 128) for l in labels) and any(G.is_chain(l, "bwd", 2, 6, 128) for l in labels)   # contexts: 8 groups of 32 rows
    if variant == "nar":
        # generator (H = 1024, 4 target measures): first-generation chain launches, a group's 64 members on two XCDs
        assert sum(l == "gru_chain_fwd ms4 np2 T4 B128 H1024" for l in labels) == 2, sorted(set(l for l in labels if "H1024" in l))
        # the frozen decoder's 512 free-running rows: ONE launch of the fused decode kernel's 64-row build (with backward saves;
        # round 3: two launches of 256 rows)
        assert sum(l == "decode_chain_train ms4 T24 B512 H512 V48" for l in labels) == 1, sorted(set(l for l in labels if l.startswith("dec")))
        assert sum(l == "gru_chain_bwd ms4 np2 T4 B128 H1024" for l in labels) == 2, sorted(set(l for l in labels if "H1024" in l))
    if free_ar:                                                     # one fused decode launch of 128 rows per generated measure
        assert sum(l.startswith("decode_chain_train") for l in labels) == n_target, sorted(set(l for l in labels if l.startswith("dec")))

    # masks in call order: encoder (T, 16B, 2H); context past (np, B, 2H); context future; then
    #   nar / ar_tf: generator (nt, B, 4H); decoder beat (4, nt*B, H); decoder tick (24, nt*B, H)
    #   ar_fr, per generated measure: generator (1, B, 4H); decoder beat (4, B, H); tick (24, B, H); re-encoding (24, B, 2H)
    shapes = [sh for sh, _ in rec]
    head = [(24, 16 * B, 2 * H), (n_past, B, 2 * H), (n_future, B, 2 * H)]
    if not free_ar:
        assert shapes == head + [(n_target, B, 4 * H), (4, n_target * B, H), (24, n_target * B, H)], shapes
    else:
        assert shapes == head + [(1, B, 4 * H), (4, B, H), (24, B, H), (24, B, 2 * H)] * n_target, shapes
    ms = [m.cpu() for _, m in rec]
    m_enc, m_cp, m_cf = ms[:3]
    # encoder rows are ordered (sequence, measure) with measures = past | target | future
    me = m_enc.permute(1, 0, 2).reshape(B, 16, 24, 2 * H)

    def enc_rows(lo, hi):
        return me[:, lo:hi].reshape(B * (hi - lo), 24, 2 * H)
    masks = {"enc_past": enc_rows(0, n_past), "enc_target": enc_rows(n_past, n_past + n_target),
             "enc_future": enc_rows(n_past + n_target, 16),
             "ctx_past": m_cp.permute(1, 0, 2), "ctx_future": m_cf.permute(1, 0, 2)}
    if not free_ar:
        m_gen, m_beat, m_tick = ms[3:6]
        # decoder rows are ordered (sequence, measure)
        mb = m_beat.permute(1, 0, 2).reshape(B, n_target, 4, H)
        mt = m_tick.permute(1, 0, 2).reshape(B, n_target, 24, H)
        masks["gen"] = m_gen.permute(1, 0, 2)
        masks["dec"] = [{"beat": mb[:, i], "tick": mt[:, i]} for i in range(n_target)]
    else:
        per = [ms[3 + 4 * i: 7 + 4 * i] for i in range(n_target)]
        masks["gen"] = [q[0].permute(1, 0, 2) for q in per]
        masks["dec"] = [{"beat": q[1].permute(1, 0, 2), "tick": q[2].permute(1, 0, 2)} for q in per]
        masks["enc_ar"] = [q[3].permute(1, 0, 2) for q in per]
    own = [k for k in P if not k.startswith("vae_model.")]
    for k in own:
        P[k].requires_grad_(True)
    hs = s.cpu().view(B, n_target, 24)
    O.kink_stats_reset()
    wo, so, gzo = O.latent_forward(P, past.cpu(), future.cpu(), target.cpu(), eps[0].reshape(-1, Z), eps[1].reshape(-1, Z),
                                   eps[2].reshape(-1, Z), auto_reg=auto_reg, teacher_forcing=tf, eps_ar=eps_ar,
                                   masks=masks, feed_tokens=hs, dec_kinks=dec_kinks)
    _assert_kinks(f"latent step {variant}")
    lo, ao = O.latent_loss(wo, target.cpu())
    lo.backward()
    assert G.rel_err(gz.detach().cpu(), gzo.detach()) < 5e-5
    assert G.rel_err(w.detach().cpu(), wo.detach()) < 5e-5
    ok = _comparable_rows(wo.detach(), f"latent {variant}").reshape(B, -1)
    assert ok.mean() > 0.99
    assert np.array_equal(s.cpu().numpy()[:, 0][ok], so.numpy()[:, 0][ok])
    assert abs(float(loss.detach()) - lo.item()) <= 1e-4 * abs(lo.item())
    assert abs(float(acc) - ao.item()) <= 2.0 / (B * n_target * 24)
    errs = {k: float((model.param_grad(k).cpu() - P[k].grad).abs().max() / (P[k].grad.abs().max() + 1e-12)) for k in own
            if P[k].grad is not None}
    worst = max(errs, key=errs.get)
    print(f"worst gradient tensor {worst}: {errs[worst]:.2e} of its max")
    assert errs[worst] < 2e-4, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    assert float(vae.grad.abs().max()) == 0.0


# (iii) every tile configuration x split over the big products of the step: gi1 = x1 W_ih^T (6144x1536x1024), the
# k-major x k-major weight-gradient product (1536x1024x6144) and dx1 = dgi1 W_ih (6144x1024x1536, B k-major)
@pytest.mark.parametrize("cfg_i", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("split", [1, 4])
@pytest.mark.parametrize("M,N,K,akm,bkm", [(6144, 1536, 1024, 0, 0), (1536, 1024, 6144, 1, 1), (6144, 1024, 1536, 0, 1)])
def test_gemm_forced_tiles_on_step_shapes(cfg_i, split, M, N, K, akm, bkm, tmp_path):
    g = torch.Generator().manual_seed(cfg_i * 100 + split * 10 + akm + 2 * bkm)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(N, K, generator=g)
    ref = (A.double() @ Bm.double().t())
    Ad = (A.t().contiguous() if akm else A).to(DEV)
    Bd = (Bm.t().contiguous() if bkm else Bm).to(DEV)
    tiles = ["t64x64", "t128x128", "t192x64", "t192x128", "t192x192"]
    try:
        ops.set_option(2, cfg_i)
        ops.set_option(3, split)
        ops.prof_enable(True)
        C = ops.gemm(Ad, Bd, M, N, K, a_kmajor=akm, b_kmajor=bkm)
        C0 = torch.randn(M, N, generator=g)
        C1 = C0.to(DEV).clone()
        ops.gemm(Ad, Bd, M, N, K, a_kmajor=akm, b_kmajor=bkm, out=C1, accumulate=True)
        torch.cuda.synchronize()
        ops.prof_dump(tmp_path / "g.csv")
    finally:
        ops.prof_enable(False)
        ops.set_option(2, -1)
        ops.set_option(3, 0)
    labels = _labels(tmp_path / "g.csv")
    assert len(labels) == 2 and all(f" {tiles[cfg_i]} s{split} " in l for l in labels), labels
    assert relmax(C, ref) < 2e-5
    assert relmax(C1, ref + C0.double()) < 2e-5


def test_gemm_repeatability_bound():
    """Split-K sums with f32 atomics are order-dependent: repeated runs may differ in the last bits, never by more
    than a few ulp of the largest partial sum (pin the tolerance)."""
    g = torch.Generator().manual_seed(12)
    M, N, K = 1536, 512, 6144
    A = torch.randn(K, M, generator=g).to(DEV)
    Bm = torch.randn(K, N, generator=g).to(DEV)
    outs = [ops.gemm(A, Bm, M, N, K, a_kmajor=True, b_kmajor=True).cpu() for _ in range(3)]
    scale = float(outs[0].abs().max())
    for o in outs[1:]:
        assert float((o - outs[0]).abs().max()) <= 4e-6 * scale


@pytest.mark.parametrize("B", [256, 512])
def test_piece_outputs_written_by_chains_or_by_split_launches_give_the_same_step(B):
    """The bf16 pieces of the encoder's products are the same whoever writes them -- the chain kernels (ChainEmit) or bf3_split
    launches over the f32 arrays --, and so is the step: loss, logits and every gradient of the teacher-forced step (B = 256:
    one chain launch per layer; 512: row chunks) under inet_set_option key 9 = 0 / 1 / 3 / 4 agree with the default's to the order
    of the f32 atomics (split-K weight gradients); with the products on the f32-input kernels (key 8 = 0) to the usual f32 bound."""
    T = 24
    c = G.CFGS["full"]
    H = c["H"]
    cfg = ops.vae_config(c["V"], c["E"], c["H"], c["Z"], c["H"])
    table, total = ops.vae_param_table(cfg)
    params = pack(table, total, G.vae_params("full"))
    g = torch.Generator().manual_seed(4048 + B)
    tok = torch.randint(0, c["V"], (B, T), generator=g).to(DEV)
    eps = torch.randn(B, c["Z"], generator=g).to(DEV)
    masks = {"enc": ops.dropout_mask((T, B, 2 * H), 0.5, 78, 0, DEV), "beat": ops.dropout_mask((4, B, H), 0.5, 78, 10 ** 8, DEV),
             "tick": ops.dropout_mask((T, B, H), 0.5, 78, 2 * 10 ** 8, DEV)}

    def step():
        grads = torch.zeros_like(params)
        hl, hce, hkl, hacc, hw, hs, hz, _ = _vae_step_with_kinks(cfg, params, grads, tok, eps, True, masks)
        torch.cuda.synchronize()
        return hl, hw.clone(), grads

    default = {8: 9, 9: 7}
    try:
        l0, w0, g0 = step()
        # key 9: nothing / only the forward rows / rows + transposed / only the BPTT kernel's rows written by the chains
        for key, val, bwd2 in ((9, 0, 0), (9, 1, 0), (9, 3, 0), (9, 4, 0), (8, 0, 0)):
            ops.set_option(key, val)
            try:
                l, w, gr = step()
            finally:
                ops.set_option(key, default[key])
            tol = 2e-5 if key == 8 or bwd2 else 2e-6           # (another kernel's f32 summation order: the usual f32 bound)
            # (the loss is a sum of B * T row terms accumulated with f32 atomics: its last bits depend on their order)
            assert abs(l - l0) <= 4e-6 * abs(l0), (key, val, l, l0)
            assert relmax(w, w0) < tol, (key, val)
            bad = []
            for pname, off, shape in table:
                a, b = unpack(table, gr, pname), unpack(table, g0, pname)
                err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
                if not err < tol * 5:
                    bad.append((pname, err))
            assert not bad, (key, val, bad)
    finally:
        for k, v in default.items():
            ops.set_option(k, v)
    assert ops.chain_status() == 0
