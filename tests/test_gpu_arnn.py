"""GPU parity of the AnticipationRNN (BASELINE config 5) against golden vectors captured
from the reference's ConstraintModelGaussianReg + AnticipationRNNGaussianRegTrainer, and against the oracle's autograd
for the free-running path (whose backward the reference cannot run on CPU under torch 2.x)."""
import types

import numpy as np
import pytest
import torch

from oracle import torch_ref as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from inpaintnet_amd import ops, synthetic
    from inpaintnet_amd.arnn import AnticipationRNNGaussianRegTrainer, ConstraintModelGaussianReg


def build(name):
    c = G.ARNN_CFGS[name]
    ds = synthetic.SyntheticFolkDataset(num_notes=c["V"])
    ds.metadatas = [types.SimpleNamespace(num_values=6), types.SimpleNamespace(num_values=6)]
    model = ConstraintModelGaussianReg(ds, note_embedding_dim=c["E"], metadata_embedding_dim=c["Em"],
                                       num_lstm_constraints_units=c["H"], num_lstm_generation_units=c["H"],
                                       linear_hidden_size=c["LH"], num_layers=2, dropout_input_prob=0.0,
                                       dropout_prob=0.0, unary_constraint=True, teacher_forcing=True)
    model.load_state_dict(G.arnn_params(name))
    return ds, model


def test_lstm_layer_kernel_vs_oracle():
    g = torch.Generator().manual_seed(4)
    # H in {256, 512} runs the chain kernels (one persistent launch per layer, csrc/chain.h): one group (B <= 64, ragged
    # last row block), several groups (B = 70, 130), both directions, both register geometries
    for (B, T, K, H, rev) in [(3, 5, 6, 16, False), (33, 7, 20, 32, True), (32, 12, 266, 256, False),
                              (37, 9, 20, 256, True), (70, 5, 8, 256, False), (130, 4, 16, 256, True),
                              (5, 6, 8, 512, True), (24, 30, 12, 512, False)]:
        case = (B, T, K, H, rev)
        whh_scale = 0.2 if H <= 256 and T <= 12 else 1.0 / np.sqrt(H)    # long / wide cases: keep the recurrence contractive
        P = {"l.weight_ih_l0": torch.randn(4 * H, K, generator=g) * 0.2, "l.weight_hh_l0": torch.randn(4 * H, H, generator=g) * whh_scale,
             "l.bias_ih_l0": torch.randn(4 * H, generator=g) * 0.1, "l.bias_hh_l0": torch.randn(4 * H, generator=g) * 0.1}
        for p in P.values():
            p.requires_grad_(True)
        x = torch.randn(B, T, K, generator=g).requires_grad_(True)
        h0 = torch.randn(B, H, generator=g).requires_grad_(True)
        c0 = torch.randn(B, H, generator=g).requires_grad_(True)
        out, (hT, cT) = O.lstm_layer(x, h0, c0, P, "l", reverse=rev)
        wo = torch.randn(B, T, H, generator=g)
        wh, wc = torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)
        ((out * wo).sum() + (hT * wh).sum() + (cT * wc).sum()).backward()
        d = {k: v.detach().cuda() for k, v in P.items()}
        x_tm = x.detach().permute(1, 0, 2).contiguous().cuda()
        gi = ops.linear_fwd(x_tm.view(T * B, K), d["l.weight_ih_l0"], d["l.bias_ih_l0"]).view(T, B, 4 * H)
        o, h, c, ws = ops.lstm_fwd(gi, d["l.weight_hh_l0"], d["l.bias_hh_l0"], H, reverse=rev, h0=h0.detach().cuda(),
                                   c0=c0.detach().cuda(), save=True, want_state=True)
        assert G.rel_err(o.permute(1, 0, 2).cpu(), out.detach()) < 5e-5, case
        assert G.rel_err(h.cpu(), hT.detach()) < 5e-5 and G.rel_err(c.cpu(), cT.detach()) < 5e-5, case
        dW = torch.zeros(4 * H, H, device="cuda"); dbi = torch.zeros(4 * H, device="cuda"); dbh = torch.zeros(4 * H, device="cuda")
        dgi, dh0, dc0 = ops.lstm_bwd(d["l.weight_hh_l0"], o, wo.permute(1, 0, 2).contiguous().cuda(), H, rev, ws, dW, dbi, dbh,
                                     h0=h0.detach().cuda(), dhT=wh.cuda(), dcT=wc.cuda(), want_dstate=True)
        torch.cuda.synchronize()
        assert G.rel_err(dW.cpu(), P["l.weight_hh_l0"].grad) < 5e-4, case
        assert G.rel_err(dbh.cpu(), P["l.bias_hh_l0"].grad) < 5e-4 and G.rel_err(dbi.cpu(), P["l.bias_ih_l0"].grad) < 5e-4, case
        assert G.rel_err(dh0.cpu(), h0.grad) < 5e-4 and G.rel_err(dc0.cpu(), c0.grad) < 5e-4, case
        dx = ops.linear_bwd(dgi.view(T * B, 4 * H), x_tm.view(T * B, K), d["l.weight_ih_l0"], need_dx=True)
        assert G.rel_err(dx.view(T, B, K).permute(1, 0, 2).cpu(), x.grad) < 5e-4, case
        assert ops.chain_status() == 0, case


def test_lstm_chain_matches_per_step_launches():
    """Chain kernel vs one launch per step on the AnticipationRNN shape (B=32, H=256, 384 ticks): same contraction order per
    output element; results agree to fp32 round-off over all 384 steps."""
    g = torch.Generator().manual_seed(8)
    B, T, H = 32, 384, 256
    gi = (torch.randn(T, B, 4 * H, generator=g) * 0.5).cuda()
    W = (torch.randn(4 * H, H, generator=g) / 16).cuda()
    b = (torch.randn(4 * H, generator=g) * 0.1).cuda()
    dout = (torch.randn(T, B, H, generator=g) * 0.1).cuda()
    res = {}
    for chain in (1, 0):
        ops.set_option(4, chain)
        try:
            o, h, c, ws = ops.lstm_fwd(gi, W, b, H, reverse=False, save=True, want_state=True)
            dW = torch.zeros(4 * H, H, device="cuda"); dbi = torch.zeros(4 * H, device="cuda"); dbh = torch.zeros(4 * H, device="cuda")
            dgi, dh0, dc0 = ops.lstm_bwd(W, o, dout, H, False, ws, dW, dbi, dbh, want_dstate=True)
            torch.cuda.synchronize()
            res[chain] = [x.cpu() for x in (o, h, c, dgi, dh0, dc0, dW, dbi)]
        finally:
            ops.set_option(4, 1)
    assert ops.chain_status() == 0
    for i in range(8):          # (not bit-identical: the two kernels' epilogues are FMA-contracted differently)
        assert G.rel_err(res[1][i], res[0][i]) < 2e-5, i


@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("B,T", [(32, 200), (7, 97)])
def test_lstm_two_layer_pipeline_matches_layer_by_layer(B, T, reverse):
    """inet_lstm2_fwd / _bwd (layer 1 one chunk of time steps behind layer 0 on a second stream, the backward pass the other
    way round) against the same two layers run one after the other through inet_lstm_fwd / _bwd: outputs, input-side gate
    gradients and all seven weight / bias gradients; T not a multiple of the chunk, both time directions, a ragged batch."""
    g = torch.Generator().manual_seed(B * 1000 + T + int(reverse))
    H = 256
    assert ops.lstm2_ok(B, T, H)
    gi0 = (torch.randn(T, B, 4 * H, generator=g) * 0.5).cuda()
    Wh0, Wi1, Wh1 = [(torch.randn(4 * H, H, generator=g) / 16).cuda() for _ in range(3)]
    bh0, bi1, bh1 = [(torch.randn(4 * H, generator=g) * 0.1).cuda() for _ in range(3)]
    dout1 = (torch.randn(T, B, H, generator=g) * 0.1).cuda()
    # layer by layer
    o0, _, _, ws0 = ops.lstm_fwd(gi0, Wh0, bh0, H, reverse=reverse, save=True)
    gi1 = ops.linear_fwd(o0.view(T * B, H), Wi1, bi1).view(T, B, 4 * H)
    o1, _, _, ws1 = ops.lstm_fwd(gi1, Wh1, bh1, H, reverse=reverse, save=True)
    ref_g = [torch.zeros_like(x) for x in (Wh0, bh0, bh0, Wi1, Wh1, bh1, bh1)]
    dgi1, _, _ = ops.lstm_bwd(Wh1, o1, dout1, H, reverse, ws1, ref_g[4], ref_g[5], ref_g[6])
    do0 = ops.linear_bwd(dgi1.view(T * B, 4 * H), o0.view(T * B, H), Wi1, ref_g[3], None, need_dx=True).view(T, B, H)
    dgi0, _, _ = ops.lstm_bwd(Wh0, o0, do0.contiguous(), H, reverse, ws0, ref_g[0], ref_g[1], ref_g[2])
    torch.cuda.synchronize()
    # pipelined
    p0, p1, pw0, pw1 = ops.lstm2_fwd(gi0, Wh0, bh0, Wi1, bi1, Wh1, bh1, H, reverse=reverse, save=True)
    got_g = [torch.zeros_like(x) for x in ref_g]
    pdgi0 = ops.lstm2_bwd(Wh0, Wi1, Wh1, p0, p1, dout1, H, reverse, pw0, pw1, grads=got_g)
    torch.cuda.synchronize()
    assert ops.chain_status() == 0
    assert G.rel_err(p0.cpu(), o0.cpu()) < 2e-5 and G.rel_err(p1.cpu(), o1.cpu()) < 2e-5
    assert G.rel_err(pdgi0.cpu(), dgi0.cpu()) < 1e-4
    for i, (a, b) in enumerate(zip(got_g, ref_g)):
        assert G.rel_err(a.cpu(), b.cpu()) < 1e-4, i


@pytest.mark.parametrize("name", ["small", "full"])
def test_arnn_teacher_forced_step_golden(name):
    fx = G.load("arnn_" + name)
    ds, model = build(name)
    trainer = AnticipationRNNGaussianRegTrainer(ds, model, lr=1e-4)
    model.train()
    score = torch.from_numpy(fx["score"]).cuda()
    md = torch.from_numpy(fx["metadata"]).cuda()
    loc = torch.from_numpy(fx["constraints_loc"]).cuda()
    a, b = [int(x) for x in fx["ticks"]]
    with torch.no_grad():
        w_all, _ = model._forward_tf(score, md, loc)
    assert w_all[0].shape == fx["tf_weights_all"].shape
    assert G.rel_err(w_all[0].cpu(), fx["tf_weights_all"]) < 1e-4
    trainer.zero_grad()
    weights, _ = model(score, md, loc, a, b, train=True, teacher_forcing=True)
    assert weights[0].shape == (score.shape[0], b - a, G.ARNN_CFGS[name]["V"])
    targets = score[:, :, a:b].transpose(0, 1)
    loss, acc = trainer.mean_crossentropy_loss_and_accuracy_voices(weights, targets)
    loss.backward()
    assert abs(float(loss.detach()) - fx["tf_loss_acc"][0]) < 1e-4 * abs(fx["tf_loss_acc"][0])
    assert abs(float(acc) - fx["tf_loss_acc"][1]) < 1e-6
    bad = []
    for k, _ in model.named_parameters():
        g = model.param_grad(k).cpu().numpy()
        if name == "small":
            key = "tf_grad/" + k
            ref = fx[key] if key in fx.files else np.zeros_like(g)
            err = np.abs(g - ref).max() / (np.abs(ref).max() + 1e-7)
        else:
            # full-size fixture: per-tensor element slices (first / last 64 elements against the tensor's own scale:
            # norm / sqrt(n) = its rms), the signed sum, and the norm
            key = "tf_gradnorm/" + k
            rn = float(fx[key]) if key in fx.files else 0.0
            flat = g.reshape(-1).astype(np.float64)
            err = abs(float(np.sqrt((flat ** 2).sum())) - rn) / (rn + 1e-9)
            if key in fx.files:
                rms = rn / np.sqrt(flat.size) + 1e-12
                scale = max(float(np.abs(fx["tf_gradhead/" + k]).max()), float(np.abs(fx["tf_gradtail/" + k]).max()), rms)
                e_head = float(np.abs(flat[:64] - fx["tf_gradhead/" + k]).max()) / scale
                e_tail = float(np.abs(flat[-64:] - fx["tf_gradtail/" + k]).max()) / scale
                e_sum = abs(float(flat.sum()) - float(fx["tf_gradsum/" + k])) / (rn * np.sqrt(flat.size) + 1e-12)
                err = max(err, e_head, e_tail, e_sum)
        if not err < 1e-3:
            bad.append((k, float(err)))
    assert not bad, bad
    trainer.step()
    for k, _ in model.named_parameters():
        v = model.param(k).cpu().numpy()
        if name == "small":
            assert np.abs(v - fx["tf_after1/" + k]).max() < 1e-5, k
        else:
            assert np.abs(v.reshape(-1)[:64] - fx["tf_after1head/" + k]).max() < 1e-5, k


@pytest.mark.parametrize("name", ["small", "full"])
def test_arnn_free_running_forward_golden(name):
    fx = G.load("arnn_" + name)
    ds, model = build(name)
    model.train()
    score = torch.from_numpy(fx["score"]).cuda()
    md = torch.from_numpy(fx["metadata"]).cuda()
    loc = torch.from_numpy(fx["constraints_loc"]).cuda()
    a, b = [int(x) for x in fx["ticks"]]
    with torch.no_grad():
        weights, gen = model(score, md, loc, a, b, train=True, teacher_forcing=False)
    ok = fx["fr_margin_row0"] > 1e-4
    first_bad = int(np.argmin(ok)) if not ok.all() else len(ok)
    assert gen.shape == fx["fr_gen"].shape
    assert np.array_equal(gen.cpu().numpy()[:, 0, :first_bad], fx["fr_gen"][:, 0, :first_bad])
    if first_bad == len(ok):
        assert G.rel_err(weights[0].cpu(), fx["fr_weights_free"]) < 2e-4


def test_arnn_free_running_backward_vs_oracle():
    """The reference trains through this path on GPU; its backward cannot run on CPU (in-place on a saved view), so
    the gradient check is against the oracle's autograd over the same arithmetic."""
    name = "small"
    fx = G.load("arnn_" + name)
    ds, model = build(name)
    model.train()
    L = 48                                                       # a shorter window keeps the CPU side quick
    score = torch.from_numpy(fx["score"])[:, :, :L]
    md = torch.from_numpy(fx["metadata"])[:, :, :L]
    loc = torch.zeros_like(score)
    loc[:, :, :12] = 1
    loc[:, :, 36:] = 1
    P = G.arnn_params(name)
    for p in P.values():
        p.requires_grad_(True)
    w_all, gen = O.arnn_forward(P, score, md, loc, teacher_forcing=False)
    loss, acc = O.arnn_loss(w_all[:, 12:36], score[:, 0, 12:36])
    loss.backward()
    model.zero_grad()
    weights, g2 = model(score.cuda(), md.cuda(), loc.cuda(), train=True, teacher_forcing=False)
    trainer = AnticipationRNNGaussianRegTrainer(ds, model)
    l2, a2 = trainer.mean_crossentropy_loss_and_accuracy_voices(weights, score.cuda()[:, :, 12:36].transpose(0, 1))
    l2.backward()
    if np.array_equal(g2.cpu().numpy()[:, 0], gen.numpy()):
        assert abs(float(l2.detach()) - loss.item()) < 1e-4 * abs(loss.item())
        bad = []
        for k in P:
            gr = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
            err = float((model.param_grad(k).cpu() - gr).abs().max() / (gr.abs().max() + 1e-7))
            if not err < 2e-3:
                bad.append((k, err))
        assert not bad, bad


def test_arnn_free_running_batched_form_equals_the_per_tick_loop(monkeypatch):
    """The free-running pass as [one sequential pass over batch element 0 for the tokens (inet_arnn_generate) + the batched kernels
    over the whole batch] against the per-tick loop of the reference's shape (INET_ARNN_FREE_RUN=loop): the same tokens, logits and
    gradients (the reference feeds back only batch element 0's argmax, anticipation_rnn_gauss_reg_model.py:253-256)."""
    from inpaintnet_amd import arnn
    fx = G.load("arnn_small")
    score = torch.from_numpy(fx["score"]).cuda()
    md = torch.from_numpy(fx["metadata"]).cuda()
    loc = torch.from_numpy(fx["constraints_loc"]).cuda()
    a, b = [int(x) for x in fx["ticks"]]
    res = []
    for batched in (True, False):
        monkeypatch.setattr(arnn, "_FREE_RUN_BATCHED", batched)
        ds, model = build("small")
        model.train()
        model.zero_grad()
        trainer = AnticipationRNNGaussianRegTrainer(ds, model)
        weights, gen = model(score, md, loc, a, b, train=True, teacher_forcing=False)
        loss, _ = trainer.mean_crossentropy_loss_and_accuracy_voices(weights, score[:, :, a:b].transpose(0, 1))
        loss.backward()
        ops.side_join()
        torch.cuda.synchronize()
        res.append((weights[0].detach().clone(), gen.clone(), model.grad.clone(), float(loss.detach())))
    (w1, g1, gr1, l1), (w0, g0, gr0, l0) = res
    assert torch.equal(g1, g0)
    assert G.rel_err(w1.cpu(), w0.cpu()) < 2e-5
    assert abs(l1 - l0) < 1e-5 * abs(l0)
    assert G.rel_err(gr1.cpu(), gr0.cpu()) < 1e-4


@pytest.mark.parametrize("window", ["golden", "from_zero", "to_end"])
def test_arnn_forward_inpaint_batched_form_equals_the_per_tick_loop(window, monkeypatch):
    """forward_inpaint the same way: the window's tokens from one sequential pass over batch element 0 behind the teacher-forced
    prefix (its state, its token in front of the window), the window for the whole batch in one batched pass -- against the loop."""
    from inpaintnet_amd import arnn
    fx = G.load("arnn_small")
    score = torch.from_numpy(fx["score"]).cuda()
    md = torch.from_numpy(fx["metadata"]).cuda()
    loc = torch.from_numpy(fx["constraints_loc"]).cuda()
    L = score.shape[2]
    a, b = {"golden": [int(x) for x in fx["ticks"]], "from_zero": (0, 30), "to_end": (L - 40, L)}[window]
    ds, model = build("small")
    model.eval()
    res = []
    for batched in (True, False):
        monkeypatch.setattr(arnn, "_FREE_RUN_BATCHED", batched)
        with torch.no_grad():
            w, gen = model.forward_inpaint(score, md, loc, a, b)
        torch.cuda.synchronize()
        res.append((w[0].clone(), gen.clone()))
    assert torch.equal(res[0][1], res[1][1])
    assert res[0][0].shape == res[1][0].shape == (score.shape[0], b - a, res[0][0].shape[-1])
    assert G.rel_err(res[0][0].cpu(), res[1][0].cpu()) < 2e-5


def test_arnn_bench_shape_step_with_input_dropout_vs_oracle(tmp_path, monkeypatch):
    """The shape bench.py times (BASELINE.json configs[4]): 32 sequences x 384 ticks, H = 256, 2 + 2 LSTM layers, teacher
    forced, Dropout2d(0.2) on the shifted note embeddings -- the mask the product drew is recorded and replayed in the
    oracle (`input_mask`), every gradient tensor is compared element-wise, and the profile labels prove that both LSTM
    stacks ran as the two-layer chunk pipeline (inet_lstm2_*: chain launches of 32 steps)."""
    import csv
    name = "full"
    c = G.ARNN_CFGS[name]
    B, L, V = 32, 384, c["V"]
    ds = synthetic.SyntheticFolkDataset(num_notes=V)
    ds.metadatas = [types.SimpleNamespace(num_values=6), types.SimpleNamespace(num_values=6)]
    model = ConstraintModelGaussianReg(ds, note_embedding_dim=c["E"], metadata_embedding_dim=c["Em"],
                                       num_lstm_constraints_units=c["H"], num_lstm_generation_units=c["H"],
                                       linear_hidden_size=c["LH"], num_layers=2, dropout_input_prob=0.2,
                                       dropout_prob=0.2, unary_constraint=True, teacher_forcing=True)
    model.load_state_dict(G.arnn_params(name))
    trainer = AnticipationRNNGaussianRegTrainer(ds, model, lr=1e-4)
    model.train()
    score = torch.from_numpy(synthetic.folk_score(B, V, seed=21)).long()
    md = torch.from_numpy(synthetic.folk_metadata(B)).long()
    a, b = 7 * 24, 11 * 24
    loc = torch.zeros_like(score)
    loc[:, :, :a] = 1
    loc[:, :, b:] = 1
    rec = []
    real_mask = ops.dropout_mask

    def recording_mask(shape, p, seed, offset, device):
        m = real_mask(shape, p, seed, offset, device)
        rec.append((tuple(shape), float(p), m.clone()))
        return m
    monkeypatch.setattr(ops, "dropout_mask", recording_mask)
    trainer.zero_grad()
    ops.prof_enable(True)
    weights, _ = model(score.cuda(), md.cuda(), loc.cuda(), a, b, train=True, teacher_forcing=True)
    loss, acc = trainer.mean_crossentropy_loss_and_accuracy_voices(weights, score.cuda()[:, :, a:b].transpose(0, 1))
    loss.backward()
    ops.side_join()
    torch.cuda.synchronize()
    ops.prof_dump(tmp_path / "l.csv")
    ops.prof_enable(False)
    assert ops.chain_status() == 0
    with open(tmp_path / "l.csv") as f:
        labels = [r["label"] for r in csv.DictReader(f)]
    # two stacks x two layers x 12 chunks of 32 steps, forward and backward
    nf = sum(l.startswith("lstm_chain_fwd") and " T32 B32 H256" in l for l in labels)
    nb_ = sum(l.startswith("lstm_chain_bwd") and " T32 B32 H256" in l for l in labels)
    assert nf == 48 and nb_ == 48, sorted(set(l for l in labels if l.startswith("lstm")))
    assert [(sh, p) for sh, p, _ in rec] == [((L, B), 0.2)], [(sh, p) for sh, p, _ in rec]
    m = rec[0][2].cpu()                                            # [L,B], pre-scaled {0, 1/0.8}
    assert set(np.unique(m.numpy()).round(4)) <= {0.0, 1.25}
    assert 0.1 < float((m == 0).float().mean()) < 0.3
    P = G.arnn_params(name)
    for p in P.values():
        p.requires_grad_(True)
    w_all, _ = O.arnn_forward(P, score, md, loc, teacher_forcing=True, input_mask=m.t().unsqueeze(-1))
    lo, ao = O.arnn_loss(w_all[:, a:b], score[:, 0, a:b])
    lo.backward()
    assert G.rel_err(weights[0].detach().cpu(), w_all[:, a:b].detach()) < 1e-4
    assert abs(float(loss.detach()) - lo.item()) < 1e-4 * abs(lo.item())
    assert abs(float(acc) - ao.item()) <= 2.0 / (B * (b - a))
    errs = {}
    for k in P:
        gr = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
        errs[k] = float((model.param_grad(k).cpu() - gr).abs().max() / (gr.abs().max() + 1e-9))
    worst = max(errs, key=errs.get)
    print(f"worst ARNN gradient tensor {worst}: {errs[worst]:.2e} of its max")
    assert errs[worst] < 5e-4, sorted(errs.items(), key=lambda kv: -kv[1])[:5]


def _token_pass_inputs(L=384, V=48, E=10, Hc=256, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, s=0.08):
        return ((torch.rand(*shape, generator=g) * 2 - 1) * s * scale).cuda()
    H = U = 256
    return dict(emb=rnd(V + 1, E, s=1.0), oc=rnd(L, Hc, s=1.0), W_ih0=rnd(4 * H, E + Hc), b_ih0=rnd(4 * H), W_hh0=rnd(4 * H, H),
                b_hh0=rnd(4 * H), W_ih1=rnd(4 * H, H), b_ih1=rnd(4 * H), W_hh1=rnd(4 * H, H), b_hh1=rnd(4 * H), W1=rnd(U, H),
                b1=rnd(U), W2=rnd(V, U, s=0.5), b2=rnd(V))


def _token_pass(x, mode, hc_init=None, first_tok=None):
    ops.set_option(14, mode)
    try:
        t = ops.arnn_generate(x["emb"], x["oc"], x["W_ih0"], x["b_ih0"], x["W_hh0"], x["b_hh0"], x["W_ih1"], x["b_ih1"], x["W_hh1"],
                              x["b_hh1"], x["W1"], x["b1"], x["W2"], x["b2"], hc_init=hc_init, first_tok=first_tok)
        torch.cuda.synchronize()
    finally:
        ops.set_option(14, 3)
    return t


@pytest.mark.parametrize("V,L", [(48, 384), (61, 100), (128, 37)])
def test_arnn_token_pass_persistent_kernel_equals_the_per_tick_launches(V, L):
    """inet_arnn_generate as ONE persistent launch (csrc/arnn_gen.hip: 13 resident workgroups, weights in registers, 8-byte {value, tick}
    granules between them) against round 4's four launches per tick: the same L tokens (fp32 both, other summation orders: a
    difference may only appear on a tick whose top two logits agree to rounding, and the test says so), for both placements of the
    workgroups, from zero state and -- forward_inpaint's use -- from a given state behind a given token."""
    x = _token_pass_inputs(L=L, V=V, seed=V)
    ref = _token_pass(x, 0)
    assert ops.chain_status() == 0
    assert int(ref.min()) >= 0 and int(ref.max()) < V and len(torch.unique(ref)) > 3      # (a sequence worth comparing)
    # 3 = the default: one XCD + XCD-local granule stores; 4 = the same REQUEST on consecutive workgroup ids, where the workgroups
    # sit on eight XCDs: their XCC-id check has to say no (a plain store is never seen across XCDs: the pass would time out)
    for mode in (1, 2, 3, 4):
        got = _token_pass(x, mode)
        assert ops.chain_status() == 0
        same = (got == ref)
        first = L if bool(same.all()) else int((~same).int().argmax())
        assert first == L, f"mode {mode}: token {first} differs ({int(got[first])} vs {int(ref[first])}): a near-tie of the two sums?"
    g = torch.Generator().manual_seed(5)
    hc = ((torch.rand(2, 2, 256, generator=g) * 2 - 1) * 0.5).cuda()
    ft = torch.tensor([V], dtype=torch.int64).cuda()             # a token the head cannot produce (row V of the embedding table)
    ref = _token_pass(x, 0, hc_init=hc, first_tok=ft)
    got = _token_pass(x, 1, hc_init=hc, first_tok=ft)
    assert torch.equal(got, _token_pass(x, 3, hc_init=hc, first_tok=ft))
    assert torch.equal(got, ref)
    assert not torch.equal(got, _token_pass(x, 1))                # (the state and the token matter)


def test_arnn_token_pass_with_nan_weights_stays_inside_the_vocabulary():
    """an all-NaN logit row used to yield token INT_MAX and the next tick's embedding gather faulted.  Both forms follow
    np.argmax now (a NaN is the maximum, lowest index among equals): tokens stay in [0, V), nothing faults, and the NaN shows up where
    the reference reports it -- Trainer.finish() / check_steps() raise ValueError('... has become nan')."""
    x = _token_pass_inputs(L=64, V=48)
    x["W2"][:] = float("nan")
    for mode in (0, 1, 3):
        t = _token_pass(x, mode)
        assert int(t.min()) == 0 and int(t.max()) == 0
    x = _token_pass_inputs(L=64, V=48)
    x["W_hh1"][5, 7] = float("nan")                              # the state becomes NaN at tick 1, the logits with it
    for mode in (0, 1, 3):
        t = _token_pass(x, mode)
        assert 0 <= int(t.min()) and int(t.max()) < 48
    assert ops.chain_status() == 0
    # through the trainer: a free-running step on NaN weights raises the reference's ValueError, no memory fault
    ds, model = build("full")
    model.train()
    trainer = AnticipationRNNGaussianRegTrainer(ds, model)
    fx = G.load("arnn_full")
    score = torch.from_numpy(fx["score"]).cuda()
    md = torch.from_numpy(fx["metadata"]).cuda()
    loc = torch.from_numpy(fx["constraints_loc"]).cuda()
    a, b = [int(v) for v in fx["ticks"]]
    model.param("linear_ouput_notes.0.weight")[:] = float("nan")
    trainer.zero_grad()
    weights, gen = model(score, md, loc, a, b, train=True, teacher_forcing=False)
    assert int(gen.min()) >= 0 and int(gen.max()) < 48
    loss, _ = trainer.mean_crossentropy_loss_and_accuracy_voices(weights, score[:, :, a:b].transpose(0, 1))
    loss.backward()
    trainer.step()
    with pytest.raises(ValueError, match="has become nan"):
        trainer.finish()


@pytest.mark.parametrize("tf", [True, False])
@pytest.mark.parametrize("window", [(48, 120), (0, 72), (300, 384)])
def test_arnn_trimmed_forward_changes_nothing(tf, window):
    """The trainers call the model with trim=True: the generation LSTMs stop behind the last unconstrained tick and the head runs on
    the unconstrained ticks only.  What the reference computes behind the window is read by nobody -- forward, and backward (zero
    gradient flows into those ticks) --, so weights, loss and EVERY gradient are those of the untrimmed pass
    (anticipation_rnn_gauss_reg_model.py:348-435), teacher-forced and free-running, for windows at the start, inside and at the end."""
    fx = G.load("arnn_small")
    score = torch.from_numpy(fx["score"]).cuda()
    md = torch.from_numpy(fx["metadata"]).cuda()
    L = score.shape[2]
    a, b = window
    b = min(b, L)
    loc = torch.ones_like(score)
    loc[:, :, a:b] = 0
    res = []
    for trim in (False, True):
        ds, model = build("small")
        model.train()
        trainer = AnticipationRNNGaussianRegTrainer(ds, model)
        trainer.zero_grad()
        weights, gen = model(score, md, loc, a, b, train=True, teacher_forcing=tf, trim=trim)
        loss, _ = trainer.mean_crossentropy_loss_and_accuracy_voices(weights, score[:, :, a:b].transpose(0, 1))
        loss.backward()
        ops.side_join()
        torch.cuda.synchronize()
        res.append((weights[0].detach().clone(), float(loss.detach()), model.grad.clone(), None if gen is None else gen.clone()))
    (w0, l0, g0, gen0), (w1, l1, g1, gen1) = res
    assert w0.shape == w1.shape == (score.shape[0], b - a, w0.shape[-1])
    assert G.rel_err(w1.cpu(), w0.cpu()) < 2e-6 and abs(l1 - l0) <= 1e-6 * abs(l0)
    assert G.rel_err(g1.cpu(), g0.cpu()) < 2e-5
    if not tf:
        assert torch.equal(gen1[:, :, :b], gen0[:, :, :b])        # (the trimmed pass does not generate the ticks behind the window)
