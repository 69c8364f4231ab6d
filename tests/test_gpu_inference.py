"""GPU parity of the widened rows through the public classes, against golden vectors
captured from the reference: LatentRNNAblations (past-only / future-only), MeasureVAE.forward_test,
VAETester.decode_mid_point, LatentRNNTester.generate (B = 1 inpainting, non-AR and auto-regressive),
ConstraintModelGaussianReg.forward_inpaint, AnticipationRNNBaseline(+Trainer)."""
import random
import types

import numpy as np
import pytest
import torch

from tests import golden_util as G

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from inpaintnet_amd import ops, synthetic
    from inpaintnet_amd.arnn import AnticipationRNNBaseline, AnticipationRNNBaselineTrainer
    from inpaintnet_amd.latent_rnn import LatentRNN
    from inpaintnet_amd.latent_rnn_ablations import LatentRNNAblations
    from inpaintnet_amd.latent_rnn_tester import LatentRNNTester
    from inpaintnet_amd.latent_rnn_trainer import LatentRNNTrainer
    from inpaintnet_amd.measure_vae import MeasureVAE
    from inpaintnet_amd.vae_tester import VAETester


def small_vae():
    c = G.CFGS["small"]
    ds = synthetic.SyntheticFolkDataset(num_notes=c["V"])
    vae = MeasureVAE(ds, note_embedding_dim=c["E"], encoder_hidden_size=c["H"], latent_space_dim=c["Z"],
                     decoder_hidden_size=c["H"], encoder_dropout_prob=0.0, decoder_dropout_prob=0.0)
    vae.load_state_dict(G.vae_params("small"))
    return c, ds, vae


@pytest.mark.parametrize("kind", ["past", "future"])
def test_latent_ablation_matches_reference(kind):
    fx = G.load(f"latent_small_abl_{kind}")
    c, ds, vae = small_vae()
    model = LatentRNNAblations(ds, vae, num_rnn_layers=2, rnn_hidden_size=c["H"], dropout=0.0, rnn_class=torch.nn.GRU,
                               auto_reg=False, teacher_forcing=True, type=kind)
    sd = G.latent_params_from_fixture(fx)
    assert set(model.state_dict()) == set(sd)
    model.load_state_dict(sd)
    assert "LatentRNN(" + kind in repr(model) and model.gen_hidden == c["H"]
    trainer = LatentRNNTrainer(ds, model, lr=1e-4)
    model.train()
    score = torch.from_numpy(fx["score"])
    n_past, n_target, n_future = [int(x) for x in fx["split"]]
    past, future, target = LatentRNNTrainer.split_score(score, n_past, n_future, n_target, 24)
    eps = tuple(torch.from_numpy(fx[k]).cuda() for k in ("eps_past", "eps_future", "eps_target"))
    trainer.zero_grad()
    w, s, gz = model(past, future, target, n_target, train=True, eps=eps)
    assert G.rel_err(gz.detach().cpu(), fx["gen_z"]) < 2e-4 and G.rel_err(w.detach().cpu(), fx["weights"]) < 2e-4
    ok = G.unique_rows(fx["margin"], 1e-3).reshape(score.shape[0], -1)
    assert np.array_equal(s.cpu().numpy()[:, 0][ok], fx["samples"][:, 0][ok])
    loss, acc = trainer.mean_crossentropy_loss_and_accuracy(w, target)
    loss.backward()
    assert abs(float(loss.detach()) - fx["loss_acc"][0]) < 1e-4 * abs(fx["loss_acc"][0]) and abs(float(acc) - fx["loss_acc"][1]) < 1e-6
    bad = []
    for k, _ in model.named_parameters():
        g, ref = model.param_grad(k).cpu().numpy(), fx["grad/" + k]
        err = np.abs(g - ref).max() / (np.abs(ref).max() + 1e-7)
        if not err < 1e-3:
            bad.append((k, float(err)))
    assert not bad, bad
    trainer.step()
    for k, _ in model.named_parameters():
        assert np.abs(model.param(k).cpu().numpy() - fx["after1/" + k]).max() < 1e-5, k


def test_forward_test_and_decode_mid_point(monkeypatch):
    fx = G.load("inference_small")
    c, ds, vae = small_vae()
    vae.eval()
    tok = torch.from_numpy(fx["ft_tokens"]).cuda()
    B, M = tok.shape[:2]
    # forward_test encodes all B*M measures in one call, rows ordered (b, measure): interleave the per-measure eps
    eps = torch.stack([torch.from_numpy(fx[f"ft_eps{i}"]) for i in range(M)], 1).reshape(B * M, -1).cuda()
    monkeypatch.setattr(torch, "randn_like", lambda t: eps)
    with torch.no_grad():
        w, s = vae.forward_test(tok)
    assert w.shape == fx["ft_weights"].shape and s.shape == fx["ft_samples"].shape and s.dtype == torch.int64
    assert G.rel_err(w.cpu(), fx["ft_weights"]) < 1e-4
    ok = G.unique_rows(fx["ft_margin"]).reshape(B, -1)
    assert np.array_equal(s.cpu().numpy()[:, 0][ok], fx["ft_samples"][:, 0][ok])
    monkeypatch.undo()
    tester = VAETester(ds, vae)
    mid = tester.decode_mid_point(torch.from_numpy(fx["mid_z1"]).cuda(), torch.from_numpy(fx["mid_z2"]).cuda(), 3)
    okm = G.unique_rows(fx["mid_margin"]).reshape(1, -1)
    assert mid.shape == fx["mid_tokens"].shape and np.array_equal(mid.cpu().numpy()[okm], fx["mid_tokens"][okm])
    _, inter = tester.test_interpolation(tok[0, :1], tok[1, :1], n=2)
    assert inter.shape == (1, 4 * 24)


@pytest.mark.parametrize("auto_reg", [False, True])
def test_generate_inpaints_like_the_reference(auto_reg, monkeypatch):
    fx = G.load("inference_small")
    tag = "gen_ar" if auto_reg else "gen_nar"
    c, ds, vae = small_vae()
    model = LatentRNN(ds, vae, num_rnn_layers=2, rnn_hidden_size=c["H"], dropout=0.0, rnn_class=torch.nn.GRU,
                      auto_reg=auto_reg, teacher_forcing=True)
    model.load_state_dict(G.latent_params("small", auto_reg))
    tester = LatentRNNTester(ds, model)
    score = torch.from_numpy(fx[f"{tag}_score"])
    past, future, target = LatentRNNTrainer.split_score(score, 5, 8, 3, 24)
    # generate() encodes past | future (no target); the auto-regressive path then re-encodes each generated measure
    queue = [torch.cat((torch.from_numpy(fx[f"{tag}_eps_past"]), torch.from_numpy(fx[f"{tag}_eps_future"])), 0).cuda()]
    if auto_reg:
        queue += [torch.from_numpy(fx[f"{tag}_eps_ar{i}"]).cuda() for i in range(3)]
    monkeypatch.setattr(torch, "randn_like", lambda t: queue.pop(0))
    _, full, _ = tester.generate(past, future, target, 3, eval=True)
    assert full.shape == (1, 16, 24) and full.dtype == torch.int64
    w = tester.last_weights
    okg = G.unique_rows(fx[f"{tag}_margin"], 1e-4).reshape(1, 3, 24)
    got = full.cpu().numpy()
    assert np.array_equal(got[:, :5], fx[f"{tag}_full"][:, :5]) and np.array_equal(got[:, 8:], fx[f"{tag}_full"][:, 8:])
    if not auto_reg or np.array_equal(got, fx[f"{tag}_full"]):
        assert G.rel_err(w.cpu(), fx[f"{tag}_weights"]) < 2e-4
        assert np.array_equal(got[:, 5:8][okg], fx[f"{tag}_full"][:, 5:8][okg])
    else:
        assert G.rel_err(w.cpu()[:, 0], fx[f"{tag}_weights"][:, 0]) < 2e-4
    assert 0.0 <= tester.last_eval[1] <= 1.0
    # contexts may be omitted: 3 START measures / 1 END measure are substituted (latent_rnn_tester.py:268-296)
    ds.note2index_dicts[0].update({"START": 1, "END": 2, "rest": 3})
    queue[:] = [torch.zeros(4, c["Z"]).cuda()] + [torch.zeros(1, c["Z"]).cuda()] * 3
    _, full2, orig = tester.generate(None, None, None, 2)
    assert full2.shape == (1, 3 + 2 + 1, 24) and orig is None
    assert int(full2[0, 0, 0]) == 1 and int(full2[0, -1, 0]) == 2


def test_arnn_forward_inpaint_and_baseline():
    fx = G.load("arnn_inpaint_small")
    c = G.ARNN_CFGS["small"]
    ds = synthetic.SyntheticFolkDataset(num_notes=c["V"])
    ds.metadatas = [types.SimpleNamespace(num_values=6), types.SimpleNamespace(num_values=6)]
    model = AnticipationRNNBaseline(ds, note_embedding_dim=c["E"], metadata_embedding_dim=c["Em"],
                                    num_lstm_constraints_units=c["H"], num_lstm_generation_units=c["H"],
                                    linear_hidden_size=c["LH"], num_layers=2, dropout_input_prob=0.0, dropout_prob=0.0,
                                    unary_constraint=True, teacher_forcing=True)
    model.load_state_dict(G.arnn_params("small"))
    assert repr(model).startswith("AnticipationRNNBaseline(") and repr(model).endswith(",tf")
    model.eval()
    score, md, loc = (torch.from_numpy(fx[k]).cuda() for k in ("score", "metadata", "constraints_loc"))
    a, b = [int(x) for x in fx["ticks"]]
    with torch.no_grad():
        w, gen = model.forward_inpaint(score, md, loc, a, b)
    assert w[0].shape == fx["inpaint_weights"].shape and gen.shape == fx["inpaint_gen"].shape
    g = gen.cpu().numpy()
    if np.array_equal(g, fx["inpaint_gen"]):
        assert G.rel_err(w[0].cpu(), fx["inpaint_weights"]) < 2e-4
    else:
        first = int(np.argmax(g[0, 0] != fx["inpaint_gen"][0, 0]))
        assert fx["inpaint_margin_row0"][first - a] < 1e-4
        assert G.rel_err(w[0].cpu()[:, :first - a], fx["inpaint_weights"][:, :first - a]) < 2e-4
    assert ops.chain_status() == 0
    # baseline trainer: Bernoulli constraint mask, the reference's draws under the same seeds
    trainer = AnticipationRNNBaselineTrainer(ds, model)
    random.seed(int(fx["baseline_seed"]))
    torch.manual_seed(int(fx["baseline_seed"]))
    for i in range(3):
        out = trainer.process_batch_data((score.cpu().int(), md.cpu().int()))
        assert out[3] is None and out[4] is None and out[2].dtype == torch.int64 and out[2].is_cuda
        assert np.array_equal(out[2].cpu().numpy(), fx["baseline_locs"][i])
    model.train()
    trainer.zero_grad()
    loss, acc = trainer.loss_and_acc_for_batch(out, 0, train=True)
    loss.backward()
    trainer.step()
    assert np.isfinite(float(loss.detach()))


@pytest.mark.parametrize("name,B", [("full", 5), ("full", 1), ("full", 16), ("full", 29), ("pk", 7), ("pk", 32),
                                    ("v61", 5), ("v93", 33), ("pk20", 9), ("full125", 3),
                                    ("full", 300), ("full", 512), ("pk", 450)])     # 64-row groups (MS = 4), ragged and full
def test_fused_decode_kernel_matches_per_tick_path(name, B):
    """The fused free-running decode (csrc/decode_chain.hip: 24 ticks x [layer 0, layer 1, projection + argmax] in one
    launch) against the per-tick launches and the oracle: logits to fp32 round-off, tokens exact on rows with a margin."""
    from oracle import torch_ref as O
    from tests.test_gpu_kernels import pack
    # vocabularies that are not a multiple of 16 (20, 61, 93, 125): the kernel pads its last column block (csrc/decode_chain.hip)
    c = dict(G.CFGS[{"pk20": "pk", "full125": "full"}.get(name, name)])
    if name == "pk":
        c["V"] = 32
    elif name == "full125":
        c["V"] = 125
    cfg = ops.vae_config(c["V"], c["E"], c["H"], c["Z"], c["H"])
    V = cfg.num_notes
    table, total = ops.vae_param_table(cfg)
    from inpaintnet_amd import layout
    shapes = layout.vae_param_shapes(V, c["E"], c["H"], c["Z"], c["H"])
    P = {k: torch.from_numpy(synthetic.det_param(k, s)) for k, s in shapes.items()}
    params = pack(table, total, P)
    z = torch.from_numpy(synthetic.det_normal(f"fused/{name}/{B}", (B, c["Z"]))).cuda()
    ops.prof_enable(True)
    w1, s1, _ = ops.decoder_fwd(cfg, z, None, False, params)
    torch.cuda.synchronize()
    import csv, tempfile, os
    with tempfile.TemporaryDirectory() as td:
        ops.prof_dump(os.path.join(td, "l.csv"))
        labels = [r["label"] for r in csv.DictReader(open(os.path.join(td, "l.csv")))]
    ops.prof_enable(False)
    # (up to sixteen measures at H = 512 run the register-resident persistent launch of csrc/decode_b1.hip, everything else decode_chain.hip)
    want = "decode_b1" if (B <= 16 and c["H"] == 512) else "decode_chain"
    assert any(l.startswith(want) for l in labels), sorted(set(labels))
    assert ops.chain_status() == 0
    ops.set_option(4, 0)
    try:
        w0, s0, _ = ops.decoder_fwd(cfg, z, None, False, params)
        torch.cuda.synchronize()
    finally:
        ops.set_option(4, 1)
    with torch.no_grad():
        wr, sr = O.decoder_forward(P, z.cpu(), None, False, feed_tokens=s1.cpu()[:, 0])
    top2 = torch.topk(wr, 2, dim=-1).values
    ok = ((top2[..., 0] - top2[..., 1]) > 1e-4).numpy()
    assert G.rel_err(w1.cpu(), wr) < 2e-5
    assert np.array_equal(s1.cpu().numpy()[:, 0][ok], sr.numpy()[:, 0][ok])
    same = np.array_equal(s1.cpu().numpy(), s0.cpu().numpy())
    if same:
        assert G.rel_err(w1.cpu(), w0.cpu()) < 2e-5


def test_fused_training_decode_of_512_rows_in_one_launch():
    """The free-running half of a training step over 512 rows (LatentRNN decodes 128 x 4 target measures per step): ONE launch
    of the 64-row build (`decode_chain_train ms4`) -- forward outputs, tokens and every gradient against the per-tick launches
    (inet_set_option key 4 = 0) with the same dropout masks."""
    import csv, os, tempfile
    from tests.test_gpu_kernels import pack
    c = G.CFGS["full"]
    cfg = ops.vae_config(c["V"], c["E"], c["H"], c["Z"], c["H"])
    table, total = ops.vae_param_table(cfg)
    params = pack(table, total, G.vae_params("full"))
    B, H, T = 512, c["H"], 24
    z = torch.from_numpy(synthetic.det_normal("fused_train/512", (B, c["Z"]))).cuda()
    mb = ops.dropout_mask((4, B, H), 0.5, 11, 0, "cuda")
    mt = ops.dropout_mask((T, B, H), 0.5, 11, 4 * B * H, "cuda")
    res = []
    try:
        for chain in (1, 0):
            ops.set_option(4, chain)
            ops.prof_enable(True)
            w, smp, ws = ops.decoder_fwd(cfg, z, None, False, params, mask_beat=mb, mask_tick=mt, save=True)
            torch.cuda.synchronize()
            with tempfile.TemporaryDirectory() as td:
                ops.prof_dump(os.path.join(td, "l.csv"))
                labels = [r["label"] for r in csv.DictReader(open(os.path.join(td, "l.csv")))]
            ops.prof_enable(False)
            fused = [l for l in labels if l.startswith("decode_chain_train")]
            assert (fused == [f"decode_chain_train ms4 T{T} B{B} H{H} V{c['V']}"]) if chain else not fused, fused
            g = torch.zeros_like(params)
            dw = torch.from_numpy(synthetic.det_normal("fused_train/dw", (B, T, c["V"]))).cuda() * 1e-3
            dz = ops.decoder_bwd(cfg, dw, w, smp, params, g, mb, mt, ws)
            torch.cuda.synchronize()
            # the branch every SELU / ReLU element took (tests/test_gpu_kernels.py::_vae_step_with_kinks), per row
            br = torch.cat([(ops.ws_field(cfg, ws, B, 1, "hb0").view(B, 2 * H) > 0),
                            (ops.ws_field(cfg, ws, B, 1, "ht0").view(4, B, 2 * H) > 0).permute(1, 0, 2).reshape(B, -1),
                            (ops.ws_field(cfg, ws, B, 1, "c_all").view(4, B, H) > 0).permute(1, 0, 2).reshape(B, -1),
                            (w > 0).flatten(1)], 1)
            res.append((w, smp, g, dz, br))
    finally:
        ops.set_option(4, 1)
        ops.prof_enable(False)
    assert ops.chain_status() == 0
    (w1, s1, g1, dz1, br1), (w0, s0, g0, dz0, br0) = res
    # comparable rows: the same 24 tokens (an argmax tie may flip a row) and the same branch at every SELU / ReLU (a
    # pre-activation within round-off of 0 -- the beat chains differ by that between the two runs -- sits on the other side of
    # the kink: its gradient is another one; row 148 of this input)
    same = (s1 == s0).all(dim=-1).squeeze(1) & (br1 == br0).all(dim=1)
    assert same.float().mean().item() > 0.99
    assert G.rel_err(w1[same].cpu(), w0[same].cpu()) < 2e-5
    assert G.rel_err(dz1[same].cpu(), dz0[same].cpu()) < 5e-5
    assert G.rel_err(g1.cpu(), g0.cpu()) < (5e-5 if bool(same.all()) else 5e-3)


def test_decoder_multinomial_sampling():
    """HierarchicalDecoder.sampling = 'multinomial' (decoder.py:506-509): the fed-back token of every free-running tick is
    drawn from softmax(weights).  Tick 0 does not depend on any draw, so its logits equal the argmax run's; the draws are
    valid tokens, reproducible for one dropout seed / call counter, and follow the model's own softmax on tick 0."""
    from inpaintnet_amd import measure_vae as MV
    c, ds, vae = small_vae()
    vae.eval()
    B = 512
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, c["Z"], generator=g).cuda()
    dummy = torch.zeros(B, 24, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        w_arg, s_arg = vae.decoder(z, dummy, train=False)
        vae.decoder.sampling = 'multinomial'
        MV.set_dropout_seed(11)
        w1, s1 = vae.decoder(z, dummy, train=False)
        MV.set_dropout_seed(11)
        w2, s2 = vae.decoder(z, dummy, train=False)
        w3, s3 = vae.decoder(z, dummy, train=False)              # next call: another stream
        vae.decoder.sampling = 'argmax'
    assert torch.equal(s1, s2) and torch.equal(w1, w2) and not torch.equal(s1, s3)
    assert G.rel_err(w1[:, 0].cpu(), w_arg[:, 0].cpu().numpy()) < 1e-6
    V = w1.shape[-1]
    assert int(s1.min()) >= 0 and int(s1.max()) < V and s1.shape == s_arg.shape
    assert not torch.equal(s1, s_arg)
    # tick 0: the same z in every row -> same distribution in every row -> frequencies follow its softmax
    zc = z[:1].repeat(4096, 1).contiguous()
    with torch.no_grad():
        vae.decoder.sampling = 'multinomial'
        wc, sc = vae.decoder(zc, torch.zeros(4096, 24, dtype=torch.int64, device="cuda"), train=False)
        vae.decoder.sampling = 'argmax'
    p = torch.softmax(wc[0, 0].double(), 0).cpu().numpy()
    counts = np.bincount(sc[:, 0, 0].cpu().numpy(), minlength=V).astype(np.float64)
    keep = p * 4096 > 5
    chi2 = float((((counts - 4096 * p) ** 2) / (4096 * p))[keep].sum())
    assert chi2 < keep.sum() + 6.0 * np.sqrt(2.0 * keep.sum()), chi2


def test_fused_decode_matches_per_tick_path_repeatedly():
    """Stress for the fused decode kernel's hand-off protocol: the same free-running decode (inference and training form:
    backward saves, dropout masks) through the fused kernel and through the per-tick launches, many times, over batch sizes
    that give 1..8 groups, with the allocator's pool dirtied in between.  A premature release in the third hand-off of
    a tick (members without a logits tile arriving a third time before slow members had arrived twice) showed up as one
    wrong 16 x 16 logits tile at tick 0 in ~5 % of the B = 256 calls; the full-size fixtures alone hit it once in ~7
    suite runs."""
    from tests.test_gpu_kernels import pack
    c = G.CFGS["full"]
    cfg = ops.vae_config(c["V"], c["E"], c["H"], c["Z"], c["H"])
    table, total = ops.vae_param_table(cfg)
    params = pack(table, total, G.vae_params("full"))
    g = torch.Generator().manual_seed(17)
    rng = np.random.RandomState(3)
    try:
        for it in range(240):
            B = [256, 5, 256, 96, 192, 32][it % 6]
            junk = torch.empty(int(rng.randint(1, 48)) << 20, device="cuda").uniform_(-100, 100)
            del junk
            z = torch.randn(B, c["Z"], generator=g).cuda()
            train = it % 3 != 0
            mb = ops.dropout_mask((4, B, c["H"]), 0.5, it, 0, "cuda") if it % 2 and train else None
            mt = ops.dropout_mask((24, B, c["H"]), 0.5, it, 10 ** 6, "cuda") if it % 2 and train else None
            outs = []
            for chain in (1, 0):
                ops.set_option(4, chain)
                w, s, _ = ops.decoder_fwd(cfg, z, None, False, params, mb, mt, save=train)
                outs.append((w, s))
            (w1, s1), (w0, s0) = outs
            scale = float(w0.abs().max())
            # tick 0 depends on no sampled token: every row must agree.  Later ticks: a row whose top-2 logits tie within
            # round-off may feed back another token on the two paths (the beat layers in front of the decode also run as
            # chain / per-step kernels and differ in the last bits) and then legitimately follows another trajectory
            assert float((w1[:, 0] - w0[:, 0]).abs().max()) / scale < 2e-5, (it, B, train)
            same = (s1 == s0).all(dim=-1).reshape(-1)
            assert float(same.float().mean()) >= 0.97, (it, B, float(same.float().mean()))
            assert float((w1[same] - w0[same]).abs().max()) / scale < 2e-5, (it, B, train)
    finally:
        ops.set_option(4, 1)
    assert ops.chain_status() == 0


@pytest.mark.parametrize("V,B", [(48, 1), (20, 1), (61, 1), (93, 1), (128, 1), (48, 2), (20, 2), (48, 3), (48, 4), (61, 4), (125, 3),
                                 (48, 5), (48, 8), (20, 7), (48, 10), (48, 11), (61, 13), (48, 16), (100, 16),
                                 (20, 12), (32, 16), (128, 14)])     # (shared recurrent groups with the merged build: V <= 32; with the widest head)
def test_decode_b1_persistent_kernel_matches_decode_chain_and_oracle(V, B):
    """(One row with V <= 64 / two rows with V <= 32: the merged build, where every layer-1 workgroup also runs layer 0's cell, the head and
    the argmax for itself -- one hand-off per tick; three to sixteen rows: teams of the tick path's 49 workgroups behind the beat path's
    launches, two rows each up to ten rows, four rows each beyond.)
    One measure (the call the north_star prices: LatentRNNTester.generate, decode_mid_point) -- and the two to four measures of the
    reference's non-auto-regressive inpainting call -- through csrc/decode_b1.hip -- 129
    resident workgroups (49 for the ticks, 80 for the beat path folded into the same launch), every weight matrix in registers, 8-byte {value, tick} granules, two hand-offs per tick, the rows looped inside every phase -- against the
    32-member exchange kernel of csrc/decode_chain.hip (inet_set_option key 15 = 0) and the oracle: logits to fp32 round-off, tokens
    exact on ticks with a margin; both workgroup placements; repeated with a dirtied allocator pool (the granules are zeroed per
    call by the prologue launch: stale tags of an earlier call must never match)."""
    from oracle import torch_ref as O
    from tests.test_gpu_kernels import pack
    from inpaintnet_amd import layout
    c = dict(G.CFGS["full"], V=V)
    cfg = ops.vae_config(c["V"], c["E"], c["H"], c["Z"], c["H"])
    table, total = ops.vae_param_table(cfg)
    shapes = layout.vae_param_shapes(V, c["E"], c["H"], c["Z"], c["H"])
    P = {k: torch.from_numpy(synthetic.det_param(k, s)) for k, s in shapes.items()}
    params = pack(table, total, P)
    rng = np.random.RandomState(V)
    try:
        for it in range(6 if B == 1 else 3):
            z = torch.from_numpy(synthetic.det_normal(f"b1/{V}/{B}/{it}", (B, c["Z"]))).cuda()
            junk = torch.empty(int(rng.randint(1, 32)) << 20, device="cuda").uniform_(-100, 100)
            del junk
            outs = {}
            for mode in (5, 4, 3, 2, 1, 0):                  # 5 = test hook: 4's request for XCD-local copies on consecutive workgroup ids (the XCC-id check must refuse); 4 = default: 3 with the critical workgroups on one XCD and XCD-local copies; 3 = the beat path folded into the launch, 2 / 1 = tick path only
                ops.set_option(15, mode)
                w, s_, _ = ops.decoder_fwd(cfg, z, None, False, params)
                torch.cuda.synchronize()
                outs[mode] = (w.clone(), s_.clone())
            assert ops.chain_status() == 0
            # placement and the XCD-local copies change no value: the same instructions on the same operands (bit for bit where the
            # beat path is folded into the launch; with its own launches in front their split-K sums differ by round-off per run)
            if B <= 6:
                assert torch.equal(outs[4][1], outs[3][1]) and torch.equal(outs[4][0], outs[3][0]), (V, B, it)
                # ... and where the critical workgroups do NOT share an XCD the check refuses the XCD-local copies: had it not, plain stores
                # would never be seen across XCDs and every wait would have run into its bound (chain_status above)
                assert torch.equal(outs[5][1], outs[3][1]) and torch.equal(outs[5][0], outs[3][0]), (V, B, it)
            else:
                assert float((outs[4][0][:, 0] - outs[3][0][:, 0]).abs().max()) < 2e-5 * float(outs[3][0].abs().max())
            with torch.no_grad():
                wr, sr = O.decoder_forward(P, z.cpu(), None, False, feed_tokens=outs[3][1].cpu()[:, 0])
            top2 = torch.topk(wr, 2, dim=-1).values
            ok = ((top2[..., 0] - top2[..., 1]) > 1e-4).numpy()
            for mode in (3, 2, 1):
                w, s_ = outs[mode]
                if not torch.equal(s_, outs[3][1]):          # (another trajectory behind a near-tie: compared against the oracle alone)
                    continue
                assert G.rel_err(w.cpu(), wr) < 2e-5, (V, it, mode)
                assert np.array_equal(s_.cpu().numpy()[:, 0][ok], sr.numpy()[:, 0][ok]), (V, it, mode)
                assert int(s_.min()) >= 0 and int(s_.max()) < V
            if torch.equal(outs[3][1], outs[0][1]):
                assert G.rel_err(outs[3][0].cpu(), outs[0][0].cpu()) < 2e-5
            # placement changes nothing: bit-identical where the beat path's launches in front are deterministic (one row: wave-per-
            # column products); with more rows their split-K products sum in launch order, so round-off there
            assert torch.equal(outs[2][1], outs[1][1])
            if B == 1:
                assert torch.equal(outs[2][0], outs[1][0])
            else:
                assert G.rel_err(outs[2][0].cpu(), outs[1][0].cpu()) < 2e-6
            # tick 0 depends on no sampled token: the folded beat path must reproduce the launches' beat 0 to round-off
            assert float((outs[3][0][:, 0] - outs[2][0][:, 0]).abs().max()) < 2e-5 * float(outs[2][0].abs().max())
    finally:
        ops.set_option(15, 4)


def test_decode_every_batch_size_up_to_sixteen_against_the_exchange_kernel():
    """Every B from 1 to 16 through csrc/decode_b1.hip's plan for it (one team with the beat path folded in; two-row teams, with or
    without the beat path's workgroups in the launch; four-row teams) against decode_chain.hip's exchange kernel (option key 15 = 0):
    tick 0 equal to round-off on every row (it depends on no sampled token), whole rows equal wherever the two paths sampled the same
    tokens, and that is nearly everywhere; twice per size, the second time on a dirtied allocator pool."""
    from tests.test_gpu_kernels import pack
    from inpaintnet_amd import layout
    c = G.CFGS["full"]
    V = c["V"]
    cfg = ops.vae_config(V, c["E"], c["H"], c["Z"], c["H"])
    table, total = ops.vae_param_table(cfg)
    shapes = layout.vae_param_shapes(V, c["E"], c["H"], c["Z"], c["H"])
    P = {k: torch.from_numpy(synthetic.det_param(k, s)) for k, s in shapes.items()}
    params = pack(table, total, P)
    same_rows = total_rows = 0
    try:
        for B in range(1, 17):
            for it in range(2):
                z = torch.from_numpy(synthetic.det_normal(f"every/{B}/{it}", (B, c["Z"]))).cuda()
                if it:
                    junk = torch.empty(16 << 20, device="cuda").uniform_(-100, 100)
                    del junk
                out = {}
                for mode in (4, 3, 0):
                    ops.set_option(15, mode)
                    w, s_, _ = ops.decoder_fwd(cfg, z, None, False, params)
                    torch.cuda.synchronize()
                    out[mode] = (w.clone(), s_.clone())
                assert ops.chain_status() == 0, B
                if B <= 6:                                   # (the default: 3 + placement; bit-identical with the beat path folded in)
                    assert torch.equal(out[4][0], out[3][0]) and torch.equal(out[4][1], out[3][1]), B
                else:
                    assert float((out[4][0][:, 0] - out[3][0][:, 0]).abs().max()) < 2e-5 * float(out[0][0].abs().max()), B
                (w3, s3), (w0, s0) = out[3], out[0]
                scale = float(w0.abs().max())
                assert int(s3.min()) >= 0 and int(s3.max()) < V and s3.shape == s0.shape
                assert float((w3[:, 0] - w0[:, 0]).abs().max()) < 2e-5 * scale, B
                same = (s3 == s0).all(dim=-1).reshape(-1)
                same_rows += int(same.sum()); total_rows += B
                if bool(same.any()):
                    assert float((w3[same] - w0[same]).abs().max()) < 2e-5 * scale, B
    finally:
        ops.set_option(15, 4)
    assert same_rows >= 0.9 * total_rows, (same_rows, total_rows)
