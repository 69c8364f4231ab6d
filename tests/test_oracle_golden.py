"""The CPU oracle (oracle/torch_ref.py) against golden vectors captured from the
upstream reference (oracle/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as O
from tests import golden_util as G

torch.set_num_threads(4)


@pytest.mark.parametrize("name", ["small", "mid", "full"])
def test_encoder(name):
    fx = G.load("vae_" + name)
    P = G.vae_params(name, fx)
    tok = torch.from_numpy(fx["tokens"])
    with torch.no_grad():
        mu, ls = O.encoder_forward(P, tok)
        mu2, ls2 = O.encoder_forward(P, tok, fast=True)
    assert G.rel_err(mu, fx["enc_mu"]) < 2e-5
    assert G.rel_err(ls, fx["enc_logsigma"]) < 2e-5
    assert G.rel_err(mu2, fx["enc_mu"]) < 2e-5
    assert G.rel_err(ls2, fx["enc_logsigma"]) < 2e-5


@pytest.mark.parametrize("name", ["small", "mid", "full"])
def test_decoder_eval_and_teacher_forced(name):
    fx = G.load("vae_" + name)
    P = G.vae_params(name, fx)
    tok = torch.from_numpy(fx["tokens"])
    z = torch.from_numpy(fx["dec_z"])
    with torch.no_grad():
        w, s = O.decoder_forward(P, z, tok, teacher_forced=False)
    assert G.rel_err(w, fx["dec_eval_weights"]) < 5e-5
    ok = G.unique_rows(fx["dec_eval_margin"])
    assert ok.mean() > 0.5
    assert s.shape == fx["dec_eval_samples"].shape
    assert np.array_equal(s.numpy()[:, 0][ok], fx["dec_eval_samples"][:, 0][ok])
    with torch.no_grad():
        w, s = O.decoder_forward(P, z, tok, teacher_forced=True)
    assert G.rel_err(w, fx["dec_tf_weights"]) < 5e-5
    assert np.array_equal(s.numpy(), fx["dec_tf_samples"])


@pytest.mark.parametrize("name", ["small", "mid", "full"])
@pytest.mark.parametrize("mode", ["tf", "fr"])
def test_vae_train_steps(name, mode):
    """5 Adam steps on one batch: loss/CE/KL/accuracy trajectory, first-step
    gradients, parameters after step 1 and 5."""
    fx = G.load("vae_" + name)
    P = G.vae_params(name, fx)
    for p in P.values():
        p.requires_grad_(True)
    tok = torch.from_numpy(fx["tokens"])
    m = {k: torch.zeros_like(p) for k, p in P.items()}
    v = {k: torch.zeros_like(p) for k, p in P.items()}
    ref_losses = fx[f"step_{mode}_losses"]
    for step in range(5):
        eps = torch.from_numpy(fx[f"step_{mode}_eps{step}"])
        for p in P.values():
            p.grad = None
        w, s, mu, ls, z = O.vae_forward(P, tok, eps, teacher_forced=(mode == "tf"))
        loss, ce, kl, acc = O.vae_loss(w, tok, mu, ls)
        loss.backward()
        got = np.array([loss.item(), ce.item(), kl.item(), acc.item()])
        assert np.allclose(got[:3], ref_losses[step][:3], rtol=2e-5, atol=1e-7), (step, got, ref_losses[step])
        if step == 0:
            assert G.rel_err(w.detach(), fx[f"step_{mode}_weights"]) < 5e-5
            assert G.rel_err(z.detach(), fx[f"step_{mode}_z"]) < 5e-5
            ok = G.unique_rows(fx[f"step_{mode}_margin"])
            assert np.array_equal(s.numpy()[:, 0][ok], fx[f"step_{mode}_samples"][:, 0][ok])
            assert abs(got[3] - ref_losses[0][3]) < 1e-6
            for k, p in P.items():
                g = p.grad.numpy()
                if name != "full":
                    ref = fx[f"step_{mode}_grad/{k}"]
                    assert np.abs(g - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-6), k
                else:
                    rn = fx[f"step_{mode}_gradnorm/{k}"]
                    gn = np.sqrt((g.astype(np.float64) ** 2).sum())
                    assert abs(gn - rn) <= 2e-4 * rn + 1e-9, k
                    ref = fx[f"step_{mode}_gradhead/{k}"]
                    got_h = g.reshape(-1)[:64]
                    assert np.abs(got_h - ref).max() <= 2e-4 * (np.abs(g).max() + 1e-9), k
        with torch.no_grad():
            O.adam_step(P, {k: p.grad for k, p in P.items()}, m, v, step + 1)
        if step in (0, 4):
            for k, p in P.items():
                # Adam moves every parameter by ~lr per step: compare the displacement
                if name == "small":
                    ref = fx[f"step_{mode}_after{step + 1}/{k}"]
                    assert np.abs(p.detach().numpy() - ref).max() < 2e-6, (k, step)
                else:
                    ref = fx[f"step_{mode}_after{step + 1}/head/{k}"]
                    assert np.abs(p.detach().numpy().reshape(-1)[:64] - ref).max() < 5e-6, (k, step)


@pytest.mark.parametrize("name", ["small", "full"])
@pytest.mark.parametrize("variant", ["nar_fr", "ar_tf", "ar_fr"])
def test_latent_rnn(name, variant):
    fx = G.load(f"latent_{name}_{variant}")
    auto_reg = variant.startswith("ar")
    tf = variant.endswith("tf")
    P = G.latent_params(name, auto_reg)
    train_keys = [k for k in P if not k.startswith("vae_model.")]
    for k in train_keys:
        P[k].requires_grad_(True)
    score = torch.from_numpy(fx["score"])
    n_past, n_target, n_future = [int(x) for x in fx["split"]]
    past, future, target = O.split_score(score, n_past, n_future, n_target)
    eps_ar = None
    if auto_reg and not tf:
        eps_ar = [torch.from_numpy(fx[f"eps_ar{i}"]) for i in range(n_target)]
    w, s, gz = O.latent_forward(P, past, future, target,
                                torch.from_numpy(fx["eps_past"]), torch.from_numpy(fx["eps_future"]),
                                torch.from_numpy(fx["eps_target"]), auto_reg=auto_reg,
                                teacher_forcing=tf, eps_ar=eps_ar)
    assert G.rel_err(gz.detach(), fx["gen_z"]) < 1e-4
    ok = G.unique_rows(fx["margin"], 1e-3).reshape(s.shape[0], -1)
    if not (auto_reg and not tf):
        # free-running AR feeds sampled tokens back through the encoder: one near-tie flips everything after
        assert G.rel_err(w.detach(), fx["weights"]) < 1e-4
        assert np.array_equal(s.numpy()[:, 0][ok], fx["samples"][:, 0][ok])
    # latent-MSE / L1 of the generated latents against the encoder's z of the target measures, as the reference's
    # Trainer.mean_mse_loss_rnn / mean_l1_loss_rnn computed them (utils/trainer.py:308-342; north_star: within 1e-4 rel)
    with torch.no_grad():
        z_t = O.latent_get_z(P, target, torch.from_numpy(fx["eps_target"]))
    assert G.rel_err(z_t, fx["z_target"]) < 1e-4
    if not (auto_reg and not tf) or np.array_equal(s.numpy(), fx["samples"]):
        mse = float(((gz.detach() - z_t) ** 2).mean())
        l1 = float((gz.detach() - z_t).abs().mean())
        assert abs(mse - float(fx["mse_gen_target"])) <= 1e-4 * float(fx["mse_gen_target"])
        assert abs(l1 - float(fx["l1_gen_target"])) <= 1e-4 * float(fx["l1_gen_target"])
    loss, acc = O.latent_loss(w, target)
    loss.backward()
    if not (auto_reg and not tf) or np.array_equal(s.numpy(), fx["samples"]):
        assert abs(loss.item() - fx["loss_acc"][0]) < 2e-5 * abs(fx["loss_acc"][0])
        assert abs(acc.item() - fx["loss_acc"][1]) < 1e-6
        for k in train_keys:
            g = P[k].grad.numpy()
            if name == "small":
                ref = fx["grad/" + k]
                assert np.abs(g - ref).max() <= 5e-4 * (np.abs(ref).max() + 1e-7), k
            else:
                rn = fx["gradnorm/" + k]
                gn = np.sqrt((g.astype(np.float64) ** 2).sum())
                assert abs(gn - rn) <= 5e-4 * rn + 1e-9, k
    for k in P:
        if k.startswith("vae_model."):
            assert P[k].grad is None


def test_split_helpers():
    fx = G.load("split_helpers")
    score = torch.from_numpy(fx["score"])
    for (p, t, f) in [(6, 4, 6), (1, 2, 13), (8, 6, 2)]:
        a, b, c = O.split_score(score, p, f, t)
        assert np.array_equal(a.numpy(), fx[f"past_{p}_{t}_{f}"])
        assert np.array_equal(b.numpy(), fx[f"future_{p}_{t}_{f}"])
        assert np.array_equal(c.numpy(), fx[f"target_{p}_{t}_{f}"])
        assert a.dtype == torch.int64


@pytest.mark.parametrize("name", ["small", "full"])
def test_arnn_teacher_forced_and_free_running(name):
    """AnticipationRNN (config 5): teacher-forced loss/accuracy/gradients/Adam step, free-running forward."""
    fx = G.load("arnn_" + name)
    P = G.arnn_params(name, fx)
    for p in P.values():
        p.requires_grad_(True)
    score = torch.from_numpy(fx["score"])
    md = torch.from_numpy(fx["metadata"])
    loc = torch.from_numpy(fx["constraints_loc"])
    a, b = [int(x) for x in fx["ticks"]]
    w_all, _ = O.arnn_forward(P, score, md, loc, teacher_forcing=True)
    assert G.rel_err(w_all.detach(), fx["tf_weights_all"]) < 5e-5
    loss, acc = O.arnn_loss(w_all[:, a:b], score[:, 0, a:b])
    assert abs(loss.item() - fx["tf_loss_acc"][0]) < 2e-5 * abs(fx["tf_loss_acc"][0])
    assert abs(acc.item() - fx["tf_loss_acc"][1]) < 1e-6
    loss.backward()
    m = {k: torch.zeros_like(p) for k, p in P.items()}
    v = {k: torch.zeros_like(p) for k, p in P.items()}
    for k, p in P.items():
        g = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, dtype=np.float32)
        if name == "small":
            key = "tf_grad/" + k
            if key in fx.files:
                ref = fx[key]
                assert np.abs(g - ref).max() <= 5e-4 * (np.abs(ref).max() + 1e-7), k
        else:
            key = "tf_gradnorm/" + k
            if key in fx.files:
                rn = float(fx[key])
                assert abs(float(np.sqrt((g.astype(np.float64) ** 2).sum())) - rn) <= 5e-4 * rn + 1e-9, k
    with torch.no_grad():
        O.adam_step(P, {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in P.items()}, m, v, 1)
    for k, p in P.items():
        if name == "small":
            assert np.abs(p.detach().numpy() - fx["tf_after1/" + k]).max() < 2e-6, k
        else:
            assert np.abs(p.detach().numpy().reshape(-1)[:64] - fx["tf_after1head/" + k]).max() < 5e-6, k
    # free running: forward only (the reference's backward of this path fails on CPU under torch 2.x)
    P2 = G.arnn_params(name)
    with torch.no_grad():
        w_all, gen = O.arnn_forward(P2, score, md, loc, teacher_forcing=False)
    # tokens are chosen from batch element 0: compare up to the first near-tie
    ok = fx["fr_margin_row0"] > 1e-4
    first_bad = int(np.argmin(ok)) if not ok.all() else len(ok)
    same = np.array_equal(gen.numpy()[:, :first_bad], fx["fr_gen"][:, 0, :first_bad]) if first_bad > 0 else True
    assert same
    if first_bad == len(ok):
        assert G.rel_err(w_all, fx["fr_weights_all"]) < 1e-4
        assert G.rel_err(w_all[:, a:b], fx["fr_weights_free"]) < 1e-4


def test_cpu_baseline_legs_run_and_agree_with_the_restatement():
    """bench.py's CPU legs for configs 3 and 5 (fused aten::gru / aten::lstm): the fast decoder equals the explicit
    restatement, and one small step of each leg runs and yields a finite loss near ln(V)."""
    P = G.vae_params("mid")
    z = torch.from_numpy(__import__("inpaintnet_amd.synthetic", fromlist=["x"]).det_normal("legs/z", (3, 24)))
    with torch.no_grad():
        w1 = O.decoder_forward_fast(P, z, None, False)
        w2, _ = O.decoder_forward(P, z, None, False)
    assert G.rel_err(w1, w2) < 1e-5
    torch.manual_seed(0)
    assert 2.0 < O.CpuLatentTrainStep(48, dropout=0.5).step(2) < 6.0
    assert 2.0 < O.CpuArnnTrainStep(48).step(2) < 6.0
