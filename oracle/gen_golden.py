#!/usr/bin/env python3
"""Golden-vector generator.  TEST INFRASTRUCTURE -- never imported by the product.

Runs ONLY in the build container, where the upstream reference is mounted
read-only at /root/reference.  It imports the reference's own Python classes
(MeasureVAE, VAETrainer, LatentRNN, LatentRNNTrainer) behind three stub modules
(music21, glob2, tensorboard_logger -- absent from this image, and only used by
the dataset / logging layers that are out of scope), drives them with
deterministic weights and inputs from inpaintnet_amd.synthetic, and writes the
inputs + the reference's outputs to tests/golden/*.npz.  Those fixtures are
data; no reference source travels.

    python oracle/gen_golden.py            # regenerates every fixture

Control of randomness in the reference:
  * eps of Normal.rsample    -> torch.distributions.normal._standard_normal patched
  * teacher-forcing coin     -> random.random patched in MeasureVAE.decoder / LatentRNN.latent_rnn
  * dropout                  -> model.eval(), or dropout prob 0.0 in the constructors
"""
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def _stub_modules():
    def mod(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m
    m21 = mod("music21")
    for s in ["interval", "note", "harmony", "expressions", "meter", "abcFormat", "stream",
              "duration", "pitch", "repeat", "exceptions21", "converter"]:
        setattr(m21, s, mod("music21." + s))
    sys.modules["music21.abcFormat"].ABCHandlerException = type("E", (Exception,), {})
    mod("glob2", glob=lambda p: [])
    mod("tensorboard_logger", configure=lambda *a, **k: None, log_value=lambda *a, **k: None)
    import matplotlib
    matplotlib.use("Agg")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)


_stub_modules()
import warnings  # noqa: E402
warnings.filterwarnings("ignore")
import torch  # noqa: E402
import torch.distributions.normal as _tdn  # noqa: E402

from MeasureVAE.measure_vae import MeasureVAE  # noqa: E402
from MeasureVAE.vae_trainer import VAETrainer  # noqa: E402
import MeasureVAE.decoder as ref_decoder_mod  # noqa: E402
from LatentRNN.latent_rnn import LatentRNN  # noqa: E402
import LatentRNN.latent_rnn as ref_latent_mod  # noqa: E402
from LatentRNN.latent_rnn_trainer import LatentRNNTrainer  # noqa: E402
from AnticipationRNN.anticipation_rnn_gauss_reg_model import ConstraintModelGaussianReg  # noqa: E402
from AnticipationRNN.anticipation_rnn_trainer import AnticipationRNNGaussianRegTrainer  # noqa: E402

from inpaintnet_amd import synthetic  # noqa: E402

torch.set_num_threads(8)
import random as _random_module  # noqa: E402
_ORIG_RANDOM_RANDOM = _random_module.random


class FakeDataset:
    def __init__(self, V):
        self.note2index_dicts = [{i: i for i in range(V)}]
        self.n_bars = 16
        self.subdivision = 6
        self.num_beats_per_bar = 4
        self.num_voices = 1

        self.metadatas = [types.SimpleNamespace(num_values=6), types.SimpleNamespace(num_values=6)]

    def empty_score_tensor(self, length):
        return torch.zeros(self.num_voices, length, dtype=torch.long)

    def __repr__(self):
        return "Fake"


class EpsQueue:
    """Replaces torch.distributions.normal._standard_normal with a FIFO of
    pre-generated eps tensors; records what it handed out."""

    def __init__(self):
        self.q = []
        self.orig = _tdn._standard_normal

    def __enter__(self):
        def fake(shape, dtype, device):
            e = self.q.pop(0)
            assert tuple(e.shape) == tuple(shape), (e.shape, shape)
            return e
        _tdn._standard_normal = fake
        return self

    def __exit__(self, *a):
        _tdn._standard_normal = self.orig

    def push(self, name, shape, seed=0):
        e = torch.from_numpy(synthetic.det_normal(name, shape, 1.0, seed))
        self.q.append(e)
        return e


def set_coin(value):
    f = (lambda: value)
    ref_decoder_mod.random.random = f
    ref_latent_mod.random.random = f


def load_det_weights(model, seed=0):
    sd = model.state_dict()
    new = {k: torch.from_numpy(synthetic.det_param(k, tuple(v.shape), seed)) for k, v in sd.items()}
    model.load_state_dict(new)
    return {k: v.numpy().copy() for k, v in new.items()}


def top2_margin(w):
    s, _ = torch.sort(w, dim=-1, descending=True)
    return (s[..., 0] - s[..., 1]).numpy()


CFGS = {
    # name: (V, E, H, Z, batch)
    "small": dict(V=12, E=4, H=16, Z=8, B=5),
    "mid": dict(V=20, E=6, H=48, Z=24, B=3),
    "full": dict(V=48, E=10, H=512, Z=256, B=5),
}


def build_vae(c, dropout=0.0):
    return MeasureVAE(FakeDataset(c["V"]), note_embedding_dim=c["E"],
                      encoder_hidden_size=c["H"], latent_space_dim=c["Z"],
                      decoder_hidden_size=c["H"], encoder_dropout_prob=dropout,
                      decoder_dropout_prob=dropout)


def grads_of(model, full):
    out = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().numpy().astype(np.float32)
        if full:
            out["grad/" + k] = g.copy()
        else:
            flat = g.reshape(-1)
            out["gradnorm/" + k] = np.float64(np.sqrt((flat.astype(np.float64) ** 2).sum()))
            out["gradsum/" + k] = np.float64(flat.astype(np.float64).sum())
            out["gradhead/" + k] = flat[:64].copy()
            out["gradtail/" + k] = flat[-64:].copy()
    return out


def param_digest(model, full):
    out = {}
    for k, p in model.named_parameters():
        v = p.detach().numpy().astype(np.float32)
        if full:
            out[k] = v.copy()
        else:
            flat = v.reshape(-1)
            out["sum/" + k] = np.float64(flat.astype(np.float64).sum())
            out["norm/" + k] = np.float64(np.sqrt((flat.astype(np.float64) ** 2).sum()))
            out["head/" + k] = flat[:64].copy()
    return out


def gen_vae(name, c):
    V, Z, B = c["V"], c["Z"], c["B"]
    full_tensors = name != "full"
    model = build_vae(c, dropout=0.0)
    weights = load_det_weights(model)
    tokens = torch.from_numpy(synthetic.det_tokens("tokens/" + name, (B, 24), V))
    fx = {"tokens": tokens.numpy()}
    if full_tensors:
        for k, v in weights.items():
            fx["param/" + k] = v

    # --- encoder (a2) --------------------------------------------------------
    model.eval()
    with torch.no_grad():
        dist = model.encoder(tokens)
    fx["enc_mu"] = dist.loc.numpy()
    fx["enc_logsigma"] = dist.scale.log().numpy()
    fx["enc_sigma"] = dist.scale.numpy()

    # --- decoder eval / free-running argmax (a4-a6) --------------------------
    z = torch.from_numpy(synthetic.det_normal("z/" + name, (B, Z), 1.0))
    fx["dec_z"] = z.numpy()
    with torch.no_grad():
        w, s = model.decoder(z, tokens, train=False)
    fx["dec_eval_weights"] = w.numpy()
    fx["dec_eval_samples"] = s.numpy()
    fx["dec_eval_margin"] = top2_margin(w)

    # --- decoder teacher-forced (train=True, coin < 0.5), dropout prob = 0 ----
    model.train()
    set_coin(0.0)
    with torch.no_grad():
        w, s = model.decoder(z, tokens, train=True)
    fx["dec_tf_weights"] = w.numpy()
    fx["dec_tf_samples"] = s.numpy()

    # --- full VAE train steps through the reference trainer (a7,a8,a10) ------
    for mode, coin in (("tf", 0.0), ("fr", 0.9)):
        model = build_vae(c, dropout=0.0)
        load_det_weights(model)
        trainer = VAETrainer(FakeDataset(V), model, lr=1e-4)
        model.train()
        set_coin(coin)
        losses = []
        with EpsQueue() as q:
            for step in range(5):
                eps = q.push(f"eps/{name}/{step}", (B, Z))
                fx[f"step_{mode}_eps{step}"] = eps.numpy()
                trainer.zero_grad()
                weights_, samples_, z_dist, prior_dist, z_tilde, z_prior = model(tokens, train=True)
                ce = trainer.mean_crossentropy_loss(weights=weights_, targets=tokens)
                kl = trainer.compute_kld_loss(z_dist, prior_dist)
                acc = trainer.mean_accuracy(weights=weights_, targets=tokens)
                loss = ce + kl
                loss.backward()
                if step == 0:
                    fx[f"step_{mode}_weights"] = weights_.detach().numpy()
                    fx[f"step_{mode}_samples"] = samples_.detach().numpy()
                    fx[f"step_{mode}_margin"] = top2_margin(weights_.detach())
                    fx[f"step_{mode}_z"] = z_tilde.detach().numpy()
                    for k, v in grads_of(model, full_tensors).items():
                        fx[f"step_{mode}_{k}"] = v
                assert not q.q, "eps queue must be fully consumed each step"
                trainer.step()
                losses.append([loss.item(), ce.item(), kl.item(), acc.item()])
                if step in (0, 4):
                    for k, v in param_digest(model, name == "small").items():
                        fx[f"step_{mode}_after{step + 1}/{k}"] = v
        fx[f"step_{mode}_losses"] = np.array(losses, dtype=np.float64)

    np.savez_compressed(os.path.join(OUT, f"vae_{name}.npz"), **fx)
    print("wrote vae_%s.npz  (%d arrays)" % (name, len(fx)))


def gen_latent(name, c, auto_reg, coin):
    """LatentRNN forward + one trainer step (a11-a15).  Frozen VAE, dropout 0."""
    V, Z, B = c["V"], c["Z"], c["B"]
    H = c["H"]
    full_tensors = name != "full"
    vae = build_vae(c, dropout=0.0)
    load_det_weights(vae)
    model = LatentRNN(FakeDataset(V), vae, num_rnn_layers=2, rnn_hidden_size=H, dropout=0.0,
                      rnn_class=torch.nn.GRU, auto_reg=auto_reg, teacher_forcing=True)
    wts = load_det_weights(model)  # covers vae_model.* as well, same names -> same values
    tag = f"latent_{name}_{'ar' if auto_reg else 'nar'}_{'tf' if coin < 0.5 else 'fr'}"
    fx = {}
    if full_tensors:
        for k, v in wts.items():
            fx["param/" + k] = v
    score = torch.from_numpy(synthetic.folk_score(B, V, seed=3))  # (B,1,384) int32
    fx["score"] = score.numpy()
    n_past, n_target, n_future = 6, 4, 6
    trainer = LatentRNNTrainer(FakeDataset(V), model, lr=1e-4)
    past, future, target = LatentRNNTrainer.split_score(score, n_past, n_future, n_target, 24)
    fx["split"] = np.array([n_past, n_target, n_future])
    model.train()
    set_coin(coin)
    with EpsQueue() as q:
        e_p = q.push(f"eps_p/{tag}", (B * n_past, Z))
        e_f = q.push(f"eps_f/{tag}", (B * n_future, Z))
        e_t = q.push(f"eps_t/{tag}", (B * n_target, Z))
        fx["eps_past"], fx["eps_future"], fx["eps_target"] = e_p.numpy(), e_f.numpy(), e_t.numpy()
        if auto_reg and coin >= 0.5:
            # free-running AR re-encodes each generated measure but the last one's z is unused
            for i in range(n_target):
                e = q.push(f"eps_ar{i}/{tag}", (B, Z))
                fx[f"eps_ar{i}"] = e.numpy()
        trainer.zero_grad()
        weights, samples, gen_z = model(past, future, target, n_target, train=True)
    loss = trainer.mean_crossentropy_loss_alt(weights=weights, targets=target)
    acc = trainer.mean_accuracy_alt(weights=weights, targets=target)
    loss.backward()
    fx["weights"] = weights.detach().numpy()
    fx["samples"] = samples.detach().numpy()
    fx["margin"] = top2_margin(weights.detach())
    fx["gen_z"] = gen_z.detach().numpy()
    fx["loss_acc"] = np.array([loss.item(), acc.item()], dtype=np.float64)
    for k, v in grads_of(model, full_tensors).items():
        fx[k] = v
    assert all(p.grad is None for p in vae.parameters())
    # latent-space diagnostics of the reference (utils/trainer.py:308-342; north_star "latent-MSE"): the generated latents
    # against the frozen encoder's z of the target measures (same eps as the forward pass used)
    with EpsQueue() as q2:
        q2.q.append(e_t)
        with torch.no_grad():
            z_target = model.get_z_seq(target)
    fx["z_target"] = z_target.numpy()
    fx["mse_gen_target"] = np.float64(trainer.mean_mse_loss_rnn(gen_z.detach(), z_target).item())
    fx["l1_gen_target"] = np.float64(trainer.mean_l1_loss_rnn(gen_z.detach(), z_target).item())
    trainer.step()
    for k, p in model.named_parameters():
        if p.requires_grad:
            v = p.detach().numpy()
            if full_tensors:
                fx["after1/" + k] = v.copy()
            else:
                fx["after1sum/" + k] = np.float64(v.astype(np.float64).sum())
                fx["after1head/" + k] = v.reshape(-1)[:64].copy()
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **fx)
    print("wrote %s.npz (%d arrays)" % (tag, len(fx)))


ARNN_CFGS = {
    "small": dict(V=12, E=4, Em=2, H=16, LH=16, B=3),
    "full": dict(V=48, E=10, Em=2, H=256, LH=256, B=2),
}


def gen_arnn(name, c):
    """ConstraintModelGaussianReg + AnticipationRNNGaussianRegTrainer (a16), dropout 0: teacher-forced step with
    gradients and one Adam step; free-running forward."""
    V, B, L = c["V"], c["B"], 384
    full_tensors = name != "full"
    ds = FakeDataset(V)

    def build():
        m = ConstraintModelGaussianReg(ds, note_embedding_dim=c["E"], metadata_embedding_dim=c["Em"],
                                       num_lstm_constraints_units=c["H"], num_lstm_generation_units=c["H"],
                                       linear_hidden_size=c["LH"], num_layers=2, dropout_input_prob=0.0,
                                       dropout_prob=0.0, unary_constraint=True, teacher_forcing=True)
        load_det_weights(m)
        return m
    model = build()
    fx = {}
    if full_tensors:
        for k, v in model.state_dict().items():
            fx["param/" + k] = v.numpy().copy()
    fx["param_keys"] = np.array(list(model.state_dict().keys()))
    fx["param_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in model.state_dict().values()])
    score = torch.from_numpy(synthetic.folk_score(B, V, seed=11)).long()
    metadata = torch.from_numpy(synthetic.folk_metadata(B)).long()
    metadata[..., 0] = torch.from_numpy(synthetic.det_tokens("arnn/md0", (B, 1, L), 6))
    n_past, n_target = 6, 4
    start_tick, end_tick = (n_past + 1) * 24, (n_past + 1) * 24 + n_target * 24
    loc = torch.zeros_like(score)
    loc[:, :, :start_tick] = 1
    loc[:, :, end_tick:] = 1
    fx["score"], fx["metadata"], fx["constraints_loc"] = score.numpy(), metadata.numpy(), loc.numpy()
    fx["ticks"] = np.array([start_tick, end_tick])
    trainer = AnticipationRNNGaussianRegTrainer(ds, model, lr=1e-4)
    model.train()
    set_coin(0.0)                                   # random.random() <= 0.5 -> teacher forcing
    trainer.zero_grad()
    loss, acc = trainer.loss_and_acc_for_batch((score, metadata, loc, start_tick, end_tick), 0, train=True)
    loss.backward()
    with torch.no_grad():
        w_all, _ = model._forward_tf(score, metadata, loc)
    fx["tf_weights_all"] = w_all[0].numpy()
    fx["tf_loss_acc"] = np.array([loss.item(), acc.item()], dtype=np.float64)
    for k, v in grads_of(model, full_tensors).items():
        fx["tf_" + k] = v
    trainer.step()
    for k, p in model.named_parameters():
        v = p.detach().numpy()
        if full_tensors:
            fx["tf_after1/" + k] = v.copy()
        else:
            fx["tf_after1head/" + k] = v.reshape(-1)[:64].copy()
    # free-running forward (backward of this path fails on CPU under torch 2.x: in-place write on a saved view)
    model = build()
    model.train()
    set_coin(0.9)
    with torch.no_grad():
        w, _ = model(score, metadata, loc, train=True)
        w_all, gen = model._forward_no_tf(score, metadata, loc)
    fx["fr_weights_free"] = w[0].numpy()
    fx["fr_weights_all"] = w_all[0].numpy()
    fx["fr_gen"] = gen.numpy()
    fx["fr_margin_row0"] = top2_margin(w_all[0][0])
    np.savez_compressed(os.path.join(OUT, f"arnn_{name}.npz"), **fx)
    print("wrote arnn_%s.npz (%d arrays)" % (name, len(fx)))


def gen_split_helpers():
    """split_score / split_to_measures / process_batch_data index contract (a9, a15)."""
    V = 12
    score = torch.from_numpy(synthetic.folk_score(3, V, seed=5))
    fx = {"score": score.numpy()}
    for (p, t, f) in [(6, 4, 6), (1, 2, 13), (8, 6, 2)]:
        a, b, c_ = LatentRNNTrainer.split_score(score, p, f, t, 24)
        fx[f"past_{p}_{t}_{f}"] = a.numpy()
        fx[f"future_{p}_{t}_{f}"] = b.numpy()
        fx[f"target_{p}_{t}_{f}"] = c_.numpy()
    np.savez_compressed(os.path.join(OUT, "split_helpers.npz"), **fx)
    print("wrote split_helpers.npz")


def gen_latent_ablation(kind):
    """LatentRNNAblations (past-only / future-only context, generator hidden = H): forward + one trainer step (f4)."""
    from LatentRNN.latent_rnn_ablations import LatentRNNAblations
    import LatentRNN.latent_rnn_ablations as ref_abl_mod
    c = dict(CFGS["small"])
    V, Z, B, H = c["V"], c["Z"], c["B"], c["H"]
    vae = build_vae(c, dropout=0.0)
    load_det_weights(vae)
    model = LatentRNNAblations(FakeDataset(V), vae, num_rnn_layers=2, rnn_hidden_size=H, dropout=0.0,
                               rnn_class=torch.nn.GRU, auto_reg=False, teacher_forcing=True, type=kind)
    wts = load_det_weights(model)
    tag = f"latent_small_abl_{kind}"
    fx = {"param/" + k: v for k, v in wts.items()}
    score = torch.from_numpy(synthetic.folk_score(B, V, seed=3))
    fx["score"] = score.numpy()
    n_past, n_target, n_future = 6, 4, 6
    trainer = LatentRNNTrainer(FakeDataset(V), model, lr=1e-4)
    past, future, target = LatentRNNTrainer.split_score(score, n_past, n_future, n_target, 24)
    fx["split"] = np.array([n_past, n_target, n_future])
    model.train()
    ref_abl_mod.random.random = lambda: 0.9
    with EpsQueue() as q:
        e_p = q.push(f"eps_p/{tag}", (B * n_past, Z))
        e_f = q.push(f"eps_f/{tag}", (B * n_future, Z))
        e_t = q.push(f"eps_t/{tag}", (B * n_target, Z))
        fx["eps_past"], fx["eps_future"], fx["eps_target"] = e_p.numpy(), e_f.numpy(), e_t.numpy()
        trainer.zero_grad()
        weights, samples, gen_z = model(past, future, target, n_target, train=True)
    loss = trainer.mean_crossentropy_loss_alt(weights=weights, targets=target)
    acc = trainer.mean_accuracy_alt(weights=weights, targets=target)
    loss.backward()
    fx["weights"], fx["samples"] = weights.detach().numpy(), samples.detach().numpy()
    fx["margin"] = top2_margin(weights.detach())
    fx["gen_z"] = gen_z.detach().numpy()
    fx["loss_acc"] = np.array([loss.item(), acc.item()], dtype=np.float64)
    for k, p in model.named_parameters():
        if p.requires_grad:
            fx["grad/" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    trainer.step()
    for k, p in model.named_parameters():
        if p.requires_grad:
            fx["after1/" + k] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **fx)
    print("wrote %s.npz (%d arrays)" % (tag, len(fx)))


def gen_inference():
    """Inference surface (f3): MeasureVAE.forward_test, VAETester.decode_mid_point, and B = 1 inpainting as
    LatentRNNTester.generate runs it (eval mode, no teacher forcing) for the non-AR and the auto-regressive model."""
    from MeasureVAE.vae_tester import VAETester
    c = dict(CFGS["small"])
    V, Z, H = c["V"], c["Z"], c["H"]
    vae = build_vae(c, dropout=0.0)
    load_det_weights(vae)
    vae.eval()
    fx = {}
    # forward_test: (B, M, 24); one rsample per measure, in measure order
    B, M = 2, 3
    tok = torch.from_numpy(synthetic.det_tokens("inf/ft", (B, M, 24), V))
    fx["ft_tokens"] = tok.numpy()
    with EpsQueue() as q, torch.no_grad():
        for i in range(M):
            fx[f"ft_eps{i}"] = q.push(f"inf/ft_eps{i}", (B, Z)).numpy()
        w, s = vae.forward_test(tok)
    fx["ft_weights"], fx["ft_samples"], fx["ft_margin"] = w.numpy(), s.numpy(), top2_margin(w)
    # decode_mid_point
    tester = VAETester(FakeDataset(V), vae)
    z1 = torch.from_numpy(synthetic.det_normal("inf/z1", (1, Z)))
    z2 = torch.from_numpy(synthetic.det_normal("inf/z2", (1, Z)))
    with torch.no_grad():
        mid = tester.decode_mid_point(z1, z2, 3)
        ws = [vae.decoder(z1 + (z2 - z1) * i / 4, torch.zeros(1, 24), False)[0] for i in range(5)]
    fx["mid_z1"], fx["mid_z2"], fx["mid_tokens"] = z1.numpy(), z2.numpy(), mid.numpy()
    fx["mid_margin"] = top2_margin(torch.cat(ws, 0))
    # generate: B = 1, past 5 / target 3 / future 8 measures
    for auto_reg in (False, True):
        model = LatentRNN(FakeDataset(V), vae, num_rnn_layers=2, rnn_hidden_size=H, dropout=0.0,
                          rnn_class=torch.nn.GRU, auto_reg=auto_reg, teacher_forcing=True)
        load_det_weights(model)
        model.eval()
        tag = "gen_ar" if auto_reg else "gen_nar"
        score = torch.from_numpy(synthetic.folk_score(1, V, seed=41))
        past, future, target = LatentRNNTrainer.split_score(score, 5, 8, 3, 24)
        fx[f"{tag}_score"] = score.numpy()
        with EpsQueue() as q, torch.no_grad():
            fx[f"{tag}_eps_past"] = q.push(f"inf/{tag}/p", (5, Z)).numpy()
            fx[f"{tag}_eps_future"] = q.push(f"inf/{tag}/f", (8, Z)).numpy()
            fx[f"{tag}_eps_target"] = q.push(f"inf/{tag}/t", (3, Z)).numpy()
            if auto_reg:
                for i in range(3):
                    fx[f"{tag}_eps_ar{i}"] = q.push(f"inf/{tag}/ar{i}", (1, Z)).numpy()
            w, s, gz = model(past, future, target, 3, train=False)
        fx[f"{tag}_weights"], fx[f"{tag}_samples"], fx[f"{tag}_gen_z"] = w.numpy(), s.numpy(), gz.numpy()
        fx[f"{tag}_margin"] = top2_margin(w)
        fx[f"{tag}_full"] = torch.cat((past, s.view(1, 3, 24), future), 1).numpy()
    np.savez_compressed(os.path.join(OUT, "inference_small.npz"), **fx)
    print("wrote inference_small.npz (%d arrays)" % len(fx))


def gen_arnn_inpaint():
    """ConstraintModelGaussianReg.forward_inpaint (eval mode) and the baseline trainer's constraint sampling (f4)."""
    import random as pyrandom
    pyrandom.random = _ORIG_RANDOM_RANDOM            # set_coin() patches the shared `random` module; undo it here
    from AnticipationRNN.anticipation_rnn_gauss_reg_model import AnticipationRNNBaseline
    from AnticipationRNN.anticipation_rnn_trainer import AnticipationRNNBaselineTrainer
    c = ARNN_CFGS["small"]
    V, B, L = c["V"], c["B"], 384
    ds = FakeDataset(V)
    model = AnticipationRNNBaseline(ds, note_embedding_dim=c["E"], metadata_embedding_dim=c["Em"],
                                    num_lstm_constraints_units=c["H"], num_lstm_generation_units=c["H"],
                                    linear_hidden_size=c["LH"], num_layers=2, dropout_input_prob=0.0,
                                    dropout_prob=0.0, unary_constraint=True, teacher_forcing=True)
    load_det_weights(model)
    model.eval()
    fx = {"repr": np.array(repr(model))}
    score = torch.from_numpy(synthetic.folk_score(B, V, seed=11)).long()
    metadata = torch.from_numpy(synthetic.folk_metadata(B)).long()
    metadata[..., 0] = torch.from_numpy(synthetic.det_tokens("arnn/md0", (B, 1, L), 6))
    start_tick, end_tick = 7 * 24, 7 * 24 + 2 * 24
    loc = torch.zeros_like(score)
    loc[:, :, :start_tick] = 1
    loc[:, :, end_tick:] = 1
    fx["score"], fx["metadata"], fx["constraints_loc"] = score.numpy(), metadata.numpy(), loc.numpy()
    fx["ticks"] = np.array([start_tick, end_tick])
    with torch.no_grad():
        w, gen = model.forward_inpaint(score, metadata, loc, start_tick, end_tick)
    fx["inpaint_weights"], fx["inpaint_gen"] = w[0].numpy(), gen.numpy()
    fx["inpaint_margin_row0"] = top2_margin(w[0][0])
    trainer = AnticipationRNNBaselineTrainer(ds, model)
    pyrandom.seed(99)
    torch.manual_seed(99)
    locs = []
    for _ in range(3):
        out = trainer.process_batch_data((score.int(), metadata.int()))
        locs.append(out[2].numpy())
        assert out[3] is None and out[4] is None
    fx["baseline_seed"] = np.array(99)
    fx["baseline_locs"] = np.stack(locs)
    np.savez_compressed(os.path.join(OUT, "arnn_inpaint_small.npz"), **fx)
    print("wrote arnn_inpaint_small.npz (%d arrays)" % len(fx))


def gen_feed_helpers():
    """Batch preparation on the host side of the reference, with its own RNG consumption (SURVEY 8c fixture 7):
    VAETrainer.process_batch_data (vae_trainer.py:42-55) for an N-bars dataset, LatentRNNTrainer.split_score_stochastic
    under a fixed torch seed (latent_rnn_trainer.py:77-132), AnticipationRNN get_constraints_location (:93-128)."""
    from DatasetManager.the_session.folk_dataset import FolkDatasetNBars

    class FakeNBars(FolkDatasetNBars):                # isinstance(dataset, FolkDatasetNBars) switches the reshape on
        def __init__(self, V):
            self.__dict__.update(FakeDataset(V).__dict__)

        def __repr__(self):
            return "FakeNBars"

    c = CFGS["small"]
    V = c["V"]
    score = torch.from_numpy(synthetic.folk_score(4, V, seed=11))
    md = torch.from_numpy(synthetic.folk_metadata(4))
    fx = {"score": score.numpy(), "metadata": md.numpy()}
    vae = build_vae(c)
    fx["vae_batch"] = VAETrainer(FakeNBars(V), vae).process_batch_data((score, md)).numpy()
    model = LatentRNN(FakeDataset(V), vae, num_rnn_layers=2, rnn_hidden_size=c["H"], dropout=0.0,
                      rnn_class=torch.nn.GRU, auto_reg=False, teacher_forcing=True)
    lt = LatentRNNTrainer(FakeDataset(V), model)
    torch.manual_seed(1234)
    draws = []
    for i in range(8):
        past, future, target, n_past, n_target = lt.split_score_stochastic(score, extra_outs=True)
        draws.append((n_past, n_target))
        if i == 0:
            fx["split0_past"], fx["split0_future"], fx["split0_target"] = past.numpy(), future.numpy(), target.numpy()
    fx["split_seed"] = np.array(1234)
    fx["split_draws"] = np.array(draws, dtype=np.int64)
    p2, f2, t2, np2, nt2 = lt.split_score_stochastic(score, extra_outs=True, fix_num_target=3)
    fx["split_fixed3"] = np.array([np2, nt2, p2.shape[1], f2.shape[1], t2.shape[1]], dtype=np.int64)
    ca = ARNN_CFGS["small"]
    dsa = FakeDataset(ca["V"])
    arnn = ConstraintModelGaussianReg(dsa, note_embedding_dim=ca["E"], metadata_embedding_dim=ca["Em"],
                                      num_lstm_constraints_units=ca["H"], num_lstm_generation_units=ca["H"],
                                      linear_hidden_size=ca["LH"], num_layers=2, dropout_input_prob=0.0,
                                      dropout_prob=0.0, unary_constraint=True, teacher_forcing=True)
    at = AnticipationRNNGaussianRegTrainer(dsa, arnn)
    torch.manual_seed(4321)
    ticks = []
    for i in range(8):
        loc, start, end = at.get_constraints_location(score)
        ticks.append((start, end))
        if i == 0:
            fx["constraints0"] = loc.numpy()
    fx["constraints_seed"] = np.array(4321)
    fx["constraints_ticks"] = np.array(ticks, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "feed_helpers.npz"), **fx)
    print("wrote feed_helpers.npz (%d arrays)" % len(fx))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["vae", "latent", "arnn", "split", "feed", "ablation", "inference", "inpaint"]
    if "vae" in which:
        for n, c in CFGS.items():
            gen_vae(n, c)
    if "latent" in which:
        for n in ("small", "full"):
            c = dict(CFGS[n])
            if n == "full":
                c["B"] = 2
            gen_latent(n, c, auto_reg=False, coin=0.9)
            gen_latent(n, c, auto_reg=True, coin=0.0)
            gen_latent(n, c, auto_reg=True, coin=0.9)
    if "arnn" in which:
        for n, c in ARNN_CFGS.items():
            gen_arnn(n, c)
    if "split" in which:
        gen_split_helpers()
    if "feed" in which:
        gen_feed_helpers()
    if "ablation" in which:
        gen_latent_ablation("past")
        gen_latent_ablation("future")
    if "inference" in which:
        gen_inference()
    if "inpaint" in which:
        gen_arnn_inpaint()
