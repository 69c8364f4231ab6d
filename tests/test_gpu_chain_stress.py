"""Repeated-call stress for the persistent (chain) kernels: the same encoder forward + backward and decoder backward
through the chain kernels and through the per-step launches (inet_set_option(4, 0)), many times, with the allocator's
pool dirtied in between.  Single-shot parity tests cannot see a hand-off that is released early once in a hundred calls
(the fused decode kernel had one: tests/test_gpu_inference.py::test_fused_decode_matches_per_tick_path_repeatedly)."""
import numpy as np
import pytest
import torch

from inpaintnet_amd import ops
from tests import golden_util as G
from tests.test_gpu_kernels import pack

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_encoder_and_decoder_backward_chains_repeatedly():
    c = G.CFGS["full"]
    cfg = ops.vae_config(c["V"], c["E"], c["H"], c["Z"], c["H"])
    table, total = ops.vae_param_table(cfg)
    params = pack(table, total, G.vae_params("full"))
    g = torch.Generator().manual_seed(5)
    rng = np.random.RandomState(11)
    T, H, Z, V = 24, c["H"], c["Z"], c["V"]
    skipped = 0
    try:
        for it in range(200):
            B = [256, 37, 256, 128, 64][it % 5]
            junk = torch.empty(int(rng.randint(1, 48)) << 20, device="cuda").uniform_(-100, 100)
            del junk
            tok = torch.randint(0, V, (B, T), generator=g).cuda()
            m_enc = ops.dropout_mask((T, B, 2 * H), 0.5, it, 0, "cuda")
            m_beat = ops.dropout_mask((4, B, H), 0.5, it, 10 ** 7, "cuda")
            m_tick = ops.dropout_mask((T, B, H), 0.5, it, 2 * 10 ** 7, "cuda")
            dmu = torch.randn(B, Z, generator=g).cuda() * 1e-2
            dls = torch.randn(B, Z, generator=g).cuda() * 1e-2
            z = torch.randn(B, Z, generator=g).cuda()
            dW = torch.randn(B, T, V, generator=g).cuda() * 1e-3
            res = []
            for chain in (1, 0):
                ops.set_option(4, chain)
                grads = torch.zeros_like(params)
                mu, ls, ews = ops.encoder_fwd(cfg, tok, params, mask=m_enc, save=True)
                ops.encoder_bwd(cfg, tok, params, grads, m_enc, dmu, dls, ews)
                # teacher-forced decode: identical tokens on both paths, so the backward inputs are identical too
                w, s, dws = ops.decoder_fwd(cfg, z, tok, True, params, m_beat, m_tick, save=True)
                dz = ops.decoder_bwd(cfg, dW, w, s, params, grads, m_beat, m_tick, dws)
                torch.cuda.synchronize()
                # branch taken by every SELU / ReLU element: the derivative jumps at 0, so two correct paths whose
                # pre-activations differ in the last bit next to 0 disagree by O(1) in that element's gradient
                kinks = [ops.ws_field(cfg, ews, B, 0, n_) > 0 for n_ in ("a_mu", "a_ls")]
                kinks += [ops.ws_field(cfg, dws, B, 1, n_) > 0 for n_ in ("hb0", "ht0", "c_all")] + [w > 0]
                res.append((mu, ls, w, dz, grads, kinks))
            a, b = res
            assert _rel(a[0], b[0]) < 2e-5 and _rel(a[1], b[1]) < 2e-5, (it, B, "encoder forward")
            assert _rel(a[2], b[2]) < 2e-5, (it, B, "decoder forward")
            if any(not torch.equal(x, y) for x, y in zip(a[5], b[5])):
                skipped += 1                       # a branch flipped between the two paths: gradients legitimately differ
                continue
            assert _rel(a[3], b[3]) < 1e-4, (it, B, "dz", _rel(a[3], b[3]))
            for name, off, shape in table:
                n = int(np.prod(shape))
                ga, gb = a[4][off:off + n], b[4][off:off + n]
                assert _rel(ga, gb) < 2e-4, (it, B, name, _rel(ga, gb))
    finally:
        ops.set_option(4, 1)
    assert ops.chain_status() == 0
    assert skipped <= 80, skipped                 # at least 120 of the 200 calls were compared element by element


def test_chunked_chain_forward_for_large_batches():
    """More rows than one resident chain launch takes (the frozen encoder of LatentRNN: 2048 measures).  Three forms of the
    forward pass must agree: the default (B = 2048: one bf16-pipe product per time step with the GRU cell as its epilogue,
    csrc/gru_step_bf3.hip; smaller batches: the chain kernel over 256-row chunks), the chunked chain launches everywhere
    (inet_set_option key 12 = 0), and the f32-input per-step launches (key 4 = 0 as well)."""
    c = G.CFGS["full"]
    cfg = ops.vae_config(c["V"], c["E"], c["H"], c["Z"], c["H"])
    table, total = ops.vae_param_table(cfg)
    params = pack(table, total, G.vae_params("full"))
    g = torch.Generator().manual_seed(9)
    try:
        for B in (2048, 512, 768, 1536):
            tok = torch.randint(0, c["V"], (B, 24), generator=g).cuda()
            res = []
            for chain, step_tiles in ((1, 256), (1, 0), (0, 0)):
                ops.set_option(4, chain)
                ops.set_option(12, step_tiles)
                ops.prof_enable(True)
                mu, ls, _ = ops.encoder_fwd(cfg, tok, params, mask=None, save=False)
                torch.cuda.synchronize()
                ops.prof_dump("/tmp/_inet_chunk.csv")
                ops.prof_enable(False)
                labels = [l.split(",")[1] for l in open("/tmp/_inet_chunk.csv").read().strip().splitlines()[1:]]
                nstep = sum(l.startswith("gru_step_bf3") for l in labels)
                assert nstep == (48 if (B in (2048, 1536) and step_tiles) else 0), (B, chain, step_tiles, labels[:6])
                if chain and not nstep:
                    assert sum(G.is_chain(l, "fwd", 2, 24, 256) for l in labels) == 2 * (B // 256), labels[:6]   # x2: two layers
                res.append((mu, ls))
            for k in (0, 1):
                assert _rel(res[k][0], res[2][0]) < 2e-5 and _rel(res[k][1], res[2][1]) < 2e-5, (B, k)
    finally:
        ops.set_option(4, 1)
        ops.set_option(12, 256)
        ops.prof_enable(False)
    assert ops.chain_status() == 0


def test_staged_encoder_backward_equals_one_pass():
    """inet_vae_encoder_bwd stage 1 (heads + layer 1) then stage 2 (layer 0 + embedding) -- the split a data-parallel
    step uses to start the all-reduce of the early gradients -- against the one-pass call; after stage 1 the two early
    ranges of the arena already hold their final values."""
    c = G.CFGS["full"]
    cfg = ops.vae_config(c["V"], c["E"], c["H"], c["Z"], c["H"])
    table, total = ops.vae_param_table(cfg)
    params = pack(table, total, G.vae_params("full"))
    offs = {n: o for n, o, _ in table}
    dec0 = min(o for n, o, _ in table if n.startswith("decoder."))
    early = [(offs["encoder.lstm.weight_ih_l1"], offs["encoder.note_embedding_layer.weight"]),
             (offs["encoder.linear_mean.0.weight"], dec0)]
    g = torch.Generator().manual_seed(21)
    for B in (256, 37):
        tok = torch.randint(0, c["V"], (B, 24), generator=g).cuda()
        mask = ops.dropout_mask((24, B, 2 * c["H"]), 0.5, 3, 0, "cuda")
        dmu = torch.randn(B, c["Z"], generator=g).cuda() * 1e-2
        dls = torch.randn(B, c["Z"], generator=g).cuda() * 1e-2
        g0 = torch.zeros_like(params)
        mu, ls, ews = ops.encoder_fwd(cfg, tok, params, mask=mask, save=True)
        ops.encoder_bwd(cfg, tok, params, g0, mask, dmu, dls, ews)
        g1 = torch.zeros_like(params)
        mu, ls, ews = ops.encoder_fwd(cfg, tok, params, mask=mask, save=True)
        ops.encoder_bwd(cfg, tok, params, g1, mask, dmu, dls, ews, stage=1)
        torch.cuda.synchronize()
        for lo, hi in early:
            assert _rel(g1[lo:hi], g0[lo:hi]) < 1e-5, (B, lo, hi)
        assert float(g1[:early[0][0]].abs().max()) == 0.0            # layer 0 not touched yet
        ops.encoder_bwd(cfg, tok, params, g1, mask, dmu, dls, ews, stage=2)
        torch.cuda.synchronize()
        for name, off, shape in table:
            n = int(np.prod(shape))
            assert _rel(g1[off:off + n], g0[off:off + n]) < 1e-5 or float(g0[off:off + n].abs().max()) == 0.0, (B, name)


def test_lstm_tagged_handoff_repeated():
    """The LSTM forward chain of the two-layer pipeline hands the state over without a counter (members poll the exchanged
    fragments for a sentinel, csrc/lstm.hip lstm_chain_fwd_tag_kernel).  60 runs of the AnticipationRNN shape (32 x 384, two
    layers, both directions) must reproduce the first run bit for bit and agree with the counter-based single-layer chain."""
    g = torch.Generator().manual_seed(31)
    B, T, H = 32, 384, 256
    gi0 = (torch.randn(T, B, 4 * H, generator=g) * 0.5).cuda()
    Wh0, Wi1, Wh1 = [(torch.randn(4 * H, H, generator=g) / 16).cuda() for _ in range(3)]
    bh0, bi1, bh1 = [(torch.randn(4 * H, generator=g) * 0.1).cuda() for _ in range(3)]
    for reverse in (False, True):
        o0, _, _, _ = ops.lstm_fwd(gi0, Wh0, bh0, H, reverse=reverse)
        gi1 = ops.linear_fwd(o0.view(T * B, H), Wi1, bi1).view(T, B, 4 * H)
        o1, _, _, _ = ops.lstm_fwd(gi1, Wh1, bh1, H, reverse=reverse)
        torch.cuda.synchronize()
        first = None
        for it in range(30):
            p0, p1, _, _ = ops.lstm2_fwd(gi0, Wh0, bh0, Wi1, bi1, Wh1, bh1, H, reverse=reverse)
            torch.cuda.synchronize()
            assert ops.chain_status() == 0, it
            if first is None:
                first = (p0.clone(), p1.clone())
                assert float((p0 - o0).abs().max()) < 2e-5 and float((p1 - o1).abs().max()) < 2e-5
            else:
                assert torch.equal(p0, first[0]) and torch.equal(p1, first[1]), (reverse, it)


def test_injected_chain_timeout_skips_adam_and_trainer_falls_back():
    """Failure path of the persistent kernels on a healthy GPU (inet_set_option key 6 drops one workgroup of the next forward
    chain launch): the bounded spin runs out, inet_chain_status() reports it, the optimizer kernel leaves the weights
    alone while the flag is up and says so in its step report, Trainer.check_steps() raises when it reads that report
    (report_lag steps later, or at once with wait_all), and the epoch loop switches to the per-step kernels and runs every
    lost batch again (Trainer._fall_back_from_chains / _replay) -- with the bias-correction step number put right."""
    from inpaintnet_amd import synthetic
    from inpaintnet_amd.measure_vae import MeasureVAE
    from inpaintnet_amd.vae_trainer import VAETrainer
    ds = synthetic.SyntheticFolkDataset(num_notes=48)
    model = MeasureVAE(ds)
    trainer = VAETrainer(ds, model, lr=1e-4)
    model.train()
    tok = torch.from_numpy(synthetic.det_tokens("fault", (64, 24), 48)).cuda()
    assert ops.chain_status(reset=True) >= 0
    try:
        # (a) direct: a failed step never reaches the weights, and the report says it was skipped
        before = model.flat.clone()
        trainer.zero_grad()
        ops.slow_waits(reset=True)
        ops.set_option(6, 1)
        loss, acc = trainer.loss_and_acc_for_batch(tok, 0, train=True)
        loss.backward()
        trainer.step()                               # (queues the optimizer launch; its report is read later)
        with pytest.raises(ops.ChainTimeoutError) as err:
            trainer.check_steps(wait_all=True)       # waits for that launch (the spin takes ~0.4 s)
        torch.cuda.synchronize()
        assert ops.chain_status() > 0
        # the recorder names the launch: forward GRU chain workgroups that gave up on their group / row-block counter after the
        # whole bounded spin, each with its workgroup id and XCC id -- and the error message carries the same summary
        rec = ops.slow_waits()
        gave_up = [e for e in rec["entries"] if e["gave_up"]]
        assert rec["count"] >= len(gave_up) > 0
        assert all(e["kernel"] in ("gru_chain_fwd", "gru_chain2_fwd") and e["site"] in ("group counter", "row-block counter")
                   and e["polls"] >= 64 and 0 <= e["xcc"] < 8 for e in gave_up), gave_up
        assert max(e["polls"] for e in gave_up) > 100000             # at least one of them ran the whole bound
        assert "Recorder:" in str(err.value) and "GAVE UP" in str(err.value)
        assert ops.slow_waits(reset=True)["count"] == rec["count"] and ops.slow_waits()["count"] == 0
        assert torch.equal(model.flat, before)       # Adam was queued and skipped itself
        assert ops.chain_status(reset=True) > 0 and ops.chain_status() == 0
        # (b) the epoch loop: the fault hits batch 0 of 5; its report is read two steps later, by which time steps 1 and 2 were
        # skipped on the device as well (the flag is sticky): all three are run again on the per-step kernels
        trainer = VAETrainer(ds, model, lr=1e-4)
        trainer.report_lag = 2                       # (the default keeps the host 12 steps ahead; a short lag shows the mid-epoch path)
        score, md = synthetic.SyntheticFolkDataset(num_notes=48, n_seq=8, seed=1).tensors()
        loader = [(torch.from_numpy(score[:4]), torch.from_numpy(md[:4]))] * 5
        trainer.dataset.n_bars = 16
        ops.set_option(6, 1)
        l, a = trainer.loss_and_acc_on_epoch(loader, 0, train=True)
        assert np.isfinite(l) and trainer.chain_fallbacks == 1
        assert trainer.lost_steps == trainer.report_lag + 1
        assert trainer.adam_t == 5                   # five batches, five applied updates: the skipped launches do not count
        assert not torch.equal(model.flat, before)
        assert ops.chain_status() == 0
    finally:
        ops.set_option(6, 0)
        ops.set_option(4, 1)
        ops.chain_status(reset=True)


def test_nonfinite_parameters_and_bad_tokens_raise_valueerror():
    """Reference error semantics without host scans: MeasureVAE/encoder.py:111-116 and decoder.py:424-429 raise ValueError
    ("... has become nan") when a weight is NaN, decoder.py:36-45 (check_index) for an index outside the vocabulary.  Here the
    optimizer kernel flags a parameter that leaves the finite range in its step report and the prologue kernels count bad
    tokens into a host-mapped word; Trainer.check_steps() raises at its next read."""
    from inpaintnet_amd import synthetic
    from inpaintnet_amd.measure_vae import MeasureVAE
    from inpaintnet_amd.vae_trainer import VAETrainer
    ds = synthetic.SyntheticFolkDataset(num_notes=48)
    model = MeasureVAE(ds)
    trainer = VAETrainer(ds, model, lr=1e-4)
    model.train()
    tok = torch.from_numpy(synthetic.det_tokens("nan", (32, 24), 48)).cuda()
    ops.token_status(reset=True)

    def one_step(tokens):
        trainer.zero_grad()
        loss, acc = trainer.loss_and_acc_for_batch(tokens, 0, train=True)
        loss.backward()
        trainer.step()
    one_step(tok)
    trainer.check_steps(wait_all=True)               # healthy: nothing raised
    # a token == V (one past the vocabulary)
    bad = tok.clone()
    bad[3, 7] = 48
    with pytest.raises(ValueError, match="Invalid Value of index"):
        one_step(bad)                                # (step() itself raises if the prologue has already run)
        trainer.check_steps(wait_all=True)
    torch.cuda.synchronize()
    assert ops.token_status() == 0                   # raising consumed the count
    model.load_state_dict({k: torch.from_numpy(synthetic.det_param(k, tuple(v.shape))) for k, v in model.state_dict().items()})
    trainer = VAETrainer(ds, model, lr=1e-4)
    one_step(tok)
    trainer.check_steps(wait_all=True)
    # a NaN weight: the next optimizer step sees a non-finite parameter
    model.param("encoder.linear_mean.0.bias")[5] = float("nan")
    with pytest.raises(ValueError, match="has become nan"):
        one_step(tok)
        trainer.check_steps(wait_all=True)
    # ... and a state_dict with a NaN is refused at load time (inference-only users never run the optimizer)
    sd = model.state_dict()
    with pytest.raises(ValueError, match="has become nan"):
        model.load_state_dict(sd)


@pytest.mark.gpu
def test_error_in_the_last_step_of_a_manual_loop_surfaces_at_finish():
    """Step reports are read report_lag = 12 steps late (the host must run ahead of the GPU), so a manual loop's LAST steps have
    nobody to read them: Trainer.finish() / `with trainer:` does.  The reference raises at the next forward
    pass (MeasureVAE/encoder.py:111-116); a script whose loop has ended has no next forward pass."""
    from inpaintnet_amd import synthetic
    from inpaintnet_amd.measure_vae import MeasureVAE
    from inpaintnet_amd.vae_trainer import VAETrainer
    ds = synthetic.SyntheticFolkDataset(num_notes=48)
    model = MeasureVAE(ds)
    tok = torch.from_numpy(synthetic.det_tokens("finish", (32, 24), 48)).cuda()

    def one_step(trainer):
        trainer.zero_grad()
        loss, acc = trainer.loss_and_acc_for_batch(tok, 0, train=True)
        loss.backward()
        trainer.step()
    trainer = VAETrainer(ds, model, lr=1e-4)
    model.train()
    for _ in range(3):
        one_step(trainer)
    model.param("decoder.x_0")[2] = float("nan")     # ... and the loop's LAST step runs on it
    one_step(trainer)                                # nothing raised: its report is 12 steps away
    assert len(trainer._inflight) == 4
    with pytest.raises(ValueError, match="has become nan"):
        trainer.finish()
    assert not trainer._inflight
    # the context-manager form, healthy and faulty
    sd = {k: torch.from_numpy(synthetic.det_param(k, tuple(v.shape))) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    with VAETrainer(ds, model, lr=1e-4) as t2:
        one_step(t2)
    assert not t2._inflight
    with pytest.raises(ValueError, match="has become nan"):
        with VAETrainer(ds, model, lr=1e-4) as t3:
            one_step(t3)
            model.param("encoder.linear_mean.0.bias")[1] = float("inf")
            one_step(t3)
    # an exception of the loop's own is not masked by what finish() finds
    model.load_state_dict(sd)
    with pytest.raises(KeyError):
        with VAETrainer(ds, model, lr=1e-4) as t4:
            one_step(t4)
            raise KeyError("the loop's own")


def test_preload_touches_every_kernel_once_and_a_healthy_step_records_no_slow_wait():
    """csrc/preload.hip: the library files every kernel handle hipcc registers (template instantiations included) and
    inet_preload() loads them all on the current device without a launch -- idempotent.  A healthy training step of both
    coin branches files no entry in the slow-wait recorder: no wait inside a persistent kernel reached the entry threshold (waits
    of 64+ polls are noted, and there are hundreds per step: chain launches become resident group by group beside the
    weight-gradient products of the side streams, and the early members of a group poll until its last one is there).  With the
    threshold lowered to 64 polls (set_option 16) the same steps DO file entries, all of that kind: a first arrival (expected =
    one round of the group's members) awaited at a group counter."""
    from inpaintnet_amd import _lib, synthetic
    from inpaintnet_amd.measure_vae import MeasureVAE
    from inpaintnet_amd.vae_trainer import VAETrainer
    L = _lib.lib()
    assert L.inet_kernel_count() >= 200
    first = ops.preload()
    assert first in (0, L.inet_kernel_count())       # (0: a trainer of an earlier test did it)
    assert ops.preload() == 0
    ds = synthetic.SyntheticFolkDataset(num_notes=48)
    model = MeasureVAE(ds)
    trainer = VAETrainer(ds, model, lr=1e-4)
    model.train()
    tok = torch.from_numpy(synthetic.det_tokens("preload", (256, 24), 48)).cuda()
    eps = torch.from_numpy(synthetic.det_normal("preload/eps", (256, 256))).cuda()
    ops.slow_waits(reset=True)
    for tf in (True, False, True, False):
        trainer.zero_grad()
        w, s_, z_dist, prior, z, zp = model(tok, train=True, eps=eps, teacher_forced=tf)
        ce, acc = trainer.mean_crossentropy_loss_and_accuracy(w, tok)
        (ce + trainer.compute_kld_loss(z_dist, prior)).backward()
        trainer.step()
    trainer.finish()
    rec = ops.slow_waits(reset=True)
    assert ops.chain_status() == 0
    assert rec["count"] == 0 and rec["entries"] == [], rec
    assert rec["noted"] >= 0
    try:
        ops.set_option(16, 64)
        for tf in (True, False):
            trainer.zero_grad()
            w, s_, z_dist, prior, z, zp = model(tok, train=True, eps=eps, teacher_forced=tf)
            ce, acc = trainer.mean_crossentropy_loss_and_accuracy(w, tok)
            (ce + trainer.compute_kld_loss(z_dist, prior)).backward()
            trainer.step()
        trainer.finish()
        low = ops.slow_waits(reset=True)
    finally:
        ops.set_option(16, 16384)
    assert low["count"] == 0 or all(not e["gave_up"] and e["polls"] >= 64 for e in low["entries"]), low
