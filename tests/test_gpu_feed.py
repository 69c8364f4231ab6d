"""Device side of the input feed: the int32 -> int64 widening and the
past/target/future split are HIP kernels; DeviceFeed hands batches over on a copy stream.  Expected values come from the
reference's own trainers (tests/golden/feed_helpers.npz, split_helpers.npz)."""
import numpy as np
import pytest
import torch

from tests import golden_util as G

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from inpaintnet_amd import ops, synthetic
    from inpaintnet_amd.feed import BatchLoader, DeviceFeed
    from inpaintnet_amd.latent_rnn import LatentRNN
    from inpaintnet_amd.latent_rnn_trainer import LatentRNNTrainer
    from inpaintnet_amd.measure_vae import MeasureVAE
    from inpaintnet_amd.vae_trainer import VAETrainer


def _small():
    c = G.CFGS["small"]
    ds = synthetic.SyntheticFolkDataset(num_notes=c["V"])
    vae = MeasureVAE(ds, note_embedding_dim=c["E"], encoder_hidden_size=c["H"], latent_space_dim=c["Z"],
                     decoder_hidden_size=c["H"], encoder_dropout_prob=0.0, decoder_dropout_prob=0.0)
    return c, ds, vae


def test_vae_process_batch_data_matches_reference():
    fx = G.load("feed_helpers")
    c, ds, vae = _small()
    tr = VAETrainer(ds, vae)
    score, md = torch.from_numpy(fx["score"]), torch.from_numpy(fx["metadata"])
    out = tr.process_batch_data((score, md))
    assert out.is_cuda and out.dtype == torch.int64 and out.is_contiguous()
    assert np.array_equal(out.cpu().numpy(), fx["vae_batch"])
    # the feed hands over (device int32, None): same result, no metadata needed
    out2 = tr.process_batch_data((score.cuda(), None))
    assert torch.equal(out, out2)


def test_split_score_stochastic_matches_reference_under_seed():
    fx = G.load("feed_helpers")
    c, ds, vae = _small()
    model = LatentRNN(ds, vae, num_rnn_layers=2, rnn_hidden_size=c["H"], dropout=0.0, rnn_class=torch.nn.GRU)
    tr = LatentRNNTrainer(ds, model)
    score = torch.from_numpy(fx["score"])
    torch.manual_seed(int(fx["split_seed"]))
    draws = []
    for i in range(8):
        past, future, target, n_past, n_target = tr.split_score_stochastic(score if i % 2 else score.cuda(), extra_outs=True)
        draws.append((n_past, n_target))
        assert past.shape == (4, n_past, 24) and target.shape == (4, n_target, 24) and future.shape[1] == 16 - n_past - n_target
        if i == 0:
            for t, k in ((past, "split0_past"), (future, "split0_future"), (target, "split0_target")):
                assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()
                assert np.array_equal(t.cpu().numpy(), fx[k])
    assert np.array_equal(np.array(draws), fx["split_draws"])
    p, f, t = tr.process_batch_data((score, None))
    assert p.shape[1] + f.shape[1] + t.shape[1] == 16


@pytest.mark.parametrize("p,t,f", [(6, 4, 6), (1, 2, 13), (8, 6, 2), (0, 16, 0), (15, 1, 0)])
def test_split_kernel_is_exact(p, t, f):
    fx = G.load("split_helpers")
    score = torch.from_numpy(fx["score"])
    past, future, target = LatentRNNTrainer.split_score(score, p, f, t, 24)
    m = score.long().view(score.shape[0], 16, 24)
    assert torch.equal(past.cpu(), m[:, :p]) and torch.equal(target.cpu(), m[:, p:p + t]) and torch.equal(future.cpu(), m[:, p + t:])
    key = f"past_{p}_{t}_{f}"
    if key in fx.files:
        assert np.array_equal(past.cpu().numpy(), fx[key]) and np.array_equal(future.cpu().numpy(), fx[f"future_{p}_{t}_{f}"])
        assert np.array_equal(target.cpu().numpy(), fx[f"target_{p}_{t}_{f}"])


def test_tokens_to_long_large():
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 48, (4096, 1, 384), generator=g, dtype=torch.int32)
    y = ops.tokens_to_long(x.cuda())
    assert y.dtype == torch.int64 and torch.equal(y.cpu(), x.long())


def test_device_feed_delivers_every_batch_in_order():
    n, bs = 11 * 8 + 3, 8
    score = torch.arange(n * 384, dtype=torch.int32).view(n, 1, 384) % 977
    md = torch.arange(n * 6, dtype=torch.int32).view(n, 1, 2, 3)
    torch.manual_seed(3)
    host = [(a.clone(), b.clone()) for a, b in BatchLoader((score, md), bs, shuffle=True)]
    assert len(host) == 11
    torch.manual_seed(3)
    feed = DeviceFeed(BatchLoader((score, md), bs, shuffle=True), fields=(0,), depth=2)
    got = []
    for s_dev, m_dev in feed:                        # more batches than staging slots: the ring is reused
        assert m_dev is None and s_dev.is_cuda and s_dev.dtype == torch.int32
        got.append(s_dev)
        torch.cuda._sleep(2_000_000)                 # a slow consumer: copies run ahead of the compute stream
    assert len(got) == len(host)
    for g_, (h, _) in zip(got, host):
        assert torch.equal(g_.cpu(), h)
    both = list(DeviceFeed(BatchLoader((score, md), bs, shuffle=False), fields=(0, 1)))
    assert torch.equal(both[2][1].cpu(), md[16:24])


def test_epoch_loop_over_the_feed_trains():
    c = G.CFGS["mid"]
    ds = synthetic.SyntheticFolkDataset(num_notes=c["V"], n_seq=48)
    model = MeasureVAE(ds, note_embedding_dim=c["E"], encoder_hidden_size=c["H"], latent_space_dim=c["Z"],
                       decoder_hidden_size=c["H"])
    tr = VAETrainer(ds, model, lr=1e-3)
    torch.manual_seed(0)
    train, val, _ = ds.data_loaders(batch_size=4, split=(0.70, 0.20))
    model.train()
    l0, _ = tr.loss_and_acc_on_epoch(train, 0, train=True)
    for _ in range(3):
        l1, a1 = tr.loss_and_acc_on_epoch(train, 0, train=True)
    model.eval()
    lv, av = tr.loss_and_acc_on_epoch(val, 0, train=False)
    assert np.isfinite([l0, l1, lv]).all() and l1 < l0 and 0.0 <= av <= 1.0
    assert tr.adam_t == 4 * len(train)
