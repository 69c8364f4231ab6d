"""Host side of the input feed -- runs without a GPU.

  * BatchLoader == torch DataLoader(TensorDataset, shuffle, drop_last) batch for batch under the same torch seed, and
    it leaves the global CPU generator in the same state (so the stochastic splits drawn between batches line up too);
  * the split / constraint-window draws consume torch's CPU generator exactly as the reference does
    (tests/golden/feed_helpers.npz was written by the reference's own trainers under a fixed seed);
  * EarlyStopping follows the reference's patience rule; rank sharding of batches.
"""
import types

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

from inpaintnet_amd import dp, synthetic
from inpaintnet_amd.feed import BatchLoader
from tests import golden_util as G


@pytest.mark.parametrize("shuffle", [True, False])
@pytest.mark.parametrize("n,bs", [(37, 8), (64, 16), (5, 8)])
def test_batch_loader_matches_torch_dataloader(shuffle, n, bs):
    a = torch.arange(n * 6, dtype=torch.int32).view(n, 1, 6)
    b = torch.arange(n * 2, dtype=torch.int32).view(n, 2) * 7
    ref = DataLoader(TensorDataset(a, b), batch_size=bs, shuffle=shuffle, drop_last=True)
    mine = BatchLoader((a, b), bs, shuffle=shuffle, drop_last=True, pin_memory=False)
    assert len(mine) == len(ref) == n // bs
    torch.manual_seed(99)
    want, want_draws = [], []
    for _ in range(3):                                   # three epochs: the permutation changes, the RNG advances
        want.append([(x.clone(), y.clone()) for x, y in ref])
        want_draws.append(int(torch.randint(0, 1000, (1,))))
    torch.manual_seed(99)
    for e in range(3):
        got = [(x.clone(), y.clone()) for x, y in mine]
        assert len(got) == len(want[e])
        for (x, y), (rx, ry) in zip(got, want[e]):
            assert torch.equal(x, rx) and torch.equal(y, ry)
        assert int(torch.randint(0, 1000, (1,))) == want_draws[e]


def test_batch_loader_keeps_last_partial_batch_when_asked():
    a = torch.arange(10).view(10, 1)
    ld = BatchLoader((a,), 4, shuffle=False, drop_last=False, pin_memory=False)
    got = [x[0].clone() for x in ld]
    assert len(ld) == 3 and [g.shape[0] for g in got] == [4, 4, 2] and torch.equal(torch.cat(got), a)


def test_synthetic_dataset_loaders_follow_the_reference_contract():
    ds = synthetic.SyntheticFolkDataset(num_notes=12, n_seq=40)
    tr, va, ev = ds.data_loaders(batch_size=8, split=(0.70, 0.20))
    assert (len(tr), len(va), len(ev)) == (28 // 8, 8 // 8, 4 // 8)       # drop_last on every loader
    assert tr.shuffle and not va.shuffle and not ev.shuffle
    score, md = next(iter(va))
    full_score, full_md = ds.tensors()
    assert score.dtype == torch.int32 and tuple(score.shape) == (8, 1, 384) and tuple(md.shape) == (8, 1, 384, 3)
    assert np.array_equal(score.numpy(), full_score[28:36])


def test_split_draws_follow_the_reference_stream():
    from inpaintnet_amd.latent_rnn_trainer import LatentRNNTrainer
    fx = G.load("feed_helpers")
    me = types.SimpleNamespace(min_num_measures_target=2, max_num_measure_target=6)
    torch.manual_seed(int(fx["split_seed"]))
    draws = [LatentRNNTrainer.draw_split(me, 16) for _ in range(8)]
    assert np.array_equal(np.array(draws), fx["split_draws"])
    n_past, n_target = LatentRNNTrainer.draw_split(me, 16, fix_num_target=3)
    assert [n_past, n_target] == list(fx["split_fixed3"][:2])


def test_constraint_window_draws_follow_the_reference_stream():
    from inpaintnet_amd.arnn import AnticipationRNNGaussianRegTrainer as T
    fx = G.load("feed_helpers")
    me = types.SimpleNamespace(min_num_measures_target=2, max_num_measure_target=6, measure_seq_len=24,
                               dataset=types.SimpleNamespace(n_bars=16))
    score = torch.from_numpy(fx["score"])
    torch.manual_seed(int(fx["constraints_seed"]))
    ticks = []
    for i in range(8):
        loc, start, end = T.get_constraints_location(me, score)
        ticks.append((start, end))
        if i == 0:
            assert np.array_equal(loc.numpy(), fx["constraints0"])
    assert np.array_equal(np.array(ticks), fx["constraints_ticks"])


def test_early_stopping_patience_rule():
    """utils/trainer.py:379-413: an epoch only counts as an improvement if the validation loss drops by >= 1e-5;
    five epochs in a row without one stop the run."""
    from inpaintnet_amd.trainer import EarlyStopping
    es = EarlyStopping()
    losses = [1.0, 0.9, 0.9, 0.95, 0.9 - 5e-6, 0.8, 0.8, 0.81, 0.82, 0.83, 0.84]
    flags = [bool(es(v, None)) for v in losses]
    #        best  impr  +1   +2    +3 (tiny)  impr  +1   +2    +3    +4    +5 -> stop
    assert flags == [False] * 10 + [True]
    assert es.early_stop and es.counter == 5 and abs(es.val_loss_min - 0.8) < 1e-12
    es2 = EarlyStopping(patience=2)
    assert [bool(es2(v)) for v in [3.0, 3.0, 3.0]] == [False, False, True]


def test_shard_batch_single_process_is_identity():
    x = torch.arange(12).view(6, 2)
    assert dp.shard_batch(x) is x
    t = dp.shard_batch((x, None))
    assert t[0] is x and t[1] is None
