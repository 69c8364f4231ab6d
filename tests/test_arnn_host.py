"""Host-side logic of the AnticipationRNN surface that needs no GPU."""
import torch

from inpaintnet_amd.arnn import free_positions


def _ref_free(loc):
    # AnticipationRNN/anticipation_rnn_gauss_reg_model.py:433
    return (loc[0, 0, :] == 0).nonzero().squeeze(-1)


def test_free_positions_matches_the_reference_expression_and_is_cached():
    loc = torch.ones(3, 1, 40, dtype=torch.int64)
    loc[:, :, 12:20] = 0
    f = free_positions(loc)
    assert torch.equal(f, _ref_free(loc))
    assert free_positions(loc) is f                          # second call: no nonzero()


def test_free_positions_cache_is_dropped_when_the_tensor_is_written():
    loc = torch.ones(2, 1, 16, dtype=torch.int64)
    loc[:, :, 4:8] = 0
    f = free_positions(loc)
    loc[:, :, 8:10] = 0                                      # in-place write: _version moves
    g = free_positions(loc)
    assert g is not f and torch.equal(g, _ref_free(loc)) and g.numel() == 6


def test_free_positions_from_a_host_copy():
    host = torch.rand(4, 1, 32) < 0.4                        # the baseline trainer's Bernoulli mask (bool)
    host = host[:1].repeat(4, 1, 1)
    dev = host.to(torch.int64)
    f = free_positions(dev, host_copy=host)
    assert torch.equal(f, _ref_free(dev))
    assert free_positions(dev) is f


def test_trainer_fills_the_cache(monkeypatch):
    """process_batch_data leaves the device tensor with its free positions attached (no device round trip in the step)."""
    import types
    from inpaintnet_amd import arnn, synthetic
    monkeypatch.setattr(arnn, "to_cuda_variable_long", lambda t: t.to(torch.int64))
    ds = synthetic.SyntheticFolkDataset(num_notes=48)
    ds.metadatas = [types.SimpleNamespace(num_values=6), types.SimpleNamespace(num_values=6)]
    tr = arnn.AnticipationRNNGaussianRegTrainer.__new__(arnn.AnticipationRNNGaussianRegTrainer)
    tr.dataset = ds
    tr.min_num_measures_target, tr.max_num_measure_target = 2, 6
    tr.measure_seq_len = ds.subdivision * ds.num_beats_per_bar
    score = torch.from_numpy(synthetic.folk_score(2, 48, seed=3))
    md = torch.from_numpy(synthetic.folk_metadata(2))
    torch.manual_seed(1)
    _, _, loc, start, end = tr.process_batch_data((score, md))
    cached = getattr(loc, "_inet_free", None)
    assert cached is not None and cached[0] == loc._version
    assert torch.equal(cached[1], _ref_free(loc))
    assert cached[1][0].item() == start and cached[1][-1].item() == end - 1
