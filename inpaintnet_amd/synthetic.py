"""Deterministic synthetic weights / FolkDB-shaped token tensors.

Everything here is a pure function of (name, shape, seed) built on numpy's
Philox bit generator, so full-size weights and inputs never have to be
committed: the golden generator (oracle/gen_golden.py, run where the reference
is importable) and the tests (run anywhere) regenerate identical values.

Shapes follow the reference's dataset tensors:
  score    (N, 1, 384) int32   -- DatasetManager/the_session/folk_dataset.py:852-861
  metadata (N, 1, 384, 3) int32
"""
import zlib

import numpy as np


def _gen(name, seed):
    key = (zlib.crc32(name.encode("utf-8")) << 32) | (seed & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))


def det_normal(name, shape, std=1.0, seed=0):
    g = _gen(name, seed)
    return (g.standard_normal(size=shape, dtype=np.float64) * std).astype(np.float32)


def det_uniform(name, shape, lo=-1.0, hi=1.0, seed=0):
    g = _gen(name, seed)
    return (g.random(size=shape, dtype=np.float64) * (hi - lo) + lo).astype(np.float32)


def det_tokens(name, shape, num_notes, seed=0):
    g = _gen(name, seed)
    return g.integers(0, num_notes, size=shape, dtype=np.int64)


def det_param(name, shape, seed=0):
    """Xavier-normal-shaped value for 'weight' tensors (as the reference's
    xavier_initialization does for every parameter whose name contains
    'weight': MeasureVAE/encoder.py:71-78), small uniform for everything else
    (biases, b_0, x_0) so that no term of the forward pass is trivially zero."""
    shape = tuple(int(s) for s in shape)
    if "weight" in name and len(shape) == 2:
        fan_out, fan_in = shape
        std = float(np.sqrt(2.0 / (fan_in + fan_out)))
        return det_normal(name, shape, std, seed)
    fan = shape[-1] if len(shape) else 1
    bound = 1.0 / float(np.sqrt(max(fan, 4)))
    return det_uniform(name, shape, -bound, bound, seed)


def det_state_dict(shapes, seed=0):
    """shapes: {key: shape} -> {key: float32 ndarray}"""
    return {k: det_param(k, s, seed) for k, s in shapes.items()}


def folk_score(n_seq, num_notes, n_bars=16, ticks_per_bar=24, seed=0):
    """(N,1,384) int32 tokens, uniform iid in [0,V)."""
    t = det_tokens("folk_score", (n_seq, 1, n_bars * ticks_per_bar), num_notes, seed)
    return t.astype(np.int32)


def folk_metadata(n_seq, n_bars=16, ticks_per_bar=24, subdivision=6):
    L = n_bars * ticks_per_bar
    md = np.zeros((n_seq, 1, L, 3), dtype=np.int32)
    md[..., 0] = 2
    md[..., 1] = (np.arange(L) % subdivision)[None, None, :]
    return md


class SyntheticFolkDataset:
    """Stand-in for FolkDatasetNBars: carries exactly the attributes the model
    constructors and trainers read (SURVEY.md section 8b 'dataset argument')."""

    def __init__(self, num_notes=48, n_bars=16, n_seq=1024, seed=0):
        self.num_notes = num_notes
        self.note2index_dicts = [{i: i for i in range(num_notes)}]
        self.index2note_dicts = [{i: i for i in range(num_notes)}]
        self.n_bars = n_bars
        self.subdivision = 6
        self.num_beats_per_bar = 4
        self.num_voices = 1
        self.n_seq = n_seq
        self.seed = seed

    def __repr__(self):
        return f"SyntheticFolk(V={self.num_notes},bars={self.n_bars},n={self.n_seq},seed={self.seed})"

    def tensors(self):
        return (folk_score(self.n_seq, self.num_notes, self.n_bars, seed=self.seed),
                folk_metadata(self.n_seq, self.n_bars))

    def data_loaders(self, batch_size, split=(0.85, 0.10)):
        """Same contract as MusicDataset.data_loaders (music_dataset.py:177-221): (train, val, eval) loaders over
        the (score, metadata) tensors split by fraction, drop_last=True, only the train loader shuffles."""
        import torch
        from .feed import BatchLoader
        assert sum(split) < 1
        score, md = (torch.from_numpy(t) for t in self.tensors())
        n = score.shape[0]
        a = int(split[0] * n)
        b = int((split[0] + split[1]) * n)
        return (BatchLoader((score[:a], md[:a]), batch_size, shuffle=True),
                BatchLoader((score[a:b], md[a:b]), batch_size, shuffle=False),
                BatchLoader((score[b:], md[b:]), batch_size, shuffle=False))
