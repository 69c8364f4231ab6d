"""Helpers shared by the oracle and GPU parity tests."""
import os

import numpy as np
import torch

from inpaintnet_amd import layout, synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CFGS = {
    "small": dict(V=12, E=4, H=16, Z=8),
    "mid": dict(V=20, E=6, H=48, Z=24),
    "full": dict(V=48, E=10, H=512, Z=256),
    # no fixture: H % 256 == 0 turns on the fragment-major operand path of the step kernels; used against the oracle
    "pk": dict(V=20, E=6, H=256, Z=24),
    # no fixture: a vocabulary / embedding wider than the token-sum kernels take (V > 64, E > 16): one-hot GEMM path
    "wide": dict(V=70, E=20, H=48, Z=24),
    # no fixture: vocabularies that are not a multiple of 16 (the real one is data-derived: MeasureVAE/measure_vae.py:56) on the
    # fragment-major path: the fused decode kernel pads its last 16-column block, the token segment-sum takes up to 128 rows
    "v61": dict(V=61, E=10, H=256, Z=24),
    "v93": dict(V=93, E=10, H=256, Z=24),
    # no fixture: beyond every fast path's limit (V > 128): per-tick decode, one-hot products for the embedding gradients
    "v140": dict(V=140, E=10, H=256, Z=24),
}


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def vae_params(name, fx=None, prefix=""):
    """state_dict-keyed float32 torch tensors for the fixture's model (regenerated
    from the deterministic generator; cross-checked against stored copies when present)."""
    c = CFGS[name]
    shapes = layout.vae_param_shapes(c["V"], c["E"], c["H"], c["Z"], c["H"], prefix=prefix)
    P = {k: torch.from_numpy(synthetic.det_param(k, s)) for k, s in shapes.items()}
    if fx is not None:
        for k in P:
            key = "param/" + k
            if key in fx.files:
                assert np.array_equal(fx[key], P[k].numpy()), k
    return P


def latent_params(name, auto_reg):
    c = CFGS[name]
    P = vae_params(name, prefix="vae_model.")
    shapes = layout.latent_param_shapes(c["Z"], c["H"], auto_reg)
    for k, s in shapes.items():
        P[k] = torch.from_numpy(synthetic.det_param(k, s))
    return P


def unique_rows(margin, tol=1e-4):
    return margin > tol


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

ARNN_CFGS = {
    "small": dict(V=12, E=4, Em=2, H=16, LH=16),
    "full": dict(V=48, E=10, Em=2, H=256, LH=256),
}


def arnn_params(name, fx=None):
    c = ARNN_CFGS[name]
    shapes = layout.arnn_param_shapes(c["V"], c["E"], c["Em"], c["H"], c["LH"])
    P = {k: torch.from_numpy(synthetic.det_param(k, s)) for k, s in shapes.items()}
    if fx is not None:
        assert list(fx["param_keys"]) == list(shapes)
        for k, shp in zip(fx["param_keys"], fx["param_shapes"]):
            assert tuple(int(d) for d in str(shp).split(",")) == tuple(shapes[str(k)]), k
        for k in P:
            if "param/" + k in fx.files:
                assert np.array_equal(fx["param/" + k], P[k].numpy()), k
    return P


def latent_params_from_fixture(fx, prefix="param/"):
    """state_dict of a fixture that stores its full weights (small configs)."""
    return {k[len(prefix):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(prefix)}


def is_chain(label, kind, nprob, T, B, H=512):
    """A chain-kernel profile label of either generation: 'gru_chain_fwd ms4 np2 T24 B256 H512' (first: csrc/gru_chain.hip; 'ms4x2'
    = the build for two launches per CU) or 'gru_chain_fwd v2w4 p9 np2 T24 B256 H512' (second: csrc/gru_chain2.hip, waves per
    workgroup and piece products)."""
    return label.startswith(f"gru_chain_{kind} ") and label.endswith(f" np{nprob} T{T} B{B} H{H}")
