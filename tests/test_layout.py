"""Host-only checks of the C-ABI library: it loads, exports every symbol declared in
include/inpaintnet_hip.h, and its parameter-arena tables agree with inpaintnet_amd/layout.py
(and therefore with the reference's state_dict order captured in the golden fixtures)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from inpaintnet_amd import _lib, layout

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    _lib.build(verbose=False)
    return _lib.lib()


def test_exports_match_header(L):
    hdr = open(os.path.join(REPO, "include", "inpaintnet_hip.h")).read()
    declared = set(re.findall(r"\b(inet_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name)
    assert L.inet_abi_version() == 1


def test_kernel_handles_are_collected_without_a_gpu(L):
    """csrc/preload.hip: the library defines __hipRegisterFunction itself (it is linked -Bsymbolic-functions), files every kernel
    handle its translation units register at load time and passes the call on to the HIP runtime -- so inet_preload() can first-touch
    ALL kernels, template instantiations included, with no list to keep in step by hand.  No GPU needed for the count; without a
    device inet_preload() refuses (-2) instead of pretending."""
    import subprocess
    assert L.inet_kernel_count() >= 200
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert " T __hipRegisterFunction" in syms
    import torch
    if not torch.cuda.is_available():
        assert L.inet_preload() == -2


def _entries(cfg, count_fn, info_fn):
    n = count_fn(C.byref(cfg))
    out = []
    for i in range(n):
        name = C.create_string_buffer(128)
        off = C.c_int64()
        dims = (C.c_int64 * 4)()
        nd = C.c_int()
        assert info_fn(C.byref(cfg), i, name, 128, C.byref(off), dims, C.byref(nd)) == 0
        out.append((name.value.decode(), off.value, tuple(dims[j] for j in range(nd.value))))
    return out


@pytest.mark.parametrize("dims", [(12, 4, 16, 8), (48, 10, 512, 256)])
def test_vae_layout_matches_python(L, dims):
    V, E, H, Z = dims
    cfg = _lib.VaeConfig(V, E, H, Z, H, 4, 6)
    ents = _entries(cfg, L.inet_vae_param_count, L.inet_vae_param_info)
    offs, total = layout.arena_offsets(layout.vae_param_shapes(V, E, H, Z, H))
    assert [e[0] for e in ents] == list(offs)
    for name, off, shape in ents:
        assert offs[name] == (off, shape), name
    assert L.inet_vae_param_floats(C.byref(cfg)) == total
    if V == 48:
        assert sum(int(np.prod(s)) for _, _, s in ents) == 17666555


@pytest.mark.parametrize("auto_reg", [0, 1])
def test_latent_layout_matches_python(L, auto_reg):
    cfg = _lib.LatentConfig(256, 512, auto_reg)
    ents = _entries(cfg, L.inet_latent_param_count, L.inet_latent_param_info)
    offs, total = layout.arena_offsets(layout.latent_param_shapes(256, 512, bool(auto_reg)))
    assert [e[0] for e in ents] == list(offs)
    for name, off, shape in ents:
        assert offs[name] == (off, shape), name
    assert L.inet_latent_param_floats(C.byref(cfg)) == total
    n = sum(int(np.prod(s)) for _, _, s in ents)
    assert n == (41468160 if auto_reg else 39901441)


def test_golden_state_dict_order():
    """The fixture stores the reference's own state_dict(): same keys, same shapes as layout.py."""
    fx = np.load(os.path.join(REPO, "tests", "golden", "vae_small.npz"))
    keys = [k[6:] for k in fx.files if k.startswith("param/")]
    shapes = layout.vae_param_shapes(12, 4, 16, 8, 16)
    assert keys == list(shapes)
    for k in keys:
        assert tuple(fx["param/" + k].shape) == tuple(shapes[k])


def test_invalid_args_rejected(L):
    bad = _lib.VaeConfig(48, 10, 500, 256, 512, 4, 6)      # hidden not a multiple of 16
    assert L.inet_vae_param_count(C.byref(bad)) == -1
    assert L.inet_vae_encoder_ws_bytes(C.byref(bad), 4, 0) == -1
    ok = _lib.VaeConfig(48, 10, 512, 256, 512, 4, 6)
    assert L.inet_vae_encoder_ws_bytes(C.byref(ok), 0, 0) == -1
    assert L.inet_vae_encoder_ws_bytes(C.byref(ok), 256, 1) > 0
    assert L.inet_vae_decoder_ws_bytes(C.byref(ok), 256, 1) > 0


def test_option_keys_validate_their_values(L):
    """inet_set_option rejects what it does not know (no GPU needed: the options are host state)."""
    assert L.inet_set_option(13, 1) == 0 and L.inet_set_option(13, 0) == 0      # side streams in rotation: 0 (all) .. 3
    assert L.inet_set_option(13, 4) == -1 and L.inet_set_option(13, -1) == -1
    assert L.inet_set_option(10, 1) == -1 and L.inet_set_option(11, 1) == -1     # removed in round 4
    assert L.inet_set_option(7, 6) == -1 and L.inet_set_option(8, 6) == -1       # six piece products: removed
    assert L.inet_set_option(99, 0) == -1
