"""The reference-side binding INTEGRATION.md shows (section 2: MeasureVAE/decoder.py:412-453 replaced by one ctypes call) is
EXECUTED here as written: the fenced block is extracted from the file, run in a fresh interpreter against the built library
on a non-default stream, and its output compared with the reference's golden decoder vectors (tests/golden/vae_full.npz:
logits within 2e-5 of the tensor max, tokens exact).  (VERDICT r04 row b: the quoted stub had gone stale unnoticed.)"""
import os
import re
import subprocess
import sys
import textwrap

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = textwrap.dedent('''
    import sys, numpy as np
    sys.path.insert(0, {repo!r})
    from tests import golden_util as G
    fx = G.load("vae_full")
    c = G.CFGS["full"]
    cfg = VaeConfig(c["V"], c["E"], c["H"], c["Z"], c["H"], 4, 6)
    dev = torch.device("cuda", 0)
    flat = pack_params(cfg, G.vae_params("full", fx), dev)
    z = torch.from_numpy(fx["dec_z"]).to(dev)
    side = torch.cuda.Stream()                       # NOT the default stream: a stale binding passed the stream as the seed
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        w, s = hip_decode(cfg, flat, z)
        tgt = torch.from_numpy(fx["dec_tf_samples"]).to(dev).reshape(z.shape[0], -1).contiguous()
        wt, st = hip_decode(cfg, flat, z, tgt, True)
    side.synchronize()
    err = float((w.cpu() - torch.from_numpy(fx["dec_eval_weights"])).abs().max()) / float(np.abs(fx["dec_eval_weights"]).max())
    errt = float((wt.cpu() - torch.from_numpy(fx["dec_tf_weights"])).abs().max()) / float(np.abs(fx["dec_tf_weights"]).max())
    clear = fx["dec_eval_margin"] > 1e-4
    same = (s.cpu().numpy()[:, 0] == fx["dec_eval_samples"][:, 0])
    print("RESULT", err, errt, int(clear.sum()), int((same | ~clear).all()), int(np.array_equal(st.cpu().numpy(), fx["dec_tf_samples"])))
''')


def stub_source():
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = [b for b in blocks if "def hip_decode" in b]
    assert len(stub) == 1
    return stub[0]


@pytest.mark.gpu
def test_the_quoted_decoder_binding_runs_and_matches_the_reference(tmp_path):
    from inpaintnet_amd import _lib
    script = tmp_path / "hip_decoder_stub.py"
    script.write_text(stub_source() + DRIVER.format(repo=REPO))
    env = dict(os.environ, INET_LIB_PATH=_lib.LIB_PATH)
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    err, errt, nclear, tokens_ok, tf_ok = float(line[1]), float(line[2]), int(line[3]), int(line[4]), int(line[5])
    assert err < 2e-5 and errt < 2e-5, (err, errt)
    assert nclear > 0 and tokens_ok == 1 and tf_ok == 1
