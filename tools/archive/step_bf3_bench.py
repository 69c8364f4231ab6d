"""Per-launch time of the big-batch GRU step kernels (csrc/gru_step_bf3.hip) inside the MeasureVAE encoder's forward pass:
python tools/step_bf3_bench.py [B ...]   (HIP events per launch, inet_prof_*)"""
import csv
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from inpaintnet_amd import ops, synthetic  # noqa: E402

cfg = ops.vae_config(48)
table, total = ops.vae_param_table(cfg)
params = torch.zeros(total, device="cuda")
for name, off, shape in table:
    n = 1
    for s_ in shape:
        n *= s_
    params[off:off + n] = torch.from_numpy(synthetic.det_param(name, tuple(shape))).reshape(-1).cuda()
for B in [int(a) for a in sys.argv[1:]] or [2048, 4096]:
    tok = torch.from_numpy(synthetic.det_tokens(f"stepbench{B}", (B, 24), 48)).cuda()
    for save in (False, True):
        for _ in range(2):
            ops.encoder_fwd(cfg, tok, params, save=save)
        torch.cuda.synchronize()
        ops.prof_enable(True)
        for _ in range(4):
            ops.encoder_fwd(cfg, tok, params, save=save)
        torch.cuda.synchronize()
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "l.csv")
            ops.prof_dump(p)
            rows = [r for r in csv.DictReader(open(p))]
        ops.prof_enable(False)
        us = [float(r["us"]) for r in rows if r["label"].startswith("gru_step_bf3")]
        per_call = len(us) // 4
        l0 = [u for k, u in enumerate(us) if (k % per_call) < 24 and (k % per_call) > 0]
        l1 = [u for k, u in enumerate(us) if (k % per_call) >= 24]
        tot = sum(float(r["us"]) for r in rows) / 4
        print(f"B={B} save={int(save)}: layer 0 steps {sum(l0) / len(l0):6.1f} us (step 0: {us[0]:5.1f}), layer 1 steps "
              f"{sum(l1) / len(l1):6.1f} us; encoder forward {tot:8.1f} us of kernel time", flush=True)
