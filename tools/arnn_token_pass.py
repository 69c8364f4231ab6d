#!/usr/bin/env python3
"""The token pass of AnticipationRNN's free-running step alone (ops.arnn_generate, L = 384): python tools/arnn_token_pass.py
With INET_ARNN_GEN_STAMPS=1 the persistent kernel (csrc/arnn_gen.hip) also leaves wall-clock stamps of the phases of a tick in its
workspace: the anatomy of a tick as workgroup C (layer-0 cell, linear_1, head, argmax) and workgroup Bi_0 (layer-1 product + cell)
see it is printed below the timings."""
import os, sys, time, types
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from inpaintnet_amd import ops, synthetic
from inpaintnet_amd.arnn import ConstraintModelGaussianReg
sys.stdout = sys.stderr
NOTES = int(sys.argv[1]) if len(sys.argv) > 1 else bench.NUM_NOTES          # (V > 64: the two-register-set build of the token pass)
ds = synthetic.SyntheticFolkDataset(num_notes=NOTES)
ds.metadatas = [types.SimpleNamespace(num_values=6), types.SimpleNamespace(num_values=6)]
m = ConstraintModelGaussianReg(ds, note_embedding_dim=10, metadata_embedding_dim=2, num_lstm_constraints_units=256,
                               num_lstm_generation_units=256, linear_hidden_size=256, num_layers=2, dropout_input_prob=0.2,
                               dropout_prob=0.2, unary_constraint=True, teacher_forcing=True)
pr = m.param
oc0 = torch.randn(384, 256, device="cuda") * 0.1
args = (pr("note_embeddings.0.weight"), oc0, pr("lstm_generation.0.weight_ih_l0"), pr("lstm_generation.0.bias_ih_l0"),
        pr("lstm_generation.0.weight_hh_l0"), pr("lstm_generation.0.bias_hh_l0"), pr("lstm_generation.1.weight_ih_l0"),
        pr("lstm_generation.1.bias_ih_l0"), pr("lstm_generation.1.weight_hh_l0"), pr("lstm_generation.1.bias_hh_l0"),
        pr("linear_1.weight"), pr("linear_1.bias"), pr("linear_ouput_notes.0.weight"), pr("linear_ouput_notes.0.bias"))
# option key 14: 0 = four launches per tick (round 4), 1 = one persistent launch (csrc/arnn_gen.hip), 2 = ... on one XCD
for mode in (0, 1, 2, 3):
    ops.set_option(14, mode)
    for _ in range(3): t = ops.arnn_generate(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): t = ops.arnn_generate(*args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"V = {NOTES}, mode {mode}: token pass, 384 ticks: {1e3 * dt:.2f} ms  ({1e6 * dt / 384:.2f} us per tick)  tokens {t[:8].tolist()} "
          f"chain status {ops.chain_status()}")
ops.set_option(14, 2)

if os.environ.get("INET_ARNN_GEN_STAMPS") == "1":
    L, V = 384, NOTES
    ops._ARNN_KEEP_WS.append(None)
    t = ops.arnn_generate(*args)
    torch.cuda.synchronize()
    ws = ops._ARNN_KEEP_WS[0]
    off = L * 1024 + V * 1024 + (2 * (4 * 256 + 2 * 1024 + 16) + 64) + 64      # csrc/arnn_gen.hip arnn_token_pass_stamps_offset
    st = ws[off:off + 32 * L].cpu().view(torch.int64).view(2, L, 8).double() * 0.01        # us (100 MHz wall clock)
    c, b = st[0, 8:L - 1], st[1, 8:L - 1]
    names_c = ["tok known -> gates, cell, publish h0", "wait for h1 (Bi's product + cell + two hand-offs)", "barrier", "linear_1 product + barrier",
               "head product + barrier", "argmax + barrier", "look at next tick's hh0 (requested under the head)"]
    print("C, mean us per phase over ticks 8..L-2:")
    for i, n in enumerate(names_c):
        print(f"  {float((c[:, i + 1] - c[:, i]).mean()):6.2f}  {n}")
    print(f"  {float((st[0, 9:L, 0] - st[0, 8:L - 1, 0]).mean()):6.2f}  tick period")
    names_b = ["wait for hh1 (requested early)", "wait for h0", "barrier", "W_ih1 product + barrier", "cell + publish h1"]
    print("Bi_0:")
    for i, n in enumerate(names_b):
        print(f"  {float((b[:, i + 1] - b[:, i]).mean()):6.2f}  {n}")
    # one-way hand-off estimates from the two clocks (the wall clock is chip-wide): C publishes h0 (stamp 1) -> Bi_0 has it (stamp 2)
    print(f"  h0: C published -> Bi_0 holds it {float((b[:, 2] - c[:, 1]).mean()):6.2f} us;  h1: Bi_0 published -> C holds it "
          f"{float((c[:, 2] - b[:, 5]).mean()):6.2f} us")
