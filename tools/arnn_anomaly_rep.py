#!/usr/bin/env python3
"""One repetition of the BENCH_r05 anomaly hunt (under per-launch events the driver's box read lstm_chain_bwd at
470 us per 32-step chunk, 110 us everywhere else): AnticipationRNN's teacher-forced step un-profiled, then bench.py's per-kernel table
of the same step (inet_prof_enable: a hipEventRecord pair around every launch), with the slow-wait recorder's threshold lowered to
64 polls so that waits inside the chains leave their coordinates.  One JSON line per run; run it on many fresh leases."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def main():
    from inpaintnet_amd import ops
    torch.cuda.set_device(0)
    ops.preload()
    ops.set_option(16, 64)
    ops.slow_waits(reset=True)
    r = bench.arnn_extra(tables=True)["anticipation_rnn_train"]
    k = r["kernels"]
    rows = {x["kernel"].split()[0]: x for x in k.get("top", [])} if "top" in k else {}
    out = {"ms_per_step": r["ms_per_step"], "ms_per_step_free_running": r["ms_per_step_free_running"],
           "table": k.get("kernels", "consistent"), "step_kernel_ms": k["step_kernel_ms"],
           "lstm_chain_bwd_us": rows.get("lstm_chain_bwd", {}).get("avg_us") or (k.get("largest") or {}).get("avg_us"),
           "lstm_chain_fwd_us": rows.get("lstm_chain_fwd", {}).get("avg_us"),
           "noted_while_profiled": k.get("waits_noted_while_profiled"), "slow64_while_profiled": k.get("slow_waits_while_profiled"),
           "entries": k.get("slow_wait_entries", [])[:4]}
    rest = ops.slow_waits(reset=True)
    out["noted_unprofiled_steps"] = rest["noted"]
    out["slow64_unprofiled_steps"] = rest["count"]
    by_xcc = {}
    for e in rest["entries"]:
        key = f"{e['kernel']}@xcc{e['xcc']}"
        by_xcc[key] = max(by_xcc.get(key, 0), e["polls"])
    out["longest_polls_by_kernel_xcc"] = by_xcc
    print(json.dumps(out))


if __name__ == "__main__":
    main()
