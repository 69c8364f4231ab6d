#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of `bench.py` into the files committed under profiles/.

    python tools/pmc_summary.py stats  <kernel_trace.csv> [seq.json]          -> per-kernel time table (text, stdout)
    python tools/pmc_summary.py pmc    <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [seq.json]

PMC conventions (MI355X_MICROARCH.md, section HBM): FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts
128-byte requests as 64 bytes for wide coalesced reads, so the read side is DOUBLED here; WRITE_SIZE is taken as is.  The two
counters are collected in separate passes (TCC slots).  Kernels are keyed by short name + launch grid (threads), because
one template instantiation serves several problem shapes."""
import csv
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"((?:gru_step_fwd|gru_step_bwd|gemm\w*|lstm_step_fwd|lstm_step_bwd|logits_argmax|gru_chain\w*|lstm_chain\w*|"
                  r"decode_\w+)_kernel<[^>]*>)", name)
    if m:
        return m.group(1)
    m = re.search(r"\(anonymous namespace\)::(\w+)\(", name)
    if m:
        return m.group(1)
    m = re.search(r"(\w+)<", name)
    return (m.group(1) if m else name)[:60]


def stats(path, top=40, seq_json=None):
    """Per-kernel time table of a kernel trace.  With the launch-order sequences of bench.py (INET_BENCH_SEQ) a kernel
    whose instantiation + grid serves several shapes is also listed per shape (`name #label`)."""
    seqs = json.load(open(seq_json)) if seq_json else {}
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    seen = defaultdict(int)
    with open(path, newline="") as f:
        rows = sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        k = short(r["Kernel_Name"])
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        keys = [k]
        gkey = f'{k}|g{int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)}'
        if gkey in seqs:
            keys.append(k + " #" + seqs[gkey][seen[gkey] % len(seqs[gkey])])
            seen[gkey] += 1
        for kk in keys:
            a = agg[kk]
            a[0] += 1; a[1] += us; a[2] = min(a[2], us); a[3] = max(a[3], us)
    tot = sum(a[1] for k, a in agg.items() if " #" not in k)
    print(f"# {sum(a[0] for k, a in agg.items() if ' #' not in k)} dispatches, {tot:.1f} us total kernel time")
    print(f"{'kernel':<58} {'calls':>7} {'total_us':>11} {'pct':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:<58} {a[0]:>7d} {a[1]:>11.1f} {100 * a[1] / tot:>6.2f} {a[1] / a[0]:>9.2f} {a[2]:>9.2f} {a[3]:>9.2f}")


def read_counter(path, counter, seqs=None):
    """key -> [launches, sum counter, sum us].  `seqs` {key: [label of the 1st, 2nd, ... launch of that key in a step]}
    (bench.py INET_BENCH_SEQ) splits a kernel|grid key that serves several shapes into key#label entries by launch order
    (dispatch ids are in host launch order; every step launches the same sequence)."""
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), f'{short(r["Kernel_Name"])}|g{r["Grid_Size"]}', float(r["Counter_Value"]),
                             (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    out = defaultdict(lambda: [0, 0.0, 0.0])
    seen = defaultdict(int)
    for _, key, val, us in rows:
        targets = [key]
        if seqs and key in seqs:
            targets.append(key + "#" + seqs[key][seen[key] % len(seqs[key])])
            seen[key] += 1
        for k in targets:
            a = out[k]
            a[0] += 1
            a[1] += val
            a[2] += us
    return out


def pmc(fetch_csv, write_csv, out_json, seq_json=None):
    seqs = json.load(open(seq_json)) if seq_json else None
    fe = read_counter(fetch_csv, "FETCH_SIZE", seqs)
    wr = read_counter(write_csv, "WRITE_SIZE", seqs)
    kernels = {}
    for key in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0, 0])[2])):
        f, w = fe.get(key), wr.get(key)
        fetch_kb = f[1] / f[0] if f else 0.0
        write_kb = w[1] / w[0] if w else 0.0
        kernels[key] = {"launches": f[0] if f else w[0], "avg_us_profiled": round((f or w)[2] / (f or w)[0], 3),
                        "fetch_size_kb_raw": round(fetch_kb, 2), "write_size_kb": round(write_kb, 2),
                        "hbm_mbytes_per_launch": round((2.0 * fetch_kb + write_kb) * 1024 / 1e6, 4)}
    labels = sorted(seqs.get("__labels__", [])) if seqs else []
    doc = {"labels": labels, "label_hash": _label_hash(labels),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 30 "
                     "--warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline",
           "corrections": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); KB -> bytes x1024",
           "kernels": kernels}
    json.dump(doc, open(out_json, "w"), indent=1)
    print(f"wrote {out_json}: {len(kernels)} kernel/grid keys")
    for k, v in list(kernels.items())[:14]:
        print(f"  {k:<62} n={v['launches']:<6} {v['avg_us_profiled']:>9.2f} us  {v['hbm_mbytes_per_launch']:>9.3f} MB/launch")


def _label_hash(labels):
    import hashlib
    return hashlib.sha1("\n".join(sorted(set(labels))).encode()).hexdigest()[:16]


def merge(out_json, *parts):
    """Union of several pmc() outputs (teacher-forced and free-running passes): a key keeps the entry of the first file that
    has it, per-shape entries (key#label) of every file are kept."""
    doc = None
    for p in parts:
        d = json.load(open(p))
        if doc is None:
            doc = d
            continue
        for k, v in d["kernels"].items():
            doc["kernels"].setdefault(k, v)
        doc["labels"] = sorted(set(doc.get("labels", [])) | set(d.get("labels", [])))
    doc["label_hash"] = _label_hash(doc.get("labels", []))
    doc["source"] += " ; passes with INET_BENCH_COIN=tf and =fr merged (every step of a pass launches the same kernel sequence)"
    json.dump(doc, open(out_json, "w"), indent=1)
    print(f"wrote {out_json}: {len(doc['kernels'])} kernel/grid keys")


def busy(csv_path, traffic_json=None, seq_json=None, simds=1024, xccs=8):
    """MFMA-busy pass (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE): per kernel|grid key (split by
    launch order like the traffic passes) the share of SIMD-cycles in which the MFMA pipe was busy,
        mfma_busy = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * SIMDs),
    the clock the kernel ran at (GRBM_GUI_ACTIVE / wall time: the chip clocks to its power budget, MI355X_MICROARCH.md DVFS), and
    SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE.  The CSV holds every counter SUMMED over its hardware instances: GRBM_GUI_ACTIVE over the 8
    XCCs (rocprofv3's own MfmaUtil takes the max over instances), so it is divided by `xccs` here -- the first table of round 4
    showed 17-20 "GHz" and MFMA-busy fractions 8x too small without that.  Kernels of a few tens of microseconds read too high a
    clock: the counter window is wider than the kernel's timestamps, so GRBM_GUI_ACTIVE -- the denominator of mfma_busy too --
    is inflated (3.2-15.8 "GHz" for everything under ~50 us).  A row whose derived clock exceeds the part's 2.4 GHz
    (+2 %) is marked `unreliable`: its clock is not reported, and its MFMA-busy share is given as a LOWER BOUND -- the larger of
    busy / (inflated window) and busy / (duration x 2.4 GHz x SIMDs); bench.py does not quote unreliable rows.  Printed as a
    table; with `traffic_json` the fractions are merged into that file's entries."""
    MAX_GHZ = 2.4
    seqs = json.load(open(seq_json)) if seq_json else None
    cols = {}
    for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
        cols[name] = read_counter(csv_path, name, seqs)
    gui = cols["GRBM_GUI_ACTIVE"]
    rows = []
    for key, (n, gsum, us) in gui.items():
        if n == 0 or gsum <= 0:
            continue
        gsum = gsum / xccs
        mf = cols["SQ_VALU_MFMA_BUSY_CYCLES"].get(key, [0, 0.0, 0.0])[1]
        sq = cols["SQ_BUSY_CYCLES"].get(key, [0, 0.0, 0.0])[1]
        ghz = gsum / n / (us / n * 1e3)
        mfb = mf / (gsum * simds)
        bad = ghz > 1.02 * MAX_GHZ
        if bad:                                                   # the window, not the kernel: lower bound from both denominators
            mfb = max(mfb, mf / (us * 1e3 * MAX_GHZ * simds))
        rows.append((us, key, n, us / n, mfb, ghz, sq / gsum, bad))
    rows.sort(reverse=True)
    print(f"{'kernel|grid[#label]':<78} {'calls':>6} {'avg_us':>9} {'mfma_busy':>10} {'clock_GHz':>10} {'sq_busy/gui':>12}")
    for us, key, n, avg, mfb, ghz, sqb, bad in rows[:40]:
        if bad:
            print(f"{key:<78} {n:>6d} {avg:>9.2f} {'>=' + format(mfb, '.3f'):>10} {'unreliable':>10} {'-':>12}")
        else:
            print(f"{key:<78} {n:>6d} {avg:>9.2f} {mfb:>10.3f} {ghz:>10.2f} {sqb:>12.2f}")
    if traffic_json:
        doc = json.load(open(traffic_json))
        for us, key, n, avg, mfb, ghz, sqb, bad in rows:
            if key in doc["kernels"]:
                ent = doc["kernels"][key]
                for k in ("mfma_busy_frac", "clock_ghz_profiled", "mfma_busy_frac_lower_bound", "busy_counters_unreliable"):
                    ent.pop(k, None)
                if bad:
                    ent["mfma_busy_frac_lower_bound"] = round(mfb, 4)
                    ent["busy_counters_unreliable"] = True
                else:
                    ent["mfma_busy_frac"] = round(mfb, 4)
                    ent["clock_ghz_profiled"] = round(ghz, 3)
        doc["source"] += (" ; mfma_busy_frac / clock_ghz_profiled from a pass with --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
                          "GRBM_GUI_ACTIVE (rows whose counter window exceeds the kernel -- derived clock > 2.4 GHz -- carry "
                          "mfma_busy_frac_lower_bound instead)")
        json.dump(doc, open(traffic_json, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "busy":
        busy(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None, sys.argv[4] if len(sys.argv) > 4 else None)
    elif sys.argv[1] == "merge":
        merge(sys.argv[2], *sys.argv[3:])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], seq_json=sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
