#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: per-stream busy time, the idle gaps of the main
stream (host-bound launches, cross-stream joins) and the phases between them.

    rocprofv3 --kernel-trace --output-format csv -d out -o run -- python3 bench.py --steps 10 --warmup 2 ...
    python tools/timeline.py out/run_kernel_trace.csv [step_index]
"""
import collections
import csv
import sys


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void at::native::", "")
    return n[:58]


def main(path, which=5, gap_us=12.0, full=False):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    i0, i1 = adam[which] + 1, adam[which + 1] + 1
    step = rows[i0:i1]
    T0 = int(step[0]["Start_Timestamp"])
    T1 = int(step[-1]["End_Timestamp"])
    print(f"step {which}: wall {(T1 - T0) / 1e3:.1f} us, {len(step)} kernels")
    streams = collections.Counter(r["Stream_Id"] for r in step)
    main_sid = streams.most_common(1)[0][0]
    for sid, n in streams.most_common():
        ks = [r for r in step if r["Stream_Id"] == sid]
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks)
        gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(ks, ks[1:])]
        print(f"  stream {sid}{' (main)' if sid == main_sid else ''}: {n} kernels, busy {busy / 1e3:.1f} us, "
              f"idle between kernels {sum(g for g in gaps if g > 0) / 1e3:.1f} us")
    ks = [r for r in step if r["Stream_Id"] == main_sid]
    print(f"main-stream gaps > {gap_us} us:")
    prev_end = None
    small = 0.0
    for idx, r in enumerate(ks):
        s = (int(r["Start_Timestamp"]) - T0) / 1e3
        e = (int(r["End_Timestamp"]) - T0) / 1e3
        if prev_end is not None:
            g = s - prev_end
            if g > gap_us:
                print(f"  t={prev_end:8.1f}  gap {g:6.1f} us   {short(ks[idx - 1]['Kernel_Name'])}  ->  {short(r['Kernel_Name'])}")
            elif g > 0:
                small += g
        prev_end = e
    print(f"  (+ {small:.1f} us in gaps below the threshold)")
    # the end of the step in detail: what the optimizer is waiting for
    print("last 1200 us of the step, both streams (start, end, stream, kernel, grid):")
    for r in step:
        s0 = (int(r["Start_Timestamp"]) - T0) / 1e3
        e0 = (int(r["End_Timestamp"]) - T0) / 1e3
        if e0 > (T1 - T0) / 1e3 - 1200 and e0 - s0 > 8:
            print(f"  {s0:8.1f} {e0:8.1f}  s{r['Stream_Id']}  {short(r['Kernel_Name'])[:44]:<44} g={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
    if full:
        print("every kernel of the step (start, end, gap to the previous kernel of the same stream, stream, kernel, grid):")
        last = {}
        for r in step:
            s0 = (int(r["Start_Timestamp"]) - T0) / 1e3
            e0 = (int(r["End_Timestamp"]) - T0) / 1e3
            sid = r["Stream_Id"]
            gap = s0 - last[sid] if sid in last else 0.0
            last[sid] = e0
            print(f"  {s0:8.1f} {e0:8.1f} {gap:7.1f}  s{sid}  {short(r['Kernel_Name'])[:50]:<50} g={r['Grid_Size_X']}")
    by = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        k = short(r["Kernel_Name"])
        by[k][0] += 1
        by[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("top kernels of this step:")
    for k, (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"  {us:8.1f} us  n={n:<4d} avg {us / n:7.1f}  {k}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5, full=len(sys.argv) > 3 and sys.argv[3] == "full")
