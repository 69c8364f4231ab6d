#!/usr/bin/env python3
"""Host-bound or device-bound?  Joins a rocprofv3 kernel trace with its HIP API trace on the correlation id and prints, for
every main-stream gap of one training step, how long before the kernel started its launch call had returned:
a small lag means the GPU was waiting for the host.
    rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d out -o t -- python3 bench.py ...
    python tools/launch_lag.py out/..._kernel_trace.csv out/..._hip_api_trace.csv [step]"""
import csv
import sys


def main(kpath, apath, which=5, gap_us=8.0):
    ks = sorted(csv.DictReader(open(kpath)), key=lambda r: int(r["Start_Timestamp"]))
    api = {}
    for r in csv.DictReader(open(apath)):
        if "Launch" in r["Function"]:
            api[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"])
    adam = [i for i, r in enumerate(ks) if "adam_kernel" in r["Kernel_Name"]]
    step = ks[adam[which] + 1: adam[which + 1] + 1]
    T0 = int(step[0]["Start_Timestamp"])
    main_sid = max(set(r["Stream_Id"] for r in step), key=lambda s: sum(1 for r in step if r["Stream_Id"] == s))
    prev_end = None
    lags = []
    host_bound = 0.0
    print("gap_us  lag_us(kernel start - launch call returned)  kernel")
    for r in step:
        if r["Stream_Id"] != main_sid:
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        a = api.get(r["Correlation_Id"])
        lag = (s - a[1]) / 1e3 if a else float("nan")
        lags.append(lag)
        if prev_end is not None and (s - prev_end) / 1e3 > gap_us:
            g = (s - prev_end) / 1e3
            hb = a is not None and lag < 15.0
            host_bound += g if hb else 0.0
            print(f"{g:7.1f} {lag:9.1f}  t={(s - T0) / 1e3:8.1f}  {'HOST' if hb else 'dev '}  {r['Kernel_Name'][:70]}")
        prev_end = e
    lags = sorted(x for x in lags if x == x)
    print(f"main-stream kernels {len(lags)}; launch lag min {lags[0]:.1f} median {lags[len(lags) // 2]:.1f} max {lags[-1]:.1f} us; "
          f"gap time attributed to the host {host_bound:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5)
