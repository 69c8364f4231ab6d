cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4j/tr -o t -- python3 $GRAFT_REPO_ROOT/tools/step_bf3_bench.py 2048 > $GRAFT_REPO_ROOT/gpurun_out/r4j/trace.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find gpurun_out/r4j/tr -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
out = []
for r in rows:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "gru_step_bf3" in n:
        out.append(((e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0, n[-40:]))
    prev_end = e
# last encoder call: 48 launches
last = out[-48:]
print("dur_us gap_before_us (last save=1 call)")
for d, g, n in last[:6] + last[22:28]:
    print(f"{d:8.1f} {g:8.1f}  {n}")
import statistics
print("median duration", statistics.median(d for d, g, n in last[1:24]), statistics.median(d for d, g, n in last[25:]), "median gap", statistics.median(g for d, g, n in last[1:]))
ns = out[48 * 2: 48 * 3]
print("save=0 call: median duration L0", statistics.median(d for d, g, n in ns[1:24]), "L1", statistics.median(d for d, g, n in ns[25:]), "gap", statistics.median(g for d, g, n in ns[1:]))
PY
rm -rf gpurun_out/r4j/tr
