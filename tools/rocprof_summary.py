#!/usr/bin/env python3
"""Per-kernel summary (count, total, avg, min, max) from a rocprofv3 rocpd database
(rocprofv3 --kernel-trace --stats -d DIR -o NAME  ->  DIR/NAME_results.db)."""
import sqlite3
import sys


def main(path, top=40):
    con = sqlite3.connect(path)
    rows = con.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# {path}: {sum(r[1] for r in rows)} dispatches, {tot:.1f} us total kernel time")
    print(f"{'kernel':<100} {'calls':>7} {'total_us':>11} {'pct':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9}")
    for r in rows[:top]:
        print(f"{r[0][:100]:<100} {r[1]:>7d} {r[2]:>11.1f} {100 * r[2] / tot:>6.2f} {r[3]:>9.2f} {r[4]:>9.2f} {r[5]:>9.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
