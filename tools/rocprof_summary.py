#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2) `*_results.db` into the per-kernel summary kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_xxx_kernel_stats.txt

Equivalent to the `--stats` table of `rocprofv3 --kernel-trace --stats` (view `top_kernels`,
durations in the database are nanoseconds)."""
import sqlite3
import sys


def main(path: str) -> None:
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                            "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"# total kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':84s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, mn, mx in rows:
        print(f"{name[:84]:84s} {calls:6d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {tot / total * 100:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
