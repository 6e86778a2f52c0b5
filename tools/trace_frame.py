#!/usr/bin/env python3
"""Timeline of one frame out of a rocprofv3 kernel trace of `tools/bench_latency.py`:

    rocprofv3 --kernel-trace --output-format csv -d out -- python tools/bench_latency.py
    python tools/trace_frame.py out/*/*_kernel_trace.csv [frame=40]

start (us, from the frame's first kernel), duration (us), hardware queue, kernel, grid -- shows which stream a kernel ran
on and where a branch waited (the HIP graph of the host-pointer path is bypassed under the profiler, so these are the
direct launches of both streams)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frame = int(sys.argv[2]) if len(sys.argv) > 2 else 40
stems = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"]]
a, b = stems[frame] - 3, stems[frame + 1] - 3          # three pyramid resizes precede the stem
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} q{r['Queue_Id']} {r['Kernel_Name'][:56]:56s} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']} vgpr={r['VGPR_Count']}")
