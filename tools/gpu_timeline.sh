# usage (GPU box): bash tools/gpu_timeline.sh <out name> <script.py> [args] -- kernel timeline (start / end relative to the frame's first kernel) of the LAST frame a script ran
NAME=$1; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ktl
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktl -- python $GRAFT_REPO_ROOT/"$@" > /tmp/ktl.log 2>&1
tail -1 /tmp/ktl.log
python - $GRAFT_REPO_ROOT/gpurun_out/$NAME.txt <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/ktl/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
first = rows[0]["Kernel_Name"]
# frames start with the first kernel name of the script's steady state: take the last complete frame
starts = [i for i, r in enumerate(rows) if "k_resize_u8" in r["Kernel_Name"] or "k_pyramid_chain" in r["Kernel_Name"]]
# three resizes per frame: a frame starts at the first of a group
grp = [i for j, i in enumerate(starts) if j == 0 or i - starts[j - 1] > 3]
a = grp[-2]; b = grp[-1]
t0 = int(rows[a]["Start_Timestamp"])
out = open(sys.argv[1], "w")
prev_end = {}
for r in rows[a:b]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = r.get("Queue_Id", "?")
    gap = s - prev_end.get(q, 0.0)
    prev_end[q] = e
    line = "%8.1f %8.1f  dur %6.1f  gap %6.1f  q%-3s g%-8s %s" % (s, e, e - s, gap, q, r.get("Grid_Size", "?"), r["Kernel_Name"].replace("hfnet::", "")[:70])
    print(line); out.write(line + "\n")
PY
