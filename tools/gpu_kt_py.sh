# usage (GPU box): bash tools/gpu_kt_py.sh <script.py> [args]  -- rocprofv3 kernel stats (calls, avg us, total) of a python script
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ktp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktp -- python $GRAFT_REPO_ROOT/"$@" > /tmp/ktp.log 2>&1
tail -2 /tmp/ktp.log
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ktp/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    print("%-84s n %6s avg %8.1f us  total %9.1f us  %5.1f%%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total us", tot / 1e3)
PY
