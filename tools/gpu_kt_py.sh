# usage (GPU box): bash tools/gpu_kt_py.sh <script.py> [args]  -- rocprofv3 kernel stats (avg us per kernel) of a python script
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ktp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktp -- python $GRAFT_REPO_ROOT/"$@" > /tmp/ktp.log 2>&1
tail -2 /tmp/ktp.log
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ktp/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-90s n %5s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
