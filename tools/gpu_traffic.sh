# usage (GPU box): bash tools/gpu_traffic.sh <out.json> [bench args]  -- FETCH_SIZE / WRITE_SIZE passes of one short bench run, folded per launch name
OUTJ=$1; shift
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --configs none --no-cpu-baseline "$@" > /tmp/pmc_$C.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/make_traffic.py $(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) 64 $GRAFT_REPO_ROOT/$OUTJ > /dev/null 2>&1
python - <<PY
import json
t = json.load(open("$GRAFT_REPO_ROOT/$OUTJ"))
for n, k in t["kernels"].items():
    print("%-22s fetch %7.0f MB write %6.0f MB  hbm-side %5.2f GB" % (n, k["fetch_kb"] / 1e3, k["write_kb"] / 1e3, (2 * k["fetch_kb"] + k["write_kb"]) * 1024 / 1e9))
PY
