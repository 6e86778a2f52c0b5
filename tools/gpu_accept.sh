#!/bin/bash
# Acceptance run at the final HEAD of a round -- the driver's own round-end commands, in its form, with the logs kept:
#   gpurun --timeout 2400 -- 'bash tools/gpu_accept.sh r05 [N=4]'
# 1. N x `python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider`   (GPUTEST_rNN's command)
# 2. `__graft_entry__.smoke()`
# 3. `python3 bench.py --gpus 1 --steps 20 --warmup 5`                 (BENCH_rNN's command; the final stdout line must parse and be < 8000 bytes)
# Everything goes to gpurun_out/<round>_accept_<box>.log (copy it to profiles/ and commit it; no source commit after it).
R=${1:-r05}; N=${2:-4}
mkdir -p gpurun_out
BOX=$(cat /proc/sys/kernel/random/boot_id 2>/dev/null | cut -c1-8)
L=gpurun_out/${R}_accept_${BOX}.log
{
  echo "== accept $R  box $BOX  $(date -u +%FT%TZ)  build $(python3 -c 'from hfnet_slam_amd import build; print(build.library_id(), build.source_id())')"
  fail=0
  for i in $(seq 1 $N); do
    HFNET_SOAK_LOG=gpurun_out/${R}_accept_soak_$i.cases python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/${R}_accept_pytest_$i.log 2>&1
    rc=$?; [ $rc != 0 ] && fail=1
    echo "pytest run $i rc=$rc  $(tail -1 gpurun_out/${R}_accept_pytest_$i.log)"
    if [ $rc != 0 ]; then tail -40 gpurun_out/${R}_accept_pytest_$i.log | cut -c1-300; tail -2 gpurun_out/${R}_accept_soak_$i.cases; fi
  done
  python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2; rc=${PIPESTATUS[0]}; [ $rc != 0 ] && fail=1; echo "smoke rc=$rc"
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_accept_bench.out 2> gpurun_out/${R}_accept_bench.err; rc=$?; [ $rc != 0 ] && fail=1
  echo "bench rc=$rc"
  python3 - <<PY
import json
line = open("gpurun_out/${R}_accept_bench.out").read().strip().splitlines()[-1]
d = json.loads(line)
print("bench line %d bytes; value %.1f %s; roofline.frac %s; cpu_baseline %s; verified %s" % (len(line), d["value"], d["unit"], d.get("roofline", {}).get("frac"), d.get("cpu_baseline", {}).get("value"), d.get("verified")))
assert len(line) < 8000
PY
  [ $? != 0 ] && fail=1
  echo "== accept $R: $([ $fail = 0 ] && echo GREEN || echo RED)"
} 2>&1 | tee $L
