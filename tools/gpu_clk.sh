# usage: bash tools/gpu_clk.sh <outdir> <bench args...>  -- effective shader clock per kernel: GRBM_GUI_ACTIVE / duration
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmcc
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmcc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/clk.log 2>&1
f=$(find /tmp/pmcc -name '*counter_collection.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f > $OUT/clk.txt
python3 - $OUT/clk.txt <<'PY'
import sys
for l in open(sys.argv[1]).read().split('\n')[1:30]:
    p=l.split()
    if len(p)<6: continue
    try:
        us=float(p[-5]); gui=float(p[-4]); busy=float(p[-3]); mf=float(p[-2]); wc=float(p[-1])
    except: continue
    print("%-48s %8.1f us  clk %.2f GHz  cu_busy/gui %.2f  mfma_busy %.0f%%  waves/SIMD %.2f"%(l[:48],us,gui/us/1e3, busy/256/max(gui,1), 100*mf/1024/max(gui,1), 4*wc/1024/max(gui,1)))
PY
