# usage (GPU box): bash tools/gpu_kt.sh <name> <bench args...>  -- rocprofv3 kernel stats of one bench run, top kernels as us per chunk
NAME=$1; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/kt_$NAME
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$NAME -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --configs none --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/kt_$NAME.log 2>&1
cp $(find /tmp/kt_$NAME -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/kt_$NAME.csv
python $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/kt_$NAME.csv 3 24 | cut -c1-140
