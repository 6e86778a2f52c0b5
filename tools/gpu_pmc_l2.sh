# usage: bash tools/gpu_pmc_l2.sh <outdir> <bench args...>  -- L1 <-> L2 traffic per kernel (TCP counters), one short bench run on the GPU box
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TCP|TCC)_[A-Z0-9_]+(_sum)?\b" | sort -u > $OUT/avail.txt
i=1
for P in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pl$i
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pl$i -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --configs none --no-cpu-baseline "$@" > $OUT/pass$i.log 2>&1
  f=$(find /tmp/pl$i -name '*counter_collection.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f > $OUT/l2_$i.txt
  i=$((i+1))
done
cut -c1-200 $OUT/l2_1.txt | head -30; cut -c1-200 $OUT/l2_2.txt | head -30; wc -l $OUT/avail.txt; tail -3 $OUT/pass1.log
