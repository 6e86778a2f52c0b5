# usage (GPU box): bash tools/gpu_traffic_chunk.sh <out.json relative to the repo> "<note>" [bench args]  -- whole-call HBM bytes of one short bench run
OUTJ=$1; NOTE=$2; shift; shift
CH=$(cd $GRAFT_REPO_ROOT && python -c 'import bench; print(bench.DEFAULT_CHUNK)')
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcc_$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcc_$C -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --configs none --no-cpu-baseline --no-natural --no-verify "$@" > /tmp/pmcc_$C.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/make_chunk_traffic.py $(find /tmp/pmcc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/pmcc_WRITE_SIZE -name '*counter_collection.csv' | head -1) $CH $GRAFT_REPO_ROOT/$OUTJ "$NOTE"
