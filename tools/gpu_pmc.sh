# usage: bash tools/gpu_pmc.sh <outdir> <bench args...>   -- two SQ counter passes of one short bench run (run on the GPU box)
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P1="SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"
i=1
for P in "$P1" "$P2"; do
  rm -rf /tmp/pmc$i
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc$i -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pass$i.log 2>&1
  f=$(find /tmp/pmc$i -name '*counter_collection.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f > $OUT/pmc$i.txt
  i=$((i+1))
done
grep -E "kernel|fused" $OUT/pmc1.txt | cut -c1-220; grep -E "kernel|fused" $OUT/pmc2.txt | cut -c1-220
