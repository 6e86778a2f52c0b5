# usage (GPU box): bash tools/gpu_pmc_py.sh <script.py> [args]  -- SQ busy / instruction counters per kernel of a python script (two passes)
cd /tmp; export TMPDIR=/tmp
P1="SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
i=1
for P in "$P1" "$P2"; do
  rm -rf /tmp/pmcp$i
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmcp$i -- python $GRAFT_REPO_ROOT/"$@" > /tmp/pmcp$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pmcp$i -name '*counter_collection.csv' | head -1) | cut -c1-220 | head -12
  i=$((i+1))
done
