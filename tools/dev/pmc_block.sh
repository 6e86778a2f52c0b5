#!/bin/bash
# LDS / instruction-mix counters of one fused block launch:  tools/dev/pmc_block.sh <binary> <layer> <frames> <variant> <tag>
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
bin=$1; L=$2; F=$3; V=$4; tag=$5
cd /tmp
P="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
rm -rf /tmp/pmcb; timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmcb -- $R/tools/dev/$bin $L $F $V 3 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/pmcb -name '*counter_collection.csv' | head -1) fused > $O/pmc_${tag}.txt
cut -c1-250 $O/pmc_${tag}.txt
