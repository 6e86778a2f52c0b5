# usage (on the GPU box, via gpurun):  bash tools/gpu_profiles.sh <round tag, e.g. r02>
# Writes gpurun_out/<tag>_profiles/: the bench line, rocprofv3 kernel stats of the same command, SQ counter passes and the
# FETCH_SIZE / WRITE_SIZE passes folded into <tag>_traffic_b<chunk>.json (chunk = bench.py's default frames per call).  Copy what is to be judged into profiles/.
set -x
TAG=$1
CH=$(cd $GRAFT_REPO_ROOT && python -c 'import bench; print(bench.DEFAULT_CHUNK)')   # bench.py default --chunk
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_profiles
mkdir -p $OUT
export TMPDIR=/tmp
# traffic passes first: the bench line quotes roofline.traffic from profiles/<tag>_traffic_b<chunk>.json of the SAME source state
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --configs none --no-cpu-baseline > $OUT/pmc_$C.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/make_traffic.py $(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) $CH $OUT/${TAG}_traffic_b${CH}.json > $OUT/traffic.log 2>&1
cp $OUT/${TAG}_traffic_b${CH}.json $GRAFT_REPO_ROOT/profiles/
bash $GRAFT_REPO_ROOT/tools/gpu_traffic_chunk.sh profiles/${TAG}_chunk_traffic_bf16x3_b${CH}.json "Tolerance mode: --opt scores_bf16x3=1 --opt desc_bf16x3=1 --opt global_bf16x3=1 --no-verify --no-natural." --opt scores_bf16x3=1 --opt desc_bf16x3=1 --opt global_bf16x3=1 > $OUT/traffic_bf16x3_pre.log 2>&1
cd $GRAFT_REPO_ROOT
# stdout: the compact line the driver parses; the record in full (tables, notes, sub-records) goes to BENCH_DETAIL
BENCH_DETAIL=$OUT/${TAG}_bench_detail.json timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --configs none --no-cpu-baseline > $OUT/kt.log 2>&1
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_bench_kernel_stats.csv
P1="SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
i=1
for P in "$P1" "$P2"; do
  rm -rf /tmp/pmc$i
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc$i -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --configs none --no-cpu-baseline > $OUT/pmc$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pmc$i -name '*counter_collection.csv' | head -1) > $OUT/${TAG}_pmc_$( [ $i = 1 ] && echo sq || echo instmix )_b${CH}.txt
  i=$((i+1))
done
# ---- the tolerance mode (engine options scores_bf16x3 + desc_bf16x3 + global_bf16x3): whole-call HBM bytes (bench.py's roofline_bf16x3 reads the file),
#      kernel stats and the two SQ passes of the same command
TOL="--opt scores_bf16x3=1 --opt desc_bf16x3=1 --opt global_bf16x3=1 --no-verify --no-natural"
bash $GRAFT_REPO_ROOT/tools/gpu_traffic_chunk.sh gpurun_out/${TAG}_profiles/${TAG}_chunk_traffic_bf16x3_b${CH}.json "Tolerance mode: $TOL." $TOL > $OUT/traffic_bf16x3.log 2>&1
bash $GRAFT_REPO_ROOT/tools/gpu_traffic_chunk.sh gpurun_out/${TAG}_profiles/${TAG}_chunk_traffic_b${CH}.json "Exact mode (defaults)." > $OUT/traffic_chunk.log 2>&1
cd /tmp
rm -rf /tmp/ktt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --configs none --no-cpu-baseline $TOL > $OUT/ktt.log 2>&1
cp $(find /tmp/ktt -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_bench_bf16x3_kernel_stats.csv
i=1
for P in "$P1" "$P2"; do
  rm -rf /tmp/pmct$i
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmct$i -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --configs none --no-cpu-baseline $TOL > $OUT/pmct$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pmct$i -name '*counter_collection.csv' | head -1) > $OUT/${TAG}_pmc_$( [ $i = 1 ] && echo sq || echo instmix )_bf16x3_b${CH}.txt
  i=$((i+1))
done
# ---- the two exact paths beside the extractor that changed in round 6, alone: kernel times of the batched database query (64 queries x 10 000
#      keyframes x 4096) and of SearchByBoW at the headline call's 255 pairs
cd /tmp
rm -rf /tmp/ktd && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktd -- python $GRAFT_REPO_ROOT/tools/dev/db_only.py 64 4096 > $OUT/ktd.log 2>&1
cp $(find /tmp/ktd -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_db_q64_kernel_stats.csv
rm -rf /tmp/ktm && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktm -- python $GRAFT_REPO_ROOT/tools/dev/match_only.py 10 255 > $OUT/ktm.log 2>&1
cp $(find /tmp/ktm -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_match_255_kernel_stats.csv
rm -f $OUT/*.log
ls -la $OUT
head -c 1500 $OUT/${TAG}_bench.json
