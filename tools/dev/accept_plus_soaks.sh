# usage (GPU box, via gpurun): bash tools/dev/accept_plus_soaks.sh <tag> <suite runs> <soak seconds> "<seeds>"
# tools/gpu_accept.sh, then the randomised soak as a script under the given seeds and the pipeline / threads soaks once; one summary line each
R=$1; N=$2; SEC=$3; SEEDS=$4
cd $GRAFT_REPO_ROOT
bash tools/gpu_accept.sh $R $N
BOX=$(cat /proc/sys/kernel/random/boot_id 2>/dev/null | cut -c1-8)
L=gpurun_out/${R}_accept_${BOX}.log
{
  for S in $SEEDS; do
    HFNET_SOAK_LOG=gpurun_out/${R}_soak_$S.cases timeout 900 python3 tools/dev/soak.py $SEC $S > gpurun_out/${R}_soak_$S.out 2>&1
    echo "soak seed $S ($SEC s) rc=$?  $(grep '^soak:' gpurun_out/${R}_soak_$S.out)  $(grep -c '^FAIL' gpurun_out/${R}_soak_$S.out) FAIL lines"
  done
  timeout 600 python3 tools/dev/soak_pipeline.py 40 > gpurun_out/${R}_soak_pipeline.out 2>&1; echo "pipeline soak (40 s) rc=$?  $(tail -1 gpurun_out/${R}_soak_pipeline.out | cut -c1-200)"
  timeout 600 python3 tools/dev/soak_threads.py 30 > gpurun_out/${R}_soak_threads.out 2>&1; echo "three-thread soak (30 s) rc=$?  $(tail -1 gpurun_out/${R}_soak_threads.out | cut -c1-200)"
} 2>&1 | tee -a $L
