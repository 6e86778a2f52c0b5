# usage (GPU box, via gpurun): bash tools/gpu_ab.sh <tag> "<pytest -k expr or none>" "<bench args A>" ["<bench args B>" ...]
# parity subset, then one short bench run per argument set with the per-launch table (blocks / heads only)
TAG=$1; KEXPR=$2; shift; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
if [ "$KEXPR" != "none" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > gpurun_out/ab_${TAG}_test.log 2>&1; echo "pytest rc $?" >> gpurun_out/ab_${TAG}_test.log
  tail -4 gpurun_out/ab_${TAG}_test.log
fi
i=0
for ARGS in "$@"; do
  timeout 600 python bench.py --steps 6 --warmup 2 --configs none --no-cpu-baseline --profile-all $ARGS > gpurun_out/ab_${TAG}_$i.json 2> gpurun_out/ab_${TAG}_$i.err
  echo "== run $i: $ARGS"
  grep "launches" gpurun_out/ab_${TAG}_$i.err | head -${AB_ROWS:-24} | awk '{printf "%s %s | ", $1, $5} END {print ""}'
  python -c "
import json;d=json.load(open('gpurun_out/ab_${TAG}_$i.json'));print('VALUE %.0f frames/s, single-stream chunk %.3f ms'%(d['value'],d['roofline']['profiled_chunk_ms_single_stream']))"
  i=$((i+1))
done
