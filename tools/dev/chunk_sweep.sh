# usage (GPU box, via gpurun): bash tools/dev/chunk_sweep.sh "<chunk sizes>" [extra bench args]   -- headline workload at several frames-per-call sizes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CH=$1; shift
for C in $CH; do
  timeout 600 python bench.py --steps 6 --warmup 2 --configs none --no-cpu-baseline --no-natural --chunk $C --batch $((C*6)) "$@" > gpurun_out/sweep_$C.json 2> gpurun_out/sweep_$C.err
  python -c "
import json;d=json.load(open('gpurun_out/sweep_$C.json'));print('chunk $C: VALUE %.0f frames/s, ms/step %.2f, verified %s'%(d['value'],d['ms_per_step'],d.get('verified')))"
done
