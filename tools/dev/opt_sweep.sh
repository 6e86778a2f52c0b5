# usage (GPU box, via gpurun): bash tools/dev/opt_sweep.sh "<opt=val ...>" ...   -- headline workload (default call size) once per argument (a space-separated option set; "" = defaults)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
i=0
for SET in "$@"; do
  ARGS=""; for o in $SET; do ARGS="$ARGS --opt $o"; done
  timeout 600 python bench.py --steps 6 --warmup 2 --configs none --no-cpu-baseline --no-natural $ARGS > gpurun_out/opt_$i.json 2> gpurun_out/opt_$i.err
  python -c "
import json;d=json.load(open('gpurun_out/opt_$i.json'));print('[$SET]: VALUE %.0f frames/s, verified %s'%(d['value'],d.get('verified',{}).get('equal')))"
  i=$((i+1))
done
