# usage (GPU box): bash tools/dev/soaks_r06.sh <seconds per soak> <seed>  -- the four soaks back to back, one summary line each into gpurun_out/soaks_<seed>.log
S=${1:-300}; SEED=${2:-1}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/soaks_${SEED}.log
{
  echo "== soaks seed $SEED, $S s each, build $(python3 -c 'from hfnet_slam_amd import build; print(build.library_id())'), box $(cut -c1-8 /proc/sys/kernel/random/boot_id)"
  HFNET_SOAK_LOG=/dev/null python3 tools/dev/soak.py $S $SEED 2>&1 | tail -6
  python3 tools/dev/soak_tolerance.py $S $SEED 2>&1 | tail -6
  python3 - <<PY
import importlib.util, os, sys
sys.path.insert(0, os.getcwd())
import torch; torch.cuda.init()
def load(n):
    s = importlib.util.spec_from_file_location(n, os.path.join("tools", "dev", n + ".py")); m = importlib.util.module_from_spec(s); s.loader.exec_module(m); return m
r, f = load("soak_pipeline").run($S / 3.0, $SEED + 100); print("pipeline soak:", r, "rounds,", len(f), "failures", f[:3])
c, f = load("soak_threads").run($S / 3.0, $SEED + 200); print("thread soak:", c, len(f), "failures", f[:3])
PY
} 2>&1 | tee $L
