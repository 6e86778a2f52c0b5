set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_ablate
for a in 0 1 2 4 3 5 6 7 8; do
  BENCH_NO_KP_CHECK=1 HFNET_FUSE_ABLATE=$a timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-all > gpurun_out/r02_ablate/ab$a.json 2> gpurun_out/r02_ablate/ab$a.txt
done
tail -n 45 gpurun_out/r02_ablate/ab0.txt
