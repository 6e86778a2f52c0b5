set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_a/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_a/pytest.txt
tail -n 30 gpurun_out/r02_a/pytest.txt
for v in 4 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-all --opt fused_variant=$v > gpurun_out/r02_a/bench_v$v.json 2> gpurun_out/r02_a/bench_v$v.txt
  cat gpurun_out/r02_a/bench_v$v.json | cut -c1-400
done
head -n 45 gpurun_out/r02_a/bench_v4.txt
