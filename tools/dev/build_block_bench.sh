# builds tools/dev/block_bench against the current kernels_block.hip (in-tree binary: travels to the GPU box, git-ignored)
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -x hip tools/dev/block_bench.hip hfnet_slam_amd/csrc/kernels_block.hip ${BB_DEFS} -o tools/dev/block_bench
