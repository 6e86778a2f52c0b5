# builds tools/dev/block_bench against the current kernels_block.hip (in-tree binary: travels to the GPU box, git-ignored)
#   BB_MAIN=<file>  another harness (tools/dev/stem_bench.hip);  BB_SRC=<file>  a modified copy of kernels_block.hip to time instead (stage ablations);  BB_OUT=<name>  binary name
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -w -I hfnet_slam_amd/csrc -I include -x hip ${BB_MAIN:-tools/dev/block_bench.hip} ${BB_SRC:-hfnet_slam_amd/csrc/kernels_block.hip} ${BB_DEFS} -o tools/dev/${BB_OUT:-block_bench}
