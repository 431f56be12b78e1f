# call 16 (round 4): the ceiling of the Fq2 kernels -- the shipped out-of-line lane-pair product run as nothing else, at 1-4 waves per SIMD
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do timeout 300 tools/ubench_product; done 2>&1 | grep -v amdgpu | tee gpurun_out/r04_ubench_product.txt
