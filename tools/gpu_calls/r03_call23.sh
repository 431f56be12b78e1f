# call 23: soak at the build with the base-4 G1 ladder (two seeds)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 400 python tests/soak.py 240 51; timeout 400 python tests/soak.py 240 52) 2>&1 | grep -v amdgpu | tail -2 | tee gpurun_out/r03_soak4.txt
