cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/ab_probe.py > gpurun_out/r03_14_ab.txt 2>&1; grep -v amdgpu gpurun_out/r03_14_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "hash or golden or config4 or ciphertext or sign" 2>&1 | grep -E "passed|failed" | tail -3
timeout 300 python tests/soak.py 60 21 2>&1 | grep -v amdgpu | tail -2
