# call 25: GPU suite + smoke at the round's last commit, then a soak with the pairing forms forced once more (new G1 ladder underneath)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_25_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_25_gpu_tests.log
grep -E "passed|failed|rc" gpurun_out/r03_25_gpu_tests.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
(TC_PAIRING_FORM=pair timeout 300 python tests/soak.py 180 61; timeout 300 python tests/soak.py 180 62) 2>&1 | grep -v amdgpu | tail -2 | tee gpurun_out/r03_soak5.txt
