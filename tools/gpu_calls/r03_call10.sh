cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "rlc" > gpurun_out/r03_10_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_10_tests.log
tail -5 gpurun_out/r03_10_tests.log
bash tools/capture_r03.sh r03_b
