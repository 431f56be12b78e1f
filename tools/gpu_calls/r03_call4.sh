cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -x -k "rlc or large_threshold or lincomb or config5_shape or combine_g1" > gpurun_out/r03_4_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_4_tests.log
tail -12 gpurun_out/r03_4_tests.log
timeout 300 python tools/g1_large_t_probe.py 131072 > gpurun_out/r03_4_g1_large_t.txt 2>&1; tail -2 gpurun_out/r03_4_g1_large_t.txt
timeout 300 python tools/g1_large_t_probe.py 4096 >> gpurun_out/r03_4_g1_large_t.txt 2>&1; tail -1 gpurun_out/r03_4_g1_large_t.txt
timeout 300 python tools/rlc_samekey_probe.py > gpurun_out/r03_4_rlc_samekey.txt 2>&1; tail -2 gpurun_out/r03_4_rlc_samekey.txt
