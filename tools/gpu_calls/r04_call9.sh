cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TC_TEST_CONFIG5_JOBS=${JOBS:-131072} timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5_full" 2>&1 | tail -40 | tee gpurun_out/r04_c9_tests.txt
