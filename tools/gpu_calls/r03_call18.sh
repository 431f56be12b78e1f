# call 18: the launcher-over-RCCL tests (one rank), the deepened oracle comparisons
timeout 1500 python -m pytest tests -m gpu -x -q -k "launcher or config5_one_gpu or comb_signing or bench_line" 2>&1 | grep -v amdgpu | tee gpurun_out/r03_18_tests.log | tail -15
