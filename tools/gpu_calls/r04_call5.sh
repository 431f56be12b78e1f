# call 5 (round 4): the issue-cost microbenchmark (roofline denominator), pending GPU tests of the ADVICE fixes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 tools/ubench_issue 2>&1 | tee gpurun_out/r04_ubench_issue.txt | head -50
timeout 120 tools/ubench_clock --peak 2>&1 | tail -1 | tee gpurun_out/r04_ubench_clock_peak.txt
timeout 120 tools/ubench_chain --peak 2>&1 | tail -1 | tee gpurun_out/r04_ubench_chain_peak.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "rlc_decryption or rlc_share or rlc_same" 2>&1 | tail -3 | tee gpurun_out/r04_c5_tests.txt
