# call 22 (round 4): G1 combination with its jobs grouped by denominator class from 65 537 jobs on (a wave of D = 1 jobs skips the
# [1 / D] ladder) against the same build without the grouping; random 4-of-10 subsets; then the tests that cover it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for lib in _nogroup default; do
  if [ $lib = default ]; then unset TC_AMD_LIB; else export TC_AMD_LIB=$GRAFT_REPO_ROOT/threshold_crypto_amd/libtc_amd$lib.so; fi
  timeout 300 python tools/g1_group_probe.py 2>&1 | grep -v amdgpu | tail -1
done; done | tee gpurun_out/r04_g1_group_ab.txt
unset TC_AMD_LIB
timeout 900 python -m pytest tests/test_gpu_wire.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "g1 or decrypt or config4 or combine" > gpurun_out/r04_g1_group_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r04_g1_group_tests.txt | tail -2
