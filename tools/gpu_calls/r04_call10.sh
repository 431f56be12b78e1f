# call 10 (round 4): operand exchange of the lane-pair multiplier through ds_swizzle (-DTC_MUL_SWIZZLE) against DPP moves, same box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for lib in default _swz; do
  if [ $lib = default ]; then unset TC_AMD_LIB; else export TC_AMD_LIB=$GRAFT_REPO_ROOT/threshold_crypto_amd/libtc_amd$lib.so; fi
  timeout 300 python tools/ab_probe.py 2>&1 | grep -v amdgpu | tail -1
done; done | tee gpurun_out/r04_swizzle_ab.txt
unset TC_AMD_LIB
TC_TEST_CONFIG5_JOBS=131072 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5" 2>&1 | tail -3 | tee gpurun_out/r04_c10_tests.txt
