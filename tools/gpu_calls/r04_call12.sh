# call 12 (round 4): the compressed cyclotomic squaring chain with its six squarings inlined, against the out-of-line build (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for lib in default _noinl; do
  if [ $lib = default ]; then unset TC_AMD_LIB; else export TC_AMD_LIB=$GRAFT_REPO_ROOT/threshold_crypto_amd/libtc_amd$lib.so; fi
  PROBE_NOCHECKS=1 timeout 300 python tools/pairing_probe.py 5 2>&1 | grep -v amdgpu | tail -1
  PROBE_B=4096 PROBE_NOCHECKS=1 timeout 300 python tools/pairing_probe.py 5 2>&1 | grep -v amdgpu | tail -1
done; done | tee gpurun_out/r04_cyclo_inline_ab.txt
unset TC_AMD_LIB
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_cpp_api.py -x -q -m gpu -k "pairing_forms or config3 or cpp_api" 2>&1 | tail -3 | tee gpurun_out/r04_c12_tests.txt
