# call 11 (round 4): the secondary G1 kernels built for two waves per SIMD (-DTC_WAVES_G1=2) against the 512-register builds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for lib in default _g1w2; do
  if [ $lib = default ]; then unset TC_AMD_LIB; else export TC_AMD_LIB=$GRAFT_REPO_ROOT/threshold_crypto_amd/libtc_amd$lib.so; fi
  timeout 600 python tools/g1_aux_probe.py 2>&1 | grep -v amdgpu | tail -1
done | tee gpurun_out/r04_g1_aux_probe.txt
