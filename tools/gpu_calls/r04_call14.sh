# call 14 (round 4): stage P with its loads issued earlier (first pair's line behind the new line's constant term, G1 coordinates unpinned) against the committed stage P
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2 3; do
for lib in default _prevP; do
  if [ $lib = default ]; then unset TC_AMD_LIB; else export TC_AMD_LIB=$GRAFT_REPO_ROOT/threshold_crypto_amd/libtc_amd$lib.so; fi
  PROBE_NOCHECKS=1 timeout 300 python tools/pairing_probe.py 5 2>&1 | grep -v amdgpu | tail -1
done; done | tee gpurun_out/r04_p_loads_ab.txt
