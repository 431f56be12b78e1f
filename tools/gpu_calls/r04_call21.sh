# call 21 (round 4): the two-waves-per-SIMD (arena) forms of the G1 kernels at EVERY batch size (-DTC_G1_ARENA_MIN=0) against the
# shipped threshold (arena above 65 536 jobs): 1 024 ... 65 536 jobs, same box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for B in 1024 8192 32768 65536; do
for lib in default _a0; do
  if [ $lib = default ]; then unset TC_AMD_LIB; else export TC_AMD_LIB=$GRAFT_REPO_ROOT/threshold_crypto_amd/libtc_amd$lib.so; fi
  echo -n "$lib "; PROBE_B=$B PROBE_TRUSTED=1 timeout 300 python tools/g1_probe.py 2>&1 | grep -v amdgpu | tail -1
done; done; done | tee gpurun_out/r04_g1_arena_all_sizes.txt
