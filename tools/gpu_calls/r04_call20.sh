# call 20 (round 4): the lane parity computed inside the lane-pair multipliers (two v_mbcnt) instead of passed as an argument that the
# 256-register kernels kept in scratch and reloaded before every product -- against the previous build, same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
for lib in _prev default; do
  if [ $lib = default ]; then unset TC_AMD_LIB; else export TC_AMD_LIB=$GRAFT_REPO_ROOT/threshold_crypto_amd/libtc_amd$lib.so; fi
  PROBE_NOCHECKS=1 timeout 300 python tools/pairing_probe.py 5 2>&1 | grep -v amdgpu | tail -1
  timeout 300 python tools/ab_probe.py 2>&1 | grep -v amdgpu | tail -1
done; done | tee gpurun_out/r04_parity_in_callee_ab.txt
