# call 2 (round 4): prepared pairing form after the register work (P: operands + first line in rows, M: coefficients streamed at
# their use): parity, same-box timing against the one-loop form and a 3-waves-per-SIMD stage P, PMC passes of both forms
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "pairing_forms or config3" 2>&1 | tail -3 | tee gpurun_out/r04_c2_tests.txt
for rep in 1 2; do
for v in auto:default pair:default auto:_p3; do
  form=${v%%:*}; lib=${v##*:}
  if [ $form = auto ]; then unset TC_PAIRING_FORM; else export TC_PAIRING_FORM=$form; fi
  if [ $lib = default ]; then unset TC_AMD_LIB; else export TC_AMD_LIB=$GRAFT_REPO_ROOT/threshold_crypto_amd/libtc_amd$lib.so; fi
  PROBE_NOCHECKS=1 timeout 300 python tools/pairing_probe.py 5 2>&1 | grep -v amdgpu | tail -1
done; done | tee gpurun_out/r04_c2_pairing_probe.txt
unset TC_PAIRING_FORM TC_AMD_LIB
bash tools/capture_pairing_r04.sh r04_prepared ""
bash tools/capture_pairing_r04.sh r04_oneloop pair
head -40 gpurun_out/summary_pairing_r04_prepared.csv
