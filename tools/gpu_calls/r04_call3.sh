# call 3 (round 4): prepared pairing form with the two-pass stage P (global-address-space rows): parity, timing, PMC
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "pairing_forms or config3" 2>&1 | tail -3 | tee gpurun_out/r04_c3_tests.txt
for rep in 1 2; do
for form in auto pair; do
  if [ $form = auto ]; then unset TC_PAIRING_FORM; else export TC_PAIRING_FORM=$form; fi
  PROBE_NOCHECKS=1 timeout 300 python tools/pairing_probe.py 5 2>&1 | grep -v amdgpu | tail -1
done; done | tee gpurun_out/r04_c3_pairing_probe.txt
unset TC_PAIRING_FORM
bash tools/capture_pairing_r04.sh r04_prepared2 ""
grep -v "^#" gpurun_out/summary_pairing_r04_prepared2.csv | head -30
