# call 1 (round 4): the prepared pairing form (k_miller_lines + k_miller_accumulate + k_final_exp) -- parity of the forms,
# config-3 full-size parity, timing against r03's two-kernel form on the same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "pairing_forms or config3 or config4 or rlc_same_key" 2>&1 | tail -5 | tee gpurun_out/r04_c1_tests.txt
for form in auto pair; do
  if [ $form = auto ]; then unset TC_PAIRING_FORM; else export TC_PAIRING_FORM=$form; fi
  PROBE_NOCHECKS=1 timeout 300 python tools/pairing_probe.py 4 2>&1 | grep -v amdgpu | tail -1
done | tee gpurun_out/r04_c1_pairing_probe.txt
unset TC_PAIRING_FORM
cd /tmp && export TMPDIR=/tmp
PROBE_NOCHECKS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04_c1_prof -o prof -- python $GRAFT_REPO_ROOT/tools/pairing_probe.py 4 > $GRAFT_REPO_ROOT/gpurun_out/r04_c1_prof.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py gpurun_out/r04_c1_prof 2>&1 | head -20 | tee gpurun_out/r04_c1_prof_summary.txt
