# call 6 (round 4): issue-cost microbenchmark at sustained launch lengths; G1 multiplication: register-table kernel vs arena form
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 tools/ubench_issue 2>&1 > gpurun_out/r04_ubench_issue.txt; head -6 gpurun_out/r04_ubench_issue.txt
for form in regs arena auto; do
  if [ $form = auto ]; then unset TC_G1_MUL_FORM; else export TC_G1_MUL_FORM=$form; fi
  timeout 300 python tools/g1_mul_probe.py 2>&1 | grep -v amdgpu | tail -1
done | tee gpurun_out/r04_g1_mul_probe.txt
