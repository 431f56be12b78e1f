# call 26: G1 kernels with the base-4 ladder at 65 536 / 131 072 / 262 144 jobs, operands trusted
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for b in 65536 131072 262144; do PROBE_TRUSTED=1 PROBE_B=$b timeout 300 python tools/g1_probe.py 2>&1 | grep -v amdgpu | tail -1; done | tee gpurun_out/r03_g1_base4_probe.txt
