# call 7 (round 4): the two-waves-per-SIMD G1 combine kernel: parity + timing of threshold decryption's combination at 65 536 - 262 144 jobs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wire.py -x -q -m gpu -k "two_wave" 2>&1 | tail -5 | tee gpurun_out/r04_c7_tests.txt
for b in 65536 131072 262144; do PROBE_TRUSTED=1 PROBE_B=$b timeout 300 python tools/g1_probe.py 2>&1 | grep -v amdgpu | tail -1; done | tee gpurun_out/r04_g1_probe.txt
