cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
( for i in 1 2; do PROBE_NOCHECKS=1 TC_PAIRING_FORM=pair timeout 300 python tools/pairing_probe.py 3; PROBE_NOCHECKS=1 TC_PAIRING_FORM=parked timeout 300 python tools/pairing_probe.py 3; done ) > gpurun_out/r03_11_parked.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03_11_parked.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_11_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_11_gpu_tests.log
grep -E "passed|failed|rc" gpurun_out/r03_11_gpu_tests.log | tail -3
