cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
( TC_PAIRING_QUAD=1 timeout 300 python tools/pairing_probe.py 4; timeout 300 python tools/pairing_probe.py 4; TC_PAIRING_QUAD=1 PROBE_B=1024 timeout 300 python tools/pairing_probe.py 4; PROBE_B=1024 timeout 300 python tools/pairing_probe.py 4 ) > gpurun_out/r03_6_quad_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03_6_quad_ab.txt
TC_PAIRING_QUAD=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -m gpu -q -x -k "pairing or verify or ciphertext or config3 or config4 or rlc or golden or threshold_sig or simple" > gpurun_out/r03_6_quad_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_6_quad_tests.log
tail -6 gpurun_out/r03_6_quad_tests.log
