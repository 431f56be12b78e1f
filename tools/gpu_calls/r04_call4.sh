# call 4 (round 4): wire-level combiners and IntoFr entries: parity tests + timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_wire.py -x -q -m gpu -k "into_fr" 2>&1 | tail -15 | tee gpurun_out/r04_c4_tests.txt
