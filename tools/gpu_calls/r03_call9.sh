cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_9_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_9_gpu_tests.log
tail -6 gpurun_out/r03_9_gpu_tests.log
bash tools/capture_r03.sh r03_a
timeout 600 python bench.py --config 5 > gpurun_out/bench_r03_config5.txt 2> gpurun_out/bench_r03_config5.err; echo "config5 rc $?"; tail -c 600 gpurun_out/bench_r03_config5.txt
( timeout 300 python tools/g1_large_t_probe.py 131072; timeout 300 python tools/g1_large_t_probe.py 32768; timeout 300 python tools/g1_large_t_probe.py 4096; timeout 300 python tools/g1_large_t_probe.py 512 ) > gpurun_out/r03_9_g1_large_t.txt 2>&1
grep -v amdgpu gpurun_out/r03_9_g1_large_t.txt
timeout 300 python tools/rlc_samekey_probe.py > gpurun_out/r03_9_rlc_samekey.txt 2>&1; grep -v amdgpu gpurun_out/r03_9_rlc_samekey.txt
