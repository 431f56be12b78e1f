# call 20: the whole GPU suite, smoke and the default bench line at HEAD; then a 10-minute soak
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_20_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_20_gpu_tests.log
grep -E "passed|failed|rc" gpurun_out/r03_20_gpu_tests.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 600 python bench.py > gpurun_out/bench_r03_20.txt 2> gpurun_out/bench_r03_20.err; echo "bench rc $?"
timeout 700 python tests/soak.py 600 41 2>&1 | grep -v amdgpu | tail -2 | tee gpurun_out/r03_soak3.txt
