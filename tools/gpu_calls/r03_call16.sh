cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_16_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_16_gpu_tests.log
grep -E "passed|failed|rc" gpurun_out/r03_16_gpu_tests.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 600 python bench.py > gpurun_out/bench_r03_16.txt 2> gpurun_out/bench_r03_16.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_r03_16.txt") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["streaming"]["value"], d["config3"]["kernel_ms"], d["config4"]["value"], d["roofline"]["traffic"])
PY
