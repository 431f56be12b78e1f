# call 21: the base-4 sign-aligned G1 GLV ladder: GPU suite, G1 probe, bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03_21_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_21_gpu_tests.log
grep -E "passed|failed|rc|Error|assert" gpurun_out/r03_21_gpu_tests.log | tail -6
timeout 300 python tools/g1_probe.py 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/r03_21_g1_probe.txt
timeout 600 python bench.py > gpurun_out/bench_r03_21.txt 2> gpurun_out/bench_r03_21.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_r03_21.txt") if l.startswith("{")][-1])
print(d["value"], d["config4"]["value"], d["config4"]["kernel_ms"], d["extras"]["threshold_decrypts_per_s"], d["secondary_rooflines"].get("threshold_decrypt", {}).get("frac"))
PY
