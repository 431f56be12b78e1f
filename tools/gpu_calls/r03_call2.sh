# round 3, GPU call 2: the whole GPU suite (input checks on by default), bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03_2_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_2_gpu_tests.log
tail -15 gpurun_out/r03_2_gpu_tests.log
timeout 600 python bench.py > gpurun_out/r03_2_bench.txt 2> gpurun_out/r03_2_bench.err; echo "bench rc $?"; tail -3 gpurun_out/r03_2_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03_2_bench.txt") if l.startswith("{")][-1])
for k in ("value", "ms_per_step", "streaming", "sustained", "pairing_verifies_per_s", "pairing_verify_kernel_ms"):
    print(k, d[k])
print("general", d["general_path"]["value"], d["general_path"]["roofline"]["frac"], "config4", d["config4"]["value"], d["config4"]["kernel_ms"])
print("roofline", d["roofline"]["kernel_ms"], d["roofline"]["frac"], {k: (v["kernel_ms"], v["frac"]) for k, v in d["secondary_rooflines"].items()})
print(d["extras"])
PY
