# call 24: the default bench line with the 3 s sustained leg
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
S=$(date +%s.%N); timeout 600 python bench.py > gpurun_out/bench_r03_24.txt 2> gpurun_out/bench_r03_24.err; echo "bench rc $? wall $(echo "$(date +%s.%N) - $S" | bc) s"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_r03_24.txt") if l.startswith("{")][-1])
print(d["value"], d["sustained"], d["streaming"]["value"], d["config4"]["value"])
PY
