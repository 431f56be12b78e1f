# call 18 (round 4): the bench line with the clock / power its legs ran at; the line-contract test
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r04_b_bench.txt 2> gpurun_out/r04_b_bench.err; tail -c 400 gpurun_out/r04_b_bench.txt; tail -3 gpurun_out/r04_b_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_b_bench.txt') if l.startswith('{')][-1])
print('value',d['value'],'frac',d['roofline']['frac'],'at kernel clock',d['roofline'].get('frac_at_kernel_clock'),d['sustained'].get('clock'))
print('config3',d['config3']['value'],d['config3']['roofline']['frac'],d['config3']['roofline'].get('frac_at_kernel_clock'),d['config3']['sustained'])
PY
timeout 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k bench 2>&1 | tail -3
