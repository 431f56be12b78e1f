# call 8 (round 4): the whole GPU suite at this commit, the latency table, the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04_c8_tests.txt
timeout 600 python bench.py --latency-table 2>gpurun_out/r04_latency.err | tee gpurun_out/r04_latency_table.txt | cut -c1-400
timeout 900 python bench.py > gpurun_out/r04_c8_bench.txt 2>gpurun_out/r04_c8_bench.err; tail -c 1500 gpurun_out/r04_c8_bench.txt; tail -5 gpurun_out/r04_c8_bench.err
