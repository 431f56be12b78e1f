# call 13 (round 4): the whole GPU suite, the config-2 capture (profile passes, constants, bench line on one lease), config 5 on one GPU:
# one rank's slice and the full 1 048 576-job batch as eight slices
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r04_c13_tests.txt
bash tools/capture_r04.sh r04_a
timeout 600 python bench.py --config 5 --steps 3 --warmup 1 > gpurun_out/r04_config5_1gpu_bench.txt 2>gpurun_out/r04_config5.err; tail -c 400 gpurun_out/r04_config5_1gpu_bench.txt
timeout 900 python bench.py --config 5 --emulate-world 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_config5_full.txt 2>gpurun_out/r04_config5_full.err; tail -c 600 gpurun_out/r04_config5_full.txt; tail -3 gpurun_out/r04_config5_full.err
