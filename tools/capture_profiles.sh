set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
timeout 600 python bench.py > gpurun_out/bench.txt 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -- python $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/prof_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write > gpurun_out/summary.csv 2>&1
tail -c 600 gpurun_out/bench.txt
