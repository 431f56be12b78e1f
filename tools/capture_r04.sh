# rocprofv3 capture for profiles/r04_*: a stats pass and separate PMC passes (TCC fetch, TCC write, two SQ passes) of the bench command,
# THEN the constants of that capture (tools/profile_constants.py) and the bench line that embeds them -- profile and line on ONE lease, the
# line's `traffic` from the capture beside it.  usage (on the GPU box): bash tools/capture_r04.sh <tag> [extra bench.py args]
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
for p in stats fetch write sq1 sq2; do rm -rf gpurun_out/prof_${tag}_$p; done
cd /tmp
B="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --in-flight 1 --sustain-seconds 0 --profile-run $*"   # profiled launches do not overlap: per-launch durations
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_stats -- $B > $R/gpurun_out/prof_${tag}_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_${tag}_fetch -- $B > $R/gpurun_out/prof_${tag}_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_${tag}_write -- $B > $R/gpurun_out/prof_${tag}_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/prof_${tag}_sq1 -- $B > $R/gpurun_out/prof_${tag}_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/prof_${tag}_sq2 -- $B > $R/gpurun_out/prof_${tag}_sq2.log 2>&1
cd $R
( echo "# box while profiling: $(timeout 120 tools/ubench_issue --peak 2>/dev/null | tail -1)"; python tools/rocpd_summary.py gpurun_out/prof_${tag}_stats gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write gpurun_out/prof_${tag}_sq1 gpurun_out/prof_${tag}_sq2 ) > gpurun_out/summary_${tag}.csv 2>&1
for p in stats fetch write sq1 sq2; do rm -rf gpurun_out/prof_${tag}_$p; done
# the constants of THIS capture, then the line that carries them
mkdir -p profiles; cp gpurun_out/summary_${tag}.csv profiles/${tag}_rocprofv3_summary.csv
python tools/profile_constants.py profiles/${tag}_rocprofv3_summary.csv > /dev/null && cp profiles/profile_constants.json gpurun_out/profile_constants_${tag}.json
timeout 900 python bench.py "$@" > gpurun_out/bench_${tag}.txt 2>gpurun_out/bench_${tag}.err
tail -c 300 gpurun_out/bench_${tag}.txt
