# rocprofv3 capture for profiles/<tag>_* (r05_c, r06_c): ONE capture per leg of the bench line (tools/profile_legs.py: the leg's timed launches are
# the last launches of its kernels in the process; tools/rocpd_summary.py --last averages exactly those), each a stats pass and
# separate PMC passes (TCC fetch, TCC write, two SQ passes -- never combined with a trace domain other than --kernel-trace), then
# the same for `bench.py --config 5`, THEN the constants of this capture (tools/profile_constants.py) and the bench line that
# embeds them -- profiles and line on ONE lease.  usage (on the GPU box): bash tools/capture_legs.sh <tag> [extra bench.py args]
tag=$1; shift
R=$GRAFT_REPO_ROOT
REPS=4
cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out profiles
PEAK="# box while profiling: $(timeout 120 tools/ubench_issue --peak 2>/dev/null | tail -1)"
passes() {   # $1 = name, $2... = command
  name=$1; shift
  for p in stats fetch write sq1 sq2; do rm -rf gpurun_out/prof_${name}_$p; done
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${name}_stats -- "$@" > $R/gpurun_out/prof_${name}_stats.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_${name}_fetch -- "$@" > $R/gpurun_out/prof_${name}_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_${name}_write -- "$@" > $R/gpurun_out/prof_${name}_write.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/prof_${name}_sq1 -- "$@" > $R/gpurun_out/prof_${name}_sq1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/prof_${name}_sq2 -- "$@" > $R/gpurun_out/prof_${name}_sq2.log 2>&1
  cd $R
}
for leg in combine verify_g2 hash_g2 g2_sign ciphertext_verify threshold_decrypt wire general_path; do
  passes ${tag}_$leg python $R/tools/profile_legs.py $leg $REPS
  ( echo "$PEAK"; echo "# leg $leg: python tools/profile_legs.py $leg $REPS; per kernel the LAST $REPS dispatches of the process"
    python tools/rocpd_summary.py --last $REPS gpurun_out/prof_${tag}_${leg}_stats gpurun_out/prof_${tag}_${leg}_fetch gpurun_out/prof_${tag}_${leg}_write gpurun_out/prof_${tag}_${leg}_sq1 gpurun_out/prof_${tag}_${leg}_sq2 ) > profiles/${tag}_${leg}_rocprofv3_summary.csv 2>&1
  grep -h '^{' gpurun_out/prof_${tag}_${leg}_stats.log | tail -1 >> gpurun_out/legs_${tag}.txt
  for p in stats fetch write sq1 sq2; do rm -rf gpurun_out/prof_${tag}_${leg}_$p; done
done
# BASELINE config 5, one rank's slice on this GPU (every kernel of the step is launched once per step: plain per-kernel averages)
passes ${tag}_config5 python $R/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline
( echo "$PEAK"; echo "# python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline"
  python tools/rocpd_summary.py gpurun_out/prof_${tag}_config5_stats gpurun_out/prof_${tag}_config5_fetch gpurun_out/prof_${tag}_config5_write gpurun_out/prof_${tag}_config5_sq1 gpurun_out/prof_${tag}_config5_sq2 ) > profiles/${tag}_config5_rocprofv3_summary.csv 2>&1
for p in stats fetch write sq1 sq2; do rm -rf gpurun_out/prof_${tag}_config5_$p; done
# the bench command itself under --kernel-trace --stats (the kernel averages of the WHOLE line's process: headline steps, config 3 / 4
# legs, work-load generation -- what round 4's captures were; the per-leg files above are the ones profile_constants.json reads)
rm -rf gpurun_out/prof_${tag}_bench_stats
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_bench_stats -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --in-flight 1 --sustain-seconds 0 --profile-run > $R/gpurun_out/prof_${tag}_bench_stats.log 2>&1 )
( echo "$PEAK"; echo "# python bench.py --steps 8 --warmup 2 --no-cpu-baseline --in-flight 1 --sustain-seconds 0 --profile-run"
  python tools/rocpd_summary.py gpurun_out/prof_${tag}_bench_stats ) > profiles/${tag}_bench_rocprofv3_summary.csv 2>&1
rm -rf gpurun_out/prof_${tag}_bench_stats
cp gpurun_out/legs_${tag}.txt profiles/${tag}_legs.txt
# the constants of THIS capture, then the line that carries them
python tools/profile_constants.py $tag > /dev/null && cp profiles/profile_constants.json gpurun_out/profile_constants_${tag}.json
# (stdout = the driver's compact line, < 8 KB; the detail object goes to bench_detail.json and, tagged, to stderr)
timeout 1200 python bench.py "$@" > gpurun_out/bench_${tag}.txt 2>gpurun_out/bench_${tag}.err
cp gpurun_out/bench_${tag}.txt profiles/${tag}_bench.txt; cp bench_detail.json profiles/${tag}_bench_detail.json
timeout 600 python bench.py --config 5 --no-cpu-baseline > gpurun_out/bench_${tag}_config5.txt 2>gpurun_out/bench_${tag}_config5.err && cp gpurun_out/bench_${tag}_config5.txt profiles/${tag}_config5_bench.txt && cp bench_detail.json profiles/${tag}_config5_bench_detail.json
cp profiles/${tag}_*.csv profiles/${tag}_*.txt profiles/${tag}_*.json profiles/profile_constants.json gpurun_out/ 2>/dev/null
wc -c gpurun_out/bench_${tag}.txt; head -c 400 gpurun_out/bench_${tag}.txt
