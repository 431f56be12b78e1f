# call 15 (round 4): do the big kernels wait for INSTRUCTIONS?  k_miller_accumulate's loop is 7 862 instructions (43 KB + the 10 KB
# of the out-of-line products), k_combine_fast<Fq2> 211 KB, final_exponentiation 189 KB of straight-line code; the instruction
# cache is shared by the waves of two CUs.  One PMC pass: SQC_ICACHE_* + SQ_IFETCH(_LEVEL), a second: SALU / branch / wait figures.
R=$GRAFT_REPO_ROOT
cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
for p in ic1 ic2; do rm -rf gpurun_out/prof_r04_$p; done
cd /tmp
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --in-flight 1 --sustain-seconds 0 --profile-run"
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $R/gpurun_out/prof_r04_ic1 -- $B > $R/gpurun_out/prof_r04_ic1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES -d $R/gpurun_out/prof_r04_ic2 -- $B > $R/gpurun_out/prof_r04_ic2.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_r04_ic1 gpurun_out/prof_r04_ic2 > gpurun_out/r04_icache_pmc.csv 2>&1
tail -5 gpurun_out/prof_r04_ic1.log; tail -5 gpurun_out/prof_r04_ic2.log
for p in ic1 ic2; do rm -rf gpurun_out/prof_r04_$p; done
grep -c . gpurun_out/r04_icache_pmc.csv
