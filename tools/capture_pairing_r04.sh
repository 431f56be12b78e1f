# rocprofv3 capture of the pairing check alone (tools/pairing_probe.py: verify_g2 over 65 536 checks, operands trusted): stats + FETCH +
# WRITE + SQ passes on ONE lease.  usage (GPU box): bash tools/capture_pairing_r04.sh <tag> [TC_PAIRING_FORM value]
tag=$1; form=$2
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
if [ -n "$form" ]; then export TC_PAIRING_FORM=$form; else unset TC_PAIRING_FORM; fi
export PROBE_NOCHECKS=1
B="python $R/tools/pairing_probe.py 6"
for p in stats fetch write sq1; do rm -rf $R/gpurun_out/pp_${tag}_$p; done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pp_${tag}_stats -- $B > $R/gpurun_out/pp_${tag}_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pp_${tag}_fetch -- $B > $R/gpurun_out/pp_${tag}_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pp_${tag}_write -- $B > $R/gpurun_out/pp_${tag}_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pp_${tag}_sq1 -- $B > $R/gpurun_out/pp_${tag}_sq1.log 2>&1
cd $R
( echo "# pairing check alone, form=${form:-auto}; box clock while profiling: $(timeout 60 tools/ubench_clock --peak 2>/dev/null | tail -1)"; python tools/rocpd_summary.py gpurun_out/pp_${tag}_stats gpurun_out/pp_${tag}_fetch gpurun_out/pp_${tag}_write gpurun_out/pp_${tag}_sq1 ) | grep -v "rocclr\|k_g2_mul_shared\|k_hash_g2\|k_combine\|k_lagrange\|k_msm\|k_point_mul\|k_fill" > gpurun_out/summary_pairing_${tag}.csv 2>&1
for p in stats fetch write sq1; do rm -rf gpurun_out/pp_${tag}_$p; done
unset TC_PAIRING_FORM PROBE_NOCHECKS
