cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for mode in pair quad; do
  if [ $mode = quad ]; then export TC_PAIRING_QUAD=1; else unset TC_PAIRING_QUAD; fi
  PROBE_NOCHECKS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/prof_r03_8_${mode}_sq1 -- python $R/tools/pairing_probe.py 2 > $R/gpurun_out/r03_8_${mode}_sq1.log 2>&1
  PROBE_NOCHECKS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $R/gpurun_out/prof_r03_8_${mode}_sq2 -- python $R/tools/pairing_probe.py 2 > $R/gpurun_out/r03_8_${mode}_sq2.log 2>&1
  PROBE_NOCHECKS=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_r03_8_${mode}_fetch -- python $R/tools/pairing_probe.py 2 > $R/gpurun_out/r03_8_${mode}_f.log 2>&1
  PROBE_NOCHECKS=1 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_r03_8_${mode}_write -- python $R/tools/pairing_probe.py 2 > $R/gpurun_out/r03_8_${mode}_w.log 2>&1
  cd $R; python tools/rocpd_summary.py gpurun_out/prof_r03_8_${mode}_sq1 gpurun_out/prof_r03_8_${mode}_sq2 gpurun_out/prof_r03_8_${mode}_fetch gpurun_out/prof_r03_8_${mode}_write > gpurun_out/r03_8_${mode}_pmc.csv 2>&1
  rm -rf gpurun_out/prof_r03_8_${mode}_*; cd /tmp
  grep -E "miller|final_exp|pairing_quad" $R/gpurun_out/r03_8_${mode}_pmc.csv
done
