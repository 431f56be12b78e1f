cd /tmp; export TMPDIR=/tmp; mkdir -p $GRAFT_REPO_ROOT/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03_5 -- python $GRAFT_REPO_ROOT/tools/g1_large_t_probe.py 131072 > $GRAFT_REPO_ROOT/gpurun_out/r03_5_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof_r03_5 > gpurun_out/r03_5_g1_large_t_summary.csv 2>&1; rm -rf gpurun_out/prof_r03_5
head -12 gpurun_out/r03_5_g1_large_t_summary.csv
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03_5b -- python $GRAFT_REPO_ROOT/tools/rlc_samekey_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r03_5b_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof_r03_5b > gpurun_out/r03_5_rlc_samekey_summary.csv 2>&1; rm -rf gpurun_out/prof_r03_5b
head -16 gpurun_out/r03_5_rlc_samekey_summary.csv
