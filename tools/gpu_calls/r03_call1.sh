# round 3, GPU call 1: the GPU suite on the split pairing kernels, same-box A/B of split / fused / one-wave Miller, a
# kernel trace of the split form, one bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_1_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03_1_gpu_tests.log
tail -3 gpurun_out/r03_1_gpu_tests.log
( timeout 300 python tools/pairing_probe.py 4; TC_PAIRING_FUSED=1 timeout 300 python tools/pairing_probe.py 4; TC_MILLER_W1=1 timeout 300 python tools/pairing_probe.py 4; timeout 300 python tools/pairing_probe.py 4 ) > gpurun_out/r03_1_pairing_ab.txt 2>&1
cat gpurun_out/r03_1_pairing_ab.txt
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03_1 -- python $GRAFT_REPO_ROOT/tools/pairing_probe.py 3 > $GRAFT_REPO_ROOT/gpurun_out/r03_1_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof_r03_1 > gpurun_out/r03_1_pairing_summary.csv 2>&1; rm -rf gpurun_out/prof_r03_1
grep -E "miller|final_exp|pairing" gpurun_out/r03_1_pairing_summary.csv | head
timeout 600 python bench.py > gpurun_out/r03_1_bench.txt 2> gpurun_out/r03_1_bench.err; echo "bench rc $?"; tail -c 1500 gpurun_out/r03_1_bench.txt; tail -5 gpurun_out/r03_1_bench.err
