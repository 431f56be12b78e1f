# call 17 (round 4): shader clock and power under the real kernels against the multiply-add-only loop the roofline peak comes from
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocm-smi --showclocks --showpower --json 2>&1 | head -c 1500; echo
timeout 300 python tools/clock_probe.py 4 2>&1 | grep -v amdgpu | tee gpurun_out/r04_clock_probe.txt
