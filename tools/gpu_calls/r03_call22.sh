# call 22: bench line + rocprofv3 passes (stats, FETCH, WRITE, SQ x2) at the build with the base-4 G1 ladder -> profiles/r03_c_*
bash tools/capture_r03.sh r03_c 2>&1 | tail -3
