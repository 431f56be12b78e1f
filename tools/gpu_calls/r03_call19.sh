# call 19: rocprofv3 capture of the config-5 line at HEAD; the batch-size sweep of combine / verify at HEAD
bash tools/capture_r03.sh c5 --config 5 2>&1 | tail -3
timeout 600 python tools/batch_sweep.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03_batch_sweep.txt | tail -12
