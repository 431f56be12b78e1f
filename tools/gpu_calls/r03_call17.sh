# call 17: soak with the pairing form forced, then default
(TC_PAIRING_FORM=pair timeout 400 python tests/soak.py 240 31; TC_PAIRING_FORM=fused timeout 300 python tests/soak.py 120 32; timeout 400 python tests/soak.py 240 33) 2>&1 | grep -v amdgpu | tee gpurun_out/r03_soak2.txt | tail -4
