cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
( for w in 2 3 4; do TC_PAIRING_QUAD=1 TC_QUAD_WAVES=$w PROBE_NOCHECKS=1 timeout 300 python tools/pairing_probe.py 3; done; PROBE_NOCHECKS=1 timeout 300 python tools/pairing_probe.py 3
  for b in 32768 16384 4096; do TC_PAIRING_QUAD=1 PROBE_NOCHECKS=1 PROBE_B=$b timeout 300 python tools/pairing_probe.py 3; PROBE_NOCHECKS=1 PROBE_B=$b timeout 300 python tools/pairing_probe.py 3; done ) > gpurun_out/r03_7_quad_waves.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03_7_quad_waves.txt
