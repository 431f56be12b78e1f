# call 19 (round 4): the pairing unit with the field multipliers waiting for the caller's memory operations at their EXIT instead of
# their entry (tools/asm_entry_wait.py) against the shipped build: correctness first (the pairing tests), then same-box timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TC_AMD_LIB=$GRAFT_REPO_ROOT/threshold_crypto_amd/libtc_amd_xw.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "pairing or verify or config3 or ciphertext or config4" 2>&1 | tail -4
unset TC_AMD_LIB
for rep in 1 2 3; do
for lib in default _xw; do
  if [ $lib = default ]; then unset TC_AMD_LIB; else export TC_AMD_LIB=$GRAFT_REPO_ROOT/threshold_crypto_amd/libtc_amd$lib.so; fi
  PROBE_NOCHECKS=1 timeout 300 python tools/pairing_probe.py 5 2>&1 | grep -v amdgpu | tail -1
done; done | tee gpurun_out/r04_entry_wait_ab.txt
# how libtc_amd_xw.so was made (in the build container, from /tmp/xw):
#   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc --cuda-device-only -S csrc/k_pairing.hip -o k_pairing.s
#   python tools/asm_entry_wait.py k_pairing.s k_pairing.t.s
#   clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c k_pairing.t.s -o k_pairing.dev.o
#   lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o k_pairing.hsaco k_pairing.dev.o
#   clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=k_pairing.hsaco -output=k_pairing.hipfb
#   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang k_pairing.hipfb -c csrc/k_pairing.hip -o k_pairing.o
#   hipcc -shared -fPIC --offload-arch=gfx950 -o libtc_amd_xw.so <the other units' objects> k_pairing.o -ldl -lpthread
