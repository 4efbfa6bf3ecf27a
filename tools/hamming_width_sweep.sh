for o in hamming_fused=1 hamming_fused=0; do OPTS=$o NQS=1,8,32,64,256 python tools/hamming_latency.py 5000000 256; done 2>&1 | grep -v amdgpu
for o in hamming_fused=1 hamming_fused=0; do OPTS=$o NQS=64,256,1024 python tools/hamming_latency.py 10000000 128; done 2>&1 | grep -v amdgpu
