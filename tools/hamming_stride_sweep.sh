for s in 32 64 128 256; do STRIDES=$s OPTS=hamming_fused=1 NQS=1,32 python tools/hamming_latency.py; done 2>&1 | grep -v amdgpu
