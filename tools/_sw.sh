mkdir -p gpurun_out/r3j
for D in 3 4 5 6; do
  echo "== depth $D"
  N=1250000 STEPS=800 DEPTH=$D SQ_INT8_REPORT=1 timeout -k 10 200 python3 tools/step_sweep.py dense_graph=0,1,0,1 2>&1 | grep -v amdgpu.ids || exit 1
done
