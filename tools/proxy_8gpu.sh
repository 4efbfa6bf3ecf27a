#!/bin/bash
# Measurement helper: ONE rank at the shape an 8-GPU run gives each rank (1.25 M-row shard of the 10 M x 128 matrix, the
# all-gather + host merge forced over a world of one) for 32 / 256 / 1024 queries per step, and the same batches on the
# whole matrix on one GPU.  A proxy, not a scaling measurement.  usage (GPU box, repo root): tools/proxy_8gpu.sh out_prefix
pre=$1
port=29541
for q in 32 256 1024; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --force-collective --rows 1250000 --queries $q --no-cpu-baseline --no-other-paths --no-parity-check --extra-batches "" > ${pre}_shard_q$q.json 2> ${pre}_shard_q$q.err || exit 1
  port=$((port+1))
  python bench.py --gpus 1 --queries $q --no-cpu-baseline --no-other-paths --no-parity-check --extra-batches "" > ${pre}_one_gpu_q$q.json 2> ${pre}_one_gpu_q$q.err || exit 1
  python - ${pre}_shard_q$q.json ${pre}_one_gpu_q$q.json $q <<'PY'
import json,sys
a=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[3]:>5s} queries/step: 1.25 M-row shard + forced collective {a['ms_per_step']:.4f} ms/step ({a['value']:.0f} q/s), 10 M rows on one GPU {b['ms_per_step']:.4f} ms/step ({b['value']:.0f} q/s): proxy ratio {b['ms_per_step']/a['ms_per_step']:.2f}x")
PY
done
