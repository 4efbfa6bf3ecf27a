#!/bin/bash
# Measurement helper: the one-rank proxy of the 8-GPU step (1.25 M-row shard, collective forced, 32 queries) under bench.py flag sets.
# usage (GPU box, repo root): tools/proxy_sweep.sh out_prefix "flags1" "flags2" ...   ("-" = defaults)
pre=$1; shift
i=0; port=29600
for o in "$@"; do
  [ "$o" = "-" ] && o=""
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --force-collective --rows ${ROWS:-1250000} --queries ${Q:-32} --steps ${STEPS:-100} --no-cpu-baseline --no-other-paths --no-parity-check --extra-batches "" $o > ${pre}_$i.json 2> ${pre}_$i.err || { tail -3 ${pre}_$i.err; exit 1; }
  python - "$o" ${pre}_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1] or 'defaults':44s} ms/step {d['ms_per_step']:.4f}  ({d['value']:.0f} q/s)  gather_host_ms {d['config'].get('gather_host_ms')}  lag {d['config'].get('results_lag_steps')}")
PY
  i=$((i+1)); port=$((port+1))
done
