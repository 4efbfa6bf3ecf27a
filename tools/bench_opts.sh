#!/bin/bash
# Measurement helper: the driver's bench command under several library option sets (one JSON line each).
# usage: tools/bench_opts.sh out_prefix "opts1" "opts2" ...      (opts: name=v,name=v or "-" for the defaults)
pre=$1; shift
i=0
for o in "$@"; do
  [ "$o" = "-" ] && o=""
  python bench.py --gpus 1 --steps ${STEPS:-20} --warmup 3 --no-other-paths --no-parity-check --extra-batches "" --lib-options "$o" > ${pre}_$i.json 2> ${pre}_$i.err || exit 1
  python - "$o" ${pre}_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d['roofline']
print(f"{sys.argv[1] or 'defaults':40s} ms/step {d['ms_per_step']:.4f}  body alone {r['kernel_ms']:.4f}  in pipeline {r['kernel_ms_in_pipeline']:.4f}  cands/q {d['config']['mean_candidates_per_query']:.0f}  frac_step {r['frac_step']:.3f}")
PY
  i=$((i+1))
done
