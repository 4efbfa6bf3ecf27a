#!/bin/bash
# Measurement helper: the driver's bench command with extra bench.py flags (one line per flag set).
# usage: STEPS=20 tools/bench_flags.sh out_prefix "flags1" "flags2" ...
pre=$1; shift
i=0
for o in "$@"; do
  [ "$o" = "-" ] && o=""
  python bench.py --gpus 1 --steps ${STEPS:-20} --warmup 3 --no-other-paths --no-parity-check --extra-batches "" $o > ${pre}_$i.json 2> ${pre}_$i.err || exit 1
  python - "$o" ${pre}_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d['roofline']
print(f"{sys.argv[1] or 'defaults':40s} ms/step {d['ms_per_step']:.4f}  body alone {r['kernel_ms']:.4f}  in pipeline {r['kernel_ms_in_pipeline']:.4f}  cands/q {d['config']['mean_candidates_per_query']:.0f}  frac_step {r['frac_step']:.3f}")
PY
  i=$((i+1))
done
