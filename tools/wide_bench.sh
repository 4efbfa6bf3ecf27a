#!/bin/bash
# Measurement helper: bench.py over 1 M x 4096 float32 rows (the wide-row filter, sq_dense_wide.hpp) under library option sets.
# usage (GPU box, repo root): tools/wide_bench.sh out_prefix "opts1" "opts2" ...    ("-" = defaults)
pre=$1; shift
i=0
for o in "$@"; do
  [ "$o" = "-" ] && o=""
  python bench.py --gpus 1 --rows ${ROWS:-1000000} --dim ${DIM:-4096} --steps 20 --warmup 3 --no-other-paths --no-cpu-baseline --extra-batches "${EXTRA:-}" --lib-options "$o" > ${pre}_$i.json 2> ${pre}_$i.err || { tail -3 ${pre}_$i.err; exit 1; }
  python - "$o" ${pre}_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d['roofline']
print(f"{sys.argv[1] or 'defaults':34s} ms/step {d['ms_per_step']:.3f}  pass alone {r['kernel_ms']:.3f} ({r['frac']:.3f} of HBM)  rerank {r['rerank_kernel_ms']:.3f}  cands/q {d['config']['mean_candidates_per_query']:.0f}  fallbacks {d['config']['fallback_queries']}  parity {d.get('parity_check',{}).get('bit_identical_topk')}  build {d['index_build_ms']:.0f} ms  other {({k:round(v['ms_per_step'],2) for k,v in d.get('other_batches',{}).items()})}")
PY
  i=$((i+1))
done
