O=gpurun_out/r3k; mkdir -p $O
L="--no-cpu-baseline --no-other-paths --extra-batches= --no-parity-check"
export SQ_INT8_REPORT=1
for D in nonneg normal uniform clustered; do
  timeout -k 10 200 python3 bench.py --data $D $L > $O/data_$D.json 2> $O/data_$D.err || exit 1
  grep "int8 filter" $O/data_$D.err | head -1
  python3 -c "
import json,sys;l=json.loads(open('$O/data_$D.json').read().strip().splitlines()[-1]);c=l['config'];r=l['roofline'];print('$D',round(l['value']),round(l['ms_per_step'],4),'kernel',round(r['kernel_ms'],4),r.get('first_stage_filter'),'cands',round(c['mean_candidates_per_query'] or 0),'mid',c.get('mid_tier_queries'),'exact',c.get('fallback_queries'))"
done
