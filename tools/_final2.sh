set -u
O=gpurun_out/r3k; mkdir -p $O
L="--no-cpu-baseline --no-other-paths --extra-batches= --no-parity-check"
show() { python3 -c "
import json,sys;l=json.loads(open('$1').read().strip().splitlines()[-1]);c=l['config'];r=l['roofline'];print('$2',round(l['value']),round(l['ms_per_step'],4),'kernel',round(r['kernel_ms'],4),r.get('first_stage_filter'),'cands',round(c['mean_candidates_per_query'] or 0),'mid',c.get('mid_tier_queries'),'exact',c.get('fallback_queries'),'gather',c.get('gather_host_ms'),'merge',c.get('merge_ms'))"; }
for D in uniform clustered nonneg; do
  timeout -k 10 200 python3 bench.py --data $D $L > $O/data_$D.json 2> $O/data_$D.err || exit 1; show $O/data_$D.json $D
done
timeout -k 10 200 python3 bench.py --query-scale 300 $L > $O/mid_qs300.json 2> $O/mid_qs300.err || exit 1; show $O/mid_qs300.json qs300
timeout -k 10 200 python3 bench.py --data far_clusters $L > $O/mid_far.json 2> $O/mid_far.err || exit 1; show $O/mid_far.json far_clusters
timeout -k 10 200 python3 bench.py --no-int8 $L > $O/bench_noint8.json 2> $O/bench_noint8.err || exit 1; show $O/bench_noint8.json no_int8
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
P="--force-collective --rows 1250000 --no-cpu-baseline --no-other-paths --extra-batches="
timeout -k 10 200 python3 bench.py $P > $O/proxy_100.json 2> $O/proxy_100.err || exit 1; show $O/proxy_100.json proxy100
timeout -k 10 200 python3 bench.py $P --steps 20 --warmup 5 > $O/proxy_20.json 2> $O/proxy_20.err || exit 1; show $O/proxy_20.json proxy20
timeout -k 10 200 python3 bench.py $P --queries 256 > $O/proxy_q256.json 2> $O/proxy_q256.err || exit 1; show $O/proxy_q256.json proxy_q256
unset MASTER_ADDR MASTER_PORT RANK WORLD_SIZE LOCAL_RANK
timeout -k 10 200 python3 bench.py --queries 256 $L > $O/one_q256.json 2> $O/one_q256.err || exit 1; show $O/one_q256.json one_q256
