for o in "" "dense_tighten=2"; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-other-paths --no-cpu-baseline --no-parity-check --extra-batches "64,128,256,1024" --lib-options "$o" > gpurun_out/s30_b_${o:-default}.json 2> gpurun_out/s30_b.err
python - "$o" gpurun_out/s30_b_${o:-default}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1] or 'defaults':18s} step {d['ms_per_step']:.4f}", {k:(round(v['ms_per_step'],4), round(v['mean_candidates_per_query'])) for k,v in d['other_batches'].items()})
PY
done
for o in "" "dense_tighten=2"; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-int8 --no-other-paths --no-cpu-baseline --no-parity-check --extra-batches "" --lib-options "$o" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-int8 ${o:-defaults}', d['ms_per_step'], d['config']['mean_candidates_per_query'])"
done
