mkdir -p gpurun_out/r3j
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
P="--force-collective --rows 1250000 --no-cpu-baseline --no-other-paths --extra-batches="
for cfg in "i8_100:" "bf_100:--no-int8" "i8_20:--steps 20 --warmup 5" "bf_20:--no-int8 --steps 20 --warmup 5" "i8_nocoll:NOCOLL"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  if [ "$fl" = "NOCOLL" ]; then
    timeout -k 10 200 python3 bench.py --rows 1250000 --no-cpu-baseline --no-other-paths --extra-batches= > gpurun_out/r3j/proxy_$tag.json 2> gpurun_out/r3j/proxy_$tag.err || exit 1
  else
    timeout -k 10 200 python3 bench.py $P $fl > gpurun_out/r3j/proxy_$tag.json 2> gpurun_out/r3j/proxy_$tag.err || { tail -5 gpurun_out/r3j/proxy_$tag.err; exit 1; }
  fi
  python3 -c "
import json;l=json.loads(open('gpurun_out/r3j/proxy_$tag.json').read().strip().splitlines()[-1]);c=l['config'];print('$tag',round(l['value']),round(l['ms_per_step'],4),round(l['roofline']['kernel_ms'],4),'rerank',round(l['roofline']['rerank_kernel_ms'],4),'cands',round(c['mean_candidates_per_query']),'gather',c.get('gather_host_ms'),'merge',c.get('merge_ms'))"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r3j/prof_proxy -o proxy --output-format csv -- python3 bench.py $P > gpurun_out/r3j/prof_proxy.log 2>&1
ls gpurun_out/r3j/prof_proxy | head
