set -u
O=gpurun_out/r3k; mkdir -p $O
timeout -k 10 700 python3 -m pytest tests -q -m gpu > $O/full_gpu.log 2>&1; echo "suite rc=$?"; tail -n 3 $O/full_gpu.log
timeout -k 10 300 python3 bench.py > $O/bench_plain.json 2> $O/bench_plain.err || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || exit 1
LEAN="--steps 30 --warmup 5 --no-parity-check --no-cpu-baseline --no-other-paths --extra-batches="
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- python3 bench.py $LEAN > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || { tail -5 $O/bench_under_rocprof.err; exit 1; }
bash tools/prof_pmc.sh r3k/pmc "dense8_scan_kernel<false>" -- python3 bench.py --steps 10 --warmup 2 --no-parity-check --no-cpu-baseline --no-other-paths --extra-batches= > $O/pmc.log 2>&1
tail -n 30 $O/pmc.log
for t in plain driver_cmd under_rocprof; do python3 -c "
import json;l=json.loads(open('$O/bench_$t.json').read().strip().splitlines()[-1]);print('$t',round(l['value']),round(l['ms_per_step'],4),round(l['roofline']['frac'],3),round(l['roofline']['kernel_ms'],4))"; done
