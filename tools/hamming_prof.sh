#!/bin/bash
# Measurement helper: per-kernel averages of one Hamming search shape under rocprofv3 --kernel-trace --stats, for several
# option sets.  usage (GPU box, repo root): NQ=1 tools/hamming_prof.sh out_dir "opts1" "opts2" ...   ("-" = defaults)
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
i=0
for o in "$@"; do
  [ "$o" = "-" ] && o=""
  (cd /tmp && OPTS="$o" PROFILE=0 REPS=${REPS:-40} rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/p$i -o h --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/hamming_breakdown.py > $GRAFT_REPO_ROOT/$out/p$i.log 2>&1)
  echo "== NQ=${NQ:-32} W=${W:-1} N=${N:-10000000} opts: ${o:-defaults}"
  python3 tools/kstats.py $out/p$i/h_kernel_stats.csv | grep -v "at::native\|rocclr" | head -8
  i=$((i+1))
done
