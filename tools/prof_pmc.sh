#!/bin/bash
# PMC passes of one measurement script under rocprofv3 (run on the GPU box from the repo root):
#   tools/prof_pmc.sh <out-dir under gpurun_out> <kernel-name substring> -- python3 tools/itq_breakdown.py
# Counters are taken from the list below when `rocprofv3 -L` knows them, at most 8 SQ counters per pass;
# FETCH_SIZE and WRITE_SIZE get passes of their own (MI355X_MICROARCH.md, rocprofv3 PMC slots).  Collected with
# --pmc only (no trace domains).  Writes <out>/pmc_summary.json: per-launch means for kernels matching the substring.
set -u
OUT=gpurun_out/$1; KSUB=$2; shift 3
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
have() { grep -q -w "$1" $OUT/counters_list.txt; }
pick() { local n=0; local out=""; for c in "$@"; do if have $c && [ $n -lt 8 ]; then out="$out $c"; n=$((n+1)); fi; done; echo $out; }
P1=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16)
P2=$(pick SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS)
P3=$(pick SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT)
i=0
# optional cache-side passes: PMC_EXTRA="TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY;TCC_HIT TCC_MISS" (passes separated by ';').
# Every extra pass goes through pick() like P1-P3: a counter gfx950 does not have (TA_BUSY ...) is dropped with a
# message -- handed to `rocprofv3 --pmc` it aborts the profiler (signal 6) before the program starts.
IFS=';' read -r -a EXTRA_RAW <<< "${PMC_EXTRA:-}"
EXTRA=()
for E in "${EXTRA_RAW[@]}"; do
  [ -z "$E" ] && continue
  # shellcheck disable=SC2086
  F=$(pick $E)
  for c in $E; do have $c || echo "PMC_EXTRA: counter $c is not known to rocprofv3 -L on this device: dropped"; done
  if [ -z "$F" ]; then echo "PMC_EXTRA pass '$E': nothing left, skipped"; else EXTRA+=("$F"); fi
done
for P in "$P1" "$P2" "$P3" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "${EXTRA[@]}"; do
  i=$((i+1))
  [ -z "$P" ] && continue
  echo "pass $i: $P"
  # shellcheck disable=SC2086
  rocprofv3 --pmc $P --kernel-trace -d $OUT/pmc$i -o pmc --output-format csv -- "$@" > $OUT/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pmc$i.log; }
done
python3 - "$OUT" "$KSUB" <<'PY'
import collections, csv, glob, json, os, sys
out, ksub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if ksub in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# kernel durations of the SAME passes (--kernel-trace rides along: clocks differ between profiled passes, so a
# counter is only comparable with the duration of its own pass)
for path in glob.glob(out + "/pmc*/**/*kernel_trace.csv", recursive=True):
    tag = "duration_ns_pass_" + os.path.basename(os.path.dirname(path)).replace("pmc", "")
    for r in csv.DictReader(open(path)):
        if ksub in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:100]][tag].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
summ = {k: {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, cs in summ.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} {v['mean']:16.1f}  ({v['launches']} launches)")
PY
