#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into the small files kept in profiles/.
usage: tools/summarize_prof.py <round-tag>[:suffix] <kernel_stats.csv> [<pmc counter_collection.csv> ...]
(a tag like r01:default_cmd writes r01_kernel_stats_default_cmd.csv and touches nothing else)"""
import collections, csv, json, os, sys

def main():
    tag, stats = sys.argv[1], sys.argv[2]
    suffix = ""
    if ":" in tag:
        tag, suffix = tag.split(":", 1)
        suffix = "_" + suffix
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats{suffix}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[3:]:
        for r in csv.DictReader(open(path)):
            pmc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if pmc:
        summary = {}
        for k, cs in pmc.items():
            if not k.startswith("sq::") and "sq::" not in k:
                continue
            summary[k] = {c: {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in cs.items()}
        with open(os.path.join(out_dir, f"{tag}_pmc.json"), "w") as f:
            json.dump(summary, f, indent=1, sort_keys=True)
        scan = [k for k in summary if "dense_scan_kernel" in k]
        if scan:
            # sample pass and full pass are two instantiations: the full pass is the one that fetches most
            s = summary[max(scan, key=lambda k: summary[k].get("FETCH_SIZE", {}).get("max", 0.0))]
            # MI355X_MICROARCH.md (HBM): FETCH_SIZE is in KiB and reads exactly 1/2 of a wide coalesced
            # stream on gfx950 -> double it; WRITE_SIZE is exact.  The full pass is the larger launch.
            fetch = s.get("FETCH_SIZE", {}).get("max", 0.0) * 1024 * 2
            write = s.get("WRITE_SIZE", {}).get("max", 0.0) * 1024
            json.dump({"dense_scan_full_pass_hbm_bytes": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write,
                       "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), {tag}; FETCH_SIZE x2 per MI355X_MICROARCH.md"},
                      open(os.path.join(out_dir, "latest_traffic.json"), "w"), indent=1)
    print("wrote", out_dir)

if __name__ == "__main__":
    main()
