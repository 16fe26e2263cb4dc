#!/usr/bin/env python3
"""Condense rocprofv3 output (tools/profile_round.sh) into the small files kept under profiles/.

    python3 tools/summarize_profiles.py RAW_DIR OUT_DIR workload...

per workload:  <OUT>/r02_<wl>_kernel_stats.csv   the --stats table (top kernels)
               <OUT>/r02_<wl>_pmc.json           mean per-dispatch counters of the dominant kernel + derived HBM bytes
and            <OUT>/traffic.json                 {workload: {hbm_bytes_per_launch, ...}}  (bench.py reads profiles/traffic.json)

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B
(MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact for 16-B-per-lane streaming stores; both are reported in KiB.
"""
import csv
import glob
import json
import os
import sys

KERNEL = "vq_search_"  # vq_search_mfma<...>, vq_search_persist<...> or, for 256 < D <= 512, vq_search_pair512<...>
RND = os.environ.get("VQ_ROUND", "r03")
# dispatches of the search kernel(s) per SEARCH: rows wider than 512 dims are swept in ceil(D / 512) launches, and the HBM
# traffic of one search is the SUM over them (round 2 reported the mean per dispatch there: VERDICT r2 weak #3)
LAUNCHES_PER_SEARCH = {"wide1024": 2}


def find(raw, wl, sub, pattern):
    hits = sorted(glob.glob(os.path.join(raw, wl, sub, "**", pattern), recursive=True))
    return hits[0] if hits else None


def counter_means(path, kernel_substr):
    """mean per dispatch of every counter, over the dispatches of the kernel (summing the per-XCD / per-instance rows)."""
    per_dispatch = {}
    name_of = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            kn = row.get("Kernel_Name", "")
            if kernel_substr not in kn:
                continue
            key = (row.get("Dispatch_Id"), row.get("Counter_Name"))
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row.get("Counter_Value", 0) or 0)
            name_of[row.get("Dispatch_Id")] = kn
    by_counter = {}
    for (_d, c), v in per_dispatch.items():
        by_counter.setdefault(c, []).append(v)
    names = sorted(set(name_of.values()))
    return {c: sum(v) / len(v) for c, v in by_counter.items()}, {c: len(v) for c, v in by_counter.items()}, names


def main():
    raw, out = sys.argv[1], sys.argv[2]
    wls = sys.argv[3:]
    os.makedirs(out, exist_ok=True)
    traffic = {}
    for wl in wls:
        stats = find(raw, wl, "stats", "*kernel_stats.csv")
        if stats:
            rows = list(csv.reader(open(stats, newline="")))
            with open(os.path.join(out, f"{RND}_{wl}_kernel_stats.csv"), "w", newline="") as f:
                csv.writer(f, quoting=csv.QUOTE_MINIMAL).writerows(rows[:8])
        pmc = {}
        kernel_names = []
        for sub in ("fetch", "write", "sq"):
            path = find(raw, wl, sub, "*counter_collection.csv")
            if not path:
                continue
            means, counts, names = counter_means(path, KERNEL)
            pmc.update(means)
            pmc.setdefault("_dispatches", {}).update(counts)
            kernel_names = names or kernel_names
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            nl = LAUNCHES_PER_SEARCH.get(wl, 1)
            hbm = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 * nl
            pmc["hbm_bytes_per_launch"] = hbm
            pmc["launches_per_search"] = nl
            traffic[wl] = {
                "hbm_bytes_per_launch": int(round(hbm)), "launches_per_search": nl,
                "FETCH_SIZE_KB": pmc["FETCH_SIZE"], "WRITE_SIZE_KB": pmc["WRITE_SIZE"],
                "kernel": kernel_names,
                "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python3 bench.py --workload %s --no-legs "
                        "--no-cpu-baseline --no-sharded --steps 20 --warmup 5`, mean over the search kernel's dispatches; gfx950 "
                        "correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request -> doubled; WRITE_SIZE "
                        "exact. hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 x launches_per_search (the counters are means per dispatch)" % wl}
        pmc["kernel"] = kernel_names
        trace = find(raw, wl, "stats", "*kernel_trace.csv")
        if trace:
            # the bench command's dispatch order: settle launches, W = 5 warm-up, K = 20 timed steps, then 3 + 10 isolated
            # launches of kernel_roofline -> the timed steps are dispatches [-33:-13] of the search kernel
            durs = []
            with open(trace, newline="") as f:
                for row in csv.DictReader(f):
                    if KERNEL in row.get("Kernel_Name", ""):
                        durs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
            durs = [d for _s, d in sorted(durs)]
            nl = LAUNCHES_PER_SEARCH.get(wl, 1)
            if len(durs) >= 33 * nl:
                timed = durs[-33 * nl:-13 * nl]
                timed = [sum(timed[i * nl:(i + 1) * nl]) for i in range(20)]  # one entry per search
                pmc["kernel_trace_timed_steps"] = {"launches": len(timed), "launches_per_search": nl, "mean_ns": sum(timed) / len(timed),
                                                   "min_ns": min(timed), "max_ns": max(timed), "all_launches": len(durs),
                                                   "mean_all_ns": sum(durs) / len(durs) * nl}
        bench_line = os.path.join(raw, f"{wl}.bench_under_trace.json")
        if os.path.exists(bench_line):
            try:
                pmc["bench_line_under_kernel_trace"] = json.loads(open(bench_line).read().strip().splitlines()[-1])["roofline"]
            except Exception:
                pass
        json.dump(pmc, open(os.path.join(out, f"{RND}_{wl}_pmc.json"), "w"), indent=1)
    tpath = os.path.join(out, "traffic.json")
    merged = {}
    if os.path.exists(tpath):  # a partial re-run (some workloads only) keeps the others' entries
        try:
            merged = json.load(open(tpath))
        except Exception:
            merged = {}
    merged.update(traffic)
    json.dump(merged, open(tpath, "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
