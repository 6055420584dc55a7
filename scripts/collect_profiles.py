#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of scripts/collect_profiles.sh into the summaries kept under profiles/:
   <tag>_k1_hbm_traffic.json  per-launch FETCH_SIZE / WRITE_SIZE of K1 (gfx950 correction of MI355X_MICROARCH.md: FETCH x2),
                              for the 128-walker launches and for the 384-walker (beyond Infinity Cache) launches
   usage: collect_profiles.py <dir with fetch/ and write/ runs> <tag>"""
import csv
import glob
import json
import os
import sys


def per_launch(d, counter, kernel_sub):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]:
                rows.append((int(r["Grid_Size"]), r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out = {"correction": "MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts 64 B per 128-B request of a coalesced "
                         "streaming read -> doubled; WRITE_SIZE exact.  Separate --pmc passes with --kernel-trace only. "
                         "Units: KB per launch as rocprofv3 reports them.",
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 20 --warmup 2 --no-cpu"}
    fe = per_launch(os.path.join(d, "fetch"), "FETCH_SIZE", "k_delta_action")
    wr = per_launch(os.path.join(d, "write"), "WRITE_SIZE", "k_delta_action")
    kern = sorted({k for _, k, _ in fe})
    out["kernels_seen"] = kern
    # the two legs differ by the number of items: group by the median counter value
    def split(rows):
        vals = sorted(v for _, _, v in rows)
        if not vals:
            return [], []
        cut = 0.5 * (vals[0] + vals[-1])
        return [v for v in vals if v <= cut], [v for v in vals if v > cut]
    f_small, f_large = split(fe)
    w_small, w_large = split(wr)
    mean = lambda a: sum(a) / len(a) if a else None
    out["FETCH_SIZE"] = {"launches": len(f_small), "mean_KB": mean(f_small)}
    out["WRITE_SIZE"] = {"launches": len(w_small), "mean_KB": mean(w_small)}
    out["kernel"] = "pigs::k_delta_action_pipe2<3>"
    out["workload"] = "bench.py default: N=256, 161 beads, 128 walkers, 20608 items per launch"
    out["algorithmic_bytes_per_launch"] = 20608 * 6200
    if f_small and w_small:
        out["hbm_bytes_per_launch"] = (2 * mean(f_small) + mean(w_small)) * 1024
    if f_large and w_large:
        out["large"] = {"workload": "384 walkers (380 MB of worldlines), 61824 items per launch",
                        "FETCH_SIZE_mean_KB": mean(f_large), "WRITE_SIZE_mean_KB": mean(w_large), "launches": len(f_large),
                        "hbm_bytes_per_launch": (2 * mean(f_large) + mean(w_large)) * 1024,
                        "algorithmic_bytes_per_launch": 61824 * 6200}
    json.dump(out, open(os.path.join("profiles", tag + "_k1_hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
    kernel_stats(os.path.join(d, "stats"), os.path.join("profiles", tag + "_bench_kernel_stats.csv"))


def kernel_stats(d, dst):
    """Per (kernel, grid size) duration statistics from the kernel trace of
    `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 10 --no-cpu` (the --stats table itself
    lumps the 128-walker and the 384-walker launches of K1 together)."""
    import collections
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name, dur = r["Kernel_Name"].split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if "k_delta_action_pipe2" in name:        # persistent grid: the two legs differ only in duration (31 vs 100 us)
                name += " [20608 items: 128 walkers]" if dur < 65000 else " [61824 items: 384 walkers]"
            acc[(name, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(dur)
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    with open(dst, "w") as f:
        f.write("kernel,grid_size,workgroup_size,calls,total_ns,average_ns,min_ns,max_ns\n")
        for (k, g, wg), v in rows:
            f.write('"%s",%d,%d,%d,%d,%.1f,%d,%d\n' % (k, g, wg, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
    print(open(dst).read())


if __name__ == "__main__":
    main()
