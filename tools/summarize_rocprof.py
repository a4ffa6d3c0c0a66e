"""Condense one tools/prof_round.sh output directory into a markdown table: per kernel the --stats average duration and
the FETCH_SIZE / WRITE_SIZE counters per dispatch (KiB as rocprofv3 reports them; FETCH doubled for gfx950, see
MI355X_MICROARCH.md "HBM / rocprofv3")."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    return name.split("(")[0].replace("void ", "").replace("rc::", "").strip()


def main(out):
    stats = {}
    f = find(os.path.join(out, "kt"), "*kernel_stats.csv")
    if f:
        for row in csv.DictReader(open(f)):
            stats[short(row["Name"])] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3, float(row["Percentage"]))
    ctr = {}
    for which in ("fetch", "write"):
        f = find(os.path.join(out, which), "*counter_collection.csv")
        acc = defaultdict(list)
        if f:
            for row in csv.DictReader(open(f)):
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        ctr[which] = {k: sum(v) / len(v) for k, v in acc.items()}
    print("| kernel | calls | avg us | % | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM bytes = 2*FETCH + WRITE |")
    print("|---|---|---|---|---|---|---|")
    for k, (calls, avg, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
        fe, wr = ctr["fetch"].get(k), ctr["write"].get(k)
        hbm = (2 * fe + wr) * 1024 if fe is not None and wr is not None else None
        print("| %s | %d | %.1f | %.1f | %s | %s | %s |" % (k, calls, avg, pct, "%.0f" % fe if fe is not None else "-",
                                                         "%.0f" % wr if wr is not None else "-", "%.4g" % hbm if hbm else "-"))


if __name__ == "__main__":
    main(sys.argv[1])
