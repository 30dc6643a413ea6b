"""per-step table of every kernel of a rocprofv3 --stats run: python tools/prof_list.py <kernel_stats.csv> <steps incl. warm-up>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = 0
for r in rows:
    if "spin_kernel" in r["Name"]:
        continue
    c, ns = int(r["Calls"]), int(r["TotalDurationNs"])
    tot += ns
    print("%7.1f/step %8.1f us/step avg %7.2f us  %s" % (c / n, ns / n / 1e3, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("total kernel time per step (us): %.1f" % (tot / n / 1e3))
