"""Folds the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch) into HBM bytes per launch per kernel.
gfx950: FETCH_SIZE tallies a 128-byte request as 64 bytes, hence 2*FETCH_SIZE (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, re, sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        name = re.sub(r"^void ", "", name)
        name = name.replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*$", "", name).strip()
        acc[name][0] += 1
        acc[name][1] += float(r["Counter_Value"])
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 2 "
               "--no-roofline --no-cpu-baseline`; KB per launch averaged over all launches of the kernel; hbm_bytes_per_launch = "
               "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md 'HBM')",
       "kernels": {}}
for k in sorted(fetch, key=lambda k: -fetch[k][1]):
    n, fk = fetch[k]
    wk = write.get(k, [n, 0.0])[1]
    out["kernels"][k] = {"launches": n, "fetch_kb": round(fk / n, 1), "write_kb": round(wk / max(1, write.get(k, [n])[0]), 1),
                         "hbm_bytes_per_launch": int((2 * fk / n + wk / max(1, write.get(k, [n])[0])) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k in list(out["kernels"])[:12]:
    print(k[:80], out["kernels"][k])
