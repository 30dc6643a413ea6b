"""How much of a replayed step has BOTH lanes busy?  Reduces a `rocprofv3 --kernel-trace` CSV of bench.py to one JSON:
the two streams with the most kernel time are the lanes; per lane the union of its kernels' [start, end] intervals, over a
window of N whole steps (from the end of one generator weight repack - the step's last launch on lane A - to the end of the
N-th one after it).
    python tools/overlap_from_trace.py <kernel_trace.csv> [steps=3] > profiles/r03_overlap.json
"""
import csv
import json
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def clip(iv, lo, hi):
    return [[max(a, lo), min(b, hi)] for a, b in iv if b > lo and a < hi]


def inter(x, y):
    i = j = 0
    tot = 0
    while i < len(x) and j < len(y):
        lo, hi = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if hi > lo:
            tot += hi - lo
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    path, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = list(csv.DictReader(open(path)))
    by = defaultdict(list)
    for r in rows:
        by[r["Stream_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    busy = {s: sum(b - a for a, b, _ in v) for s, v in by.items()}
    lanes = sorted(busy, key=lambda s: -busy[s])[:2]
    # lane A is the stream of the generator chain
    la = max(lanes, key=lambda s: sum(1 for _, _, n in by[s] if "resblock_kernel" in n or "resblock_ws_kernel" in n))
    lb = [s for s in lanes if s != la][0]
    marks = sorted(b for a, b, n in by[la] if "pack_multi_kernel" in n)
    if len(marks) < nsteps + 1:
        raise SystemExit("not enough whole steps in the trace")
    lo, hi = marks[-(nsteps + 1)], marks[-1]
    ua, ub = clip(union([(a, b) for a, b, _ in by[la]]), lo, hi), clip(union([(a, b) for a, b, _ in by[lb]]), lo, hi)
    ta, tb = sum(b - a for a, b in ua), sum(b - a for a, b in ub)
    both = inter(ua, ub)
    wall = hi - lo
    others = sum(min(b, hi) - max(a, lo) for s, v in by.items() if s not in lanes for a, b, _ in v if b > lo and a < hi)
    ksum = sum(min(b, hi) - max(a, lo) for s in lanes for a, b, _ in by[s] if b > lo and a < hi)
    out = {"source": path.split("/")[-1], "steps": nsteps, "wall_ms_per_step": round(wall / nsteps / 1e6, 4),
           "lane_A_busy_frac": round(ta / wall, 4), "lane_B_busy_frac": round(tb / wall, 4),
           "both_lanes_busy_frac": round(both / wall, 4), "exactly_one_lane_busy_frac": round((ta + tb - 2 * both) / wall, 4),
           "no_lane_busy_frac": round(1.0 - (ta + tb - both) / wall, 4),
           "kernel_time_ms_per_step": round(ksum / nsteps / 1e6, 4),
           "kernel_time_on_other_streams_ms_per_step": round(others / nsteps / 1e6, 4),
           "note": "lane = HIP stream; busy = union of the kernel execution intervals rocprofv3 reports for the stream (kernels of one "
                   "stream never overlap each other; under the profiler every launch carries ~1-2 us of extra dispatch cost, so "
                   "the wall time per step here is above the unprofiled bench line)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
