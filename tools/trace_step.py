"""Anatomy of ONE step from a rocprofv3 --kernel-trace csv: per hardware queue (lane) the span, busy time, launch gaps
and the time per kernel family.  usage: python tools/trace_step.py <kernel_trace.csv> [step_index_from_end]"""
import collections, csv, re, statistics, sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
packs = [i for i, r in enumerate(rows) if "pack_multi" in r["Kernel_Name"]]
end, start = packs[-1 - 2 * (back - 1)], packs[-3 - 2 * (back - 1)] + 1
step = rows[start:end + 1]
t0 = step[0]["s"]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:64]


print(f"step wall {(step[-1]['e'] - t0) / 1e3:.0f} us, {len(step)} kernels")
for q in sorted({r["Queue_Id"] for r in step}):
    ks = [r for r in step if r["Queue_Id"] == q]
    busy = sum(r["e"] - r["s"] for r in ks) / 1e3
    gaps = [(ks[i + 1]["s"] - ks[i]["e"]) / 1e3 for i in range(len(ks) - 1)] or [0]
    print(f"queue {q}: {len(ks)} kernels, span {(ks[0]['s'] - t0) / 1e3:.0f}..{(ks[-1]['e'] - t0) / 1e3:.0f} us, busy {busy:.0f} us, "
          f"median gap {statistics.median(gaps):.2f} us, gaps<20us sum {sum(g for g in gaps if g < 20):.0f} us")
    agg = collections.OrderedDict()
    for r in ks:
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0, 0])
        a[0] += 1
        a[1] += (r["e"] - r["s"]) / 1e3
        a[2] = max(a[2], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"])))
    for k, (n, t, wg) in sorted(agg.items(), key=lambda x: -x[1][1])[:22]:
        print(f"   {k:66s} {n:4d} {t:8.1f} us  avg {t / n:6.1f}  max WGs {wg}")
