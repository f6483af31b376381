"""Summarise a rocprofv3 kernel_trace.csv: per-kernel totals for the LAST batch plus solver gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last batch = after the last k_assign_chacha
idx = max(i for i, r in enumerate(rows) if "k_assign" in r["Kernel_Name"])
rows = rows[idx:]
tot = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    t = tot.setdefault(name, [0, 0]); t[0] += 1; t[1] += d
for k, (c, d) in tot.items():
    print("%-40s calls=%4d total=%9.3f ms avg=%9.1f us" % (k, c, d / 1e6, d / c / 1e3))
sol = [r for r in rows if "k_solver" in r["Kernel_Name"]]
if sol:
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sol]
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(sol, sol[1:])]
    print("solver: span %.3f ms, sum durations %.3f ms, sum gaps %.3f ms, max dur %.1f us, median dur %.1f us, median gap %.1f us" % (
        (int(sol[-1]["End_Timestamp"]) - int(sol[0]["Start_Timestamp"])) / 1e6, sum(durs) / 1e6, sum(gaps) / 1e6, max(durs) / 1e3, sorted(durs)[len(durs) // 2] / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
    top = sorted(zip(durs, [r["Grid_Size_Y"] for r in sol]), reverse=True)[:8]
    print("longest solver levels (us, gridY):", [(round(d / 1e3, 1), g) for d, g in top])
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("batch span %.3f ms" % (span / 1e6))
if sol:
    print("first 24 solver levels (us, gridY):", [(round(d / 1e3, 1), r["Grid_Size_Y"]) for d, r in list(zip(durs, sol))[:24]])
    print("scratch/vgpr:", sol[0]["Scratch_Size"], sol[0]["VGPR_Count"])
