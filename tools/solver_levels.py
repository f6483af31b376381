"""Per-level durations of k_solver in the LAST batch of a rocprofv3 kernel trace (csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_assign" in r["Kernel_Name"])
lv = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]), r["Kernel_Name"][:40]) for r in rows[idx:] if "k_solver" in r["Kernel_Name"]]
tot = sum(d for d, _, _ in lv)
print("levels", len(lv), "total %.3f ms" % (tot / 1e6))
for i, (d, g, nme) in sorted(enumerate(lv), key=lambda kv: -kv[1][0])[:25]:
    print("level %3d  %8.1f us  grid %9d  %s" % (i, d / 1e3, g, nme))
import collections
h = collections.Counter(min(int(d / 1e3) // 50 * 50, 1000) for d, _, _ in lv)
print(sorted(h.items()))
