"""Summarise a rocprofv3 kernel_trace.csv: per-kernel totals for the LAST batch."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_assign" in r["Kernel_Name"])
rows = rows[idx:]
tot = collections.OrderedDict()
def short(n):
    m = re.search(r"(k_\w+)(<[^>]*>)?", n)
    if not m: return n[:40]
    t = m.group(1)
    if "Fp2" in n.split("(")[0]: t += "<Fp2>"
    elif "FpParams" in n.split("(")[0]: t += "<Fp>"
    return t
for r in rows:
    name = short(r["Kernel_Name"])
    if name.startswith("k_msm<") : name += " grid=%sx%s" % (int(r["Grid_Size_X"])//64, r["Grid_Size_Y"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    t = tot.setdefault(name, [0, 0]); t[0] += 1; t[1] += d
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-42s calls=%4d total=%9.3f ms (%4.1f%%) avg=%9.1f us" % (k, c, d / 1e6, 100.0 * d / span, d / c / 1e3))
print("batch span %.3f ms" % (span / 1e6))
