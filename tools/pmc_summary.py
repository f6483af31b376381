"""Sum rocprofv3 --pmc counters per kernel (short name) over a counter_collection.csv; prints one line per kernel with every
counter found and the summed kernel time of the same dispatches.  usage: pmc_summary.py <csv> [name filter]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
def short(n):
    m = re.search(r"(k_\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]
acc = collections.OrderedDict(); seen = set()
for r in rows:
    k = short(r["Kernel_Name"])
    if flt and flt not in k:
        continue
    a = acc.setdefault(k, collections.OrderedDict(calls=0, ms=0.0))
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); a["calls"] += 1; a["ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, a in acc.items():
    print(k, " ".join("%s=%s" % (n, ("%.3f" % v if isinstance(v, float) and n == "ms" else "%.6g" % v if isinstance(v, float) else v)) for n, v in a.items()))
