"""How much do the kernels of concurrent calls overlap?  Reads a rocprofv3 kernel_trace.csv and, for the second half of the run
(steady state), prints: wall span, summed kernel time, share of the span with >= 1 / 2 / 3 kernels running, busy share per HIP queue,
and the kernels with the largest summed duration.  usage: trace_concurrency.py <kernel_trace.csv>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
mid = t0 + (t1 - t0) // 2
rows = [r for r in rows if int(r["Start_Timestamp"]) >= mid]
t0 = int(rows[0]["Start_Timestamp"]); span = t1 - t0
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
level = 0; last = t0; at = collections.Counter()
for t, d in ev:
    at[level] += t - last; last = t; level += d
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("span %.2f ms, %d kernels, summed kernel time %.2f ms (x%.2f)" % (span / 1e6, len(rows), tot / 1e6, tot / span))
cum = 0
for k in sorted(at, reverse=True):
    cum += at[k]
    if k in (0, 1, 2, 3, 4): print("  >= %d kernels running: %5.1f %% of the span" % (k, 100.0 * cum / span))
q = collections.Counter()
for r in rows: q[r.get("Queue_Id", "?")] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("  busy share per queue:", {k: "%.0f%%" % (100.0 * v / span) for k, v in sorted(q.items())})
def short(n):
    m = re.search(r"(k_\w+)", n); return m.group(1) if m else n[:32]
kt = collections.Counter(); kc = collections.Counter()
for r in rows: kt[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); kc[short(r["Kernel_Name"])] += 1
for k, v in kt.most_common(12): print("  %-28s calls %6d  total %8.2f ms  avg %8.1f us" % (k, kc[k], v / 1e6, v / kc[k] / 1e3))
