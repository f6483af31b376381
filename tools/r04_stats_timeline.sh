#!/bin/bash
# rocprofv3 kernel summary + launch timeline of a bench line (round 4): W names the outputs, ARGS the bench.py arguments, WIN the timeline window in ms before the last kernel, MINMS the shortest launch listed.  Output: gpurun_out/r04sa/
set -o pipefail
export PYTHONUNBUFFERED=1
W=${W:-aes128}; ARGS=${ARGS:---workload $W --steps 4 --warmup 1}; WIN=${WIN:-330}
O=gpurun_out/r04sa; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 bench.py $ARGS --no-cpu-baseline --verify 0 > $O/stats_bench_$W.json 2> $O/stats_$W.err && echo "stats ok"
T=$(find $O -name "*kernel_trace.csv" | head -1)
# per-kernel time over the LAST step's worth of launches is awkward to cut; instead: totals of the launches after the last table-building kernel
python3 - "$T" $WIN ${MINMS:-0.3} > $O/kernel_steps_$W.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_init = max((i for i, r in enumerate(rows) if "k_build_" in r["Kernel_Name"] or "k_qb_" in r["Kernel_Name"] or "k_shift_bases" in r["Kernel_Name"]), default=-1)
tot = collections.defaultdict(lambda: [0, 0])
for r in rows[last_init + 1:]:
    n = r["Kernel_Name"].replace("gsc::(anonymous namespace)::", "").replace("void ", "").replace("bn254::", "").split("(")[0]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot[n][0] += d; tot[n][1] += 1
all_ns = sum(v[0] for v in tot.values())
print("kernels after init: total %.1f ms" % (all_ns / 1e6))
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:40]: print("%9.2f ms %6d  %5.1f%%  %s" % (d / 1e6, c, 100.0 * d / all_ns, n[:110]))
# every launch over 0.3 ms of the last 330 ms (about one step), in start order, with its queue (lane) and grid
t_end = max(int(r["End_Timestamp"]) for r in rows)
print("launches over 0.3 ms in the window:")
for r in rows:
    s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e0 < t_end - float(sys.argv[2]) * 1e6 or e0 - s0 < float(sys.argv[3]) * 1e6: continue
    n = r["Kernel_Name"].replace("gsc::(anonymous namespace)::", "").replace("void ", "").replace("bn254::", "").split("(")[0]
    print("  t=%8.2f  %8.2f ms  q%s  grid %sx%sx%s wg %s  %s" % ((s0 - t_end) / 1e6, (e0 - s0) / 1e6, r.get("Queue_Id", "?"), r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"), r.get("Workgroup_Size_X", "?"), n[:60]))
PY
rm -f $T
cat $O/kernel_steps_$W.txt | head -140
cut -c1-200 $O/stats_bench_$W.json
