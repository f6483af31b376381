"""Builds profiles/rNN_msm_z_pmc.json (what bench.py replays as roofline.traffic) from three rocprofv3 --pmc passes of
`bench.py --steps 1 --warmup 1 --no-cpu-baseline`: FETCH_SIZE, WRITE_SIZE, TCC_HIT/MISS/REQ/EA0_RDREQ.
usage: make_msm_z_pmc.py <fetch.csv> <write.csv> <tcc.csv> <out.json> [window_z] [batch] [nbases] [quotient form as gsc_describe prints it, e.g. evaluation-form+digits]
(nbases: 32767 for the coefficient-form quotient, 32768 for the evaluation form, whose batches also launch the same kernel template with narrow
digits for the wide rows of c: only the Z launch — the wide-digit instantiation at c = 17 — is counted)"""
import csv, json, math, sys


C_ARG = int(sys.argv[5]) if len(sys.argv) > 5 else 16


def per_launch(path, want):
    acc = {}; ids = set(); ms = 0.0
    for r in csv.DictReader(open(path)):
        if "k_msm_win" not in r["Kernel_Name"] or "Fp29f" not in r["Kernel_Name"]:
            continue
        if C_ARG > 16 and "true>" not in r["Kernel_Name"]:
            continue
        if r["Dispatch_Id"] not in ids:
            ids.add(r["Dispatch_Id"]); ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    n = max(1, len(ids))
    return {k: acc[k] / n for k in want if k in acc}, ms / n, n


fetch, ms_f, n_f = per_launch(sys.argv[1], ["FETCH_SIZE"])
write, ms_w, n_w = per_launch(sys.argv[2], ["WRITE_SIZE"])
tcc, ms_t, n_t = per_launch(sys.argv[3], ["TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum"])
c = int(sys.argv[5]) if len(sys.argv) > 5 else 16
batch = int(sys.argv[6]) if len(sys.argv) > 6 else 8192
nbases, nwin = int(sys.argv[7]) if len(sys.argv) > 7 else 32767, {17: 15, 16: 16, 15: 17, 14: 19, 13: 20}.get(c, math.ceil(254 / c))
E = 1 << (c - 1); g = 384 * 64.0                       # entries per row; lanes of one XCD's 384 resident waves gathering from the same row
model = 1.0 - (E / g) * (1.0 - math.exp(-g / E))
out = {
    "kernel": "k_msm_win<Fp29f> over the Z rows (one row of 2^%d multiples per base, window-parallel accumulators, one XCD per slice)" % (c - 1),
    # bench.py replays this file only for a run whose engine description matches every key here (quotient: the text after "quotient=" in gsc_describe)
    "config": {"kernel": "chacha20", "batch": batch, "window_z": c, "nbases": nbases, "quotient": sys.argv[8] if len(sys.argv) > 8 else "evaluation-form+digits" if nbases == 32768 else "coefficient-form",
               "grid_waves": (nbases + 255) // 256 * nwin * (batch // 64)},
    "FETCH_SIZE_KB_per_launch": fetch["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": write["WRITE_SIZE"],
    "hbm_bytes_per_launch": int((fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024),
    "algorithmic_bytes_per_launch": batch * nbases * 96,
    "l2": {"TCC_REQ": tcc.get("TCC_REQ_sum"), "TCC_HIT": tcc.get("TCC_HIT_sum"), "TCC_MISS": tcc.get("TCC_MISS_sum"),
           "hit_rate": round(tcc["TCC_HIT_sum"] / tcc["TCC_REQ_sum"], 4), "TCC_EA0_RDREQ": tcc.get("TCC_EA0_RDREQ_sum"),
           "hit_rate_model": round(model, 4)},
    "launch_ms_under_pmc": round((ms_f + ms_w + ms_t) / 3, 1), "launches_per_pass": n_f,
    "notes": "separate rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 1 --no-cpu-baseline`; values are per launch.  FETCH_SIZE is used "
             "uncorrected: the guide's x2 correction is for wide coalesced streams; for this kernel's pattern (random 64-byte gathers) the counter was "
             "calibrated on a known byte count with tools/ubench_gather.hip in round 2 (factor 0.9999: FETCH_SIZE = TCC_EA0_RDREQ x 64 B).  "
             "hit_rate_model: the L2 hit rate this access pattern CAN have — every resident wave of an XCD (32 CUs x 4 SIMDs x 3 waves x 64 lanes = g = 24 576 "
             "lanes) gathers a uniformly random one of the E = 2^(c-1) entries of the row the XCD is on; distinct entries touched = E (1 - exp(-g / E)), so "
             "hit = 1 - (E / g)(1 - exp(-g / E)): 0.17 at c = 17, 0.30 at c = 16, 0.83 at c = 13 — a ceiling (lanes perfectly in step on one row); measured at c = 16: 0.26 (round 2) and 0.21 (round 3), "
             "at c = 13: 0.74.  The other waves of a slice (1 920 windows x proof groups at c = 17, 384 resident) come by "
             "after the row has left the 4 MiB L2 (a slice streams 256 - 512 MiB of rows through it), so keeping the resident waves in step — they already are, by "
             "construction: same start, same work — cannot raise it; only more lanes per row visit (registers) or shorter rows (more windows, more additions) can.",
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out["l2"]), out["hbm_bytes_per_launch"], out["launch_ms_under_pmc"])
