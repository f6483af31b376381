"""VALU instructions per wave-addition of the Z-table kernel from a pmc_summary.py listing (SQ_INSTS_VALU of k_msm_win<Fp29f, true>).
usage: make_valu_per_add.py <pmc_sq_counters.txt> <out.json> [batch] [nbases] [windows]
wave-additions per launch = bases x windows x batch / 64 (every lane of a wave adds one table entry per base and window)."""
import json, re, sys
txt, out = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
nbases = int(sys.argv[4]) if len(sys.argv) > 4 else 32768
nwin = int(sys.argv[5]) if len(sys.argv) > 5 else 15
line = [l for l in open(txt) if l.startswith("k_msm_win<bn254::Fp29f, true>")][-1]
f = dict(kv.split("=") for kv in line.split()[2:] if "=" in kv)
calls = int(f["calls"]); valu = float(f["SQ_INSTS_VALU"]) / calls
wave_adds = nbases * nwin * batch // 64
json.dump({"kernel": "k_msm_win<Fp29f, true>", "config": {"batch": batch, "nbases": nbases, "windows": nwin},
           "SQ_INSTS_VALU_per_launch": valu, "wave_adds_per_launch": wave_adds, "instr_per_wave_add": round(valu / wave_adds, 2),
           "SQ_ACTIVE_INST_VALU_per_SQ_BUSY_CYCLES": round(float(f["SQ_ACTIVE_INST_VALU"]) / float(f["SQ_BUSY_CYCLES"]), 3),
           "source": txt, "launches": calls, "ms_under_pmc": round(float(f["ms"]) / calls, 3)}, open(out, "w"), indent=1)
print(open(out).read())
