#!/usr/bin/env python3
"""Per-kernel SQ counter summary from two rocprofv3 --pmc passes (sqa_<wl>, sqb_<wl>): what bounds each kernel.
usage: sq_summary.py <dir> <wl> [out.json].  Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves
(MI355X_MICROARCH.md); SQ_BUSY_CU_CYCLES, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT are cycles summed over CUs."""
import csv, glob, json, re, sys
from collections import defaultdict

d, wl = sys.argv[1], sys.argv[2]


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)", name)
    return m.group(1) if m else name


acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(int)
for tag in ("sqa", "sqb"):
    fs = sorted(glob.glob("%s/**/%s_%s_counter_collection.csv" % (d, tag, wl), recursive=True))
    if not fs:
        print("missing", tag); continue
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_") or k.startswith("k_synth"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if tag == "sqa" and r["Counter_Name"] == "SQ_WAVE_CYCLES":
            launches[k] += 1
print("# %s: SQ counters per kernel, summed over the launches of one bench pass (rocprofv3 --pmc, two passes of 8 counters)" % wl)
print("# wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave parked at s_waitcnt / barrier); issue-stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES;")
print("# active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; LDS busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES; conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;")
print("# VALU busy = SQ_ACTIVE_INST_VALU * 4 / (SQ_BUSY_CU_CYCLES * 4 SIMDs)")
print("%-18s %8s %6s %11s %7s %9s %10s %10s %10s %10s %10s" % ("kernel", "launches", "wait", "issue-stall", "active", "LDS busy", "conflicts", "VALU busy", "VALU inst", "SALU inst", "LDS inst"))
for k in sorted(acc, key=lambda x: -acc[x].get("SQ_WAVE_CYCLES", 0)):
    a = acc[k]
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    busy = a.get("SQ_BUSY_CU_CYCLES", 0) or 1
    lds = a.get("SQ_LDS_IDX_ACTIVE", 0)
    print("%-18s %8d %5.0f%% %10.0f%% %6.0f%% %8.0f%% %9.0f%% %9.0f%% %10.3g %10.3g %10.3g" % (
        k, launches[k], 100 * a.get("SQ_WAIT_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * a.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        100 * lds / busy, 100 * a.get("SQ_LDS_BANK_CONFLICT", 0) / (lds or 1), 100 * a.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (busy * 4),
        a.get("SQ_INSTS_VALU", 0), a.get("SQ_INSTS_SALU", 0), a.get("SQ_INSTS_LDS", 0)))

if len(sys.argv) > 3:
    # machine-readable form: bench.py turns the dominant kernel's entry into roofline.issue.  SQ_WAVE_CYCLES and the SQ_WAIT / SQ_ACTIVE counters are
    # quad-cycles summed over waves; cycles_per_inst = wave cycles (x4) over the wave-instructions of every class the wave executed
    out = {"workload": wl, "source": "rocprofv3 --pmc, two passes of 8 SQ counters (scripts/r06_profiles.sh)", "kernels": {}}
    for k, a in acc.items():
        wc = a.get("SQ_WAVE_CYCLES", 0) or 1
        busy = a.get("SQ_BUSY_CU_CYCLES", 0) or 1
        ninst = a.get("SQ_INSTS_VALU", 0) + a.get("SQ_INSTS_SALU", 0) + a.get("SQ_INSTS_LDS", 0) + a.get("SQ_INSTS_VMEM_RD", 0) + a.get("SQ_INSTS_VMEM_WR", 0)
        out["kernels"][k] = {
            "launches": launches[k], "valu_busy": a.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (busy * 4), "wait": a.get("SQ_WAIT_ANY", 0) / wc,
            "issue_stall": a.get("SQ_WAIT_INST_ANY", 0) / wc, "active": a.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            "lds_busy": a.get("SQ_LDS_IDX_ACTIVE", 0) / busy, "lds_conflict": a.get("SQ_LDS_BANK_CONFLICT", 0) / (a.get("SQ_LDS_IDX_ACTIVE", 0) or 1),
            "wave_inst_valu": a.get("SQ_INSTS_VALU", 0), "wave_inst_salu": a.get("SQ_INSTS_SALU", 0), "wave_inst_lds": a.get("SQ_INSTS_LDS", 0),
            "wave_inst_vmem": a.get("SQ_INSTS_VMEM_RD", 0) + a.get("SQ_INSTS_VMEM_WR", 0), "waves": a.get("SQ_WAVES", 0),
            "cycles_per_inst": 4.0 * wc / ninst if ninst else None,
        }
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
