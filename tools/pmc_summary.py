#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output: per (kernel, grid size) the mean of every counter over the dispatches, plus
the derived figures used in DESIGN.md.  python tools/pmc_summary.py <dir> [<dir> ...] [--match k_normal_sample] [--out f.json]

SQ_* cycle counters (SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*) count quad-cycles (x4 = shader
cycles, MI355X_MICROARCH.md); FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B... rocprofv3 reports FETCH_SIZE in
KB (x1024 B) and, on gfx950, half the bytes of a wide coalesced read stream (x2 correction for reads)."""
import argparse
import csv
import glob
import json
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--match", default="")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    acc = {}
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row.get("Kernel_Name", "")
                    if a.match and a.match not in name:
                        continue
                    key = (name[:120], int(row.get("Grid_Size", 0)), int(row.get("Workgroup_Size", 0)))
                    c = acc.setdefault(key, {})
                    v = c.setdefault(row["Counter_Name"], [])
                    v.append(float(row["Counter_Value"]))
    out = []
    for (name, grid, wg), c in sorted(acc.items()):
        rec = {"kernel": name, "grid_size": grid, "workgroup_size": wg,
               "counters": {k: sum(v) / len(v) for k, v in sorted(c.items())}, "dispatches": max(len(v) for v in c.values())}
        m = rec["counters"]
        d = {}
        if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"]:
            # fraction of the time the SQs were busy in which a SIMD was issuing a VALU instruction, chip average:
            # ACTIVE_INST_VALU is summed over the SIMDs (4 per CU), BUSY_CYCLES over the SQs (1 per CU)
            d["valu_issue_fraction_of_busy"] = m["SQ_ACTIVE_INST_VALU"] / (4.0 * m["SQ_BUSY_CYCLES"])
        if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m and m["SQ_WAVES"]:
            d["valu_instructions_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
        if "SQ_ACTIVE_INST_VALU" in m and "SQ_INSTS_VALU" in m and m["SQ_INSTS_VALU"]:
            d["shader_cycles_per_valu_instruction"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"]
        if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if k in m:
                    d[k.lower() + "_share_of_wave_cycles"] = m[k] / m["SQ_WAVE_CYCLES"]
        rec["derived"] = d
        out.append(rec)
        print(json.dumps(rec))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
