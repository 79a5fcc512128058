#!/usr/bin/env python3
"""Condense rocprofv3 SQ / GRBM / TCC counter passes (counter_collection.csv) into a per-kernel table
with the derived figures DESIGN.md quotes:

  clk_GHz        GRBM_GUI_ACTIVE / 8 XCDs / kernel time            (effective shader clock)
  valu_busy      4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * cycles)    (SQ_* count quad-cycles, MI355X_MICROARCH.md)
  waves_per_simd 4 * SQ_WAVE_CYCLES / (1024 * cycles)               (average resident waves per SIMD)
  wait_frac      SQ_WAIT_ANY / SQ_WAVE_CYCLES                       (share of wave time parked on s_waitcnt)
  issue_stall    SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  valu_per_wave  SQ_INSTS_VALU / SQ_WAVES

usage: python tools/sq_summary.py <kernel_trace.csv> <counter_collection.csv> [more counter csvs ...]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").replace("mgx::", "")
    return re.sub(r"\(.*$", "", name)


def clusters(values, ratio=1.6):
    # (as in tools/prof_summary.py: launches of one kernel with one workgroup count but different grids are told apart
    # by a jump of more than `ratio` in their sorted durations)
    order = sorted(range(len(values)), key=lambda i: -values[i])
    out, cur = [], [order[0]]
    for a, b in zip(order, order[1:]):
        if values[b] > 0 and values[a] / values[b] > ratio:
            out.append(cur)
            cur = []
        cur.append(b)
    out.append(cur)
    return out


def main():
    kt, ctr_paths = sys.argv[1], sys.argv[2:]
    dur = defaultdict(list)
    regs = {}
    with open(kt) as fh:
        for row in sorted(csv.DictReader(fh), key=lambda r: int(r["Dispatch_Id"])):
            key = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]) // max(int(row["Workgroup_Size_X"]), 1))
            dur[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            regs[key] = (int(row["VGPR_Count"]), int(row.get("Accum_VGPR_Count", 0) or 0), int(row["LDS_Block_Size"]))
    vals = defaultdict(lambda: defaultdict(list))
    for path in ctr_paths:
        with open(path) as fh:
            for row in sorted(csv.DictReader(fh), key=lambda r: int(r["Dispatch_Id"])):
                wg = max(int(row.get("Workgroup_Size", 1) or 1), 1)
                key = (short(row["Kernel_Name"]), int(row.get("Grid_Size", 0) or 0) // wg)
                vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("| kernel | workgroups | us (traced) | clk GHz | VGPR (rocprof x2 = arch) | LDS | waves/SIMD avg | VALU busy | wait (s_waitcnt) | issue stall | VALU instr / wave | L2 hit |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    # sub-groups by duration; every pass runs the same launch sequence, so launch i of a group is the same launch in
    # the trace and in every counter list (lists of another length are left whole)
    rows = []
    for key in vals:
        if key not in dur:
            continue
        cl = clusters(dur[key])
        for n, idx in enumerate(cl):
            tag = f" ({'abcdefgh'[n]})" if len(cl) > 1 else ""
            sub = {k: ([v[i] for i in idx] if len(v) == len(dur[key]) else v) for k, v in vals[key].items()}
            rows.append(((key[0], f"{key[1]}{tag}"), key, [dur[key][i] for i in idx], sub))
    for (name, wgs), key, d, sub in sorted(rows, key=lambda r: -sum(r[2])):
        c = {k: sum(v) / len(v) for k, v in sub.items()}
        if "SQ_WAVE_CYCLES" not in c:
            continue
        us = sum(d) / len(d) / 1e3
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        clk = cyc / (us * 1e3) if cyc else float("nan")
        if not cyc:
            cyc = us * 1e3 * 2.1
        simd_cyc = 1024.0 * cyc
        hit = c.get("TCC_HIT_sum", 0.0) / max(c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0), 1.0)
        vg, ag, lds = regs[key]
        print(f"| {name} | {wgs} | {us:.1f} | {clk:.2f} | {vg} (= {2 * vg}) | {lds} | {4 * c['SQ_WAVE_CYCLES'] / simd_cyc:.2f} | "
              f"{4 * c.get('SQ_ACTIVE_INST_VALU', 0) / simd_cyc:.2f} | {c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.2f} | "
              f"{c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.2f} | {c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_WAVES', 1), 1):.0f} | "
              f"{hit:.2f} |")


if __name__ == "__main__":
    main()
