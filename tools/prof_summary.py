#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries kept under profiles/.

  kernel trace : python tools/prof_summary.py kt  <kernel_trace.csv>  > profiles/rNN_kernel_trace_summary.md
  PMC pass     : python tools/prof_summary.py pmc <counter_collection.csv> [<more.csv> ...]
                 prints per (kernel, grid) average counter values and, when both
                 FETCH_SIZE and WRITE_SIZE are present, HBM bytes per launch with
                 the gfx950 correction of MI355X_MICROARCH.md §HBM:
                 bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
                 (FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane streaming
                 reads; WRITE_SIZE is exact for 16-B/lane streaming stores).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").replace("mgx::", "")
    return re.sub(r"\(.*$", "", name)


def clusters(values, ratio=1.6):
    """Two levels of one hierarchy can launch the same kernel with the same number of workgroups (8192^2 in one round of
    paired chunks and 4096^2 in one uniform round are both 512): split a group whose sorted values jump by more than
    `ratio` into sub-groups, largest first.  Returns a list of index lists."""
    order = sorted(range(len(values)), key=lambda i: -values[i])
    out, cur = [], [order[0]]
    for a, b in zip(order, order[1:]):
        if values[b] > 0 and values[a] / values[b] > ratio or (values[b] <= 0 < values[a]):
            out.append(cur)
            cur = []
        cur.append(b)
    out.append(cur)
    return out


def kt(path):
    groups = defaultdict(list)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            key = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]) // max(int(row["Workgroup_Size_X"]), 1),
                   int(row["VGPR_Count"]), int(row["SGPR_Count"]), int(row["LDS_Block_Size"]), int(row["Scratch_Size"]))
            groups[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    total = sum(sum(v) for v in groups.values())
    print("| kernel | workgroups | calls | avg us | min us | max us | total ms | % | VGPR | SGPR | LDS | scratch |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    split = {}
    for key, v in groups.items():
        cl = clusters(v)
        for n, idx in enumerate(cl):
            split[key[:2] + ((f" ({'abcdefgh'[n]})" if len(cl) > 1 else ""),) + key[2:]] = [v[i] for i in idx]
    for key, v in sorted(split.items(), key=lambda kv: -sum(kv[1])):
        name, wgs, tag, vg, sg, lds, scr = key
        wgs = f"{wgs}{tag}"
        print(f"| {name} | {wgs} | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} | "
              f"{sum(v) / 1e6:.3f} | {100.0 * sum(v) / total:.2f} | {vg} | {sg} | {lds} | {scr} |")


def pmc(paths):
    vals = defaultdict(lambda: defaultdict(list))
    for path in paths:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                wg = max(int(row.get("Workgroup_Size", row.get("Workgroup_Size_X", 1)) or 1), 1)
                grid = int(row.get("Grid_Size", row.get("Grid_Size_X", 0)) or 0)
                key = (short(row["Kernel_Name"]), grid // wg)
                vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    print("| kernel | workgroups | launches | counter | avg per launch |")
    print("|---|---|---|---|---|")
    split = {}
    for key, ctrs in vals.items():
        # (sub-groups by the first counter's values: launches of one group come in the same order in every counter)
        first = sorted(ctrs)[0]
        cl = clusters(ctrs[first])
        for n, idx in enumerate(cl):
            tag = f" ({'abcdefgh'[n]})" if len(cl) > 1 else ""
            split[(key[0], f"{key[1]}{tag}")] = {c: [v[i] for i in idx if i < len(v)] for c, v in ctrs.items()}
    for key, ctrs in sorted(split.items(), key=lambda kv: -max(sum(x) for x in kv[1].values())):
        rec = {}
        for c, v in sorted(ctrs.items()):
            rec[c] = sum(v) / len(v)
            print(f"| {key[0]} | {key[1]} | {len(v)} | {c} | {rec[c]:.1f} |")
        if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
            rec["hbm_bytes_per_launch"] = (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0
            rec["hbm_read_bytes_per_launch"] = 2.0 * rec["FETCH_SIZE"] * 1024.0
            rec["hbm_write_bytes_per_launch"] = rec["WRITE_SIZE"] * 1024.0
            print(f"| {key[0]} | {key[1]} | | HBM bytes (2*FETCH+WRITE)*1024 | {rec['hbm_bytes_per_launch']:.0f} |")
        out[f"{key[0]}@{key[1]}wg"] = rec
    print()
    print("```json")
    print(json.dumps(out, indent=1))
    print("```")


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("kt", "pmc"):
        print(__doc__)
        sys.exit(1)
    if sys.argv[1] == "kt":
        kt(sys.argv[2])
    else:
        pmc(sys.argv[2:])
