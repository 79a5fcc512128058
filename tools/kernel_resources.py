#!/usr/bin/env python3
"""Tabulate the kernels' resource usage from the build's remark files
(multigrid_nikhil_c-_amd/csrc/build/*.remarks, written by every compile):
    python tools/kernel_resources.py [substring ...]
prints name, VGPRs, AGPRs, scratch bytes per lane, occupancy (waves per SIMD), LDS bytes per block."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels():
    out = []
    for path in sorted(glob.glob(os.path.join(ROOT, "multigrid_nikhil_c-_amd", "csrc", "build", "*.remarks"))):
        txt = open(path).read()
        for m in re.finditer(r"Function Name: (\S+)(.*?)LDS Size \[bytes/block\]: (\d+)", txt, re.S):
            body = m.group(2)
            g = lambda k: int(re.search(k + r": (\d+)", body).group(1))
            out.append(dict(name=m.group(1), vgprs=g(r"VGPRs"), agprs=g(r"AGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                            occupancy=g(r"Occupancy \[waves/SIMD\]"), lds=int(m.group(3)), file=os.path.basename(path)))
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        d = r.stdout.splitlines()
        return d if len(d) == len(names) else names
    except Exception:
        return names


if __name__ == "__main__":
    ks = kernels()
    for k, d in zip(ks, demangle([k["name"] for k in ks])):
        d = re.sub(r"^void mgx::|\(.*$", "", d)
        if all(s in d for s in sys.argv[1:]):
            print(f"{d:60s} vgpr {k['vgprs']:3d} agpr {k['agprs']:3d} scratch {k['scratch']:4d} occ {k['occupancy']} lds {k['lds']:6d}")
