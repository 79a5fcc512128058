#!/usr/bin/env python3
"""Run bench.py once per BASELINE configuration (sequentially, one process at a
time) and print the markdown table kept in profiles/rNN_configs.md.

    python3 tools/run_configs.py > gpurun_out/configs.md
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ROWS = [
    ("metric grid, the reference's V(10,10) (bench default)", "--level 13 --mu1 10 --mu2 10"),
    ("config 2: 4096², 6-level V(2,1) Jacobi", "--level 12 --coarsest 7 --mu1 2 --mu2 1"),
    ("config 2 grid with the reference's V(10,10)", "--level 12 --coarsest 7 --mu1 10 --mu2 10"),
    ("config 3: 8192² red-black GS V(2,1)", "--level 13 --smoother rbgs --mu1 2 --mu2 1"),
    ("config 3 with V(2,2)", "--level 13 --smoother rbgs --mu1 2 --mu2 2"),
    ("8192² Jacobi V(2,1)", "--level 13 --mu1 2 --mu2 1"),
    ("config 5: 8192² mixed fp32/fp64 (FMG column), V(2,1)", "--level 13 --dtype mixed --mu1 2 --mu2 1"),
    ("config 5 with V(10,10)", "--level 13 --dtype mixed --mu1 10 --mu2 10"),
    ("8192² pure fp32 (stalls at the float floor, D11)", "--level 13 --dtype f32 --mu1 10 --mu2 10"),
    ("config 4 grid (16384²) on ONE GPU, V(10,10)", "--level 14 --mu1 10 --mu2 10"),
    ("16384² red-black GS V(2,1) on one GPU", "--level 14 --smoother rbgs --mu1 2 --mu2 1"),
    ("config 1: 256², 3-level V(10,10) (the CPU-reference case, here on the GPU)", "--level 8 --coarsest 6 --mu1 10 --mu2 10"),
    ("the reference's own hierarchy: levels 10..7 (PS:17-18), V(10,10)", "--level 10 --coarsest 7 --mu1 10 --mu2 10 --steps 50 --warmup 5"),
]


def main():
    print("| BASELINE config | flags | ms/cycle | whole-job G upd/s | cycles to 1e-8 | s to 1e-8 | reference problem (FMG): "
          "cycles, s | smoother G upd/s | 1-sweep frac |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, flags in ROWS:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + flags.split()
        if "--steps" not in flags:
            cmd += ["--steps", "10", "--warmup", "2"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
        try:
            d = json.loads(line)
        except ValueError:
            print(f"| {name} | `{flags}` | FAILED rc={out.returncode} | | | | | | |", flush=True)
            sys.stderr.write(out.stderr[-2000:])
            continue
        fmg = d.get("reference_problem_fmg") or {}
        r1 = d.get("roofline_single_sweep") or {}
        print(f"| {name} | `{flags}` | {d['ms_per_step']:.3f} | {d['value'] / 1e9:.1f} | {d.get('vcycles_to_1e-8')} | "
              f"{d.get('seconds_to_1e-8', float('nan')):.4f} | {fmg.get('cycles_to_1e-8')}, {fmg.get('seconds_to_1e-8', float('nan')):.4f} | "
              f"{d['roofline']['smoother_updates_per_s'] / 1e9:.0f} | {r1.get('frac', float('nan')):.3f} |", flush=True)


if __name__ == "__main__":
    main()
