#!/bin/bash
echo "== final A/B of the priority alternation (new = on, other = off)"
bash tools/gpu_ab.sh tools/ab/libmgx_noprio.so 13 12 14
echo "== shortest chunk of the deep passes (MGX_MIN_CHUNK) in the 8-slab budget and on one GPU"
for c in 16 12 8; do echo "min chunk $c"; MGX_MIN_CHUNK=$c python tools/slab_budget.py 14 fma 2>&1 | grep "P=8"; done
python - <<'PY'
import subprocess, json, os
for c in ("16", "12", "8"):
    env = dict(os.environ, MGX_MIN_CHUNK=c, MGX_TILE_MAX_N="0")
    for L in ("11", "10"):
        out = subprocess.run(["python", "bench.py", "--no-cpu-baseline", "--level", L, "--steps", "50", "--warmup", "5"], env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print("min chunk", c, "L" + L, "marching only:", round(d["ms_per_step"], 4))
PY
