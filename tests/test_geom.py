"""Tile map of the deep folded passes (csrc/mgx_geom.hpp) checked on the CPU: the same two functions the launcher and the
kernel use (cycle_geom_pick, cycle_tile_at), compiled with g++ into tests/geom_check.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multigrid_nikhil_c-_amd", "csrc")


@pytest.fixture(scope="module")
def geom_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("geom") / "geom_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", CSRC, "-o", exe, os.path.join(ROOT, "tests", "geom_check.cpp")], check=True)
    return exe


@pytest.mark.parametrize("env", [{}, {"PAIR": "0"}, {"RATIO": "125"}, {"RATIO": "200"}, {"FUZZ": "3000"}, {"FUZZ": "3000", "PAIRMIN": "16", "MINCHUNK": "8"},
                                 {"FUZZ": "2000", "PAIRMIN": "24", "RATIO": "170", "EDGEPCT": "0", "LASTPCT": "60"}, {"FUZZ": "2000", "PAIRMIN": "40", "EDGEPCT": "45", "LASTPCT": "5"}])
def test_every_row_of_every_strip_is_covered_exactly_once(geom_check, env):
    r = subprocess.run([geom_check], env={**os.environ, **env}, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert r.stdout.strip().endswith("cases)") and "ok" in r.stdout.splitlines()[-1]


def test_the_bench_grid_runs_in_one_round_of_paired_chunks(geom_check):
    out = subprocess.run([geom_check, "v"], capture_output=True, text=True, check=True).stdout
    line = next(l for l in out.splitlines() if l.startswith("N= 8192 rows     1.. 8192 K=10 POST=1"))
    assert " tall " in line and "tall  0" not in line and "(1.00 rounds)" in line or "(0.9" in line, line
