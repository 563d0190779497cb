"""Short, seeded runs of the random sweeps of scripts/dev (profiles/r06_random_sweeps.txt) inside `pytest -m gpu`: random geometries --
box, rings, first ring, ring step, ranges, steps 0.25 .. 3, start states, engine options, --nomirror -- through the engine API, the
host drivers' iteration loops, the drop-in symbols of the reference's library and the class-resident search, every case against the
CPU checker with the literal bar.  Each sweep is a child process (the scripts are command-line tools; one at a time)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWEEPS = [
    ("random_sweep.py", ["48", "101", "small", "state+options", "wide"]),
    ("random_sweep.py", ["32", "102", "big", "state+options", "wide"]),
    ("random_sweep.py", ["10", "103", "huge", "state+options", "wide"]),
    ("random_loop_sweep.py", ["20", "104", "small"]),
    ("random_loop_sweep.py", ["14", "105", "big"]),
    ("random_legacy_sweep.py", ["24", "106", "small"]),
    ("random_legacy_sweep.py", ["14", "107", "big"]),
    ("random_classes_sweep.py", ["20", "108", "small"]),
]


@pytest.mark.parametrize("script,args", SWEEPS, ids=["%s-%s" % (s[:-3], "-".join(a[1:3])) for s, a in SWEEPS])
def test_seeded_random_sweep(script, args):
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dev", script)] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, text=True)
    tail = "\n".join(r.stdout.splitlines()[-25:])
    assert r.returncode == 0, tail
    assert "agree" in r.stdout.splitlines()[-1], tail
