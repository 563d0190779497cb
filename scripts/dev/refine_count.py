"""how many particles the exact re-evaluation (refine_winner_kernel) takes per step of a bench workload, and how long it runs
(run on the GPU box: python scripts/dev/refine_count.py [workload])"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "reffree"
sys.argv = [sys.argv[0], "--workload", w, "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-parity", "--no-pcie", "--no-others"]
from cryo_ralib_amd import api  # noqa: E402
orig = api.Engine.align
counts = []


def align(self, *a, **k):
    r = orig(self, *a, **k)
    self.sync()
    counts.append(self.last_refine_count())
    return r


api.Engine.align = align
bench.main()
print("refined per align call:", counts)
