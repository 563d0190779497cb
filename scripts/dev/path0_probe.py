"""which geometries still run the round-1 kernel pair (ra_search_path == 0): one line per (box, radius range, reference class)"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cryo_ralib_amd import api
rows = []
for nx in (100, 112, 128, 160):
    for ou in range(21, 41):
        for xr in (0, 1, 3, 5):
            if ou + xr > (nx - 1) // 2: continue
            for nref in (17, 50, 100):
                e = api.Engine(nx, ou, xr, xr, 1.0, nref, api.RA_MODE_MREF)
                if e.search_path == 0: rows.append((nx, ou, xr, nref))
                e.close()
by = collections.defaultdict(list)
for nx, ou, xr, nref in rows: by[(nx, xr, nref)].append(ou)
for k, v in sorted(by.items()): print(k, "ou", min(v), "..", max(v), len(v))
