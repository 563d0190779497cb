"""ra_create over a grid of geometries (no search): every combination must either create an engine or be refused as a geometry error
(particle crosses the image boundary) -- never RA_ERR_STATE / RA_ERR_HIP from a plan that was promised and does not fit
(tcrop_wanted against build_device_geometry).  Prints the failures and the count per search family."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cryo_ralib_amd import api  # noqa: E402

fam = collections.Counter()
bad = []
n = 0
for nx in (64, 90, 96, 100, 112, 128, 140, 160, 192, 200, 256):
    for ou in list(range(8, 41)) + [44, 52, 56, 60, 61, 64, 70, 80, 100, 120]:
        for xr in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12):
            if ou + xr > (nx - 1) // 2:
                continue
            for nref in (1, 2, 10, 14, 15, 16, 17, 50, 100):
                for ts in (1.0, 0.5):
                    if ts == 0.5 and xr > 3:
                        continue
                    mode = api.RA_MODE_REFFREE if nref == 1 else api.RA_MODE_MREF
                    n += 1
                    try:
                        e = api.Engine(nx, ou, xr, xr, ts, nref, mode)
                        fam[(e.search_path, int(e.search_tiled), e.search_offsets_per_pass, int(e.search_skips_offsets))] += 1
                        e.close()
                    except api.EngineError as ex:
                        bad.append((nx, ou, xr, nref, ts, str(ex)[:120]))
print("%d geometries, %d failures" % (n, len(bad)))
for b in bad[:40]:
    print("FAIL", b)
for k, v in sorted(fam.items()):
    print("family (path, tiled, offsets per pass, skips offsets) %s: %d" % (k, v))
