import sys, os
sys.path.insert(0, os.getcwd())
from cryo_ralib_amd import api
for nx, ou, xr, nref in [(140,58,2,3),(140,59,2,3),(140,60,2,3),(140,61,2,3),(140,62,2,3),(140,61,3,3),(140,60,3,3),(132,60,3,10),(128,60,3,10),(140,62,1,3),(140,62,0,3),(160,64,2,3),(160,65,2,3)]:
    try:
        e = api.Engine(nx, ou, xr, xr, 1.0, nref)
        print(nx, ou, xr, nref, (e.search_path, int(e.search_tiled), e.search_offsets_per_pass), flush=True)
        e.close()
    except Exception as ex:
        print(nx, ou, xr, nref, "ERR", ex)
