"""ra_debug_spectra on an engine whose search runs the fused kernel over a crop (generic class): the polar stage bin for bin"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_solo import polar_stage_check
from cryo_ralib_amd import api
polar_stage_check(160, 34, 3, api.RA_MODE_MREF, n=3)
polar_stage_check(144, 30, 2, api.RA_MODE_REFFREE, n=3)
print("ok")
