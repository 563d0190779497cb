"""Collects the ctypes binding statements of the reference's drivers into a data fixture.

Run in the build container (needs /root/reference):  python tests/golden/make_binding_fixture.py
Writes tests/golden/binding_symbols.json: for every driver, the list of (symbol, restype expression)
pairs it sets at import time and the list of library symbols it calls.  Only the names travel, no source text.
"""
import json
import os
import re

REF = "/root/reference"
DRIVERS = ["test_mref_gpu_align.py", "test_reffree_gpu_align.py", "test_reffree.py", "test_mref_cheng_yu_bdb_cuda.py"]
HERE = os.path.dirname(os.path.abspath(__file__))

out = {}
for d in DRIVERS:
    restypes, calls = [], set()
    for no, line in enumerate(open(os.path.join(REF, d), encoding="utf-8", errors="replace"), 1):
        code = line.split("#", 1)[0]
        m = re.match(r"\s*cu_module\.(\w+)\.restype\s*=\s*(.+?)\s*$", code)
        if m:
            restypes.append({"symbol": m.group(1), "restype": m.group(2), "line": no})
            continue
        for m in re.finditer(r"cu_module\.(\w+)\s*\(", code):
            calls.add(m.group(1))
    out[d] = {"restype_statements": restypes, "called": sorted(calls)}
with open(os.path.join(HERE, "binding_symbols.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
