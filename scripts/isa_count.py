"""developer aid: instruction-class counts per kernel from `hipcc -S --cuda-device-only` output"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
cur = None; cnt = {}
for ln in lines:
    m = re.match(r'(_Z\w+):', ln)
    if m:
        cur = m.group(1); cnt[cur] = collections.Counter(); continue
    if cur is None: continue
    if ln.startswith('.Lfunc_end'): cur = None; continue
    mm = re.match(r'\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+|flat_\w+)', ln)
    if not mm: continue
    op = mm.group(1); c = cnt[cur]
    if op.startswith('v_pk'): c['v_pk'] += 1
    elif op.startswith('v_mfma'): c['mfma'] += 1
    elif op.startswith('v_'): c['valu'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith('scratch'): c['scratch'] += 1
    else: c['vmem'] += 1
    if 'f64' in op: c['f64'] += 1
for k, c in cnt.items():
    if pat in k: print(k[:70], dict(c))
