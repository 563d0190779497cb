"""LDS bank model of the tap reads of a 256-sample ring job (search_fused_kernel, 8 lanes per ring, 8 rings per wave) for two image
layouts: row-major with stride pst (shipped) and ROWS INTERLEAVED IN PAIRS -- (I[2k][x], I[2k+1][x]) as one 8-byte word, VERDICT r05
item 2 -- where a sample whose upper row is even finds its four taps in two adjacent 8-byte words (one ds_read2_b64) and a sample whose
upper row is odd needs one dword out of four different words (two ds_read2_b32 with a dword stride of 2).  Cost model: an LDS
instruction is served in groups of 32 lanes; a group costs max over the 32 banks of the distinct dwords it touches there (64 banks
are not assumed); ds_read2_* = its two accesses one after the other.  Prints LDS cycles per sample pair-of-rows and the instruction count."""
import numpy as np
rng = np.random.default_rng(1)

def cost(addr_lists):
    """addr_lists: list of arrays [64] of dword addresses (one access each); returns bank cycles summed over accesses and half-waves"""
    tot = 0
    for addr in addr_lists:
        for grp in (slice(0, 32), slice(32, 64)):
            a = addr[grp]
            a = a[a >= 0]
            if a.size == 0:
                continue
            ad = np.unique(a)
            tot += np.bincount(ad % 32, minlength=32).max()
    return tot

def run(pst, layout, trials=200, n=256):
    cyc = 0; ins = 0; samples = 0
    for tr in range(trials):
        r0 = rng.integers(21, 34)
        cx = 46 + rng.integers(-3, 4) + rng.random() * 2 - 1; cy = 46 + rng.integers(-3, 4) + rng.random() * 2 - 1
        lanes = np.arange(64); sub = lanes // 8; t = lanes % 8
        for step in range(16):
            for u in range(2):
                j = 2 * (8 * step + t) + u; r = r0 + sub
                phi = 2 * np.pi * j / n
                x = cx + r * np.sin(phi); y = cy + r * np.cos(phi)
                ix = np.floor(x).astype(int); iy = np.floor(y).astype(int)
                samples += 1
                if layout == 'rowmajor':
                    a = iy * pst + ix
                    cyc += cost([a, a + 1, a + pst, a + pst + 1]); ins += 2
                else:
                    even = (iy & 1) == 0
                    w = lambda xx, yy: ((yy >> 1) * pst + xx) * 2 + (yy & 1)
                    # even rows: words w(ix, iy) .. +1 and w(ix + 1, iy) .. +1 : one ds_read2_b64 (4 dword accesses), lanes with odd rows masked
                    e = [np.where(even, w(ix, iy) + k, -1) for k in (0, 1)] + [np.where(even, w(ix + 1, iy) + k, -1) for k in (0, 1)]
                    # odd rows: two ds_read2_b32, lanes with even rows masked
                    o = [np.where(~even, w(ix, iy), -1), np.where(~even, w(ix + 1, iy), -1), np.where(~even, w(ix, iy + 1), -1), np.where(~even, w(ix + 1, iy + 1), -1)]
                    cyc += cost(e) + cost(o); ins += 3          # both paths are issued: the lanes of a wave differ in row parity
    return cyc / samples, ins / samples

for pst in (101, 103, 105):
    a = run(pst, 'rowmajor'); b = run(pst, 'pairs')
    print('row stride %3d: row-major %.2f bank cycles, %.1f LDS instructions per sample | rows in pairs %.2f bank cycles, %.1f instructions' % (pst, a[0], a[1], b[0], b[1]))
