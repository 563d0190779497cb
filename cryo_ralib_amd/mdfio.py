"""Dependency-free reader / writer for EMAN2 "MDF" HDF5 particle stacks.

The reference reads and writes its stacks through EMAN2 (`EMData.read_image(stack, i)`,
test_mref_gpu_align.py:1358-1375; `write_image("aqm%03d.hdf", j)`, :564).  EMAN2's HDF files are plain
HDF5 written with the library defaults:

    /MDF/images                 attribute imageid_max
    /MDF/images/<i>             attributes EMAN.nx, EMAN.ny, EMAN.nz, ... (the image header)
    /MDF/images/<i>/image       float32 [ny][nx] (or [nz][ny][nx]), contiguous layout

(cuda/EMAN2_test.ipynb cell 4: `f['MDF']['images'][str(i)]['image']`).  Neither libhdf5 nor h5py ship with the
GPU image, so this module implements the small part of the HDF5 file format those files use: superblock
version 0/1, symbol-table groups (version-1 B-trees, local heaps, symbol-table nodes), version-1 object headers
with continuation blocks, dataspace / datatype / contiguous-or-compact layout / attribute messages.  Chunked or
compressed datasets and the "latest format" structures (superblock 2/3, fractal heaps) are rejected with a clear
error.  The writer emits the same subset; `tests/test_host.py` checks both directions against files produced and
dumped by the real HDF5 library.
"""
import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5FormatError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------ reader

class _File:
    def __init__(self, buf):
        self.b = buf
        base = 0
        while base < len(buf) and buf[base:base + 8] != _SIG:      # the superblock may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
        if base >= len(buf):
            raise HDF5FormatError("not an HDF5 file")
        ver = buf[base + 8]
        if ver > 1:
            raise HDF5FormatError("HDF5 superblock version %d (latest-format file) is not supported; "
                                  "EMAN2 writes version 0" % ver)
        self.O, self.L = buf[base + 13], buf[base + 14]
        if self.O != 8 or self.L != 8:
            raise HDF5FormatError("only 8-byte offsets / lengths are supported")
        p = base + 24 + (4 if ver == 1 else 0)
        self.base_addr = struct.unpack_from("<Q", buf, p)[0]
        p += 32                                              # base, free-space info, end of file, driver info
        self.root = self._symbol_entry(p)

    # -- primitives
    def _u(self, p, n):
        return int.from_bytes(self.b[p:p + n], "little")

    def _symbol_entry(self, p):
        name_off, ohdr, cache = struct.unpack_from("<QQI", self.b, p)
        ent = {"name_off": name_off, "ohdr": ohdr + self.base_addr, "cache": cache}
        if cache == 1:
            bt, heap = struct.unpack_from("<QQ", self.b, p + 24)
            ent["btree"], ent["heap"] = bt + self.base_addr, heap + self.base_addr
        return ent

    def messages(self, addr):
        """[(type, flags, payload offset, size)] of a version-1 object header, following continuations"""
        b = self.b
        if b[addr] != 1:
            raise HDF5FormatError("object header version %d is not supported (latest-format file?)" % b[addr])
        nmsg, = struct.unpack_from("<H", b, addr + 2)
        hsize, = struct.unpack_from("<I", b, addr + 8)
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", b, p)
                body = p + 8
                if mtype == 0x0010:                          # continuation
                    off, ln = struct.unpack_from("<QQ", b, body)
                    blocks.append((off + self.base_addr, ln))
                out.append((mtype, flags, body, msize))
                p = body + msize
        return out

    def _heap_data(self, heap):
        if self.b[heap:heap + 4] != b"HEAP":
            raise HDF5FormatError("bad local heap signature")
        return struct.unpack_from("<Q", self.b, heap + 24)[0] + self.base_addr

    def _walk_btree(self, node, heap_data, out):
        b = self.b
        if b[node:node + 4] != b"TREE":
            raise HDF5FormatError("bad B-tree signature")
        ntype, level, nent = b[node + 4], b[node + 5], struct.unpack_from("<H", b, node + 6)[0]
        if ntype != 0:
            raise HDF5FormatError("unexpected B-tree node type %d" % ntype)
        p = node + 24                                        # key0
        for i in range(nent):
            child = struct.unpack_from("<Q", b, p + 8)[0] + self.base_addr
            if level > 0:
                self._walk_btree(child, heap_data, out)
            else:
                if b[child:child + 4] != b"SNOD":
                    raise HDF5FormatError("bad symbol-table node signature")
                nsym, = struct.unpack_from("<H", b, child + 6)
                for s in range(nsym):
                    ent = self._symbol_entry(child + 8 + 40 * s)
                    q = heap_data + ent["name_off"]
                    raw = bytes(b[q:q + 64])
                    z = raw.find(b"\0")
                    while z < 0:                              # names longer than the first window
                        more = bytes(b[q + len(raw):q + len(raw) + 256])
                        raw += more
                        z = raw.find(b"\0")
                        if not more:
                            raise HDF5FormatError("unterminated link name")
                    name = raw[:z].decode()
                    out[name] = ent
            p += 16

    def children(self, ent):
        """{name: symbol entry} of a group"""
        if "btree" not in ent:
            for mtype, _, body, _ in self.messages(ent["ohdr"]):
                if mtype == 0x0011:
                    bt, heap = struct.unpack_from("<QQ", self.b, body)
                    ent = dict(ent, btree=bt + self.base_addr, heap=heap + self.base_addr)
                    break
                if mtype in (0x0002, 0x0006):
                    raise HDF5FormatError("compact / dense link storage (latest-format group) is not supported")
            else:
                raise HDF5FormatError("object is not a group")
        out = {}
        self._walk_btree(ent["btree"], self._heap_data(ent["heap"]), out)
        return out

    # -- messages
    def _dataspace(self, body):
        b = self.b
        ver, rank, flags = b[body], b[body + 1], b[body + 2]
        p = body + (8 if ver == 1 else 4)
        return [struct.unpack_from("<Q", b, p + 8 * i)[0] for i in range(rank)]

    def _datatype(self, body):
        b = self.b
        cls, ver = b[body] & 15, b[body] >> 4
        bits0 = b[body + 1]
        size, = struct.unpack_from("<I", b, body + 4)
        order = ">" if bits0 & 1 else "<"
        if cls == 1:
            return np.dtype(order + "f%d" % size)
        if cls == 0:
            return np.dtype(order + ("i" if bits0 & 8 else "u") + "%d" % size)
        if cls == 3:
            return np.dtype("S%d" % size)
        raise HDF5FormatError("datatype class %d is not supported" % cls)

    def read_dataset(self, ent):
        b = self.b
        dims = dtype = data = None
        for mtype, _, body, msize in self.messages(ent["ohdr"]):
            if mtype == 0x0001:
                dims = self._dataspace(body)
            elif mtype == 0x0003:
                dtype = self._datatype(body)
            elif mtype == 0x000B:
                raise HDF5FormatError("filtered (compressed) datasets are not supported; rewrite the stack "
                                      "without compression (e2proc2d.py --compressbits=-1)")
            elif mtype == 0x0008:
                ver = b[body]
                if ver == 3:
                    cls = b[body + 1]
                    if cls == 1:
                        addr, size = struct.unpack_from("<QQ", b, body + 2)
                        data = (addr + self.base_addr, size) if addr != _UNDEF else (None, 0)
                    elif cls == 0:
                        size, = struct.unpack_from("<H", b, body + 2)
                        data = (body + 4, size)
                    else:
                        raise HDF5FormatError("chunked dataset layout is not supported")
                elif ver in (1, 2):
                    rank, cls = b[body + 1], b[body + 2]
                    if cls != 1:
                        raise HDF5FormatError("only contiguous version-1/2 layouts are supported")
                    addr, = struct.unpack_from("<Q", b, body + 8)
                    data = (addr + self.base_addr, None)
                else:
                    raise HDF5FormatError("data layout message version %d is not supported" % ver)
        if dims is None or dtype is None or data is None:
            raise HDF5FormatError("dataset lacks a dataspace / datatype / layout message")
        count = int(np.prod(dims)) if dims else 1
        if data[0] is None:
            return np.zeros(dims, dtype)
        return np.frombuffer(b, dtype, count, data[0]).reshape(dims)

    def attributes(self, ent, skipped=None):
        """{name: value} of the attribute messages of an object (numbers, numeric arrays and fixed-length strings); names of
        attributes with a datatype this reader does not decode (variable-length strings, compounds) go to `skipped`"""
        out = {}
        b = self.b
        for mtype, _, body, msize in self.messages(ent["ohdr"]):
            if mtype != 0x000C:
                continue
            ver = b[body]
            nsz, tsz, ssz = struct.unpack_from("<HHH", b, body + 2)
            pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
            p = body + 8 + (1 if ver == 3 else 0)
            name = bytes(b[p:p + nsz]).split(b"\0")[0].decode()
            p += pad(nsz)
            try:
                dtype = self._datatype(p)
            except HDF5FormatError:
                if skipped is not None:
                    skipped.append(name)
                continue
            q = p + pad(tsz)
            dims = self._dataspace(q)
            q += pad(ssz)
            count = int(np.prod(dims)) if dims else 1
            val = np.frombuffer(b, dtype, count, q)
            out[name] = val.reshape(dims) if dims else val[0]
        return out


def mdf_image_count(path):
    """number of images of an EMAN2 MDF stack without touching the pixel data"""
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    f = _File(memoryview(buf))
    imgs = f.children(f.children(f.children(f.root)["MDF"])["images"])
    return sum(1 for k in imgs if k.isdigit())


def read_mdf_stack(path, with_attrs=False, first=0, last=None, report=None):
    """float32 [n][ny][nx] (images in numerical order of their group names); optionally the per-image
    attribute dictionaries (EMAN.* header items).  first / last select images [first, last) -- the file is
    memory-mapped and only the selected datasets are read (each rank reads its own slice,
    test_mref_gpu_align.py:1358-1375)."""
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    f = _File(memoryview(buf))
    root = f.children(f.root)
    if "MDF" not in root:
        raise HDF5FormatError("%s: no /MDF group (not an EMAN2 HDF stack)" % path)
    mdf = f.children(root["MDF"])
    if "images" not in mdf:
        raise HDF5FormatError("%s: no /MDF/images group" % path)
    imgs = f.children(mdf["images"])
    ids = sorted(int(k) for k in imgs if k.isdigit())
    ids = ids[first:last]
    out, attrs = [], []
    for i in ids:
        g = f.children(imgs[str(i)])
        if "image" not in g:
            raise HDF5FormatError("%s: /MDF/images/%d has no 'image' dataset" % (path, i))
        px = f.read_dataset(g["image"])
        if report is not None and np.asarray(px).dtype != np.float32:
            report.setdefault("pixel_types", set()).add(str(np.asarray(px).dtype))
        out.append(np.asarray(px, np.float32))
        if with_attrs:
            sk = [] if report is not None else None
            attrs.append(f.attributes(imgs[str(i)], sk))
            if sk:
                report.setdefault("undecoded_attributes", set()).update(sk)
    arr = np.stack(out) if out else np.zeros((0, 0, 0), np.float32)
    if arr.ndim == 4 and arr.shape[1] == 1:
        arr = arr[:, 0]
    return (arr, attrs) if with_attrs else arr


# ------------------------------------------------------------------------------------------------ writer

class _Writer:
    """lays the file out in memory: every structure is appended 8-byte aligned and addressed absolutely"""

    LEAF_K = 4                # symbol-table nodes hold up to 2 * LEAF_K entries      (the library's defaults, so that
    INTERNAL_K = 16           # B-tree nodes hold up to 2 * INTERNAL_K children         EMAN2 can extend the file)

    def __init__(self):
        self.buf = bytearray()

    def alloc(self, data):
        while len(self.buf) % 8:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr

    def reserve(self, n):
        return self.alloc(bytes(n))

    @staticmethod
    def _msg(mtype, body, flags=0):
        body = bytes(body) + bytes((-len(body)) % 8)
        return struct.pack("<HHB3x", mtype, len(body), flags) + body

    def object_header(self, msgs):
        body = b"".join(msgs)
        return self.alloc(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

    @staticmethod
    def dataspace(dims):
        return struct.pack("<BBB5x", 1, len(dims), 0) + b"".join(struct.pack("<Q", d) for d in dims)

    @staticmethod
    def datatype(dt):
        dt = np.dtype(dt)
        if dt.kind == "f":
            assert dt.itemsize in (4, 8)
            # IEEE LE: class 1 v1; bits: byte order 0, padding 0, mantissa normalisation 2 (implied), sign location 31 / 63
            if dt.itemsize == 8:
                return struct.pack("<BBBBI", 0x11, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            return struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        if dt.kind in "iu":
            return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
        if dt.kind == "S":       # fixed-length string, null-terminated, ASCII
            return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, dt.itemsize)
        raise ValueError(dt)

    def attribute(self, name, value):
        """scalar or 1-D array attribute (EMAN2 stores a Transform, e.g. xform.align2d, as 12 floats); numbers keep their
        width and signedness, str / bytes become fixed-length strings (EMAN.source_path, ...)"""
        if isinstance(value, str):
            value = value.encode()
        value = np.asarray(value)
        if value.dtype.kind == "U":
            value = np.char.encode(value)
        if value.dtype.kind == "S":
            # the longest NUL-stripped value plus its terminating NUL (a value read back from a file already carries the stored
            # NUL in its itemsize: sizing from itemsize + 1 would grow the type by one byte per round trip)
            longest = max([len(v) for v in np.atleast_1d(value).ravel().tolist()] + [0])
            dt = np.dtype("S%d" % (longest + 1))
        elif value.dtype.kind == "f":
            dt = np.dtype("<f8") if value.dtype.itemsize == 8 else np.dtype("<f4")
        elif value.dtype.kind in "iu":
            dt = np.dtype("<" + value.dtype.kind + str(value.dtype.itemsize))
        elif value.dtype.kind == "b":
            dt = np.dtype("<i4")
        else:
            raise ValueError("attribute %s: values of type %s cannot be written" % (name, value.dtype))
        if value.ndim > 1:
            raise ValueError("attribute %s: %d-dimensional attributes cannot be written" % (name, value.ndim))
        nm = name.encode() + b"\0"
        ty, sp = self.datatype(dt), self.dataspace(list(value.shape))
        pad = lambda x: x + bytes((-len(x)) % 8)
        body = struct.pack("<BBHHH", 1, 0, len(nm), len(ty), len(sp)) + pad(nm) + pad(ty) + pad(sp) + value.astype(dt).tobytes()
        return self._msg(0x000C, body)

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr, "<f4")
        data = self.alloc(arr.tobytes())
        layout = struct.pack("<BBQQ", 3, 1, data, arr.nbytes)
        fill = struct.pack("<BBBB", 2, 2, 0, 0)             # fill value v2: allocate late, write never (undefined... ) not defined
        return self.object_header([self._msg(0x0001, self.dataspace(arr.shape)), self._msg(0x0003, self.datatype("<f4"), 1),
                                   self._msg(0x0005, fill), self._msg(0x0008, layout)])

    def group(self, children, attrs=()):
        """children: {name: (object header address, is_group, btree, heap)}; returns the same tuple for the new group"""
        names = sorted(children, key=lambda s: s.encode())
        heap = bytearray(8)                                  # offset 0: the empty name
        offs = {}
        for n in names:
            offs[n] = len(heap)
            e = n.encode() + b"\0"
            heap += e + bytes((-len(e)) % 8)
        heap_data = self.alloc(bytes(heap))
        heap_hdr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, heap_data))
        def spread(items, cap):
            """cut into the fewest runs of <= cap items, sizes as even as possible (no underfull node)"""
            nrun = max(1, -(-len(items) // cap))
            q, r = divmod(len(items), nrun)
            out, i = [], 0
            for j in range(nrun):
                k = q + (1 if j < r else 0)
                out.append(items[i:i + k]); i += k
            return out

        # level 0: symbol-table nodes; every node of the tree is described by (address, offset of its largest name)
        level = []
        for chunk in spread(names, 2 * self.LEAF_K):
            body = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk)))
            for n in chunk:
                addr, is_group, bt, hp = children[n]
                body += struct.pack("<QQI4x", offs[n], addr, 1 if is_group else 0)
                body += struct.pack("<QQ", bt, hp) if is_group else bytes(16)
            body += bytes(40 * (2 * self.LEAF_K - len(chunk)))
            level.append((self.alloc(bytes(body)), offs[chunk[-1]] if chunk else 0))
        depth = 0
        while True:
            runs = spread(level, 2 * self.INTERNAL_K)
            size = 24 + 2 * self.INTERNAL_K * 16 + 8
            addrs = [self.reserve(size) for _ in runs]
            left_key = 0
            nxt = []
            for j, run in enumerate(runs):
                node = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, depth, len(run), addrs[j - 1] if j else _UNDEF,
                                                          addrs[j + 1] if j + 1 < len(runs) else _UNDEF))
                node += struct.pack("<Q", left_key)
                for child, key in run:
                    node += struct.pack("<QQ", child, key)
                node += bytes(size - len(node))
                self.buf[addrs[j]:addrs[j] + size] = node
                left_key = run[-1][1]
                nxt.append((addrs[j], left_key))
            level = nxt
            depth += 1
            if len(level) == 1:
                break
        bt = level[0][0]
        ohdr = self.object_header([self._msg(0x0011, struct.pack("<QQ", bt, heap_hdr))] + list(attrs))
        return ohdr, True, bt, heap_hdr


def transform2d_matrix(alpha, sx, sy, mirror, scale=1.0):
    """the 12 floats (3 x 4, row-major) EMAN2 keeps for Transform({"type": "2d", alpha, tx, ty, mirror, scale}):
    x' = M R(alpha) x + t with R(alpha) = [[cos, sin], [-sin, cos]] and the mirror an x-flip of the rotated
    coordinates (pinned by the combine_params2 / inverse_transform2 known answers of
    cuda/EMAN2_test.ipynb cells 23-25 as far as rotation and translation go)"""
    a = np.deg2rad(float(alpha))
    c, s = np.cos(a) * scale, np.sin(a) * scale
    m = -1.0 if mirror else 1.0
    return np.array([m * c, m * s, 0.0, sx, -s, c, 0.0, sy, 0.0, 0.0, scale, 0.0], np.float32)


def params_from_matrix(m12):
    """(alpha, sx, sy, mirror) of a 2-D transform stored as 12 floats (inverse of transform2d_matrix)"""
    m = np.asarray(m12, np.float64).reshape(3, 4)
    mirror = int(m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0] < 0)
    sgn = -1.0 if mirror else 1.0
    alpha = np.rad2deg(np.arctan2(sgn * m[0, 1], sgn * m[0, 0])) % 360.0
    return float(alpha), float(m[0, 3]), float(m[1, 3]), mirror


def write_alignment_headers(src, dst, params, assign=None, ids=None):
    """header write-back of the reference's drivers (test_mref_cheng_yu_bdb_cuda.py:114-203: output_attr /
    write_attr with list_params ["xform.align2d", "assign", "ID"]; set_params2D at test_mref_gpu_align.py:588):
    copy stack `src` to `dst` with, per image, EMAN.xform.align2d (12 floats), EMAN.assign and EMAN.ID next to the
    attributes the stack already has.  params: rows (alpha, sx, sy, mirror).

    The reference updates header items in place (EMData.write_header); this writer rebuilds the whole file, so it must be
    able to carry everything over.  It REFUSES (HDF5FormatError, nothing written) when the source holds what it cannot
    re-encode: attributes of a type it does not decode (variable-length strings such as EMAN.ctf, compounds), attributes
    with more than one dimension, or pixel data that are not float32 -- with dst == src (in-place replacement of the
    user's stack) always, with a new dst unless the attribute can simply be left out there (reported in the return value).
    Returns the sorted names of the attributes that were NOT carried into a new dst ([] when everything was), followed by a
    note per pixel type that was re-encoded as float32 there."""
    import os
    report = {}
    arr, attrs = read_mdf_stack(src, with_attrs=True, report=report)
    n = arr.shape[0]
    assert len(params) == n
    lost = set(report.get("undecoded_attributes", ()))
    for i in range(n):
        lost.update(k for k, v in attrs[i].items() if k.startswith("EMAN.") and np.ndim(v) > 1)
    inplace = os.path.abspath(src) == os.path.abspath(dst)
    if report.get("pixel_types") and inplace:
        raise HDF5FormatError("%s: pixel data of type %s would be re-encoded as float32; write the headers to a new file instead"
                              % (src, ", ".join(sorted(report["pixel_types"]))))
    if lost and inplace:
        raise HDF5FormatError("%s: attributes %s cannot be carried over by this writer; refusing to replace the stack -- "
                              "write the headers to a new file instead" % (src, ", ".join(sorted(lost))))
    out = []
    for i in range(n):
        a = {k[5:]: v for k, v in attrs[i].items() if k.startswith("EMAN.") and k[5:] not in ("nx", "ny", "nz") and np.ndim(v) <= 1}
        a["xform.align2d"] = transform2d_matrix(params[i][0], params[i][1], params[i][2], int(params[i][3]))
        if assign is not None:
            a["assign"] = np.int32(assign[i])
        a["ID"] = np.int32(ids[i] if ids is not None else i)
        out.append(a)
    tmp = dst + ".tmp"
    try:
        write_mdf_stack(tmp, np.array(arr), out)
    except Exception:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise
    os.replace(tmp, dst)
    # pixel data of another type were re-encoded as float32 in the new file: reported with the attributes that were left out
    return sorted(lost) + ["<pixel data %s re-encoded as float32>" % t for t in sorted(report.get("pixel_types", ()))]


def write_mdf_stack(path, arr, extra_attrs=None):
    """write float32 images [n][ny][nx] as an EMAN2 MDF stack; extra_attrs: optional list of {name: int | float | 1-D array}
    per image, written as EMAN.<name> next to EMAN.nx / ny / nz"""
    arr = np.asarray(arr, np.float32)
    if arr.ndim == 2:
        arr = arr[None]
    w = _Writer()
    sb = w.reserve(96)                                       # superblock v0 (56 bytes) + root symbol-table entry (40)
    images = {}
    for i, img in enumerate(arr):
        attrs = [w.attribute("EMAN.nx", np.int32(img.shape[-1])), w.attribute("EMAN.ny", np.int32(img.shape[-2])),
                 w.attribute("EMAN.nz", np.int32(1))]
        for k, v in (extra_attrs[i].items() if extra_attrs else ()):
            attrs.append(w.attribute("EMAN." + k, v))
        d = w.dataset(img)
        images[str(i)] = w.group({"image": (d, False, 0, 0)}, attrs)
    g_images = w.group(images, [w.attribute("imageid_max", np.int32(arr.shape[0] - 1))])
    g_mdf = w.group({"images": g_images})
    root = w.group({"MDF": g_mdf})
    eof = (len(w.buf) + 7) & ~7
    w.buf += bytes(eof - len(w.buf))
    head = _SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, _Writer.LEAF_K, _Writer.INTERNAL_K, 0)
    head += struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
    head += struct.pack("<QQI4xQQ", 0, root[0], 1, root[2], root[3])
    assert len(head) == 96
    w.buf[sb:sb + 96] = head
    with open(path, "wb") as f:
        f.write(w.buf)
