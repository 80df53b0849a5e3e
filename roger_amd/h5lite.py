"""A small HDF5 writer / reader for restart files (h5py and h5netcdf are not part of this image; where h5py is installed
`roger_amd.restart` uses it instead and this module is the fallback).

What it covers is what RoGeR's restart files contain (roger/restart.py:32-67, roger/io_tools/hdf5.py): a root group with named
groups ("core", one per diagnostic), each holding plain numeric datasets -- float64 / float32 / int64 / int32 / int8 / uint8 arrays of
any rank, rank 0 included -- and nothing else (no attributes on datasets, no references, no compound types).

* `write(path, {group: {name: array}})` produces the classic layout every HDF5 library reads: superblock version 0, version-1 object
  headers, symbol-table groups (one symbol node per group: the superblock's leaf-node K is set accordingly), contiguous little-endian
  datasets.
* `read(path)` reads that layout AND what h5py writes with its defaults for such files: symbol-table groups over several B-tree
  levels, object-header continuation blocks, contiguous or chunked datasets (version-1 chunk B-trees) with the deflate and shuffle
  filters, h5py's boolean enum (read as int8).

Format reference: the HDF5 File Format Specification version 1.1/2.0 (superblock 0, B-tree v1 "TREE", symbol node "SNOD", local heap
"HEAP", object header messages 0x0001 dataspace, 0x0003 datatype, 0x0005 fill value, 0x0008 layout, 0x000B filter pipeline, 0x0010
continuation, 0x0011 symbol table).  Checked both ways against h5py in the build container (tests/test_h5lite.py)."""
import struct
import zlib

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
_DTYPES = {"float64": ("<f8", 1), "float32": ("<f4", 1), "int64": ("<i8", 0), "int32": ("<i4", 0), "int8": ("<i1", 0), "uint8": ("<u1", 0),
           "int16": ("<i2", 0), "uint16": ("<u2", 0), "uint32": ("<u4", 0), "uint64": ("<u8", 0)}


def _pad8(n):
    return (n + 7) & ~7


# ------------------------------------------------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------------------------------------------------
def _datatype_message(dt):
    dt = np.dtype(dt)
    size = dt.itemsize
    if dt.kind == "f":
        exp_bits, man_bits = (11, 52) if size == 8 else (8, 23)
        bits = bytes([0x20, size * 8 - 1, 0])   # little-endian, mantissa normalisation: implied msb; sign bit location
        props = struct.pack("<HHBBBBI", 0, size * 8, man_bits, exp_bits, 0, man_bits, (1 << (exp_bits - 1)) - 1)
        cls = 1
    else:
        bits = bytes([0x08 if dt.kind == "i" else 0x00, 0, 0])   # little-endian, two's complement if signed
        props = struct.pack("<HH", 0, size * 8)
        cls = 0
    return bytes([0x10 | cls]) + bits + struct.pack("<I", size) + props


def _message(mtype, body):
    body = body + b"\0" * (_pad8(len(body)) - len(body))
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def _object_header(messages):
    data = b"".join(messages)
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(data)) + data


class _Writer:
    def __init__(self):
        self.buf = bytearray()

    def alloc(self, data, align=8):
        while len(self.buf) % align:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr

    def dataset(self, arr):
        arr = np.asarray(arr)
        if arr.dtype == np.bool_:
            arr = arr.astype(np.int8)
        if arr.dtype.name not in _DTYPES:
            raise TypeError(f"h5lite: unsupported dtype {arr.dtype}")
        le = np.ascontiguousarray(arr, dtype=_DTYPES[arr.dtype.name][0])
        raw = le.tobytes()
        data_addr = self.alloc(raw) if raw else UNDEF
        space = struct.pack("<BBB5x", 1, arr.ndim, 0) + b"".join(struct.pack("<Q", int(d)) for d in arr.shape)
        msgs = [_message(0x0001, space), _message(0x0003, _datatype_message(le.dtype)), _message(0x0005, bytes([2, 2, 0, 0])),
                _message(0x0008, struct.pack("<BBQQ", 3, 1, data_addr, len(raw)))]
        return self.alloc(_object_header(msgs))

    def group(self, entries, leaf_k):
        """entries: {name: object header address}.  One symbol node holding all of them, sorted by name."""
        names = sorted(entries)
        heap = bytearray(b"\0" * 8)   # offset 0: the empty name
        offs = {}
        for n in names:
            offs[n] = len(heap)
            b = n.encode() + b"\0"
            heap += b + b"\0" * (_pad8(len(b)) - len(b))
        # a free block at the end keeps the library's heap code happy: offset of free list head, block = (next, size)
        free_off = len(heap)
        heap += struct.pack("<QQ", 1, 16)   # next = 1 (H5HL_FREE_NULL), size of this free block
        heap_data = self.alloc(bytes(heap))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data))
        snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
        for n in names:
            snod += struct.pack("<QQII16x", offs[n], entries[n], 0, 0)
        snod += b"\0" * (40 * (2 * leaf_k - len(names)))
        snod_addr = self.alloc(snod)
        internal_k = 16
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF)
        tree += struct.pack("<QQQ", 0, snod_addr, offs[names[-1]] if names else 0)
        tree += b"\0" * ((2 * internal_k + 1) * 8 + 2 * internal_k * 8 - 24)
        tree_addr = self.alloc(tree)
        hdr = self.alloc(_object_header([_message(0x0011, struct.pack("<QQ", tree_addr, heap_addr))]))
        return hdr, tree_addr, heap_addr


def write(path, groups):
    """groups: {group name: {dataset name: array}} -> an HDF5 file with those groups under the root."""
    n_max = max([len(groups)] + [len(v) for v in groups.values()] + [1])
    leaf_k = max(4, (n_max + 1) // 2)
    w = _Writer()
    w.alloc(b"\0" * 96)   # superblock, filled in at the end
    top = {}
    for gname, dsets in groups.items():
        entries = {name: w.dataset(arr) for name, arr in dsets.items()}
        top[gname] = w.group(entries, leaf_k)[0]
    root_hdr, root_tree, root_heap = w.group(top, leaf_k)
    sb = SIG + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", leaf_k, 16, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, len(w.buf), UNDEF)
    sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", root_tree, root_heap)
    assert len(sb) == 96
    w.buf[:96] = sb
    with open(path, "wb") as f:
        f.write(bytes(w.buf))


# ------------------------------------------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------------------------------------------
class _Reader:
    def __init__(self, data):
        self.d = data
        if data[:8] != SIG:
            raise ValueError("not an HDF5 file (or a user block precedes the superblock)")
        ver = data[8]
        if ver not in (0, 1):
            raise ValueError(f"h5lite reads superblock versions 0 and 1 (classic files); this file has version {ver}: read it with h5py")
        if data[13] != 8 or data[14] != 8:
            raise ValueError("h5lite: only 8-byte offsets and lengths")
        off = 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from("<Q", data, off)[0]
        entry = off + 32
        self.root = struct.unpack_from("<Q", data, entry + 8)[0]

    # -- object headers ---------------------------------------------------------------------------------------------------------
    def messages(self, addr):
        d = self.d
        if d[addr:addr + 4] == b"OHDR":
            return self._messages_v2(addr)
        ver, nmsg, _, size = struct.unpack_from("<BxHII", d, addr)
        if ver != 1:
            raise ValueError("h5lite: unknown object header version")
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", d, pos)
                body = d[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:
                    caddr, clen = struct.unpack_from("<QQ", body, 0)
                    blocks.append((caddr, clen))
                out.append((mtype, body))
        return out

    def _messages_v2(self, addr):
        """Version-2 object header ("OHDR", continuation chunks "OCHK"): what libhdf5 writes for objects of files that track the creation
        order of links and attributes -- every netCDF-4 file (h5netcdf / netCDF4 create their groups with track_order)."""
        d = self.d
        if d[addr + 4] != 2:
            raise ValueError("h5lite: unknown object header version")
        flags = d[addr + 5]
        pos = addr + 6
        if flags & 0x20:
            pos += 16            # access, modification, change, birth time
        if flags & 0x10:
            pos += 4             # max compact / min dense attributes
        nb = 1 << (flags & 3)
        size = int.from_bytes(d[pos:pos + nb], "little")
        pos += nb
        corder = 2 if flags & 0x04 else 0
        blocks = [(pos, pos + size)]
        out = []
        while blocks:
            pos, end = blocks.pop(0)
            while pos + 4 + corder <= end:
                mtype, msize, _ = struct.unpack_from("<BHB", d, pos)
                body = d[pos + 4 + corder:pos + 4 + corder + msize]
                pos += 4 + corder + msize
                if mtype == 0x10:
                    caddr, clen = struct.unpack_from("<QQ", body, 0)
                    if d[caddr:caddr + 4] != b"OCHK":
                        raise ValueError("h5lite: object header continuation chunk expected")
                    blocks.append((caddr + 4, caddr + clen - 4))   # between the signature and the checksum
                elif mtype != 0:
                    out.append((mtype, body))
        return out

    # -- groups -----------------------------------------------------------------------------------------------------------------
    def _heap_name(self, heap_addr, off):
        assert self.d[heap_addr:heap_addr + 4] == b"HEAP"
        data_addr = struct.unpack_from("<Q", self.d, heap_addr + 24)[0]
        start = data_addr + off
        end = self.d.index(b"\0", start)
        return self.d[start:end].decode()

    def _tree_leaves(self, addr):
        d = self.d
        assert d[addr:addr + 4] == b"TREE", "group B-tree node expected"
        _, level, used = struct.unpack_from("<BBH", d, addr + 4)
        pos = addr + 24
        children = []
        for i in range(used):
            pos += 8                     # key i
            children.append(struct.unpack_from("<Q", d, pos)[0])
            pos += 8
        for c in children:
            if level > 0:
                yield from self._tree_leaves(c)
            else:
                yield c

    def group(self, hdr_addr):
        """{name: object header address} of a group: symbol table (classic) or link messages / a fractal heap of them (new style)."""
        msgs = self.messages(hdr_addr)
        st = [b for t, b in msgs if t == 0x0011]
        if st:
            tree, heap = struct.unpack_from("<QQ", st[0], 0)
            out = {}
            for snod in self._tree_leaves(tree):
                assert self.d[snod:snod + 4] == b"SNOD"
                n = struct.unpack_from("<H", self.d, snod + 6)[0]
                for i in range(n):
                    name_off, obj = struct.unpack_from("<QQ", self.d, snod + 8 + 40 * i)
                    out[self._heap_name(heap, name_off)] = obj
            return out
        out = {}
        for t, b in msgs:
            if t == 0x06:        # compact storage: the links are messages of the header
                name, obj, _ = self._link(b, 0)
                if obj is not None:
                    out[name] = obj
            elif t == 0x02:      # link info: dense storage in a fractal heap
                pos = 2 + (8 if b[1] & 1 else 0)
                heap = struct.unpack_from("<Q", b, pos)[0]
                if heap != UNDEF:
                    out.update(self._heap_links(heap))
        if not any(t in (0x02, 0x06) for t, _ in msgs):
            raise ValueError("h5lite: not a group")
        return out

    def _link(self, b, pos):
        """A link message at b[pos:]: (name, object header address or None for a soft / external link, position after it)."""
        if b[pos] != 1:
            raise ValueError("h5lite: link message version")
        flags = b[pos + 1]
        pos += 2
        ltype = 0
        if flags & 0x08:
            ltype = b[pos]
            pos += 1
        if flags & 0x04:
            pos += 8
        if flags & 0x10:
            pos += 1
        nb = 1 << (flags & 3)
        nlen = int.from_bytes(b[pos:pos + nb], "little")
        pos += nb
        name = bytes(b[pos:pos + nlen]).decode()
        pos += nlen
        if ltype == 0:
            return name, struct.unpack_from("<Q", b, pos)[0], pos + 8
        if ltype == 1:           # soft link: length + path
            n = struct.unpack_from("<H", b, pos)[0]
            return name, None, pos + 2 + n
        n = struct.unpack_from("<H", b, pos)[0]
        return name, None, pos + 2 + n

    def _heap_links(self, addr):
        """The link messages held as managed objects of a fractal heap ("FRHP").  Files written once hold them back to back from the
        start of every direct block, so the blocks are read through instead of going by the name index (a version-2 B-tree)."""
        d = self.d
        if d[addr:addr + 4] != b"FRHP":
            raise ValueError("h5lite: fractal heap expected")
        flags = d[addr + 9]
        (width,) = struct.unpack_from("<H", d, addr + 110)
        start_size, max_direct = struct.unpack_from("<QQ", d, addr + 112)
        max_bits, _, root, cur_rows = struct.unpack_from("<HHQH", d, addr + 128)
        if struct.unpack_from("<H", d, addr + 7)[0]:
            raise ValueError("h5lite: filtered fractal heap")
        off_bytes = (max_bits + 7) // 8
        dhdr = 5 + 8 + off_bytes + (4 if flags & 2 else 0)
        max_direct_rows = (max_direct.bit_length() - 1) - (start_size.bit_length() - 1) + 2
        out = {}

        def direct(baddr, size):
            if baddr == UNDEF:
                return
            if d[baddr:baddr + 4] != b"FHDB":
                raise ValueError("h5lite: fractal heap direct block expected")
            pos, end = baddr + dhdr, baddr + size
            while pos < end and d[pos] == 1:
                name, obj, nxt = self._link(d, pos)
                if obj is not None:
                    out[name] = obj
                pos = nxt

        def indirect(baddr, rows):
            if baddr == UNDEF:
                return
            if d[baddr:baddr + 4] != b"FHIB":
                raise ValueError("h5lite: fractal heap indirect block expected")
            pos = baddr + 5 + 8 + off_bytes
            for r in range(rows):
                size = start_size * (1 if r < 2 else 2 ** (r - 1))
                for _ in range(width):
                    child = struct.unpack_from("<Q", d, pos)[0]
                    pos += 8
                    if r < max_direct_rows:
                        direct(child, size)
                    else:
                        indirect(child, (size.bit_length() - 1) - (start_size * width).bit_length() + 2)

        if cur_rows == 0:
            direct(root, start_size)
        else:
            indirect(root, cur_rows)
        return out

    def is_group(self, hdr_addr):
        return any(t in (0x0011, 0x02, 0x06) for t, _ in self.messages(hdr_addr))

    # -- datasets ---------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _dtype(body):
        cls, ver = body[0] & 0x0F, body[0] >> 4
        size = struct.unpack_from("<I", body, 4)[0]
        big = body[1] & 1
        if cls == 0:
            kind = "i" if body[1] & 0x08 else "u"
        elif cls == 1:
            kind = "f"
        elif cls == 8:   # enumeration (h5py's bool): the base type follows the 8-byte prefix
            return _Reader._dtype(body[8:])
        else:
            raise ValueError(f"h5lite: datatype class {cls} is not supported")
        return np.dtype((">" if big else "<") + kind + str(size))

    def dataset(self, hdr_addr):
        msgs = self.messages(hdr_addr)
        shape = dtype = layout = None
        filters = []
        for t, b in msgs:
            if t == 0x0001:
                ver, rank = b[0], b[1]
                pos = 8 if ver == 1 else 4
                shape = tuple(struct.unpack_from("<Q", b, pos + 8 * i)[0] for i in range(rank))
            elif t == 0x0003:
                dtype = self._dtype(b)
            elif t == 0x0008:
                layout = b
            elif t == 0x000B:
                ver, nf = b[0], b[1]
                pos = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid, nlen, _, ncd = struct.unpack_from("<HHHH", b, pos)
                    pos += 8
                    if ver == 1 or fid >= 256:
                        pos += _pad8(nlen) if ver == 1 else nlen
                    cd = struct.unpack_from(f"<{ncd}I", b, pos)
                    pos += 4 * ncd
                    if ver == 1 and ncd % 2:
                        pos += 4
                    filters.append((fid, cd))
        if shape is None or dtype is None or layout is None:
            raise ValueError("h5lite: dataset without dataspace / datatype / layout")
        count = int(np.prod(shape)) if shape else 1
        ver, cls = layout[0], layout[1]
        if ver != 3:
            raise ValueError(f"h5lite: data layout message version {ver}")
        if cls == 1:
            addr, size = struct.unpack_from("<QQ", layout, 2)
            if addr == UNDEF:
                return np.zeros(shape, dtype=dtype.newbyteorder("="))
            return np.frombuffer(self.d, dtype=dtype, count=count, offset=addr).reshape(shape).astype(dtype.newbyteorder("="))
        if cls == 0:   # compact
            size = struct.unpack_from("<H", layout, 2)[0]
            return np.frombuffer(layout[4:4 + size], dtype=dtype, count=count).reshape(shape).astype(dtype.newbyteorder("="))
        if cls == 2:
            rank1 = layout[2]
            tree = struct.unpack_from("<Q", layout, 3)[0]
            cdims = struct.unpack_from(f"<{rank1}I", layout, 11)[:-1]
            out = np.zeros(shape, dtype=dtype.newbyteorder("="))
            if tree != UNDEF:
                for offs, raw, mask in self._chunks(tree, rank1):
                    for k, (fid, cd) in enumerate(reversed(filters)):
                        if mask & (1 << (len(filters) - 1 - k)):
                            continue
                        if fid == 1:
                            raw = zlib.decompress(raw)
                        elif fid == 2:   # shuffle
                            n = cd[0] if cd else dtype.itemsize
                            raw = np.frombuffer(raw, dtype=np.uint8).reshape(n, -1).T.tobytes()
                        else:
                            raise ValueError(f"h5lite: filter {fid} is not supported")
                    chunk = np.frombuffer(raw, dtype=dtype, count=int(np.prod(cdims))).reshape(cdims)
                    sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                    out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
            return out
        raise ValueError(f"h5lite: layout class {cls}")

    def _chunks(self, addr, rank1):
        d = self.d
        assert d[addr:addr + 4] == b"TREE", "chunk B-tree node expected"
        ntype, level, used = struct.unpack_from("<BBH", d, addr + 4)
        key_size = 8 + 8 * rank1
        pos = addr + 24
        for _ in range(used):
            csize, mask = struct.unpack_from("<II", d, pos)
            offs = struct.unpack_from(f"<{rank1}Q", d, pos + 8)[:-1]
            child = struct.unpack_from("<Q", d, pos + key_size)[0]
            pos += key_size + 8
            if level > 0:
                yield from self._chunks(child, rank1)
            else:
                yield offs, d[child:child + csize], mask


def read_root(path):
    """{dataset name: array} of the datasets directly under the root -- the variables of a netCDF-4 file (forcing.nc, parameters.nc,
    <identifier>.rate.nc ... as the reference writes them through h5netcdf: version-2 object headers, links in a fractal heap,
    chunked + gzip data; the dimension scales come back as datasets like any other)."""
    return read(path).get("", {})


def is_hdf5(path):
    with open(path, "rb") as f:
        return f.read(8) == SIG


def read(path):
    """{group name: {dataset name: array}} of the groups under the root (datasets directly under the root come under the key "")."""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    out = {}
    for name, addr in r.group(r.root).items():
        if r.is_group(addr):
            out[name] = {k: r.dataset(a) for k, a in r.group(addr).items() if not r.is_group(a)}
        else:
            out.setdefault("", {})[name] = r.dataset(addr)
    return out
