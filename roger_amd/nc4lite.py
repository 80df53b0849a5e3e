"""A netCDF-4 writer for the output files of the diagnostics (h5netcdf / netCDF4 / h5py are not part of this image; the reference writes
`<identifier>.<diagnostic>.nc` through h5netcdf, roger/io_tools/netcdf.py:22-72, 121-165).

netCDF-4 is HDF5 with conventions, and this module writes exactly those (NetCDF-4 file format specification, "HDF5 format"):

* every dimension is an HDF5 dataset that is a DIMENSION SCALE -- attributes `CLASS = "DIMENSION_SCALE"`, `NAME = <name>` and
  `_Netcdf4Dimid = <id>`; the coordinate variables `x`, `y`, `Time` carry their values;
* every variable is a dataset whose attribute `DIMENSION_LIST` holds, per axis, a variable-length list of object references to the
  scales of that axis (one each), and every scale lists the variables that use it in `REFERENCE_LIST` (the HDF5 dimension-scale API's
  back references: compound {object reference, axis});
* the UNLIMITED dimension (`Time`) and the variables along it are chunked datasets (one record per chunk, version-1 chunk B-tree) whose
  dataspace has an unlimited maximum extent; everything else is contiguous;
* attributes (`long_name`, `units`, `time_origin`, the global ones on the root group) are fixed-length ASCII strings.

Container: superblock version 0, version-1 object headers, symbol-table groups, version-1 attribute messages, one global heap
collection for the reference lists -- the oldest, most widely read layout (`roger_amd.h5lite` writes restart files the same way and
shares the primitives).  Checked with h5py (dimension scales attached, `maxshape`, values, attributes) in tests/test_nc4lite.py where
h5py is available, and read back by `h5lite.read_root` everywhere."""
import struct

import numpy as np

from .h5lite import SIG, UNDEF, _datatype_message, _DTYPES, _message, _object_header, _pad8, _Writer

UNLIMITED = 0xFFFFFFFFFFFFFFFF
_CHUNK_K = 32          # superblock version 0 has no field for it: the library's default "indexed storage internal node K"


def _pad(b):
    return b + b"\0" * (_pad8(len(b)) - len(b))


def _dataspace(shape, maxshape=None):
    body = struct.pack("<BBB5x", 1, len(shape), 1 if maxshape is not None else 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)
    if maxshape is not None:
        body += b"".join(struct.pack("<Q", UNLIMITED if m is None else int(m)) for m in maxshape)
    return body


def _string_type(n):
    return bytes([0x13, 0x00, 0, 0]) + struct.pack("<I", n)     # class 3, null-terminated ASCII


_REF_TYPE = bytes([0x17, 0x00, 0, 0]) + struct.pack("<I", 8)   # class 7, object reference
_VLEN_REF_TYPE = bytes([0x19, 0x00, 0, 0]) + struct.pack("<I", 16) + _REF_TYPE   # class 9: sequence of object references
_I32_TYPE = _datatype_message(np.dtype("<i4"))
# REFERENCE_LIST element: struct { hobj_ref_t dataset; int dimension; } as the dimension-scale API lays it out (16 bytes)
_REFLIST_TYPE = (bytes([0x16, 0x02, 0, 0]) + struct.pack("<I", 16)
                 + _pad(b"dataset\0") + struct.pack("<IB3xII4I", 0, 0, 0, 0, 0, 0, 0, 0) + _REF_TYPE
                 + _pad(b"dimension\0") + struct.pack("<IB3xII4I", 8, 0, 0, 0, 0, 0, 0, 0) + _I32_TYPE)


def _attribute(name, dtype_msg, space_msg, data):
    nm = name.encode() + b"\0"
    body = struct.pack("<BxHHH", 1, len(nm), len(dtype_msg), len(space_msg)) + _pad(nm) + _pad(dtype_msg) + _pad(space_msg) + data
    return _message(0x000C, body)


def _attr_string(name, value):
    raw = str(value).encode("ascii", "replace") + b"\0"
    return _attribute(name, _string_type(len(raw)), _dataspace(()), raw)


def _attr_value(name, value):
    if isinstance(value, (str, bytes)):
        return _attr_string(name, value.decode() if isinstance(value, bytes) else value)
    a = np.asarray(value)
    if a.dtype.kind in "iu":
        a = a.astype("<i8" if a.dtype.itemsize > 4 else "<i4")
    else:
        a = a.astype("<f8")
    return _attribute(name, _datatype_message(a.dtype), _dataspace(a.shape), a.tobytes())


class _Nc4Writer(_Writer):
    def chunked_data(self, arr, rank):
        """One chunk per index of the first axis; returns the address of the chunk B-tree's root (UNDEF for no records)."""
        n = arr.shape[0]
        if n == 0 or arr.size == 0:
            return UNDEF
        rank1 = rank + 1
        key_size = 8 + 8 * rank1
        node_size = 24 + (2 * _CHUNK_K + 1) * key_size + 2 * _CHUNK_K * 8
        csize = arr[0].nbytes

        def key(i, size):
            return struct.pack("<II", size, 0) + struct.pack("<Q", i) + b"\0" * (8 * (rank1 - 1))

        entries = [(i, self.alloc(arr[i].tobytes())) for i in range(n)]     # (first record of the subtree, address)
        level = 0
        while True:
            groups = [entries[k:k + 2 * _CHUNK_K] for k in range(0, len(entries), 2 * _CHUNK_K)]
            addrs = [self.alloc(b"\0" * node_size) for _ in groups]
            nxt = []
            for gi, (grp, addr) in enumerate(zip(groups, addrs)):
                node = b"TREE" + struct.pack("<BBHQQ", 1, level, len(grp), addrs[gi - 1] if gi else UNDEF,
                                             addrs[gi + 1] if gi + 1 < len(groups) else UNDEF)
                for first, child in grp:
                    node += key(first, csize) + struct.pack("<Q", child)
                last = groups[gi + 1][0][0] if gi + 1 < len(groups) else n
                node += key(last, 0)
                self.buf[addr:addr + len(node)] = node
                nxt.append((grp[0][0], addr))
            if len(nxt) == 1:
                return nxt[0][1]
            entries, level = nxt, level + 1

    def dataset_header(self, arr, maxshape, attrs):
        """Object header of a dataset; chunked along the first axis if `maxshape` says that axis is unlimited."""
        if arr.dtype.name not in _DTYPES:
            raise TypeError(f"nc4lite: unsupported dtype {arr.dtype}")
        le = np.ascontiguousarray(arr, dtype=_DTYPES[arr.dtype.name][0])
        unlimited = maxshape is not None and len(maxshape) and maxshape[0] is None
        msgs = [_message(0x0001, _dataspace(le.shape, maxshape if unlimited else None)), _message(0x0003, _datatype_message(le.dtype))]
        if unlimited:
            tree = self.chunked_data(le, le.ndim)
            cdims = (1,) + tuple(int(d) for d in le.shape[1:]) + (le.dtype.itemsize,)
            msgs.append(_message(0x0005, bytes([2, 3, 0, 0])))     # fill value: allocate incrementally, never written, undefined
            msgs.append(_message(0x0008, struct.pack("<BBB", 3, 2, le.ndim + 1) + struct.pack("<Q", tree) + b"".join(struct.pack("<I", c) for c in cdims)))
        else:
            raw = le.tobytes()
            addr = self.alloc(raw) if raw else UNDEF
            msgs.append(_message(0x0005, bytes([2, 2, 0, 0])))
            msgs.append(_message(0x0008, struct.pack("<BBQQ", 3, 1, addr, len(raw))))
        return self.alloc(_object_header(msgs + attrs))

    def group_with_attrs(self, entries, leaf_k, attrs):
        names = sorted(entries)
        heap = bytearray(b"\0" * 8)
        offs = {}
        for nm in names:
            offs[nm] = len(heap)
            heap += _pad(nm.encode() + b"\0")
        free_off = len(heap)
        heap += struct.pack("<QQ", 1, 16)
        heap_data = self.alloc(bytes(heap))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data))
        snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
        for nm in names:
            snod += struct.pack("<QQII16x", offs[nm], entries[nm], 0, 0)
        snod += b"\0" * (40 * (2 * leaf_k - len(names)))
        snod_addr = self.alloc(snod)
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_addr, offs[names[-1]] if names else 0)
        tree += b"\0" * ((2 * 16 + 1) * 8 + 2 * 16 * 8 - 24)
        tree_addr = self.alloc(tree)
        hdr = self.alloc(_object_header([_message(0x0011, struct.pack("<QQ", tree_addr, heap_addr))] + attrs))
        return hdr, tree_addr, heap_addr


def write(path, dimensions, variables, attributes=None):
    """dimensions: {name: size or None (unlimited)}, in netCDF dimension-id order.
    variables: {name: (dims tuple, array, {attribute: value})}; a variable named like a dimension is its coordinate variable.
    A dimension without a coordinate variable becomes a scale without values, as netCDF writes it.  attributes: global ones."""
    dim_names = list(dimensions)
    dim_id = {d: k for k, d in enumerate(dim_names)}
    arrays = {}
    for name, (dims, arr, _) in variables.items():
        arr = np.asarray(arr)
        if arr.dtype == np.bool_:
            arr = arr.astype(np.int8)
        if arr.ndim != len(dims):
            raise ValueError(f"{name}: {arr.ndim} axes for dimensions {dims}")
        for ax, d in enumerate(dims):
            if d not in dimensions:
                raise ValueError(f"{name}: unknown dimension {d!r}")
            if dimensions[d] is not None and arr.shape[ax] != dimensions[d]:
                raise ValueError(f"{name}: axis {ax} has {arr.shape[ax]} entries, dimension {d} has {dimensions[d]}")
            if dimensions[d] is None and ax != 0:
                raise ValueError(f"{name}: the unlimited dimension must be the first axis")
        arrays[name] = arr
    n_unlimited = {d: max([arrays[v].shape[0] for v, (dims, _, _) in variables.items() if dims and dims[0] == d] + [0])
                   for d in dim_names if dimensions[d] is None}
    users = {d: [(v, ax) for v, (dims, _, _) in variables.items() for ax, dd in enumerate(dims) if dd == d and v != d] for d in dim_names}

    def build(addr_of):
        """The whole file for given object-header addresses of the datasets (two passes: the layout does not depend on them)."""
        w = _Nc4Writer()
        w.alloc(b"\0" * 96)
        # one global heap collection with one object per (variable, axis): the reference to the scale of that axis
        pairs = [(v, ax) for v, (dims, _, _) in variables.items() if v not in dimensions for ax in range(len(dims))]
        gcol_size = max(4096, _pad8(16 + 32 * len(pairs) + 16))
        gcol = bytearray(b"GCOL" + struct.pack("<B3xQ", 1, gcol_size))
        heap_index = {}
        for k, (v, ax) in enumerate(pairs):
            heap_index[(v, ax)] = k + 1
            gcol += struct.pack("<HH4xQ", k + 1, 1, 8) + struct.pack("<Q", addr_of.get(variables[v][0][ax], 0))
        gcol += struct.pack("<HH4xQ", 0, 0, gcol_size - len(gcol))
        gcol += b"\0" * (gcol_size - len(gcol))
        gcol_addr = w.alloc(bytes(gcol))
        new = {}
        for d in dim_names:   # the scales first (netCDF creates them in dimension-id order)
            size = n_unlimited[d] if dimensions[d] is None else dimensions[d]
            if d in variables:
                _, _, vattrs = variables[d]
                arr, nm = arrays[d], d
            else:
                vattrs, arr = {}, np.zeros(size, dtype=np.float32)
                nm = f"This is a netCDF dimension but not a netCDF variable.{size:10d}"
            attrs = [_attr_string("CLASS", "DIMENSION_SCALE"), _attr_string("NAME", nm),
                     _attribute("_Netcdf4Dimid", _I32_TYPE, _dataspace(()), struct.pack("<i", dim_id[d]))]
            if users[d]:
                data = b"".join(struct.pack("<Qi4x", addr_of.get(v, 0), ax) for v, ax in users[d])
                attrs.append(_attribute("REFERENCE_LIST", _REFLIST_TYPE, _dataspace((len(users[d]),)), data))
            attrs += [_attr_value(k, v) for k, v in vattrs.items()]
            new[d] = w.dataset_header(arr, (None,) if dimensions[d] is None else None, attrs)
        for v, (dims, _, vattrs) in variables.items():
            if v in dimensions:
                continue
            data = b"".join(struct.pack("<IQI", 1, gcol_addr, heap_index[(v, ax)]) for ax in range(len(dims)))
            attrs = [_attribute("DIMENSION_LIST", _VLEN_REF_TYPE, _dataspace((len(dims),)), data)] if dims else []
            attrs += [_attr_value(k, val) for k, val in vattrs.items()]
            maxshape = tuple(None if dimensions[d] is None else dimensions[d] for d in dims)
            new[v] = w.dataset_header(arrays[v], maxshape, attrs)
        leaf_k = max(4, (len(new) + 1) // 2)
        root_hdr, root_tree, root_heap = w.group_with_attrs(new, leaf_k, [_attr_value(k, v) for k, v in (attributes or {}).items()])
        sb = SIG + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", leaf_k, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(w.buf), UNDEF)
        sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", root_tree, root_heap)
        w.buf[:96] = sb
        return bytes(w.buf), new

    _, addrs = build({})
    data, again = build(addrs)
    assert again == addrs, "nc4lite: the layout moved between the two passes"
    with open(path, "wb") as f:
        f.write(data)


# ------------------------------------------------------------------------------------------------------------------------------
# reader (of the files written above: version-1 object headers and attribute messages)
# ------------------------------------------------------------------------------------------------------------------------------
def _parse_attribute(r, body, names_by_addr):
    ver, name_size, dt_size, ds_size = struct.unpack_from("<BxHHH", body, 0)
    if ver != 1:
        raise ValueError(f"nc4lite.read: attribute message version {ver} (files of other writers: use h5py)")
    pos = 8
    name = body[pos:pos + name_size].split(b"\0")[0].decode()
    pos += _pad8(name_size)
    dt = body[pos:pos + dt_size]
    pos += _pad8(dt_size)
    ds = body[pos:pos + ds_size]
    pos += _pad8(ds_size)
    rank = ds[1]
    shape = tuple(struct.unpack_from("<Q", ds, 8 + 8 * k)[0] for k in range(rank))
    count = int(np.prod(shape)) if shape else 1
    cls, size = dt[0] & 0x0F, struct.unpack_from("<I", dt, 4)[0]
    data = body[pos:pos + count * size]
    if cls == 3:
        return name, data.split(b"\0")[0].decode()
    if cls in (0, 1):
        a = np.frombuffer(data, dtype=r._dtype(dt), count=count).reshape(shape)
        return name, (a.item() if not shape else a.copy())
    if cls == 9:     # DIMENSION_LIST: per axis the names of the scales attached to it
        out = []
        for k in range(count):
            n, gaddr, idx = struct.unpack_from("<IQI", data, 16 * k)
            out.append([names_by_addr.get(a) for a in _gcol_object(r.d, gaddr, idx, n)])
        return name, out
    if cls == 6:     # REFERENCE_LIST: (variable, axis) pairs
        return name, [(names_by_addr.get(struct.unpack_from("<Q", data, 16 * k)[0]), struct.unpack_from("<i", data, 16 * k + 8)[0]) for k in range(count)]
    return name, None


def _gcol_object(d, gaddr, idx, n):
    assert d[gaddr:gaddr + 4] == b"GCOL", "global heap collection expected"
    size = struct.unpack_from("<Q", d, gaddr + 8)[0]
    pos = gaddr + 16
    while pos < gaddr + size:
        oi, _, osize = struct.unpack_from("<HH4xQ", d, pos)
        if oi == idx:
            return [struct.unpack_from("<Q", d, pos + 16 + 8 * k)[0] for k in range(n)]
        if oi == 0:
            break
        pos += 16 + _pad8(osize)
    raise ValueError("nc4lite.read: dangling global heap reference")


def read(path):
    """{"dimensions": {name: size, None for the unlimited one}, "variables": {name: (dims, array, attributes)}, "attributes": {...}} of
    a file written by `write` (what the tests and `read_svat_output` need; files of other writers: h5lite.read_root / h5py)."""
    from .h5lite import _Reader

    with open(path, "rb") as f:
        r = _Reader(f.read())
    entries = r.group(r.root)
    names_by_addr = {a: n for n, a in entries.items()}
    out = {"dimensions": {}, "variables": {}, "attributes": {}}
    for t, b in r.messages(r.root):
        if t == 0x000C:
            k, v = _parse_attribute(r, b, names_by_addr)
            out["attributes"][k] = v
    scales = {}
    for name, addr in entries.items():
        attrs, unlimited = {}, False
        for t, b in r.messages(addr):
            if t == 0x000C:
                k, v = _parse_attribute(r, b, names_by_addr)
                attrs[k] = v
            elif t == 0x0001 and b[2] & 1:
                rank = b[1]
                unlimited = struct.unpack_from("<Q", b, 8 + 8 * rank)[0] == UNLIMITED
        arr = r.dataset(addr)
        if attrs.get("CLASS") == "DIMENSION_SCALE":
            scales[name] = (attrs.get("_Netcdf4Dimid", 0), None if unlimited else arr.shape[0], str(attrs.get("NAME", "")).startswith("This is a netCDF dimension"))
        out["variables"][name] = (attrs, arr)
    for name in sorted(scales, key=lambda n: scales[n][0]):
        out["dimensions"][name] = scales[name][1]
    final = {}
    for name, (attrs, arr) in out["variables"].items():
        if name in scales and scales[name][2]:
            continue    # a dimension without a coordinate variable
        dims = tuple(ax[0] for ax in attrs["DIMENSION_LIST"]) if "DIMENSION_LIST" in attrs else ((name,) if name in scales else ())
        user = {k: v for k, v in attrs.items() if k not in ("CLASS", "NAME", "_Netcdf4Dimid", "DIMENSION_LIST", "REFERENCE_LIST")}
        final[name] = (dims, arr, user)
    out["variables"] = final
    return out
