"""A minimal writer for HDF5 files of the *old* on-disk style (superblock version 0, version-1 object headers, the
root group as a symbol table, chunked datasets indexed by a version-1 B-tree, filter pipeline version 1) -- the
layout netCDF-4 files written by HDF5 1.8-era libraries have, e.g. compressed reanalysis files.  Written from the
HDF5 File Format Specification for the tests of mptrac_amd/host/nc_hdf5.c: the image has no HDF5 library, and
the reference's own netCDF-4 files are all contiguous and new-style.  (A file made here is laid out by this
module's reading of the specification, not by the HDF5 library.)

Writer(style="new") writes the other common combination instead -- what netCDF-4 produces with the "1.8" format
bounds: superblock version 2, version-2 object headers, the root group as compact link messages, version-2
dataspace / filter pipeline / version-3 attribute messages, and still the version-3 layout with a version-1 chunk
B-tree.  (Checksums are written as zero: the reader under test does not verify them.)"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, body):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def _dataspace(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", n) for n in shape)


def _datatype(dtype):
    dtype = np.dtype(dtype)
    big = dtype.byteorder == ">"
    if dtype.kind == "f":
        if dtype.itemsize == 4:
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            sign = 31
        else:
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            sign = 63
        return struct.pack("<BBBBI", 0x11, 0x20 | (1 if big else 0), sign, 0, dtype.itemsize) + props
    return struct.pack("<BBBBI", 0x10, 0x08 | (1 if big else 0), 0, 0, dtype.itemsize) + struct.pack("<HH", 0, 8 * dtype.itemsize)


def _attribute(name, value):
    nm = name.encode() + b"\0"
    ds = struct.pack("<BBB5x", 1, 0, 0)
    if isinstance(value, (str, bytes)):      # a fixed-length, null-terminated string
        raw = (value.encode() if isinstance(value, str) else value) + b"\0"
        dt = struct.pack("<BBBBI", 0x13, 0, 0, 0, len(raw))
    else:
        value = np.asarray(value)
        raw = value.tobytes()
        dt = _datatype(value.dtype)
    return struct.pack("<BxHHH", 1, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + raw


def dimension_scale(writer, name, values):
    """a coordinate variable the way netCDF-4 stores a dimension: a 1-D dataset with CLASS = DIMENSION_SCALE"""
    writer.dataset(name, values, attrs=(("CLASS", "DIMENSION_SCALE"), ("NAME", name)))


def _shuffle(raw, itemsize):
    a = np.frombuffer(raw, dtype=np.uint8).reshape(-1, itemsize)
    return a.T.copy().tobytes()


def _msg2(mtype, body):
    return struct.pack("<BHB", mtype, len(body), 0) + body


def _dataspace2(shape):
    return struct.pack("<BBBB", 2, len(shape), 0, 1) + b"".join(struct.pack("<Q", n) for n in shape)


def _attribute3(name, value):
    nm = name.encode() + b"\0"
    ds = struct.pack("<BBBB", 2, 0, 0, 0)
    if isinstance(value, (str, bytes)):
        raw = (value.encode() if isinstance(value, str) else value) + b"\0"
        dt = struct.pack("<BBBBI", 0x13, 0, 0, 0, len(raw))
    else:
        value = np.asarray(value)
        raw = value.tobytes()
        dt = _datatype(value.dtype)
    return struct.pack("<BBHHHB", 3, 0, len(nm), len(dt), len(ds), 0) + nm + dt + ds + raw


class Writer:
    def __init__(self, style="old"):
        assert style in ("old", "new")
        self.new = style == "new"
        self.buf = bytearray(96)      # the superblock is written last
        self.entries = []             # (name, object header address)

    def _append(self, data):
        while len(self.buf) % 8:
            self.buf.append(0)
        at = len(self.buf)
        self.buf += data
        return at

    def _object_header(self, messages):
        body = b"".join(messages)
        if self.new:   # "OHDR", version 2, flags (size of chunk 0 in 4 bytes), size, messages, checksum
            return self._append(b"OHDR" + struct.pack("<BBI", 2, 0x02, len(body)) + body + b"\0\0\0\0")
        return self._append(struct.pack("<BxHII4x", 1, len(messages), 1, len(body)) + body)

    def dataset(self, name, array, chunks=None, shuffle=False, deflate=None, attrs=(), fill=None, skip_chunks=()):
        """array: numpy array (its dtype and byte order are stored as they are); chunks: chunk shape or None for
        contiguous storage; skip_chunks: chunk indices (tuples) that are not written (they read as `fill`)"""
        array = np.ascontiguousarray(array)
        _msg = _msg2 if self.new else globals()["_msg"]
        msgs = [_msg(0x01, (_dataspace2 if self.new else _dataspace)(array.shape)), _msg(0x03, _datatype(array.dtype))]
        if fill is not None:
            fv = np.asarray(fill, dtype=array.dtype).tobytes()
            if self.new:   # version 3: flags (allocation time 2, write time 0, defined), size, value
                msgs.append(_msg(0x05, struct.pack("<BBI", 3, 0x02 | 0x20, len(fv)) + fv))
            else:
                msgs.append(_msg(0x05, struct.pack("<BBBBI", 2, 2, 0, 1, len(fv)) + fv))
        if chunks is None:
            at = self._append(array.tobytes())
            msgs.append(_msg(0x08, struct.pack("<BBQQ", 3, 1, at, array.nbytes)))
        else:
            filters = []
            if shuffle:
                filters.append(struct.pack("<HHHI", 2, 0, 1, array.itemsize) if self.new
                               else struct.pack("<HHHHI4x", 2, 0, 0, 1, array.itemsize))
            if deflate is not None:
                filters.append(struct.pack("<HHHI", 1, 0, 1, deflate) if self.new
                               else struct.pack("<HHHHI4x", 1, 0, 0, 1, deflate))
            if filters:
                msgs.append(_msg(0x0B, (struct.pack("<BB", 2, len(filters)) if self.new else struct.pack("<BB6x", 1, len(filters)))
                                 + b"".join(filters)))
            rank = array.ndim
            keys = []
            counts = [-(-array.shape[k] // chunks[k]) for k in range(rank)]
            for idx in np.ndindex(*counts):
                if idx in skip_chunks:
                    continue
                block = np.zeros(chunks, dtype=array.dtype)
                if fill is not None:
                    block[...] = fill
                sel = tuple(slice(idx[k] * chunks[k], min((idx[k] + 1) * chunks[k], array.shape[k])) for k in range(rank))
                part = array[sel]
                block[tuple(slice(0, n) for n in part.shape)] = part
                raw = block.tobytes()
                if shuffle:
                    raw = _shuffle(raw, array.itemsize)
                if deflate is not None:
                    raw = zlib.compress(raw, deflate)
                at = self._append(raw)
                keys.append((len(raw), [idx[k] * chunks[k] for k in range(rank)] + [0], at))
            node = b"TREE" + struct.pack("<BBHQQ", 1, 0, len(keys), UNDEF, UNDEF)
            for size, off, at in keys:
                node += struct.pack("<II", size, 0) + b"".join(struct.pack("<Q", o) for o in off) + struct.pack("<Q", at)
            node += struct.pack("<II", 0, 0) + b"".join(struct.pack("<Q", n) for n in list(array.shape) + [0])
            tree = self._append(node) if keys else UNDEF
            msgs.append(_msg(0x08, struct.pack("<BBBQ", 3, 2, rank + 1, tree)
                             + b"".join(struct.pack("<I", c) for c in list(chunks) + [array.itemsize])))
        for aname, value in attrs:
            msgs.append(_msg(0x0C, (_attribute3 if self.new else _attribute)(aname, value)))
        self.entries.append((name, self._object_header(msgs)))

    def close(self, path):
        if self.new:
            links = [_msg2(0x06, struct.pack("<BBB", 1, 0, len(name.encode())) + name.encode() + struct.pack("<Q", addr))
                     for name, addr in self.entries]
            root = self._object_header(links)
            eof = len(self.buf)
            sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBB", 2, 8, 8, 0) + struct.pack("<QQQQ", 0, UNDEF, eof, root) + b"\0\0\0\0"
            self.buf[:len(sb)] = sb
            with open(path, "wb") as f:
                f.write(self.buf)
            return
        # local heap: the empty name of the root at offset 0, then the link names (sorted, as the B-tree keys need)
        self.entries.sort()
        heap = bytearray(b"\0" * 8)
        offsets = []
        for name, _ in self.entries:
            offsets.append(len(heap))
            heap += _pad8(name.encode() + b"\0")
        heap_data = self._append(bytes(heap))
        heap_at = self._append(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), UNDEF, heap_data))
        snod = b"SNOD" + struct.pack("<BxH", 1, len(self.entries))
        for off, (_, addr) in zip(offsets, self.entries):
            snod += struct.pack("<QQII16x", off, addr, 0, 0)
        snod_at = self._append(snod)
        tree_at = self._append(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_at, offsets[-1]))
        root = self._object_header([_msg(0x11, struct.pack("<QQ", tree_at, heap_at))])
        eof = len(self.buf)
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBxBBBxHHI", 0, 0, 0, 0, 8, 8, 4, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF) + struct.pack("<QQII16x", 0, root, 0, 0)
        assert len(sb) == 96
        self.buf[:96] = sb
        with open(path, "wb") as f:
            f.write(self.buf)
