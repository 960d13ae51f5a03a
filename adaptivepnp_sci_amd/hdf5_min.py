"""A minimal pure-Python reader for the HDF5 subset MATLAB's `save -v7.3` writes -- enough to load the reference's scene
files (meas_bayer / mask_bayer / orig_bayer / orig, two_stage_ADMM_Online_FFD_Warm.py:164-197) where h5py is not installed.

Supported: superblock version 0 / 1 behind a user block (MATLAB's 512-byte text header), old-style groups (symbol-table
B-tree + local heap), version-1 object headers with continuation blocks, simple dataspaces, little-endian fixed-point and
IEEE floating-point datatypes, data layout version 3 (compact, contiguous, chunked with a version-1 chunk B-tree), the
deflate and shuffle filters.  Anything else raises `Hdf5Unsupported` (the caller then asks for h5py).  Arrays come back
with the axes as stored, i.e. reversed with respect to MATLAB's -- exactly what `numpy.array(h5py.File(p)[name])` returns.

Format reference: "HDF5 File Format Specification Version 2.0" (The HDF Group), sections III.A-III.D, IV.A.1-2.
"""
import struct
import zlib

import numpy as np

SIG = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF
MAX_DEPTH = 64                      # B-tree levels; a real file has 1-3
# what a damaged file can trip inside the decoder; read_mat73 re-raises all of it as Hdf5Unsupported
_DAMAGE = (struct.error, IndexError, ValueError, TypeError, OverflowError, MemoryError, zlib.error, RecursionError)


class Hdf5Unsupported(RuntimeError):
    pass


class _File:
    def __init__(self, path):
        with open(path, 'rb') as f:
            self.buf = f.read()
        off = 0
        while self.buf[off:off + 8] != SIG:                 # the superblock sits at 0, 512, 1024, 2048, ...
            off = 512 if off == 0 else off * 2
            if off + 8 > len(self.buf):
                raise Hdf5Unsupported('no HDF5 signature')
        self.base = off
        ver = self.buf[off + 8]
        if ver not in (0, 1):
            raise Hdf5Unsupported(f'superblock version {ver} (only 0 / 1, what MATLAB writes)')
        self.O, self.L = self.buf[off + 13], self.buf[off + 14]
        if (self.O, self.L) != (8, 8):
            raise Hdf5Unsupported('offsets / lengths other than 8 bytes')
        p = off + 24 + (4 if ver == 1 else 0)
        base_addr, _fs, _eof, _drv = struct.unpack_from('<4Q', self.buf, p)
        # addresses in the file are relative to the base address; with a user block that is where the superblock sits
        self.base = off if base_addr in (0, UNDEF) else base_addr
        self.root = self._symbol_entry(p + 32)

    # ---- primitives (every address is bounds-checked: a truncated or corrupt file raises Hdf5Unsupported, nothing else)
    def at(self, addr, n):
        a = self.base + addr
        if addr < 0 or n < 0 or a + n > len(self.buf):
            raise Hdf5Unsupported(f'address {addr:#x} + {n} lies outside the file (truncated or corrupt)')
        return self.buf[a:a + n]

    def u(self, addr, fmt):
        a = self.base + addr
        if addr < 0 or a + struct.calcsize('<' + fmt) > len(self.buf):
            raise Hdf5Unsupported(f'address {addr:#x} lies outside the file (truncated or corrupt)')
        return struct.unpack_from('<' + fmt, self.buf, a)

    def _symbol_entry(self, abs_pos):
        name_off, hdr, cache = struct.unpack_from('<QQI', self.buf, abs_pos)
        ent = {'name_off': name_off, 'header': hdr, 'cache': cache}
        if cache == 1:
            ent['btree'], ent['heap'] = struct.unpack_from('<QQ', self.buf, abs_pos + 24)
        return ent

    # ---- object headers (version 1)
    def messages(self, addr):
        if self.at(addr, 4) == b'OHDR':
            raise Hdf5Unsupported('version-2 object header (file not written with the old format MATLAB uses)')
        ver, _r, nmsg, _rc, size = self.u(addr, 'BBHII')
        if ver != 1:
            raise Hdf5Unsupported(f'object header version {ver}')
        out, blocks, seen = [], [(addr + 16, size)], {addr + 16}
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self.u(pos, 'HHB')
                body = pos + 8
                if mtype == 0x10:                                     # continuation
                    caddr, clen = self.u(body, 'QQ')
                    if caddr in seen:
                        raise Hdf5Unsupported('cyclic object-header continuation')
                    seen.add(caddr)
                    blocks.append((caddr, clen))
                out.append((mtype, body, msize, flags))
                pos = body + msize
        return out

    # ---- groups (symbol table)
    def _heap_name(self, heap, off):
        if self.at(heap, 4) != b'HEAP':
            raise Hdf5Unsupported('bad local heap')
        data_addr, = self.u(heap + 24, 'Q')
        a = self.base + data_addr + off
        end = self.buf.find(b'\x00', a) if 0 <= a < len(self.buf) else -1
        if end < 0:
            raise Hdf5Unsupported('unterminated name in the local heap (truncated or corrupt)')
        return self.buf[a:end].decode('ascii', 'replace')

    def _group_nodes(self, btree, seen=None, depth=0):
        seen = set() if seen is None else seen
        if btree in seen or depth > MAX_DEPTH:
            raise Hdf5Unsupported('cyclic or too deep group B-tree')
        seen.add(btree)
        if self.at(btree, 4) != b'TREE':
            raise Hdf5Unsupported('bad group B-tree node')
        ntype, level, used = self.u(btree + 4, 'BBH')
        if ntype != 0:
            raise Hdf5Unsupported('group B-tree node type')
        pos = btree + 8 + 16
        for i in range(used):
            child, = self.u(pos + 8 + i * 16, 'Q')              # key(8) child(8) key child ... key
            if level > 0:
                yield from self._group_nodes(child, seen, depth + 1)
            else:
                yield child

    def members(self, ent=None):
        """{name: object header address} of a group given by its symbol-table entry (default: the root group)"""
        ent = ent or self.root
        if 'btree' not in ent:
            for mtype, body, _sz, _fl in self.messages(ent['header']):
                if mtype == 0x11:
                    ent = dict(ent, btree=self.u(body, 'Q')[0], heap=self.u(body + 8, 'Q')[0])
                    break
            else:
                raise Hdf5Unsupported('group without a symbol table (new-style links)')
        out = {}
        for snod in self._group_nodes(ent['btree']):
            if self.at(snod, 4) != b'SNOD':
                raise Hdf5Unsupported('bad symbol node')
            n, = self.u(snod + 6, 'H')
            for i in range(n):
                e = self._symbol_entry(self.base + snod + 8 + i * 40)
                out[self._heap_name(ent['heap'], e['name_off'])] = e['header']
        return out

    # ---- datasets
    def dataset(self, addr):
        shape = dtype = layout = None
        filters = []
        for mtype, body, msize, _fl in self.messages(addr):
            if mtype == 0x01:
                ver, rank, flags = self.u(body, 'BBB')
                p = body + (8 if ver == 1 else 4)
                if ver not in (1, 2):
                    raise Hdf5Unsupported(f'dataspace version {ver}')
                shape = tuple(self.u(p, f'{rank}Q')) if rank else ()
            elif mtype == 0x03:
                cv, b0, _b1, _b2, size = self.u(body, 'BBBBI')
                cls = cv & 15
                if b0 & 1:
                    raise Hdf5Unsupported('big-endian datatype')
                if size not in (1, 2, 4, 8) or (cls == 1 and size == 1):
                    raise Hdf5Unsupported(f'datatype of {size} bytes')
                if cls == 0:
                    dtype = np.dtype(('<i' if b0 & 8 else '<u') + str(size))
                elif cls == 1:
                    dtype = np.dtype('<f' + str(size))
                else:
                    raise Hdf5Unsupported(f'datatype class {cls} (only integers and floats)')
            elif mtype == 0x08:
                ver, cls = self.u(body, 'BB')
                if ver != 3:
                    raise Hdf5Unsupported(f'data layout version {ver}')
                if cls == 0:
                    n, = self.u(body + 2, 'H')
                    layout = ('compact', body + 4, n)
                elif cls == 1:
                    a, n = self.u(body + 2, 'QQ')
                    layout = ('contiguous', a, n)
                elif cls == 2:
                    nd, = self.u(body + 2, 'B')
                    bt, = self.u(body + 3, 'Q')
                    layout = ('chunked', bt, self.u(body + 11, f'{nd}I'))
                else:
                    raise Hdf5Unsupported('virtual layout')
            elif mtype == 0x0B:
                ver, nf = self.u(body, 'BB')
                p = body + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid, = self.u(p, 'H')
                    if ver == 1 or fid >= 256:
                        nlen, _flags, ncv = self.u(p + 2, 'HHH')
                        p += 8 + ((nlen + 7) // 8 * 8 if ver == 1 else nlen)
                    else:
                        _flags, ncv = self.u(p + 2, 'HH')
                        nlen = 0
                        p += 6
                    cvs = self.u(p, f'{ncv}I')
                    p += 4 * ncv + (4 if ver == 1 and ncv % 2 else 0)
                    filters.append((fid, cvs))
        if shape is None or dtype is None or layout is None:
            raise Hdf5Unsupported('not a simple dataset')
        n = 1
        for d in shape:
            n *= int(d)
        if n * dtype.itemsize > max(1 << 30, 1000 * len(self.buf)):
            raise Hdf5Unsupported(f'dataset of {n} elements in a file of {len(self.buf)} bytes (corrupt dataspace?)')
        if layout[0] == 'compact':
            raw = bytes(self.at(layout[1], layout[2]))
            return np.frombuffer(raw, dtype, n).reshape(shape).copy()
        if layout[0] == 'contiguous':
            if layout[1] == UNDEF:
                return np.zeros(shape, dtype)
            return np.frombuffer(bytes(self.at(layout[1], n * dtype.itemsize)), dtype, n).reshape(shape).copy()
        bt, cdims = layout[1], layout[2]
        chunk = tuple(cdims[:-1])
        if len(chunk) != len(shape) or cdims[-1] != dtype.itemsize:
            raise Hdf5Unsupported('chunk dimensionality')
        out = np.zeros(shape, dtype)
        if bt != UNDEF:
            for offs, fmask, caddr, csize in self._chunks(bt, len(shape)):
                raw = bytes(self.at(caddr, csize))
                for k, (fid, cvs) in reversed(list(enumerate(filters))):
                    if fmask & (1 << k):
                        continue
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:
                        es = cvs[0] if cvs else dtype.itemsize
                        raw = np.frombuffer(raw, np.uint8).reshape(es, -1).T.tobytes()
                    else:
                        raise Hdf5Unsupported(f'filter {fid}')
                block = np.frombuffer(raw, dtype, int(np.prod(chunk))).reshape(chunk)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, shape))
                out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def _chunks(self, node, rank, seen=None, depth=0):
        seen = set() if seen is None else seen
        if node in seen or depth > MAX_DEPTH:
            raise Hdf5Unsupported('cyclic or too deep chunk B-tree')
        seen.add(node)
        if self.at(node, 4) != b'TREE':
            raise Hdf5Unsupported('bad chunk B-tree node')
        ntype, level, used = self.u(node + 4, 'BBH')
        if ntype != 1:
            raise Hdf5Unsupported('chunk B-tree node type')
        ksz = 8 + 8 * (rank + 1)
        pos = node + 24
        for i in range(used):
            k = pos + i * (ksz + 8)
            csize, fmask = self.u(k, 'II')
            offs = self.u(k + 8, f'{rank}Q')
            child, = self.u(k + ksz, 'Q')
            if level > 0:
                yield from self._chunks(child, rank, seen, depth + 1)
            else:
                yield offs, fmask, child, csize


def read_mat73(path, names=None):
    """{variable: ndarray (axes as stored = reversed MATLAB order)} of the numeric root-level variables of a v7.3 MAT-file;
    `names`: only these (missing ones are skipped).  Non-numeric variables (cells, structs, chars: groups / references) are
    skipped."""
    try:
        f = _File(path)
        members = f.members()
    except _DAMAGE as e:
        raise Hdf5Unsupported(f'{path}: truncated or corrupt HDF5 structure ({type(e).__name__}: {e})') from e
    out = {}
    for name, addr in members.items():
        if name.startswith('#') or (names is not None and name not in names):
            continue
        try:
            out[name] = f.dataset(addr)
        except _DAMAGE as e:
            # whatever a damaged file trips over inside the decoder surfaces as the one documented exception
            if names is not None:
                raise Hdf5Unsupported(f'{path}: variable {name}: truncated or corrupt ({type(e).__name__}: {e})') from e
        except Hdf5Unsupported:
            if names is not None:
                raise
    return out
