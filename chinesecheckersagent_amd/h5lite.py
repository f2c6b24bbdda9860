"""Minimal pure-Python reader (and writer) for the HDF5 subset Keras 2.1.6 / h5py write
(SURVEY.md H9): superblock v0, version-1 object headers, symbol-table groups (B-tree v1 + SNOD +
local heap), contiguous unfiltered little-endian datasets.  Enough to load the reference's
`*.h5` weight files (model.py:46-48 -> keras load_weights) and to write `board_x / pi_y / v_y`
training files (utils.py:48-56) without h5py, which is not installed on the target image.
"""
import struct

import numpy as np

SIG = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


class H5File(object):
    def __init__(self, path):
        with open(path, 'rb') as f:
            self.buf = f.read()
        b = self.buf
        if b[:8] != SIG:
            raise H5Error('not an HDF5 file: %s' % path)
        ver = b[8]
        if ver not in (0, 1):
            raise H5Error('superblock version %d not supported' % ver)
        self.O, self.L = b[13], b[14]
        if self.O != 8 or self.L != 8:
            raise H5Error('only 8-byte offsets/lengths supported')
        p = 24 if ver == 0 else 28
        self.base = self._u64(p)
        p += 32                                   # base, free-space, eof, driver
        # root symbol table entry
        self.root_header = self._u64(p + 8)
        cache = self._u32(p + 16)
        self.root_btree, self.root_heap = (self._u64(p + 24), self._u64(p + 32)) if cache == 1 else (None, None)

    # ---- primitives
    def _u16(self, p): return struct.unpack_from('<H', self.buf, p)[0]
    def _u32(self, p): return struct.unpack_from('<I', self.buf, p)[0]
    def _u64(self, p): return struct.unpack_from('<Q', self.buf, p)[0]

    def _messages(self, addr):
        """yield (type, data offset, size) of a version-1 object header incl. continuation blocks"""
        b = self.buf
        if b[addr] != 1:
            raise H5Error('object header version %d not supported' % b[addr])
        nmsg = self._u16(addr + 2)
        size = self._u32(addr + 8)
        blocks = [(addr + 16, size)]
        seen = 0
        while blocks and seen < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and seen < nmsg:
                mtype, msize = self._u16(p), self._u16(p + 2)
                data = p + 8
                seen += 1
                if mtype == 0x10:
                    blocks.append((self._u64(data), self._u64(data + 8)))
                else:
                    yield mtype, data, msize
                p = data + msize

    def _heap_name(self, heap_addr, off):
        b = self.buf
        if b[heap_addr:heap_addr + 4] != b'HEAP':
            raise H5Error('bad local heap')
        data = self._u64(heap_addr + 24)
        end = b.index(b'\0', data + off)
        return b[data + off:end].decode()

    def _btree_entries(self, addr, heap):
        """yield (name, object header address) of a group B-tree (v1, node type 0)"""
        b = self.buf
        if b[addr:addr + 4] != b'TREE':
            raise H5Error('bad B-tree node')
        level, used = b[addr + 5], self._u16(addr + 6)
        p = addr + 8 + 16                          # skip siblings
        for i in range(used):
            child = self._u64(p + 8)
            p += 16
            if level > 0:
                for x in self._btree_entries(child, heap):
                    yield x
            else:
                if b[child:child + 4] != b'SNOD':
                    raise H5Error('bad symbol node')
                n = self._u16(child + 6)
                q = child + 8
                for j in range(n):
                    yield self._heap_name(heap, self._u64(q)), self._u64(q + 8)
                    q += 40

    def _group_tables(self, header):
        for mtype, data, size in self._messages(header):
            if mtype == 0x11:
                return self._u64(data), self._u64(data + 8)
        return None

    def children(self, header=None):
        """dict name -> object header address of a group"""
        header = self.root_header if header is None else header
        t = self._group_tables(header)
        if t is None:
            return {}
        return dict(self._btree_entries(t[0], t[1]))

    def _dataset(self, header):
        shape = dtype = None
        addr = size = None
        chunked = None
        for mtype, data, msize in self._messages(header):
            b = self.buf
            if mtype == 0x01:                                   # dataspace
                ver, rank, flags = b[data], b[data + 1], b[data + 2]
                p = data + (8 if ver == 1 else 4)
                shape = tuple(self._u64(p + 8 * i) for i in range(rank))
            elif mtype == 0x03:                                 # datatype
                cls = b[data] & 0x0F
                bits0 = b[data + 1]
                sz = self._u32(data + 4)
                if bits0 & 1:
                    raise H5Error('big-endian data not supported')
                if cls == 1:
                    dtype = {2: '<f2', 4: '<f4', 8: '<f8'}[sz]
                elif cls == 0:
                    signed = (bits0 >> 3) & 1
                    dtype = ('<i%d' if signed else '<u%d') % sz
                else:
                    raise H5Error('datatype class %d not supported' % cls)
            elif mtype == 0x08:                                 # layout
                ver = b[data]
                if ver == 3 and b[data + 1] == 2:               # chunked: B-tree v1 of raw data chunks
                    nd = b[data + 2]
                    chunked = (self._u64(data + 3), tuple(self._u32(data + 11 + 4 * i) for i in range(nd)))
                    addr = 0
                elif ver == 3:
                    if b[data + 1] != 1:
                        raise H5Error('only contiguous and chunked layouts supported')
                    addr, size = self._u64(data + 2), self._u64(data + 10)
                else:
                    raise H5Error('layout version %d not supported' % ver)
            elif mtype == 0x0B:
                raise H5Error('filtered data not supported')
        if shape is None or dtype is None or addr is None:
            return None
        if chunked is not None:
            return self._read_chunked(shape, dtype, chunked[0], chunked[1])
        if addr == UNDEF:
            return np.zeros(shape, dtype=dtype)
        n = int(np.prod(shape)) if shape else 1
        return np.frombuffer(self.buf, dtype=dtype, count=n, offset=self.base + addr).reshape(shape).copy()

    def _chunks(self, addr, nd):
        """yield (offsets, address) of the raw data chunks below a type-1 B-tree node"""
        b = self.buf
        if addr == UNDEF:
            return
        if b[addr:addr + 4] != b'TREE' or b[addr + 4] != 1:
            raise H5Error('bad chunk B-tree node')
        level, used = b[addr + 5], self._u16(addr + 6)
        ks = 8 + 8 * nd
        p = addr + 24
        for i in range(used):
            key, child = p + i * (ks + 8), self._u64(p + i * (ks + 8) + ks)
            if level > 0:
                for x in self._chunks(child, nd):
                    yield x
            else:
                if self._u32(key + 4) != 0:
                    raise H5Error('filtered chunks not supported')
                yield tuple(self._u64(key + 8 + 8 * j) for j in range(nd - 1)), child

    def _read_chunked(self, shape, dtype, btree, cdims):
        nd = len(cdims)
        if nd != len(shape) + 1:
            raise H5Error('chunk rank does not match the dataspace')
        cshape = tuple(cdims[:-1])
        out = np.zeros(shape, dtype=dtype)
        n = int(np.prod(cshape))
        for off, addr in self._chunks(btree, nd):
            c = np.frombuffer(self.buf, dtype=dtype, count=n, offset=self.base + addr).reshape(cshape)
            sl = tuple(slice(o, min(o + cs, d)) for o, cs, d in zip(off, cshape, shape))
            out[sl] = c[tuple(slice(0, s_.stop - s_.start) for s_ in sl)]
        return out

    def get(self, path):
        """dataset at 'a/b/c' as a numpy array"""
        header = self.root_header
        parts = [x for x in path.split('/') if x]
        for name in parts:
            ch = self.children(header)
            if name not in ch:
                raise KeyError(path)
            header = ch[name]
        out = self._dataset(header)
        if out is None:
            raise KeyError('%s is not a dataset' % path)
        return out

    def walk(self, header=None, prefix=''):
        """yield (path, array) of every dataset"""
        for name, h in sorted(self.children(header).items()):
            if self._group_tables(h) is not None:
                for x in self.walk(h, prefix + name + '/'):
                    yield x
            else:
                d = self._dataset(h)
                if d is not None:
                    yield prefix + name, d


# ---------------------------------------------------------------------------------------------
# writer: a flat file of contiguous datasets in the root group (what utils.save_train_data makes:
# H.create_dataset('board_x' | 'pi_y' | 'v_y', data=...), utils.py:48-56)

def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == 'f':
        size = dt.itemsize
        # IEEE little-endian float: class 1, version 1; bit field: byte order 0, pad 0, mantissa norm 2 (implied)
        exp_bits, mant_bits, bias = {4: (8, 23, 127), 8: (11, 52, 1023)}[size]
        body = struct.pack('<BBBBI', 0x11, 0x20, 8 * size - 1, 0, size)
        body += struct.pack('<HHBBBBI', 0, 8 * size, mant_bits, exp_bits, 0, mant_bits, bias)
        return body
    if dt.kind in 'iu':
        size = dt.itemsize
        body = struct.pack('<BBBBI', 0x10, 0x08 if dt.kind == 'i' else 0x00, 0, 0, size)
        body += struct.pack('<HH', 0, 8 * size)
        return body
    raise H5Error('dtype %s not supported' % dt)


def write_datasets(path, datasets):
    """datasets: ordered list of (name, ndarray).  Little-endian, contiguous, no attributes."""
    names = [n for n, _ in datasets]
    arrays = [np.ascontiguousarray(a) for _, a in datasets]
    arrays = [a.astype(a.dtype.newbyteorder('<')) if a.dtype.byteorder == '>' else a for a in arrays]
    # layout: superblock (96) | root header | heap | btree | snod | dataset headers | data
    pos = 96
    root_hdr = pos
    pos += 16 + 24                                   # header + one symbol-table message (8 + 16)
    heap = pos
    heap_data_size = 8 + sum(len(n) + 1 for n in names)
    heap_data_size += -heap_data_size % 8
    heap_data_size = max(heap_data_size, 24)
    pos += 32
    heap_data = pos
    pos += heap_data_size
    btree = pos
    K, LEAF_K = 16, 4                                # group internal / leaf node K of the superblock (HDF5 defaults)
    pos += 8 + 16 + (2 * K + 1) * 8 + 2 * K * 8      # node header, siblings, keys, children
    snod = pos
    pos += 8 + 2 * LEAF_K * 40
    if len(names) > 2 * LEAF_K:
        raise H5Error('too many datasets for one symbol node')
    hdrs = []
    msgs = []
    for a in arrays:
        m = b''
        ds = struct.pack('<BBBB4x', 1, a.ndim, 0, 0) + b''.join(struct.pack('<Q', d) for d in a.shape)
        m += struct.pack('<HHB3x', 0x01, len(_pad8(ds)), 0) + _pad8(ds)
        dtm = _dtype_msg(a.dtype)
        m += struct.pack('<HHB3x', 0x03, len(_pad8(dtm)), 1) + _pad8(dtm)
        msgs.append(m)
        hdrs.append(pos)
        pos += 16 + len(m) + 8 + 24                  # + layout message (8 + pad8(18) = 24)
    pos += -pos % 8
    data_addr = []
    for a in arrays:
        data_addr.append(pos)
        pos += a.nbytes + (-a.nbytes % 8)
    eof = pos
    meta_end = data_addr[0] if arrays else eof
    out = bytearray(meta_end)                         # metadata only: the arrays go to the file straight from their buffers
    # superblock v0
    out[0:8] = SIG
    struct.pack_into('<BBBBBBBBHHI', out, 8, 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, K, 0)
    struct.pack_into('<QQQQ', out, 24, 0, UNDEF, eof, UNDEF)
    struct.pack_into('<QQII', out, 56, 0, root_hdr, 1, 0)
    struct.pack_into('<QQ', out, 80, btree, heap)
    # root object header: symbol table message
    struct.pack_into('<BBHII4x', out, root_hdr, 1, 0, 1, 1, 24)
    struct.pack_into('<HHB3xQQ', out, root_hdr + 16, 0x11, 16, 0, btree, heap)
    # local heap
    out[heap:heap + 4] = b'HEAP'
    struct.pack_into('<B3xQQQ', out, heap + 4, 0, heap_data_size, 1, heap_data)     # 1 = H5HL_FREE_NULL: no free block
    order = sorted(range(len(names)), key=lambda i: names[i])
    offs = {}
    p = 8
    for i in order:
        nb = names[i].encode() + b'\0'
        out[heap_data + p:heap_data + p + len(nb)] = nb
        offs[i] = p
        p += len(nb)
    # B-tree leaf with one child (the symbol node)
    out[btree:btree + 4] = b'TREE'
    struct.pack_into('<BBHQQ', out, btree + 4, 0, 0, 1, UNDEF, UNDEF)
    struct.pack_into('<QQQ', out, btree + 24, 0, snod, offs[order[-1]] if order else 0)
    out[snod:snod + 4] = b'SNOD'
    struct.pack_into('<BBH', out, snod + 4, 1, 0, len(names))
    q = snod + 8
    for i in order:
        struct.pack_into('<QQII16x', out, q, offs[i], hdrs[i], 0, 0)
        q += 40
    # dataset headers + data
    for i, a in enumerate(arrays):
        h = hdrs[i]
        body = msgs[i]
        lay = struct.pack('<BBQQ', 3, 1, data_addr[i], a.nbytes)
        body += struct.pack('<HHB3x', 0x08, len(_pad8(lay)), 0) + _pad8(lay)
        struct.pack_into('<BBHII4x', out, h, 1, 0, 3, 1, len(body))
        out[h + 16:h + 16 + len(body)] = body
    with open(path, 'wb') as f:
        f.write(out)
        for i, a in enumerate(arrays):
            f.seek(data_addr[i])
            if a.nbytes:
                f.write(memoryview(a.reshape(-1)).cast('B'))
        f.truncate(eof)


# ---------------------------------------------------------------------------------------------
# streaming writer: the same flat file of datasets in the root group, but CHUNKED along the first axis (layout class 2 with
# a version-1 B-tree of raw data chunks per dataset), so that rows can be appended while a run is still producing them --
# the self-play generator streams finished games to the training file while the GPU plays on, host memory bounded by
# one chunk.  h5py / the reference's train.py read it like any dataset (f['board_x'], np.array(f.get('pi_y'))).

class StreamWriter(object):
    """specs: ordered [(name, row shape, dtype)]; append(list of arrays with the same number of rows); close().
    Chunks of `chunk_rows` rows are written as they fill up; all metadata that depends on the row count (dataspace, chunk
    B-trees) is written by close().  The file appears under its name when close() has finished (temporary name before)."""

    ISTORE_K = 32                      # B-tree node K of chunked storage under a version-0 superblock (the library's default)

    def __init__(self, path, specs, chunk_rows=4096):
        import os
        self.path, self.tmp = path, '%s.tmp%d' % (path, os.getpid())
        self.names = [n for n, _, _ in specs]
        self.row_shapes = [tuple(int(x) for x in sh) for _, sh, _ in specs]
        self.dtypes = [np.dtype(dt).newbyteorder('<') for _, _, dt in specs]
        self.chunk_rows = int(chunk_rows)
        self.rows = 0                                    # rows appended
        self.flushed = 0                                 # rows written out as full chunks
        self._buf = [np.zeros((self.chunk_rows,) + sh, dtype=dt) for sh, dt in zip(self.row_shapes, self.dtypes)]
        self._fill = 0
        self._chunks = [[] for _ in specs]               # per dataset: file address of every chunk, in row order
        if len(self.names) > 8:
            raise H5Error('too many datasets for one symbol node')
        # fixed-size metadata at the start: superblock | root header | heap | group B-tree | symbol node | dataset headers
        pos = 96
        self.root_hdr = pos; pos += 16 + 24
        self.heap = pos
        self.heap_data_size = max(24, (8 + sum(len(n) + 1 for n in self.names) + 7) // 8 * 8)
        pos += 32
        self.heap_data = pos; pos += self.heap_data_size
        self.btree = pos; pos += 8 + 16 + (2 * 16 + 1) * 8 + 2 * 16 * 8
        self.snod = pos; pos += 8 + 2 * 4 * 40
        self.hdrs = []
        for sh in self.row_shapes:
            self.hdrs.append(pos)
            pos += 16 + self._header_body_size(len(sh) + 1)
        pos += -pos % 8
        self.pos = pos                                   # where the next chunk goes
        self.f = open(self.tmp, 'wb')
        self.f.write(b'\0' * pos)

    @staticmethod
    def _header_body_size(rank):
        ds = 8 + 8 * rank
        lay = 3 + 8 + 4 * (rank + 1)
        return (8 + ds) + (8 + 24) + (8 + (lay + 7) // 8 * 8)       # dataspace, datatype (16-24 bytes padded to 24), layout

    def append(self, arrays):
        n = len(arrays[0])
        done = 0
        while done < n:
            take = min(n - done, self.chunk_rows - self._fill)
            for b, a in zip(self._buf, arrays):
                b[self._fill:self._fill + take] = a[done:done + take]
            self._fill += take
            done += take
            if self._fill == self.chunk_rows:
                self._emit()
        self.rows += n

    def _emit(self):
        for i, b in enumerate(self._buf):
            self._chunks[i].append(self.pos)
            self.f.write(memoryview(b.reshape(-1)).cast('B'))
            self.pos += b.nbytes
        self.flushed += self._fill
        self._fill = 0

    def _btree(self, out, i):
        """the chunk B-tree of dataset i appended to `out` (a bytearray positioned at file offset self.pos); -> root address"""
        nd = len(self.row_shapes[i]) + 2                                  # rank of the dataset + the element-size dimension
        ks = 8 + 8 * nd
        node_size = 24 + (2 * self.ISTORE_K + 1) * ks + 2 * self.ISTORE_K * 8
        nbytes = self._buf[i].nbytes
        level = 0
        entries = [(r * self.chunk_rows, addr) for r, addr in enumerate(self._chunks[i])]      # (first row, child address)
        if not entries:
            return UNDEF
        end_row = len(entries) * self.chunk_rows
        while True:
            nodes = []
            for g in range(0, len(entries), 2 * self.ISTORE_K):
                grp = entries[g:g + 2 * self.ISTORE_K]
                addr = self.pos + len(out)
                node = bytearray(node_size)
                node[0:4] = b'TREE'
                struct.pack_into('<BBHQQ', node, 4, 1, level, len(grp), UNDEF, UNDEF)
                p = 24
                for row, child in grp:
                    struct.pack_into('<II', node, p, nbytes, 0)
                    struct.pack_into('<Q', node, p + 8, row)                 # offsets: (row, 0, ..., 0)
                    struct.pack_into('<Q', node, p + ks, child)
                    p += ks + 8
                nxt = entries[g + len(grp)][0] if g + len(grp) < len(entries) else end_row
                struct.pack_into('<II', node, p, 0, 0)
                struct.pack_into('<Q', node, p + 8, nxt)                     # the key behind the last child
                nodes.append((grp[0][0], addr))
                out += node
            # siblings of this level
            for j, (_, addr) in enumerate(nodes):
                off = addr - self.pos
                struct.pack_into('<QQ', out, off + 8, nodes[j - 1][1] if j else UNDEF, nodes[j + 1][1] if j + 1 < len(nodes) else UNDEF)
            if len(nodes) == 1:
                return nodes[0][1]
            entries, level = nodes, level + 1

    def close(self):
        import os
        if self.f is None:
            return self.path
        if self._fill:                                                       # the last, partly filled chunk (its tail is never read)
            for b in self._buf:
                b[self._fill:] = 0
            self._emit()
        tail = bytearray()
        roots = [self._btree(tail, i) for i in range(len(self.names))]
        self.f.write(tail)
        eof = self.pos + len(tail)
        out = bytearray(self.hdrs[-1] + 16 + self._header_body_size(len(self.row_shapes[-1]) + 1) if self.hdrs else self.snod + 8 + 320)
        K, LEAF_K = 16, 4
        out[0:8] = SIG
        struct.pack_into('<BBBBBBBBHHI', out, 8, 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, K, 0)
        struct.pack_into('<QQQQ', out, 24, 0, UNDEF, eof, UNDEF)
        struct.pack_into('<QQII', out, 56, 0, self.root_hdr, 1, 0)
        struct.pack_into('<QQ', out, 80, self.btree, self.heap)
        struct.pack_into('<BBHII4x', out, self.root_hdr, 1, 0, 1, 1, 24)
        struct.pack_into('<HHB3xQQ', out, self.root_hdr + 16, 0x11, 16, 0, self.btree, self.heap)
        out[self.heap:self.heap + 4] = b'HEAP'
        struct.pack_into('<B3xQQQ', out, self.heap + 4, 0, self.heap_data_size, 1, self.heap_data)
        order = sorted(range(len(self.names)), key=lambda i: self.names[i])
        offs, p = {}, 8
        for i in order:
            nb = self.names[i].encode() + b'\0'
            out[self.heap_data + p:self.heap_data + p + len(nb)] = nb
            offs[i] = p
            p += len(nb)
        out[self.btree:self.btree + 4] = b'TREE'
        struct.pack_into('<BBHQQ', out, self.btree + 4, 0, 0, 1, UNDEF, UNDEF)
        struct.pack_into('<QQQ', out, self.btree + 24, 0, self.snod, offs[order[-1]] if order else 0)
        out[self.snod:self.snod + 4] = b'SNOD'
        struct.pack_into('<BBH', out, self.snod + 4, 1, 0, len(self.names))
        q = self.snod + 8
        for i in order:
            struct.pack_into('<QQII16x', out, q, offs[i], self.hdrs[i], 0, 0)
            q += 40
        for i, sh in enumerate(self.row_shapes):
            dims = (self.rows,) + sh
            ds = struct.pack('<BBBB4x', 1, len(dims), 0, 0) + b''.join(struct.pack('<Q', d) for d in dims)
            body = struct.pack('<HHB3x', 0x01, len(_pad8(ds)), 0) + _pad8(ds)
            dtm = _dtype_msg(self.dtypes[i])
            dtm = dtm + b'\0' * (24 - len(dtm))
            body += struct.pack('<HHB3x', 0x03, 24, 1) + dtm
            cd = (self.chunk_rows,) + sh + (self.dtypes[i].itemsize,)
            lay = struct.pack('<BBBQ', 3, 2, len(cd), roots[i]) + b''.join(struct.pack('<I', d) for d in cd)
            body += struct.pack('<HHB3x', 0x08, len(_pad8(lay)), 0) + _pad8(lay)
            assert len(body) == self._header_body_size(len(dims))
            struct.pack_into('<BBHII4x', out, self.hdrs[i], 1, 0, 3, 1, len(body))
            out[self.hdrs[i] + 16:self.hdrs[i] + 16 + len(body)] = body
        self.f.seek(0)
        self.f.write(out)
        self.f.close()
        self.f = None
        os.replace(self.tmp, self.path)
        return self.path

    def abort(self):
        import os
        if self.f is not None:
            self.f.close()
            self.f = None
            if os.path.exists(self.tmp):
                os.remove(self.tmp)


# ---------------------------------------------------------------------------------------------
# writer for nested groups with attributes: the layout of a Keras 2.1.6 `save_weights` file
# (root attrs layer_names / backend / keras_version; one group per layer with attr weight_names;
# datasets at layer/layer/name:0 -- model.py:33-37), so that files written here can be read by
# keras' load_weights / h5py as well as by H5File above.

_LEAF_K, _INT_K = 64, 16           # one symbol node (<= 128 entries) and one B-tree node per group


def _attr_msg(name, value):
    """version-1 attribute message body for a numpy array of fixed-length bytes / floats / ints, or a bytes scalar"""
    if isinstance(value, (bytes, str)):
        value = np.array(value.encode() if isinstance(value, str) else value, dtype='S')
    value = np.asarray(value)
    if value.dtype.kind == 'S':
        dt = struct.pack('<BBBBI', 0x13, 0x01, 0, 0, max(value.dtype.itemsize, 1))          # string, null-padded, ASCII
    else:
        dt = _dtype_msg(value.dtype)
    if value.ndim == 0:
        ds = struct.pack('<BBBB4x', 1, 0, 0, 0)
    else:
        ds = struct.pack('<BBBB4x', 1, value.ndim, 0, 0) + b''.join(struct.pack('<Q', d) for d in value.shape)
    nm = name.encode() + b'\0'
    body = struct.pack('<BBHHH', 1, 0, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + value.tobytes()
    return _pad8(body)


class _Group(object):
    def __init__(self, attrs=None):
        self.attrs = attrs or []           # list of (name, value)
        self.children = []                 # list of (name, _Group | ndarray)


def _layout(node, pos):
    """assign file offsets depth-first; returns the next free offset"""
    if isinstance(node, _Group):
        node.msgs = b''.join(struct.pack('<HHB3x', 0x0C, len(m), 0) + m for m in (_attr_msg(n, v) for n, v in node.attrs))
        node.hdr = pos
        pos += 16 + 24 + len(node.msgs)
        pos += -pos % 8
        names = sorted(n for n, _ in node.children)
        node.heap_size = max(24, 8 + sum(len(n) + 1 for n in names))
        node.heap_size += -node.heap_size % 8
        node.heap = pos; pos += 32
        node.heap_data = pos; pos += node.heap_size
        node.btree = pos; pos += 8 + 16 + (2 * _INT_K + 1) * 8 + 2 * _INT_K * 8
        node.snod = pos; pos += 8 + 2 * _LEAF_K * 40
        if len(node.children) > 2 * _LEAF_K:
            raise H5Error('too many children for one symbol node')
        for _, ch in node.children:
            pos = _layout(ch, pos)
        return pos
    a = node['array']
    ds = struct.pack('<BBBB4x', 1, a.ndim, 0, 0) + b''.join(struct.pack('<Q', d) for d in a.shape)
    dtm = _dtype_msg(a.dtype)
    node['msgs'] = (struct.pack('<HHB3x', 0x01, len(_pad8(ds)), 0) + _pad8(ds) +
                    struct.pack('<HHB3x', 0x03, len(_pad8(dtm)), 1) + _pad8(dtm))
    node['hdr'] = pos
    pos += 16 + len(node['msgs']) + 8 + 24
    pos += -pos % 8
    node['data'] = pos
    pos += a.nbytes + (-a.nbytes % 8)
    return pos


def _emit(node, out):
    if isinstance(node, _Group):
        nmsg = 1 + len(node.attrs)
        struct.pack_into('<BBHII4x', out, node.hdr, 1, 0, nmsg, 1, 24 + len(node.msgs))
        struct.pack_into('<HHB3xQQ', out, node.hdr + 16, 0x11, 16, 0, node.btree, node.heap)
        out[node.hdr + 40:node.hdr + 40 + len(node.msgs)] = node.msgs
        out[node.heap:node.heap + 4] = b'HEAP'
        struct.pack_into('<B3xQQQ', out, node.heap + 4, 0, node.heap_size, 1, node.heap_data)
        order = sorted(range(len(node.children)), key=lambda i: node.children[i][0])
        offs, p = {}, 8
        for i in order:
            nb = node.children[i][0].encode() + b'\0'
            out[node.heap_data + p:node.heap_data + p + len(nb)] = nb
            offs[i] = p
            p += len(nb)
        out[node.btree:node.btree + 4] = b'TREE'
        struct.pack_into('<BBHQQ', out, node.btree + 4, 0, 0, 1 if order else 0, UNDEF, UNDEF)
        struct.pack_into('<QQQ', out, node.btree + 24, 0, node.snod, offs[order[-1]] if order else 0)
        out[node.snod:node.snod + 4] = b'SNOD'
        struct.pack_into('<BBH', out, node.snod + 4, 1, 0, len(order))
        q = node.snod + 8
        for i in order:
            ch = node.children[i][1]
            if isinstance(ch, _Group):      # cache the child's B-tree / heap in the entry like the library does
                struct.pack_into('<QQII', out, q, offs[i], ch.hdr, 1, 0)
                struct.pack_into('<QQ', out, q + 24, ch.btree, ch.heap)
            else:
                struct.pack_into('<QQII16x', out, q, offs[i], ch['hdr'], 0, 0)
            q += 40
        for _, ch in node.children:
            _emit(ch, out)
        return
    a = node['array']
    lay = struct.pack('<BBQQ', 3, 1, node['data'] if a.nbytes else UNDEF, a.nbytes)
    body = node['msgs'] + struct.pack('<HHB3x', 0x08, len(_pad8(lay)), 0) + _pad8(lay)
    struct.pack_into('<BBHII4x', out, node['hdr'], 1, 0, 3, 1, len(body))
    out[node['hdr'] + 16:node['hdr'] + 16 + len(body)] = body
    out[node['data']:node['data'] + a.nbytes] = a.tobytes()


def write_keras_weights(path, layers, backend='tensorflow', keras_version='2.1.6', root=None):
    """layers: ordered list of (layer_name, [(weight_name, ndarray), ...]) in Keras' layer order; weight_name like
    'conv2d_1/kernel:0'.  Layers without weights get an empty weight_names attribute, as Keras writes them.
    root='model_weights' nests everything in that group, where Keras' whole-model files (model.save) keep the weights."""
    width = max(len(n) for n, _ in layers)
    top = _Group([('layer_names', np.array([n.encode() for n, _ in layers], dtype='S%d' % width)),
                   ('backend', backend), ('keras_version', keras_version)])
    for lname, weights in layers:
        if weights:
            w = max(len(n) for n, _ in weights)
            g = _Group([('weight_names', np.array([n.encode() for n, _ in weights], dtype='S%d' % w))])
            sub = {}
            for wname, arr in weights:               # 'conv2d_1/kernel:0' -> group conv2d_1, dataset kernel:0
                parts = wname.split('/')
                node = sub.setdefault(parts[0], _Group())
                node.children.append(('/'.join(parts[1:]), {'array': np.ascontiguousarray(arr)}))
            g.children = list(sub.items())
        else:
            g = _Group([('weight_names', np.zeros((0,), dtype='<f8'))])
        top.children.append((lname, g))
    if root:
        outer = _Group()
        outer.children.append((root, top))
        top = outer
    end = _layout(top, 96)
    out = bytearray(end)
    out[0:8] = SIG
    struct.pack_into('<BBBBBBBBHHI', out, 8, 0, 0, 0, 0, 0, 8, 8, 0, _LEAF_K, _INT_K, 0)
    struct.pack_into('<QQQQ', out, 24, 0, UNDEF, end, UNDEF)
    struct.pack_into('<QQII', out, 56, 0, top.hdr, 1, 0)
    struct.pack_into('<QQ', out, 80, top.btree, top.heap)
    _emit(top, out)
    with open(path, 'wb') as f:
        f.write(bytes(out))
