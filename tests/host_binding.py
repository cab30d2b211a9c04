"""ctypes view of the library's GPU-free host entry points (xe_hosttest.c):
filter compiler, IP-list flattener, datagram pre-pass.  No flow results are
computed on the CPU by the product; these only expose its host-side planning."""
import ctypes as C

import numpy as np

import os

import xenoeye_amd

HOSTTEST_LIB = os.path.join(os.path.dirname(os.path.abspath(xenoeye_amd.__file__)), "libxe_hosttest.so")
_lib = None


def load_hosttest():
    """the test hooks of xe_hosttest.c: a library of their own, beside the product library"""
    global _lib
    if _lib is None:
        if not os.path.exists(HOSTTEST_LIB):
            raise RuntimeError("%s is missing: run `python __graft_entry__.py` (build()) first" % HOSTTEST_LIB)
        _lib = C.CDLL(HOSTTEST_LIB)
    return _lib


class ListDev(C.Structure):
    _fields_ = [("root4", C.c_uint32), ("root6", C.c_uint32), ("nodes4", C.c_uint32), ("nodes6", C.c_uint32),
                ("bit", C.c_uint32), ("chains4", C.c_uint32), ("chains6", C.c_uint32), ("rbits", C.c_uint32)]


class Set(C.Structure):
    _fields_ = [("data_off", C.c_uint64), ("first_rec", C.c_uint32), ("nrec", C.c_uint32), ("avail", C.c_int32),
                ("stride", C.c_uint32), ("tmpl", C.c_uint32), ("dgram_left", C.c_int32),
                ("exporter", C.c_uint32), ("flags", C.c_uint32)]


class Exporter(C.Structure):
    _fields_ = [("dev_ip", C.c_uint8 * 4), ("dev_id", C.c_uint8 * 4), ("sampling_rate", C.c_uint32),
                ("device", C.c_int32)]


TERM = 0x80000000


class Host:
    def __init__(self):
        L = self.L = load_hosttest()
        L.xe_ht_create.restype = C.c_void_p
        L.xe_ht_free.argtypes = [C.c_void_p]
        L.xe_ht_add_iplist.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.xe_ht_load_iplists.argtypes = [C.c_void_p, C.c_char_p]
        L.xe_ht_filter_dump.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.xe_ht_flatten.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_size_t),
                                    C.POINTER(ListDev), C.c_size_t, C.c_int]
        L.xe_ht_release.argtypes = [C.c_void_p]
        L.xe_ht_nlists.argtypes = [C.c_void_p]
        L.xe_ht_nlists.restype = C.c_size_t
        L.xe_ht_add_device.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int]
        L.xe_ht_prepass.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_int)]
        L.xe_ht_prepass.restype = C.c_void_p
        L.xe_ht_plan_free.argtypes = [C.c_void_p]
        L.xe_ht_field_parse.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        L.xe_ht_set_prepass_threads.argtypes = [C.c_void_p, C.c_int]
        L.xe_ht_set_prepass_threads.restype = None
        for f in ("dg_first_nf", "dg_first_sf"):
            fn = getattr(L, "xe_ht_plan_" + f)
            fn.argtypes = [C.c_void_p]
            fn.restype = C.POINTER(C.c_uint32)
        for f in ("nrec", "nsf", "nsets", "nexporters", "skipped", "oversized", "ndatagrams"):
            fn = getattr(L, "xe_ht_plan_" + f)
            fn.argtypes = [C.c_void_p]
            fn.restype = C.c_size_t
        L.xe_ht_plan_sets.argtypes = [C.c_void_p]
        L.xe_ht_plan_sets.restype = C.POINTER(Set)
        L.xe_ht_plan_exporters.argtypes = [C.c_void_p]
        L.xe_ht_plan_exporters.restype = C.POINTER(Exporter)
        L.xe_ht_list_mos.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.xe_ht_list_devices.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.xe_ht_geodb_set.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        L.xe_ht_geodb_list.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_char_p), C.c_size_t]
        for f in ("undo_begin", "undo_rollback", "undo_end"):
            getattr(L, "xe_ht_" + f).argtypes = [C.c_void_p]
            getattr(L, "xe_ht_" + f).restype = None
        L.xe_ht_undo_count.argtypes = [C.c_void_p]
        L.xe_ht_undo_count.restype = C.c_size_t
        self.h = L.xe_ht_create()

    def close(self):
        if self.h:
            self.L.xe_ht_free(self.h)
            self.h = None

    def add_iplist(self, name, text):
        if isinstance(text, str):
            text = text.encode()
        assert self.L.xe_ht_add_iplist(self.h, name.encode(), text, len(text)) == 0

    def load_iplists(self, d):
        return self.L.xe_ht_load_iplists(self.h, d.encode())

    def geodb_set(self, kind, family, image):
        return self.L.xe_ht_geodb_set(self.h, kind, family, bytes(image), len(image))

    def geodb_list(self, func, geo_field, family, ranges=(), strs=()):
        """index of the list made of the database for an asn (func 10) / asd (11) / geo (9) term"""
        n = max(len(ranges), len(strs))
        lo = (C.c_int32 * max(n, 1))(*[r[0] for r in ranges])
        hi = (C.c_int32 * max(n, 1))(*[r[1] for r in ranges])
        ss = (C.c_char_p * max(n, 1))(*[x.encode() for x in strs])
        return self.L.xe_ht_geodb_list(self.h, func, geo_field, family, lo, hi, ss, n)

    def undo_begin(self):
        self.L.xe_ht_undo_begin(self.h)

    def undo_count(self):
        return self.L.xe_ht_undo_count(self.h)

    def undo_rollback(self):
        self.L.xe_ht_undo_rollback(self.h)

    def undo_end(self):
        self.L.xe_ht_undo_end(self.h)

    def filter_dump(self, text):
        """(rc, dump, err)"""
        out = C.create_string_buffer(1 << 16)
        err = C.create_string_buffer(1024)
        rc = self.L.xe_ht_filter_dump(self.h, text.encode(), out, len(out), err, len(err))
        return rc, out.value.decode(), err.value.decode(errors="replace")

    def flatten(self, root_bits=0):
        """(words, per list (root4, root6, nodes4, nodes6, bit, chains4, chains6)); the roots are
        Root numbers that remember how many address bits index them"""
        n = self.L.xe_ht_nlists(self.h)
        dev = (ListDev * max(n, 1))()
        words = C.POINTER(C.c_uint32)()
        nw = C.c_size_t()
        assert self.L.xe_ht_flatten(self.h, C.byref(words), C.byref(nw), dev, n, root_bits) == 0
        arr = np.ctypeslib.as_array(words, shape=(max(nw.value, 1),)).copy() if nw.value else np.zeros(1, np.uint32)
        self.L.xe_ht_release(words)
        return arr, [(Root(d.root4, d.rbits & 0xFF), Root(d.root6, (d.rbits >> 8) & 0xFF), d.nodes4, d.nodes6, d.bit,
                      d.chains4, d.chains6) for d in dev[:n]]

    def field_parse(self, text):
        out = (C.c_int * 9)()
        err = C.create_string_buffer(1100)
        rc = self.L.xe_ht_field_parse(text.encode(), out, err, len(err))
        return rc, tuple(out), err.value.decode(errors="replace")

    def prepass(self, batch, threads=None):
        if threads is not None:
            self.L.xe_ht_set_prepass_threads(self.h, int(threads))
        data, offs, src, kinds = batch.packed()
        rc = C.c_int()
        p = self.L.xe_ht_prepass(self.h, data.ctypes.data, offs.ctypes.data, src.ctypes.data, kinds.ctypes.data,
                                 len(offs) - 1, C.byref(rc))
        assert rc.value == 0 and p, "pre-pass failed: %d" % rc.value
        L = self.L
        ns = L.xe_ht_plan_nsets(p)
        sets = [L.xe_ht_plan_sets(p)[i] for i in range(ns)]
        nd = L.xe_ht_plan_ndatagrams(p)
        out = dict(nrec=L.xe_ht_plan_nrec(p), nsf=L.xe_ht_plan_nsf(p), skipped=L.xe_ht_plan_skipped(p),
                   oversized=L.xe_ht_plan_oversized(p),
                   dg_first_nf=[L.xe_ht_plan_dg_first_nf(p)[i] for i in range(nd + 1)],
                   dg_first_sf=[L.xe_ht_plan_dg_first_sf(p)[i] for i in range(nd + 1)],
                   sets=[dict(first_rec=s.first_rec, nrec=s.nrec, avail=s.avail, stride=s.stride, tmpl=s.tmpl,
                              flags=s.flags, data_off=s.data_off, exporter=s.exporter, dgram_left=s.dgram_left)
                         for s in sets],
                   exporters=[dict(dev_ip=bytes(e.dev_ip), dev_id=bytes(e.dev_id), rate=e.sampling_rate,
                                   device=e.device)
                              for e in (L.xe_ht_plan_exporters(p)[i] for i in range(L.xe_ht_plan_nexporters(p)))])
        L.xe_ht_plan_free(p)
        return out

    def list_mos(self, d):
        out = C.create_string_buffer(1 << 18)
        rc = self.L.xe_ht_list_mos(d.encode(), out, len(out))
        rows = [ln.split("\t") for ln in out.value.decode().splitlines()]
        return rc, [(r[0], int(r[1]), r[2]) for r in rows]

    def list_devices(self, path):
        out = C.create_string_buffer(1 << 16)
        rc = self.L.xe_ht_list_devices(path.encode(), out, len(out))
        return rc, [ln.split("\t") for ln in out.value.decode().splitlines()]


CHAIN = 0x40000000
CHAIN_WORDS = 8


NODE_BITS = int(load_hosttest().xe_ht_node_bits())


def _resolve(e, addr, nbits, depth):
    """an entry that carries a short chain in itself -> 'ends here' with the
    answers; `depth` = address bits the entry stands behind"""
    if (e >> 30) != 3:
        return e
    if e & 0x20000000:
        base, lst, ln, path, sl = 0, (e >> 16) & 7, (e >> 20) & 31, e & 0xFFFF, (e >> 19) & 1
    else:
        base, lst, ln, path, sl = e & 0xFF, (e >> 8) & 7, (e >> 11) & 15, (e >> 15) & 0x1FFF, (e >> 28) & 1
    left = ((addr << (128 - nbits)) << depth) & ((1 << 128) - 1)      # bits from `depth`, left-aligned
    d = ((left >> (128 - 16)) >> (16 - ln)) ^ path
    m = (d == 0) or (sl and ((d >> (ln - 1)) & 1))
    return TERM | base | (int(m) << lst)


class Root(int):
    """word offset of a root table + the number of address bits it is indexed with (16 or 24)"""
    def __new__(cls, off, rbits):
        o = int.__new__(cls, off)
        o.rbits = rbits or 16
        return o


def walk_flat(words, root, nodes, addr_bytes, bit=0, chains=0):
    """Reference walker of the flattened list format (tests only): the
    group's root table over the first 16 (or 24) bits, then one
    directly indexed node per NODE_BITS address bits; a tail of one prefix is a
    chain record or sits in the entry itself; `bit` selects the list inside the
    group's answer bits."""
    if not root:
        return 0
    nbits = 8 * len(addr_bytes)
    addr = int.from_bytes(addr_bytes, "big")
    rbits = getattr(root, "rbits", 16)
    top = addr >> (nbits - rbits)
    e = int(words[root + top])
    e = _resolve(e, addr, nbits, rbits)
    depth = rbits
    while not (e & TERM) and depth < nbits:
        if e & CHAIN:
            c = chains + (e & ~CHAIN) * CHAIN_WORDS
            meta = int(words[c])
            ln, lst, sl = (meta >> 11) & 127, (meta >> 8) & 7, (meta >> 18) & 1
            pat = (int(words[c + 3]) << 96) | (int(words[c + 2]) << 64) | (int(words[c + 5]) << 32) | int(words[c + 4])
            rest = (addr << (128 - nbits + depth)) & ((1 << 128) - 1)   # address bits from `depth`, left-aligned
            m = (((rest ^ pat) >> (128 - ln)) == 0) or (sl and ((rest ^ pat) >> 127))
            e = TERM | (meta & 0xFF) | (int(m) << lst)
        else:
            e = int(words[nodes + e * (1 << NODE_BITS) + ((addr >> (nbits - depth - NODE_BITS)) & ((1 << NODE_BITS) - 1))])
            e = _resolve(e, addr, nbits, depth + NODE_BITS)
        depth += NODE_BITS
    return ((e >> bit) & 1) if (e & TERM) else 0
