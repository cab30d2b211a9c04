"""xenoeye_amd: MI355X flow path (Netflow v5/v9/IPFIX/sFlow decode + monitoring
object classification) behind the C ABI of include/xe_gpu.h.

This module is only a ctypes view of libxenoeye_gpu.so for tests, the
benchmark and Python callers; all work happens in the library.  There is no
Python or CPU implementation of the path here: without the built library (or
without a GPU) every entry point raises.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# XE_GPU_LIB: another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("XE_GPU_LIB") or os.path.join(_HERE, "libxenoeye_gpu.so")
FLOW_INFO_SIZE = 2312

_lib = None


class XeError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__("xe_gpu error %d: %s" % (code, detail))


def load_library(path=LIB_PATH):
    """Load the HIP library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python __graft_entry__.py` (build()) first; "
                           "there is no fallback implementation" % path)
    L = C.CDLL(path)
    vp, cp, sz, i32, u32 = C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_uint32
    sig = {
        "xe_gpu_strerror": (cp, [i32]),
        "xe_gpu_last_error": (cp, [vp]),
        "xe_gpu_build_arch": (cp, []),
        "xe_gpu_create": (i32, [C.POINTER(vp), i32]),
        "xe_gpu_destroy": (None, [vp]),
        "xe_gpu_load_iplists": (i32, [vp, cp]),
        "xe_gpu_add_iplist": (i32, [vp, cp, cp, sz]),
        "xe_gpu_add_mo": (i32, [vp, cp, i32, cp]),
        "xe_gpu_mo_host_only": (i32, [vp, i32]),
        "xe_gpu_mo_host_only_reason": (cp, [vp, i32]),
        "xe_gpu_mo_host_gated": (i32, [vp, i32]),
        "xe_gpu_set_port_names": (i32, [vp, C.POINTER(cp)]),
        "xe_gpu_batch_record_datagram": (i32, [vp, vp, sz]),
        "xe_gpu_load_mos": (i32, [vp, cp]),
        "xe_gpu_mo_count": (i32, [vp]),
        "xe_gpu_mo_name": (cp, [vp, i32]),
        "xe_gpu_mo_parent": (i32, [vp, i32]),
        "xe_gpu_mo_dump": (i32, [vp, i32, cp, sz]),
        "xe_filter_check": (i32, [vp, cp, cp, sz, cp, sz]),
        "xe_gpu_add_device": (i32, [vp, i32, u32, i32, u32, i32, i32, C.POINTER(cp), i32]),
        "xe_gpu_load_devices": (i32, [vp, cp]),
        "xe_gpu_commit": (i32, [vp]),
        "xe_gpu_set_prepass_threads": (i32, [vp, i32]),
        "xe_gpu_set_device_walk": (i32, [vp, i32]),
        "xe_gpu_set_geodb": (i32, [vp, i32, i32, vp, sz]),
        "xe_gpu_load_geodb": (i32, [vp, cp]),
        "xe_gpu_batch_device_walked": (i32, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "xe_gpu_batch_unsupported_sets": (C.c_size_t, [vp]),
        "xe_gpu_batch_oversized_datagrams": (C.c_size_t, [vp]),
        "xe_gpu_add_fwm": (i32, [vp, i32, C.POINTER(C.c_char_p), i32, C.c_size_t]),
        "xe_gpu_fwm_info": (i32, [vp, i32, C.POINTER(C.c_size_t), C.POINTER(i32), C.POINTER(C.c_size_t)]),
        "xe_gpu_fwm_read": (C.c_long, [vp, i32, vp, vp, C.c_size_t, C.POINTER(C.c_uint64)]),
        "xe_gpu_fwm_clear": (i32, [vp, i32]),
        "xe_gpu_fwm_window": (i32, [vp, i32, C.POINTER(i32), C.POINTER(i32)]),
        "xe_gpu_fwm_export": (C.c_long, [vp, i32, i32, vp, vp, C.c_size_t, vp, C.POINTER(i32)]),
        "xe_gpu_load_mos_with_tables": (i32, [vp, C.c_char_p, C.c_size_t, C.POINTER(i32)]),
        "xe_gpu_fwm_count": (i32, [vp]),
        "xe_gpu_fwm_object": (i32, [vp, i32]),
        "xe_gpu_fwm_name": (C.c_char_p, [vp, i32]),
        "xe_gpu_batch_fwm_ms": (i32, [vp, C.POINTER(C.c_float)]),
        "xe_gpu_host_alloc": (vp, [vp, C.c_size_t]),
        "xe_gpu_host_free": (None, [vp, vp]),
        "xe_gpu_batch_prepare": (i32, [vp, vp, vp, vp, vp, sz, C.POINTER(vp)]),
        "xe_gpu_batch_free": (None, [vp]),
        "xe_gpu_batch_run": (i32, [vp, vp]),
        "xe_gpu_batch_sync": (i32, [vp]),
        "xe_gpu_batch_timing": (i32, [vp, i32]),
        "xe_gpu_batch_kernel_ms": (i32, [vp, C.POINTER(C.c_float)]),
        "xe_gpu_batch_count_gathers": (i32, [vp, vp, C.POINTER(C.c_uint64)]),
        "xe_gpu_set_overlap": (i32, [vp, i32]),
        "xe_gpu_batch_input_done": (i32, [vp]),
        "xe_gpu_batch_done": (i32, [vp]),
        "xe_gpu_upload_begin": (i32, [vp, vp, C.c_size_t, C.POINTER(vp)]),
        "xe_gpu_upload_free": (None, [vp]),
        "xe_gpu_batch_prepare_uploaded": (i32, [vp, vp, vp, vp, vp, vp, C.c_size_t, C.POINTER(vp)]),
        "xe_gpu_reload_geodb": (i32, [vp, C.c_char_p]),
        "xe_gpu_batch_per_lane_records": (C.c_size_t, [vp]),
        "xe_gpu_set_walk_timing": (i32, [vp, i32]),
        "xe_gpu_set_list_root_bits": (i32, [vp, i32]),
        "xe_gpu_batch_walk_ms": (i32, [vp, C.POINTER(C.c_float)]),
        "xe_gpu_batch_records": (sz, [vp]),
        "xe_gpu_batch_mask_words": (sz, [vp]),
        "xe_gpu_batch_payload_bytes": (sz, [vp]),
        "xe_gpu_batch_algorithmic_bytes": (sz, [vp]),
        "xe_gpu_batch_masks": (i32, [vp, vp, sz]),
        "xe_gpu_batch_counters": (i32, [vp, vp, sz]),
        "xe_gpu_batch_dropped": (i32, [vp, vp, sz]),
        "xe_gpu_batch_flow_info": (i32, [vp, sz, sz, vp]),
        "xe_gpu_batch_flow_cursor": (i32, [vp, vp, C.POINTER(vp)]),
        "xe_gpu_flow_cursor_seek": (i32, [vp, sz]),
        "xe_gpu_flow_cursor_free": (None, [vp]),
        "xe_gpu_batch_column": (i32, [vp, i32, vp, sz, C.POINTER(sz)]),
        "xe_gpu_batch_device_ptrs": (i32, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "xe_field_count": (i32, []),
        "xe_field_name": (cp, [i32]),
        "xe_field_id": (i32, [i32]),
        "xe_field_size_min": (i32, [i32]),
        "xe_field_size_max": (i32, [i32]),
        "xe_field_offset": (sz, [i32]),
        "xe_gpu_iplist_match": (i32, [vp, cp, i32, vp, sz, vp]),
        "xe_gpu_iplist_match_noalloc": (i32, [vp, cp, i32, vp, sz, vp]),
        "xe_pcap_open": (i32, [cp, C.POINTER(vp)]),
        "xe_pcap_close": (None, [vp]),
        "xe_pcap_error": (cp, [vp]),
        "xe_pcap_read": (C.c_long, [vp, i32, vp, sz, vp, vp, sz]),
        "xe_pcap_stats": (None, [vp] + [C.POINTER(C.c_uint64)] * 4),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)   # AttributeError = the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = None  # filled lazily by tests from include/xe_gpu.h


def fields():
    L = load_library()
    return [dict(idx=i, name=L.xe_field_name(i).decode(), id=L.xe_field_id(i), min=L.xe_field_size_min(i),
                 max=L.xe_field_size_max(i), off=L.xe_field_offset(i)) for i in range(L.xe_field_count())]


def filter_check(text, engine=None):
    """(ok_on_device, dump_text, error_text) for a filter string."""
    L = load_library()
    dump = C.create_string_buffer(1 << 16)
    err = C.create_string_buffer(1024)
    rc = L.xe_filter_check(engine._h if engine else None, text.encode(), dump, len(dump), err, len(err))
    return rc, dump.value.decode(), err.value.decode(errors="replace")


class PcapReader:
    """Datagram batches out of a classic pcap file (xe_pcap_*): iterate over it
    for (data, offsets, src, None) tuples that Engine.prepare() takes."""

    def __init__(self, path, udp_port=0, max_bytes=64 << 20, max_datagrams=1 << 16):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.xe_pcap_open(os.fsencode(path), C.byref(h))
        self._h = h
        if rc != 0:
            detail = self.L.xe_pcap_error(h).decode(errors="replace") if h else ""
            self.close()
            raise XeError(rc, detail)
        self.udp_port, self.max_bytes, self.max_datagrams = int(udp_port), int(max_bytes), int(max_datagrams)

    def close(self):
        if self._h:
            self.L.xe_pcap_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def read(self):
        """the next batch, or None at the end of the file"""
        data = np.empty(self.max_bytes, dtype=np.uint8)
        offs = np.zeros(self.max_datagrams + 1, dtype=np.uint64)
        src = np.zeros(self.max_datagrams, dtype=np.uint32)
        n = self.L.xe_pcap_read(self._h, self.udp_port, data.ctypes.data, data.size, offs.ctypes.data,
                                src.ctypes.data, self.max_datagrams)
        if n < 0:
            raise XeError(n, self.L.xe_pcap_error(self._h).decode(errors="replace"))
        if n == 0:
            return None
        return data[:int(offs[n])], offs[:n + 1], src[:n], None

    def __iter__(self):
        while True:
            b = self.read()
            if b is None:
                return
            yield b

    def stats(self):
        v = [C.c_uint64() for _ in range(4)]
        self.L.xe_pcap_stats(self._h, *[C.byref(x) for x in v])
        return dict(zip(("frames", "taken", "ignored", "damaged"), [int(x.value) for x in v]))


class Engine:
    """One xe_gpu handle: configuration + device state on one GPU."""

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.xe_gpu_create(C.byref(h), device)
        if rc != 0:
            raise XeError(rc, "xe_gpu_create(device=%d): %s (no GPU, no fallback)" %
                          (device, self.L.xe_gpu_strerror(rc).decode()))
        self._h = h
        self._batches = weakref.WeakSet()
        self._pinned = []

    def close(self):
        """Destroys the handle; batches made from it are freed first (a batch
        must not outlive its handle)."""
        if self._h:
            for b in list(self._batches):
                b.free()
            for p in self._pinned:
                self.L.xe_gpu_host_free(self._h, p)
            self._pinned = []
            self.L.xe_gpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise XeError(rc, "%s: %s" % (self.L.xe_gpu_strerror(rc).decode(),
                                          self.L.xe_gpu_last_error(self._h).decode(errors="replace")))
        return rc

    def load_iplists(self, d):
        self._chk(self.L.xe_gpu_load_iplists(self._h, d.encode()))

    def add_iplist(self, name, text):
        if isinstance(text, str):
            text = text.encode()
        self._chk(self.L.xe_gpu_add_iplist(self._h, name.encode(), text, len(text)))

    def add_mo(self, name, filt, parent=-1):
        return self._chk(self.L.xe_gpu_add_mo(self._h, name.encode(), parent, filt.encode()))

    def mo_host_only(self, i):
        """the object's filter is left to the caller's own filter_match()"""
        return bool(self._chk(self.L.xe_gpu_mo_host_only(self._h, i)))

    def mo_host_only_reason(self, i):
        return self.L.xe_gpu_mo_host_only_reason(self._h, i).decode()

    def mo_host_gated(self, i):
        return bool(self._chk(self.L.xe_gpu_mo_host_gated(self._h, i)))

    def set_port_names(self, names):
        """names: 65536 service names ('' or None = the port has none), or None to remove the table"""
        if names is None:
            self._chk(self.L.xe_gpu_set_port_names(self._h, None))
            return
        assert len(names) == 65536
        arr = (C.c_char_p * 65536)(*[(n.encode() if isinstance(n, str) else n) if n else None for n in names])
        self._chk(self.L.xe_gpu_set_port_names(self._h, arr))

    def load_mos_with_tables(self, d, max_keys=0):
        """objects and their "fwm" entries from a mo directory; returns (tables added, entries skipped)"""
        sk = C.c_int()
        n = self._chk(self.L.xe_gpu_load_mos_with_tables(self._h, d.encode(), int(max_keys), C.byref(sk)))
        return n, int(sk.value)

    def fwm_tables(self):
        """[(object index, name)] of the registered tables"""
        return [(self.L.xe_gpu_fwm_object(self._h, i), self.L.xe_gpu_fwm_name(self._h, i).decode())
                for i in range(self.L.xe_gpu_fwm_count(self._h))]

    def load_mos(self, d):
        self._chk(self.L.xe_gpu_load_mos(self._h, d.encode()))

    def load_devices(self, path):
        self._chk(self.L.xe_gpu_load_devices(self._h, path.encode()))

    def mo_count(self):
        return self.L.xe_gpu_mo_count(self._h)

    def mo_name(self, i):
        return self.L.xe_gpu_mo_name(self._h, i).decode()

    def mo_parent(self, i):
        return self.L.xe_gpu_mo_parent(self._h, i)

    def mo_dump(self, i):
        buf = C.create_string_buffer(1 << 16)
        self._chk(self.L.xe_gpu_mo_dump(self._h, i, buf, len(buf)))
        return buf.value.decode()

    def add_device(self, ip=None, dev_id=None, sampling_rate=0, skip_unmarked=False, marks=()):
        import socket
        arr = (C.c_char_p * max(1, len(marks)))(*[m.encode() for m in marks])
        ipn = int.from_bytes(socket.inet_aton(ip), "little") if ip is not None else 0
        self._chk(self.L.xe_gpu_add_device(self._h, int(ip is not None), ipn, int(dev_id is not None),
                                           dev_id or 0, sampling_rate, int(skip_unmarked), arr, len(marks)))

    def add_fwm(self, mo, fields, max_keys=0):
        """fixed-window table of object `mo` over the given field strings; returns its index"""
        arr = (C.c_char_p * len(fields))(*[f.encode() for f in fields])
        return self._chk(self.L.xe_gpu_add_fwm(self._h, int(mo), arr, len(fields), int(max_keys)))

    def fwm_read(self, fwm):
        """{key bytes: tuple of sums}, and the number of records that found no room"""
        kb, nv, sl = C.c_size_t(), C.c_int(), C.c_size_t()
        self._chk(self.L.xe_gpu_fwm_info(self._h, fwm, C.byref(kb), C.byref(nv), C.byref(sl)))
        n = self._chk(self.L.xe_gpu_fwm_read(self._h, fwm, None, None, 0, None))
        keys = np.zeros((max(n, 1), max(kb.value, 1)), dtype=np.uint8)
        vals = np.zeros((max(n, 1), max(nv.value, 1)), dtype=np.uint64)
        dropped = C.c_uint64()
        n2 = self._chk(self.L.xe_gpu_fwm_read(self._h, fwm, keys.ctypes.data, vals.ctypes.data, n, C.byref(dropped)))
        assert n2 == n
        rows = {bytes(keys[i, :kb.value]): tuple(int(x) for x in vals[i, :nv.value]) for i in range(n)}
        if len(rows) != n:
            raise XeError(-6, "fixed-window table %d holds %d rows for %d distinct keys" % (fwm, n, len(rows)))
        return rows, int(dropped.value)

    def fwm_window(self, fwm):
        """("limit", "time") of the mo.conf entry"""
        lim, sec = C.c_int(), C.c_int()
        self._chk(self.L.xe_gpu_fwm_window(self._h, fwm, C.byref(lim), C.byref(sec)))
        return int(lim.value), int(sec.value)

    def fwm_export(self, fwm, limit=-1):
        """([(key bytes, sums)] in the reference's export order, cut after
        `limit` rows (0 = all, -1 = the entry's own limit); sums of the rows
        behind the cut; whether the cut was reached)"""
        kb, nv, sl = C.c_size_t(), C.c_int(), C.c_size_t()
        self._chk(self.L.xe_gpu_fwm_info(self._h, fwm, C.byref(kb), C.byref(nv), C.byref(sl)))
        n = self._chk(self.L.xe_gpu_fwm_read(self._h, fwm, None, None, 0, None))
        keys = np.zeros((max(n, 1), max(kb.value, 1)), dtype=np.uint8)
        vals = np.zeros((max(n, 1), max(nv.value, 1)), dtype=np.uint64)
        others = np.zeros(max(nv.value, 1), dtype=np.uint64)
        hit = C.c_int()
        m = self._chk(self.L.xe_gpu_fwm_export(self._h, fwm, int(limit), keys.ctypes.data, vals.ctypes.data, n,
                                               others.ctypes.data, C.byref(hit)))
        return ([(bytes(keys[i, :kb.value]), tuple(int(x) for x in vals[i, :nv.value])) for i in range(m)],
                tuple(int(x) for x in others[:nv.value]), bool(hit.value))

    def fwm_clear(self, fwm):
        self._chk(self.L.xe_gpu_fwm_clear(self._h, fwm))

    def set_prepass_threads(self, n):
        """host threads that walk datagram headers in prepare() (1 = calling thread only)"""
        self._chk(self.L.xe_gpu_set_prepass_threads(self._h, int(n)))

    def set_geodb(self, kind, family, image):
        """geo (kind 0) / AS (kind 1) database, family 4 / 6: the bytes of the reference's .db file"""
        buf = np.frombuffer(bytes(image), dtype=np.uint8)
        self._chk(self.L.xe_gpu_set_geodb(self._h, int(kind), int(family), buf.ctypes.data if buf.size else None, buf.size))

    def load_geodb(self, d):
        return self._chk(self.L.xe_gpu_load_geodb(self._h, d.encode()))

    def reload_geodb(self, d):
        """re-read the database files of a directory into the configured handle and commit"""
        return self._chk(self.L.xe_gpu_reload_geodb(self._h, d.encode()))

    def set_device_walk(self, mode):
        """which headers the device walks itself: bit 0 sFlow, bit 1 Netflow datagrams of the usual shape"""
        self._chk(self.L.xe_gpu_set_device_walk(self._h, int(mode)))

    def set_list_root_bits(self, bits):
        """width of the list tables' roots: 0 = chosen per table, 16 or 24 (before commit())"""
        self._chk(self.L.xe_gpu_set_list_root_bits(self._h, int(bits)))

    def set_overlap(self, on=True):
        """consecutive batches of this handle overlap: prepare() no longer waits for the datagram copy"""
        self._chk(self.L.xe_gpu_set_overlap(self._h, int(on)))

    def set_walk_timing(self, enable=True):
        """record HIP events around the device header walk of every prepare()"""
        self._chk(self.L.xe_gpu_set_walk_timing(self._h, int(enable)))

    def commit(self):
        self._chk(self.L.xe_gpu_commit(self._h))

    def iplist_match(self, name, family, addrs):
        a = np.ascontiguousarray(addrs, dtype=np.uint8)
        n = a.size // (4 if family == 4 else 16)
        out = np.zeros(n, dtype=np.uint8)
        self._chk(self.L.xe_gpu_iplist_match(self._h, name.encode(), family, a.ctypes.data, n, out.ctypes.data))
        return out

    def iplist_match_noalloc(self, name, family, addrs):
        """iplist_match() through the entry point that allocates nothing"""
        a = np.ascontiguousarray(addrs, dtype=np.uint8)
        n = a.size // (4 if family == 4 else 16)
        out = np.zeros(n, dtype=np.uint8)
        self._chk(self.L.xe_gpu_iplist_match_noalloc(self._h, name.encode(), family, a.ctypes.data, n, out.ctypes.data))
        return out

    def pinned(self, data):
        """a copy of the byte array `data` in page-locked memory (freed with the engine)"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n = max(int(data.size), 1)
        p = self.L.xe_gpu_host_alloc(self._h, n)
        if not p:
            raise XeError(-2, "xe_gpu_host_alloc(%d) failed" % n)
        self._pinned.append(p)
        out = np.ctypeslib.as_array((C.c_uint8 * n).from_address(p))[:data.size]
        out[:] = data
        return out

    def pinned_array(self, shape, dtype):
        """an uninitialised array in page-locked memory (freed with the engine),
        e.g. a result buffer for PreparedBatch.masks(out=...)"""
        dt = np.dtype(dtype)
        n = max(int(np.prod(shape)) * dt.itemsize, 1)
        p = self.L.xe_gpu_host_alloc(self._h, n)
        if not p:
            raise XeError(-2, "xe_gpu_host_alloc(%d) failed" % n)
        self._pinned.append(p)
        raw = np.ctypeslib.as_array((C.c_uint8 * n).from_address(p))
        return raw[:int(np.prod(shape)) * dt.itemsize].view(dt).reshape(shape)

    def upload_begin(self, data):
        """start sending a batch's datagram bytes (uint8 array, page-locked for an asynchronous copy)
        ahead of prepare(batch, upload=...); returns the upload"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        up = C.c_void_p()
        self._chk(self.L.xe_gpu_upload_begin(self._h, data.ctypes.data, data.size, C.byref(up)))
        return up

    def upload_free(self, up):
        self.L.xe_gpu_upload_free(up)

    def prepare(self, batch, upload=None):
        """batch: synth.Batch or (data, offsets, src, kinds) arrays; upload: what upload_begin()
        returned for the same data (the batch takes it over)."""
        if hasattr(batch, "packed"):
            data, offs, src, kinds = batch.packed()
        else:
            data, offs, src, kinds = batch
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        src = np.ascontiguousarray(src, dtype=np.uint32)
        kinds = None if kinds is None else np.ascontiguousarray(kinds, dtype=np.uint8)
        n = len(offs) - 1
        b = C.c_void_p()
        if upload is not None:
            self._chk(self.L.xe_gpu_batch_prepare_uploaded(self._h, upload, data.ctypes.data, offs.ctypes.data, src.ctypes.data,
                                                           kinds.ctypes.data if kinds is not None else None, n, C.byref(b)))
        else:
            self._chk(self.L.xe_gpu_batch_prepare(self._h, data.ctypes.data, offs.ctypes.data, src.ctypes.data,
                                                  kinds.ctypes.data if kinds is not None else None, n, C.byref(b)))
        pb = PreparedBatch(self, b)
        self._batches.add(pb)
        return pb


class PreparedBatch:
    def __init__(self, eng, handle):
        self.eng = eng
        self.L = eng.L
        self._b = handle
        self.records = self.L.xe_gpu_batch_records(handle)
        self.unsupported_sets = self.L.xe_gpu_batch_unsupported_sets(handle)
        self.oversized_datagrams = self.L.xe_gpu_batch_oversized_datagrams(handle)
        self.mask_words = self.L.xe_gpu_batch_mask_words(handle)
        self.payload_bytes = self.L.xe_gpu_batch_payload_bytes(handle)
        self.algorithmic_bytes = self.L.xe_gpu_batch_algorithmic_bytes(handle)

    def free(self):
        if self._b:
            self.L.xe_gpu_batch_free(self._b)
            self._b = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def run(self, stream=None):
        self.eng._chk(self.L.xe_gpu_batch_run(self._b, stream))

    def sync(self):
        self.eng._chk(self.L.xe_gpu_batch_sync(self._b))

    def timing(self, enable=True):
        self.eng._chk(self.L.xe_gpu_batch_timing(self._b, int(enable)))

    def fwm_ms(self):
        ms = C.c_float()
        self.eng._chk(self.L.xe_gpu_batch_fwm_ms(self._b, C.byref(ms)))
        return float(ms.value)

    def kernel_ms(self):
        """(fast netflow, general netflow, sflow) kernel time of the last run, -1 = not launched"""
        ms = (C.c_float * 3)()
        self.eng._chk(self.L.xe_gpu_batch_kernel_ms(self._b, ms))
        return tuple(ms)

    def input_done(self):
        """overlap mode: blocks until the datagram buffer given to prepare() may be reused"""
        self.eng._chk(self.L.xe_gpu_batch_input_done(self._b))

    @property
    def per_lane_records(self):
        """Netflow records left to the per-lane kernel"""
        return int(self.L.xe_gpu_batch_per_lane_records(self._b))

    def done(self):
        """has the last run finished? (never blocks)"""
        rc = self.L.xe_gpu_batch_done(self._b)
        if rc < 0:
            self.eng._chk(rc)
        return bool(rc)

    def count_gathers(self, stream=None):
        """(root-table loads, node / chain-record loads) of the list walks of one more run"""
        out = (C.c_uint64 * 2)()
        self.eng._chk(self.L.xe_gpu_batch_count_gathers(self._b, stream, out))
        return int(out[0]), int(out[1])

    def walk_ms(self):
        """device time of the header walk of this batch's prepare(), -1 = not timed / walked by the host"""
        ms = C.c_float()
        self.eng._chk(self.L.xe_gpu_batch_walk_ms(self._b, C.byref(ms)))
        return float(ms.value)

    def masks(self, out=None):
        if out is None:
            out = np.zeros((self.records, self.mask_words), dtype=np.uint64)
        else:   # a caller's buffer (page-locked memory copies at link speed)
            assert out.dtype == np.uint64 and out.flags.c_contiguous and out.size >= self.records * self.mask_words
            out = out.reshape(-1)[:self.records * self.mask_words].reshape(self.records, self.mask_words)
        self.eng._chk(self.L.xe_gpu_batch_masks(self._b, out.ctypes.data, out.size))
        return out

    def counters(self):
        n = self.eng.mo_count()
        out = np.zeros((max(n, 1), 3), dtype=np.uint64)
        self.eng._chk(self.L.xe_gpu_batch_counters(self._b, out.ctypes.data, out.size))
        return out[:n]

    def dropped(self):
        out = np.zeros(max(self.records, 1), dtype=np.uint8)
        self.eng._chk(self.L.xe_gpu_batch_dropped(self._b, out.ctypes.data, out.size))
        return out[:self.records]

    def flow_info(self, first=0, count=None):
        count = self.records - first if count is None else count
        out = np.zeros((count, FLOW_INFO_SIZE), dtype=np.uint8)
        if count:
            self.eng._chk(self.L.xe_gpu_batch_flow_info(self._b, first, count, out.ctypes.data))
        return out

    def flow_records(self, records):
        """struct flow_info of the records named (rising indices), made one after the other in one
        buffer through xe_gpu_batch_flow_cursor / xe_gpu_flow_cursor_seek; returns (len(records), FLOW_INFO_SIZE) bytes"""
        buf = np.zeros(FLOW_INFO_SIZE, dtype=np.uint8)
        buf[:] = 0xA5                              # the cursor clears it
        out = np.zeros((len(records), FLOW_INFO_SIZE), dtype=np.uint8)
        cur = C.c_void_p()
        self.eng._chk(self.L.xe_gpu_batch_flow_cursor(self._b, buf.ctypes.data, C.byref(cur)))
        try:
            for i, r in enumerate(records):
                self.eng._chk(self.L.xe_gpu_flow_cursor_seek(cur, int(r)))
                out[i] = buf
        finally:
            self.L.xe_gpu_flow_cursor_free(cur)
        return out

    def device_walked(self):
        """(Netflow datagrams, their records, sFlow records) whose headers the device walked"""
        a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self.eng._chk(self.L.xe_gpu_batch_device_walked(self._b, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def record_datagram(self):
        """per record the index of the datagram it was decoded from"""
        out = np.zeros(max(self.records, 1), dtype=np.uint32)
        self.eng._chk(self.L.xe_gpu_batch_record_datagram(self._b, out.ctypes.data, out.size))
        return out[:self.records]

    def column(self, field):
        w = C.c_size_t()
        probe = np.zeros(1, dtype=np.uint8)
        # first call only learns the width
        self.L.xe_gpu_batch_column(self._b, field, probe.ctypes.data, 0, C.byref(w))
        if w.value == 0:
            raise XeError(-1, "field %d has no column in this batch" % field)
        out = np.zeros((self.records, w.value), dtype=np.uint8)
        self.eng._chk(self.L.xe_gpu_batch_column(self._b, field, out.ctypes.data, out.size, C.byref(w)))
        return out
