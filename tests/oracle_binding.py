"""ctypes binding of the CPU oracle (oracle/libxe_oracle.so) and of the
reference units prebuilt under oracle/_ref.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libxe_oracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

FLOW_INFO_SIZE = 2312


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
        L = self.L = C.CDLL(path)
        L.xo_field_name.restype = C.c_char_p
        for f in ("xo_field_off", "xo_field_off_size", "xo_field_off_has", "xo_off_sampling_rate",
                  "xo_off_payload_ptr", "xo_log_nrecords", "xo_log_mask_words"):
            getattr(L, f).restype = C.c_size_t
        L.xo_iplist_add_text.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.xo_iplist_load_dir.argtypes = [C.c_char_p]
        L.xo_iplist_match4.argtypes = [C.c_char_p, C.c_uint32]
        L.xo_iplist_match6.argtypes = [C.c_char_p, C.c_char_p]
        L.xo_iplist_nodes.argtypes = [C.c_char_p, C.c_int]
        L.xo_iplist_nodes.restype = C.c_size_t
        L.xo_filter_parse.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.xo_filter_dump.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        L.xo_filter_match.argtypes = [C.c_int, C.c_void_p]
        L.xo_mo_add.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]
        L.xo_field_parse.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        L.xo_fwm_add.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_size_t]
        for f in ("xo_fwm_rows", "xo_fwm_keysize", "xo_fwm_nvals"):
            getattr(L, f).argtypes = [C.c_int]
            getattr(L, f).restype = C.c_size_t
        L.xo_fwm_dump.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.xo_fwm_dump.restype = None
        L.xo_fwm_export.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.xo_fwm_export.restype = C.c_long
        L.xo_device_add.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                    C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_size_t]
        L.xo_netflow_process.argtypes = [C.c_char_p, C.c_int, C.c_uint32]
        L.xo_sflow_process.argtypes = [C.c_char_p, C.c_int, C.c_uint32]
        L.xo_log_config.argtypes = [C.c_int, C.c_size_t]
        L.xo_log_mask.argtypes = [C.c_size_t]
        L.xo_log_mask.restype = C.POINTER(C.c_uint64)
        L.xo_log_dropped.argtypes = [C.c_size_t]
        L.xo_log_flow.argtypes = [C.c_size_t]
        L.xo_log_flow.restype = C.c_void_p
        L.xo_counters.restype = C.POINTER(C.c_uint64)
        L.xo_mac_addr_read.argtypes = [C.c_char_p, C.c_char_p]
        self.reset()

    def reset(self):
        self.L.xo_reset()

    # field table
    def fields(self):
        out = []
        for i in range(self.L.xo_nfields()):
            out.append(dict(idx=i, name=self.L.xo_field_name(i).decode(), id=self.L.xo_field_id(i),
                            type=self.L.xo_field_type(i), min=self.L.xo_field_min(i),
                            max=self.L.xo_field_max(i), off=self.L.xo_field_off(i),
                            off_size=self.L.xo_field_off_size(i), off_has=self.L.xo_field_off_has(i)))
        return out

    def iplist_add(self, name, text):
        if isinstance(text, str):
            text = text.encode()
        return self.L.xo_iplist_add_text(name.encode(), text, len(text))

    def iplist_load_dir(self, d):
        return self.L.xo_iplist_load_dir(d.encode())

    def match4(self, name, addr_bytes):
        return self.L.xo_iplist_match4(name.encode(), int.from_bytes(addr_bytes, "little"))

    def match6(self, name, addr_bytes):
        return self.L.xo_iplist_match6(name.encode(), bytes(addr_bytes))

    def filter_parse(self, text):
        err = C.create_string_buffer(1100)
        h = self.L.xo_filter_parse(text.encode(), err, len(err))
        if h < 0:
            raise ValueError(err.value.decode(errors="replace"))
        return h

    def filter_dump(self, h):
        buf = C.create_string_buffer(1 << 16)
        n = self.L.xo_filter_dump(h, buf, len(buf))
        assert n >= 0
        return buf.value.decode()

    def filter_match(self, h, flow_bytes):
        b = C.create_string_buffer(bytes(flow_bytes), FLOW_INFO_SIZE)
        return self.L.xo_filter_match(h, C.cast(b, C.c_void_p))

    def field_parse(self, text):
        """(aggr, offset, size, scale, descending, function, offset of its second argument, k, argument sizes) of a fixed-window field text"""
        out = (C.c_int * 9)()
        err = C.create_string_buffer(1100)
        if not self.L.xo_field_parse(text.encode(), out, err, len(err)):
            raise ValueError(err.value.decode(errors="replace"))
        return tuple(out)

    def fwm_add(self, mo, fields):
        err = C.create_string_buffer(1100)
        arr = (C.c_char_p * len(fields))(*[f.encode() for f in fields])
        i = self.L.xo_fwm_add(mo, arr, len(fields), err, len(err))
        if i < 0:
            raise ValueError(err.value.decode(errors="replace"))
        return i

    def fwm_read(self, fwm):
        """{key bytes: tuple of sums}"""
        import numpy as np
        n, ks, nv = self.L.xo_fwm_rows(fwm), self.L.xo_fwm_keysize(fwm), self.L.xo_fwm_nvals(fwm)
        keys = np.zeros((max(n, 1), max(ks, 1)), dtype=np.uint8)
        vals = np.zeros((max(n, 1), max(nv, 1)), dtype=np.uint64)
        if n:
            self.L.xo_fwm_dump(fwm, keys.ctypes.data, vals.ctypes.data)
        return {bytes(keys[i, :ks]): tuple(int(x) for x in vals[i, :nv]) for i in range(n)}

    def fwm_export(self, fwm, limit=0):
        """([(key bytes, sums)] in export order, others, hit_limit)"""
        import numpy as np
        n, ks, nv = self.L.xo_fwm_rows(fwm), self.L.xo_fwm_keysize(fwm), self.L.xo_fwm_nvals(fwm)
        keys = np.zeros((max(n, 1), max(ks, 1)), dtype=np.uint8)
        vals = np.zeros((max(n, 1), max(nv, 1)), dtype=np.uint64)
        others = np.zeros(max(nv, 1), dtype=np.uint64)
        hit = C.c_int()
        m = self.L.xo_fwm_export(fwm, limit, keys.ctypes.data, vals.ctypes.data, others.ctypes.data, C.byref(hit))
        return ([(bytes(keys[i, :ks]), tuple(int(x) for x in vals[i, :nv])) for i in range(m)],
                tuple(int(x) for x in others[:nv]), bool(hit.value))

    def mo_add(self, filt, parent=-1):
        err = C.create_string_buffer(1100)
        i = self.L.xo_mo_add(parent, filt.encode(), err, len(err))
        if i < 0:
            raise ValueError(err.value.decode(errors="replace"))
        return i

    def device_add(self, ip=None, dev_id=None, sampling_rate=0, skip_unmarked=False, marks=()):
        import socket
        err = C.create_string_buffer(1100)
        arr = (C.c_char_p * max(1, len(marks)))(*[m.encode() for m in marks])
        ipn = int.from_bytes(socket.inet_aton(ip), "little") if ip is not None else 0
        ok = self.L.xo_device_add(ip is not None, ipn, dev_id is not None, dev_id or 0,
                                  sampling_rate, int(skip_unmarked), arr, len(marks), err, len(err))
        if not ok:
            raise ValueError(err.value.decode(errors="replace"))

    def geodb_set(self, kind, family, image):
        """kind 0 geo / 1 AS, family 4 / 6, image: bytes of the reference's .db file"""
        self.L.xo_geodb_set.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        assert self.L.xo_geodb_set(kind, family, bytes(image), len(image))

    def geodb_lookup(self, kind, family, addr):
        """payload bytes of the node the database walk ends on, or None (geoip.c:33-88)"""
        self.L.xo_geodb_lookup.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        buf = C.create_string_buffer(216)
        if not self.L.xo_geodb_lookup(kind, family, bytes(addr), buf):
            return None
        return buf.raw[:204 if kind else 214]

    def set_port_names(self, names):
        """65536 service names ('' / None = none) for portstr() / ppstr(); None removes them"""
        if names is None:
            self.L.xo_set_port_names(None)
            return
        arr = (C.c_char_p * 65536)(*[(n.encode() if isinstance(n, str) else n) if n else None for n in names])
        self.L.xo_set_port_names(arr)

    def port_to_str(self, port):
        buf = C.create_string_buffer(400)
        self.L.xo_port_to_str(buf, C.c_uint16(port))
        return buf.value.decode()

    def ports_pair_to_str(self, p1, p2):
        buf = C.create_string_buffer(800)
        self.L.xo_ports_pair_to_str(buf, C.c_uint16(p1), C.c_uint16(p2))
        return buf.value.decode()

    def log_config(self, keep_flows=True, max_records=0):
        self.L.xo_log_config(int(keep_flows), max_records)

    def netflow(self, pkt, src_ip4_bytes):
        return self.L.xo_netflow_process(bytes(pkt), len(pkt), int.from_bytes(src_ip4_bytes, "little"))

    def sflow(self, pkt, src_ip4_bytes=b"\0\0\0\0"):
        return self.L.xo_sflow_process(bytes(pkt), len(pkt), int.from_bytes(src_ip4_bytes, "little"))

    def nrecords(self):
        return self.L.xo_log_nrecords()

    def masks(self):
        n, w = self.nrecords(), self.L.xo_log_mask_words()
        if n == 0:
            return np.zeros((0, w), dtype=np.uint64)
        p = self.L.xo_log_mask(0)
        return np.ctypeslib.as_array(p, shape=(n, w)).copy()

    def dropped(self):
        return np.array([self.L.xo_log_dropped(i) for i in range(self.nrecords())], dtype=np.uint8)

    def flows(self):
        n = self.nrecords()
        if n == 0:
            return np.zeros((0, FLOW_INFO_SIZE), dtype=np.uint8)
        p = self.L.xo_log_flow(0)
        assert p, "log_config(keep_flows=True) first"
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n, FLOW_INFO_SIZE)).copy()

    def counters(self):
        n = self.L.xo_nmo()
        p = self.L.xo_counters()
        return np.ctypeslib.as_array(p, shape=(max(n, 1), 3)).copy()[:n]

    def log_clear(self):
        self.L.xo_log_clear()


def have_ref(name):
    return os.path.exists(os.path.join(REF_DIR, name))


class RefIplist:
    """The reference's own iplist.c (oracle/_ref/libref_iplist.so).  Its list
    registry is process-global and load-once, so every call loads a fresh
    directory of uniquely named lists."""

    def __init__(self):
        L = self.L = C.CDLL(os.path.join(REF_DIR, "libref_iplist.so"))
        L.iplists_load.argtypes = [C.c_char_p]
        L.iplist_get_by_name.argtypes = [C.c_char_p]
        L.iplist_get_by_name.restype = C.c_void_p
        L.iplist_match4.argtypes = [C.c_void_p, C.c_uint32]
        L.iplist_match6.argtypes = [C.c_void_p, C.c_char_p]

    def load_dir(self, d):
        return self.L.iplists_load(d.encode())

    def get(self, name):
        return self.L.iplist_get_by_name(name.encode())

    def match4(self, h, addr_bytes):
        return self.L.iplist_match4(h, int.from_bytes(addr_bytes, "little"))

    def match6(self, h, addr_bytes):
        return self.L.iplist_match6(h, bytes(addr_bytes))


class RefSflow:
    def __init__(self):
        L = self.L = C.CDLL(os.path.join(REF_DIR, "libref_sflow.so"))
        L.ref_sflow_process.argtypes = [C.c_char_p, C.c_int]
        L.ref_sflow_nrec.restype = C.c_size_t
        L.ref_sflow_rec.argtypes = [C.c_size_t]
        L.ref_sflow_rec.restype = C.c_void_p
        L.ref_flow_info_size.restype = C.c_size_t
        assert L.ref_flow_info_size() == FLOW_INFO_SIZE

    def process(self, pkt):
        rc = self.L.ref_sflow_process(bytes(pkt), len(pkt))
        n = self.L.ref_sflow_nrec()
        recs = np.zeros((n, FLOW_INFO_SIZE), dtype=np.uint8)
        for i in range(n):
            recs[i] = np.frombuffer(C.string_at(self.L.ref_sflow_rec(i), FLOW_INFO_SIZE), dtype=np.uint8)
        return rc, recs


class RefGeo:
    """The reference's geo / AS database structures and trie macros (geoip.h,
    ip-btrie.h under oracle/ref_geo_harness.c -> oracle/_ref/libref_geo.so):
    images built with IP_BTRIE_ADD, looked up with IP_BTRIE_LOOKUP as geoip.c does."""

    def __init__(self):
        L = self.L = C.CDLL(os.path.join(REF_DIR, "libref_geo.so"))
        L.ref_geo_node_size.restype = C.c_size_t
        L.ref_geo_payload_offset.restype = C.c_size_t
        L.ref_geo_field_offset.restype = C.c_size_t
        L.ref_geo_db_new.restype = C.c_void_p
        L.ref_geo_db_add.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p]
        L.ref_geo_db_image.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.ref_geo_db_image.restype = C.c_void_p
        L.ref_geo_db_free.argtypes = [C.c_void_p, C.c_int]
        for fn in ("ref_geo_lookup4", "ref_as_lookup4"):
            getattr(L, fn).argtypes = [C.c_char_p, C.c_uint32, C.c_char_p]
        for fn in ("ref_geo_lookup6", "ref_as_lookup6"):
            getattr(L, fn).argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]

    def node_size(self, kind):
        return self.L.ref_geo_node_size(kind)

    def payload_offset(self, kind):
        return self.L.ref_geo_payload_offset(kind)

    def field_layout(self):
        """(offset, size) of the 8 fields of struct geoip_info"""
        return [(self.L.ref_geo_field_offset(f), self.L.ref_geo_field_size(f)) for f in range(8)]

    def build(self, kind, entries):
        """image of a database made of (address bytes, mask length, payload bytes) with the reference's insertion macro"""
        db = self.L.ref_geo_db_new(kind)
        psize = self.node_size(kind) - 12
        for addr, mask, payload in entries:
            assert self.L.ref_geo_db_add(db, kind, bytes(addr), mask, bytes(payload).ljust(psize, b"\0")) == 1
        n = C.c_size_t()
        p = self.L.ref_geo_db_image(db, kind, C.byref(n))
        img = C.string_at(p, n.value) if n.value else b""
        self.L.ref_geo_db_free(db, kind)
        return img

    def lookup(self, kind, family, image, addr):
        """payload bytes or None"""
        out = C.create_string_buffer(256)
        if family == 4:
            fn = self.L.ref_as_lookup4 if kind else self.L.ref_geo_lookup4
            ok = fn(image, int.from_bytes(bytes(addr), "little"), out)
        else:
            fn = self.L.ref_as_lookup6 if kind else self.L.ref_geo_lookup6
            ok = fn(image, bytes(addr), out)
        return out.raw[:204 if kind else 214] if ok else None


class RefUtils:
    def __init__(self):
        L = self.L = C.CDLL(os.path.join(REF_DIR, "libref_utils.so"))
        L.mac_addr_read.argtypes = [C.c_char_p, C.c_char_p]
        L.tcp_flags_to_str.argtypes = [C.c_uint8]
        L.tcp_flags_to_str.restype = C.c_char_p
        L.port_to_str.argtypes = [C.c_char_p, C.c_uint16]
        L.ports_pair_to_str.argtypes = [C.c_char_p, C.c_uint16, C.c_uint16]

    def port_to_str(self, port):
        buf = C.create_string_buffer(400)
        self.L.port_to_str(buf, port)
        return buf.value.decode()

    def ports_pair_to_str(self, p1, p2):
        buf = C.create_string_buffer(800)
        self.L.ports_pair_to_str(buf, p1, p2)
        return buf.value.decode()

    def port_names(self):
        """the service name of every port, the way a binding gets them: port_to_str() prints
        'name (port)' for a port with a name and the number alone otherwise (utils.c:28-37)"""
        names = []
        for p in range(65536):
            t = self.port_to_str(p)
            tail = " (%d)" % p
            names.append(t[:-len(tail)] if t.endswith(tail) else "")
        return names


def port_names():
    """the reference's service name of every port ('' = none): what its port_to_str() (utils.c:28-37)
    printed when its utils.c was built in the build container, kept as a fixture
    (tests/golden/port_names.json, written by tests/golden/make_fixtures.py)"""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "port_names.json")) as f:
        named = json.load(f)
    names = [""] * 65536
    for p, n in named.items():
        names[int(p)] = n
    return names
