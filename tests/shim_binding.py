"""ctypes view of the drop-in shim (xenoeye_amd/libxenoeye_gpu_shim.so): the
reference's entry points netflow_process() / sflow_process() fed one datagram
at a time, and a recorder standing in for monit_object_process_nf().  Test
infrastructure only."""
import ctypes as C
import os

import numpy as np

import xenoeye_amd

SHIM_SO = os.path.join(os.path.dirname(os.path.abspath(xenoeye_amd.__file__)), "libxenoeye_gpu_shim.so")
FLOW_INFO_SIZE = 2312


class FlowPacketInfo(C.Structure):
    """struct flow_packet_info, flow-info.h:35-46"""
    _fields_ = [("src_addr", C.c_uint8 * 16), ("src_addr_ipv4", C.c_uint32), ("source_id", C.c_uint32),
                ("epoch", C.c_uint32), ("time_ns", C.c_uint64), ("rawpacket", C.c_uint8 * 65536),
                ("sampling_rate", C.c_int)]


PROCESS_NF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p)
FILTER_MATCH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class ShimConfig(C.Structure):
    _fields_ = [("nthreads", C.c_size_t), ("gpu", C.POINTER(C.c_void_p)), ("mo", C.POINTER(C.c_void_p)),
                ("nmo", C.c_size_t), ("expr", C.POINTER(C.c_void_p)), ("process_nf", PROCESS_NF),
                ("filter_match", FILTER_MATCH), ("max_datagrams", C.c_size_t), ("max_bytes", C.c_size_t),
                ("max_wait_ns", C.c_uint64), ("wake_port", C.POINTER(C.c_uint16)), ("overlap", C.c_int)]


class Shim:
    """engines: one committed xenoeye_amd.Engine per capture thread (the same
    engine may be given several times).  Every exit call is recorded as
    (object index, thread_id, time_ns, flow_info bytes).  host_match(object
    index, flow_info address) answers for host-only objects."""

    GLOBL = 0xABCD0000   # stands in for struct xe_data *: handed through untouched

    def __init__(self, engines, host_match=None, max_datagrams=0, max_bytes=0, max_wait_ns=0, keep_flows=True, overlap=0):
        if not os.path.exists(SHIM_SO):
            raise RuntimeError("%s is missing: run build() first" % SHIM_SO)
        xenoeye_amd.load_library()
        L = self.L = C.CDLL(SHIM_SO)
        L.xenoeye_gpu_shim_init.argtypes = [C.POINTER(ShimConfig)]
        L.xenoeye_gpu_shim_flush.argtypes = [C.c_void_p, C.c_size_t]
        L.xenoeye_gpu_shim_poll.argtypes = [C.c_void_p, C.c_size_t]
        L.xenoeye_gpu_shim_stats.argtypes = [C.c_size_t, C.POINTER(C.c_uint64)]
        L.xenoeye_gpu_shim_error.argtypes = [C.c_size_t]
        L.xenoeye_gpu_shim_error.restype = C.c_char_p
        L.xenoeye_gpu_shim_shutdown.restype = None
        for f in (L.netflow_process, L.sflow_process):
            f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(FlowPacketInfo), C.c_int]
        self.calls = []
        self.keep_flows = keep_flows
        self.host_evals = 0
        nmo = engines[0].mo_count()

        def process_nf(globl, mo, thread_id, time_ns, flow):
            assert globl == self.GLOBL
            fb = C.string_at(flow, FLOW_INFO_SIZE) if self.keep_flows else None
            self.calls.append((int(mo) - 1, int(thread_id), int(time_ns), fb))
            return 1

        def filter_match(expr, flow):
            self.host_evals += 1
            return int(bool(host_match(int(expr) - 1, flow)))

        self._cb = (PROCESS_NF(process_nf), FILTER_MATCH(filter_match) if host_match else FILTER_MATCH())
        # objects and expressions are opaque to the shim: index + 1 as the "pointer"
        self._gpu = (C.c_void_p * len(engines))(*[e._h for e in engines])
        self._mo = (C.c_void_p * max(nmo, 1))(*[m + 1 for m in range(nmo)])
        self._expr = (C.c_void_p * max(nmo, 1))(*[m + 1 for m in range(nmo)])
        cfg = ShimConfig(len(engines), self._gpu, self._mo, nmo, self._expr, self._cb[0], self._cb[1],
                         max_datagrams, max_bytes, max_wait_ns, None, overlap)
        rc = L.xenoeye_gpu_shim_init(C.byref(cfg))
        if rc != 0:
            raise xenoeye_amd.XeError(rc, "xenoeye_gpu_shim_init")
        L.netflow_process_init()
        self._fpi = {}     # one struct flow_packet_info per capture thread, as on their stacks

    def _feed(self, fn, pkt, src, thread_id):
        f = self._fpi.get(thread_id)
        if f is None:
            f = self._fpi[thread_id] = FlowPacketInfo()
        n = len(pkt)
        C.memmove(f.rawpacket, bytes(pkt), n)
        f.src_addr_ipv4 = int.from_bytes(src, "little")
        f.time_ns = 0
        rc = fn(self.GLOBL, thread_id, C.byref(f), n)
        return rc, int(f.time_ns)

    def netflow(self, pkt, src, thread_id=0):
        """(return value, the time the shim stamped the datagram with)"""
        return self._feed(self.L.netflow_process, pkt, src, thread_id)

    def sflow(self, pkt, src=b"\0\0\0\0", thread_id=0):
        return self._feed(self.L.sflow_process, pkt, src, thread_id)

    def flush(self, thread_id=0):
        rc = self.L.xenoeye_gpu_shim_flush(self.GLOBL, thread_id)
        if rc != 0:
            raise xenoeye_amd.XeError(rc, self.L.xenoeye_gpu_shim_error(thread_id).decode(errors="replace"))

    def poll(self, thread_id=0):
        return self.L.xenoeye_gpu_shim_poll(self.GLOBL, thread_id)

    def stats(self, thread_id=0):
        v = (C.c_uint64 * 5)()
        self.L.xenoeye_gpu_shim_stats(thread_id, v)
        return dict(zip(("datagrams", "batches", "records", "calls", "failed"), [int(x) for x in v]))

    def error(self, thread_id=0):
        return self.L.xenoeye_gpu_shim_error(thread_id).decode(errors="replace")

    def close(self):
        self.L.xenoeye_gpu_shim_shutdown()
