"""The record-after-record read-out (xe_gpu_batch_flow_cursor /
xe_gpu_flow_cursor_seek) hands out the same struct flow_info bytes as the
block read-out xe_gpu_batch_flow_info - whose bytes the parity tests compare
with the oracle's - whatever records are skipped in between: Netflow records
of several templates (fast and per-lane kernels, device marks), sFlow
samples, host-walked and device-walked batches."""
import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth

pytestmark = pytest.mark.gpu


def mixed_workload():
    w = synth.workload_c2(6000, prefixes=200)
    c3 = synth.workload_c3(1200)
    for p, a, k in zip(c3["batch"].pkts, c3["batch"].src, c3["batch"].kind):
        w["batch"].add(p, a, k)
    c1 = synth.workload_c1(1500, per_packet=24)
    for p, a, k in zip(c1["batch"].pkts, c1["batch"].src, c1["batch"].kind):
        w["batch"].add(p, a, k)
    w["devices"] = [dict(ip="192.0.2.1", sampling_rate=7, marks=("proto 17", "dst port 443"))]
    return w


@pytest.mark.parametrize("device_walk", [0, 1])
def test_cursor_equals_block_readout(device_walk):
    w = mixed_workload()
    e = xenoeye_amd.Engine(device=0)
    try:
        for name, text in w["iplists"].items():
            e.add_iplist(name, text)
        for d in w["devices"]:
            e.add_device(**d)
        for filt, parent in w["mos"]:
            e.add_mo("mo", filt, parent)
        e.commit()
        e.set_device_walk(device_walk)
        for _ in range(2):                       # the second batch reuses the handle's host copy
            b = e.prepare(w["batch"])
            b.run()
            n = b.records
            want = b.flow_info()
            assert n > 8000 and want.any()
            rng = np.random.default_rng(5 + device_walk)
            for picks in (np.arange(n), np.flatnonzero(rng.random(n) < 0.3), np.array([0, n - 1]), np.array([n // 2] * 3)):
                got = b.flow_records(picks)
                assert np.array_equal(got, want[picks])
            # read-outs while no cursor is open come from the device again
            assert np.array_equal(b.flow_info(10, 50), want[10:60])
            masks = b.masks()
            assert masks.shape[0] == n
            b.free()
    finally:
        e.close()


def test_one_cursor_per_handle():
    import ctypes as C
    w = synth.workload_c1(500)
    e = xenoeye_amd.Engine(device=0)
    try:
        for name, text in (w.get("iplists") or {}).items():
            e.add_iplist(name, text)
        for filt, parent in w["mos"]:
            e.add_mo("mo", filt, parent)
        e.commit()
        b = e.prepare(w["batch"])
        b.run()
        buf = np.zeros(xenoeye_amd.FLOW_INFO_SIZE, dtype=np.uint8)
        c1, c2 = C.c_void_p(), C.c_void_p()
        L = e.L
        assert L.xe_gpu_batch_flow_cursor(b._b, buf.ctypes.data, C.byref(c1)) == 0
        assert L.xe_gpu_batch_flow_cursor(b._b, buf.ctypes.data, C.byref(c2)) < 0      # the first is still open
        assert L.xe_gpu_flow_cursor_seek(c1, 5) == 0
        assert L.xe_gpu_flow_cursor_seek(c1, 4) < 0                                    # never backwards
        assert L.xe_gpu_flow_cursor_seek(c1, b.records) < 0
        # with a cursor open the other read-outs are served from its copy: same answers
        m_open = b.masks()
        L.xe_gpu_flow_cursor_free(c1)
        assert np.array_equal(m_open, b.masks())
        assert L.xe_gpu_batch_flow_cursor(b._b, buf.ctypes.data, C.byref(c2)) == 0
        L.xe_gpu_flow_cursor_free(c2)
        b.free()
    finally:
        e.close()
