"""The drop-in boundary: datagrams go in one at a time through the reference's
entry points (netflow_process / sflow_process of the shim), and the sequence of
exit calls (object, thread, arrival time of the record's datagram, struct
flow_info bytes) must be the sequence in which the reference would call
monit_object_process_nf(): the oracle's records in arrival order, objects in
pre-order, children only under a parent that received the record, nothing for
records the device rules drop."""
import ctypes as C
import os
import shutil
import tempfile

import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from helpers import configure
from oracle_binding import Oracle, ORACLE_SO
from shim_binding import Shim

pytestmark = pytest.mark.gpu

EXP = bytes([192, 0, 2, 1])


@pytest.fixture()
def eng():
    e = xenoeye_amd.Engine(device=0)
    yield e
    e.close()


def expected_calls(orc, batch, times, thread_id=0):
    """what the reference would hand its consumers: per datagram (oracle run one
    datagram at a time) the records it yields, per record the objects whose bit
    is set, low to high"""
    orc.log_config(keep_flows=True)
    want = []
    for i, (p, s, k) in enumerate(zip(batch.pkts, batch.src, batch.kind)):
        n0 = orc.nrecords()
        if k == 1:
            orc.sflow(p, s)
        else:
            orc.netflow(p, s)
        n1 = orc.nrecords()
        if n1 == n0:
            continue
        masks, flows = orc.masks()[n0:n1], orc.flows()[n0:n1]
        dropped = [orc.L.xo_log_dropped(r) for r in range(n0, n1)]
        for r in range(n1 - n0):
            if dropped[r]:
                continue
            for w in range(masks.shape[1]):
                x = int(masks[r, w])
                while x:
                    b = (x & -x).bit_length() - 1
                    x &= x - 1
                    want.append((64 * w + b, thread_id, times[i], flows[r].tobytes()))
    return want


def feed(shim, batch, thread_id=0):
    times = []
    for p, s, k in zip(batch.pkts, batch.src, batch.kind):
        rc, t = (shim.sflow if k == 1 else shim.netflow)(p, s, thread_id)
        assert rc == 1
        assert t > 1_600_000_000 * 10 ** 9   # CLOCK_REALTIME, nanoseconds
        times.append(t)
    shim.flush(thread_id)
    return times


def check(shim, orc, batch, thread_id=0, first_call=0):
    times = feed(shim, batch, thread_id)
    want = expected_calls(orc, batch, times, thread_id)
    got = shim.calls[first_call:]
    assert len(got) == len(want), "%d exit calls, the reference makes %d" % (len(got), len(want))
    for i, (g, w) in enumerate(zip(got, want)):
        assert g[:3] == w[:3], "call %d: (object, thread, time) %s, reference %s" % (i, g[:3], w[:3])
        if g[3] != w[3]:
            a, b = np.frombuffer(g[3], np.uint8), np.frombuffer(w[3], np.uint8)
            raise AssertionError("call %d (object %d): flow_info differs at bytes %s" % (i, g[0], np.nonzero(a != b)[0][:12]))
    return want


@pytest.mark.parametrize("overlap", [0, 1])
def test_c1_sequence(eng, orc, overlap):
    w = synth.workload_c1(5000, per_packet=24)
    configure(eng, orc, w["iplists"], w["mos"])
    # several batches, the last one partial; with overlap a full batch runs while the next is collected
    shim = Shim([eng], max_datagrams=64, overlap=overlap)
    try:
        want = check(shim, orc, w["batch"])
        assert 0 < len(want) < 5000
        st = shim.stats()
        assert st["datagrams"] == len(w["batch"]) and st["records"] == 5000 and st["calls"] == len(want)
        assert st["batches"] >= 3 and st["failed"] == 0
        # the time of a call is the arrival time of ITS datagram (netflow.c:802, :227-228)
        assert {c[2] for c in shim.calls} == {c[2] for c in want}
    finally:
        shim.close()


@pytest.mark.parametrize("overlap", [0, 1])
def test_c2_nested_objects_and_batch_limits(eng, orc, overlap):
    w = synth.workload_c2(4000, prefixes=300, n_exporters=8)
    mos = list(w["mos"])
    # children and grandchildren under the list objects, and an object that takes everything
    mos += [("proto 6", 0), ("dst port 0-1023", len(mos)), ("not proto 6", 0), ("", -1), ("proto 17", len(mos) + 3)]
    # pre-order: a child directly follows its parent's subtree; rebuild in that order
    order, remap = [], {}

    def add(i):
        remap[i] = len(order)
        order.append(i)
        for j, (_, p) in enumerate(mos):
            if p == i:
                add(j)

    for j, (_, p) in enumerate(mos):
        if p < 0:
            add(j)
    mos = [(mos[i][0], remap[mos[i][1]] if mos[i][1] >= 0 else -1) for i in order]
    configure(eng, orc, w["iplists"], mos)
    shim = Shim([eng], max_bytes=65536, overlap=overlap)       # a few datagrams per batch
    try:
        want = check(shim, orc, w["batch"])
        objs = {c[0] for c in want}
        assert len(objs) > 10
        assert shim.stats()["batches"] >= 3
    finally:
        shim.close()


@pytest.mark.parametrize("overlap", [0, 1])
def test_c3_sflow_and_mixed_kinds(eng, orc, overlap):
    w = synth.workload_c3(700)
    c1 = synth.workload_c1(600, per_packet=10)
    mos = list(w["mos"]) + [("", -1)]
    lists = dict(w["iplists"])
    lists.update(c1["iplists"])
    mos += [(c1["mos"][0][0], len(mos) - 1)]
    configure(eng, orc, lists, mos)
    # one thread that is handed both kinds: a batch holds one kind, arrival order is kept
    b = synth.Batch()
    sf, nf = list(zip(w["batch"].pkts, w["batch"].src)), list(zip(c1["batch"].pkts, c1["batch"].src))
    b.add(*nf.pop(0), kind=0)                 # the template datagram first
    while sf or nf:
        for _ in range(3):
            if sf:
                b.add(*sf.pop(0), kind=1)
        for _ in range(2):
            if nf:
                b.add(*nf.pop(0), kind=0)
    shim = Shim([eng], overlap=overlap)
    try:
        want = check(shim, orc, b)
        assert len(want) > 700
        assert shim.stats()["batches"] > 20    # every change of kind ends a batch
    finally:
        shim.close()


def test_device_rules_and_unknown_versions(eng, orc):
    w = synth.workload_c1(900, per_packet=30)
    devices = [dict(ip="192.0.2.1", sampling_rate=100, skip_unmarked=True, marks=["proto 6", "dst port 0-20000"])]
    configure(eng, orc, w["iplists"], [("", -1), (w["mos"][0][0], 0), ("dev-mark 2", -1)], devices)
    shim = Shim([eng])
    try:
        rc, t = shim.netflow(b"\x00\x07" + bytes(40), EXP)     # version 7: refused like netflow.c:816-818
        assert rc == 0 and t > 0
        assert shim.netflow(b"", EXP)[0] == 0
        want = check(shim, orc, w["batch"])
        assert 0 < len(want)
        assert shim.stats()["datagrams"] == len(w["batch"])
        # sampling rate of the device reaches the consumers through flow->sampling_rate
        off = 2304
        assert all(int.from_bytes(c[3][off:off + 4], "little") == 100 for c in shim.calls)
    finally:
        shim.close()


@pytest.mark.parametrize("overlap", [0, 1])
def test_two_threads_keep_their_own_order(eng, orc, overlap):
    import threading
    e2 = xenoeye_amd.Engine(device=0)
    try:
        w = synth.workload_c2(3000, prefixes=200, n_exporters=4)
        configure(eng, orc, w["iplists"], w["mos"])
        configure(e2, _Null(), w["iplists"], w["mos"])
        half = len(w["batch"]) // 2
        parts = [synth.Batch(), synth.Batch()]
        # exporters stay on one thread, as the operator pins them to capture ports
        for p, s, k in zip(w["batch"].pkts, w["batch"].src, w["batch"].kind):
            parts[s[3] & 1].add(p, s, k)
        shim = Shim([eng, e2], max_datagrams=50, overlap=overlap)
        try:
            times = [None, None]

            def run(t):
                times[t] = feed(shim, parts[t], t)

            ths = [threading.Thread(target=run, args=(t,)) for t in range(2)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            for t in range(2):
                orc.log_clear()
                want = expected_calls(orc, parts[t], times[t], t)
                got = [c for c in shim.calls if c[1] == t]
                assert [g[:3] for g in got] == [x[:3] for x in want]
                assert [g[3] for g in got] == [x[3] for x in want]
            assert half > 0
        finally:
            shim.close()
    finally:
        e2.close()


class _Null:
    """configure() drives an engine and an oracle alike; this stands in where only the engine is wanted"""

    def __getattr__(self, name):
        return lambda *a, **k: 0


def test_host_only_objects_are_decided_by_the_callers_filter_match(eng, orc):
    """objects whose filter the device cannot decide keep their bit and place; the shim asks the
    caller's filter_match() for them and gates the objects below them on its answer"""
    # a second, independent copy of the oracle plays the reference's filter_match(): mfreq() keeps
    # state between records, so the copy that answers the shim must not be the one that predicts
    tmp = tempfile.mkdtemp()
    try:
        so2 = os.path.join(tmp, "liboracle_copy.so")
        shutil.copy(ORACLE_SO, so2)
        orc2 = Oracle(so2)
        w = synth.workload_c1(3000, per_packet=30)
        mos = [(w["mos"][0][0], -1),                              # 0 device
               ("mfreq(src port, dst port) 0-32767", -1),          # 1 host-only (state between records)
               ("proto 6", 1),                                     # 2 device, under a host-only object
               ("vashost 'x' or 'y'", 2),                          # 3 host-only (absent strings equal anything)
               ("dst port 0-9999", 3),                             # 4 device, two host-only objects above
               ("country(src host) 'NL'", -1),                     # 5 host-only: no geo database, never matches
               ("proto 17", 5),                                    # 6 never reached
               ("", -1)]                                           # 7 everything
        configure(eng, orc, w["iplists"], mos)
        configure(None, orc2, w["iplists"], mos)
        assert [eng.mo_host_only(i) for i in range(8)] == [False, True, False, True, False, True, False, False]
        assert [eng.mo_host_gated(i) for i in range(8)] == [False, True, True, True, True, True, True, False]
        assert "state" in eng.mo_host_only_reason(1) and "string" in eng.mo_host_only_reason(3)
        with pytest.raises(xenoeye_amd.XeError):
            eng.add_fwm(2, ["proto", "octets"])               # its records are not known on the device
        evals = []

        def host_match(mo, flow):
            evals.append(mo)
            return orc2.L.xo_mo_match(mo, C.c_void_p(flow))

        shim = Shim([eng], host_match=host_match, max_datagrams=40)
        try:
            want = check(shim, orc, w["batch"])
            objs = [c[0] for c in want]
            assert objs.count(7) == 3000 and 0 < objs.count(1) < 3000 and objs.count(4) > 0
            assert objs.count(5) == 0 and objs.count(6) == 0
            # only host-only objects are ever asked, each once per record that reached it
            assert set(evals) == {1, 3, 5} and evals.count(1) == 3000 and evals.count(5) == 3000
            assert evals.count(3) == objs.count(2)
        finally:
            shim.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_parked_batch_is_delivered_without_more_datagrams(eng, orc):
    """Overlap: a full batch is left running; when nothing else arrives, the idle path (poll, which is
    also what the wake-up datagram triggers) delivers it, then the partial batch behind it once that is due."""
    import time
    w = synth.workload_c1(1500, per_packet=10)
    configure(eng, orc, w["iplists"], w["mos"])
    shim = Shim([eng], max_datagrams=100, max_wait_ns=30_000_000, overlap=1)
    try:
        times = []
        for p, s in zip(w["batch"].pkts, w["batch"].src):
            rc, t = shim.netflow(p, s)
            assert rc == 1
            times.append(t)
        want = expected_calls(orc, w["batch"], times)
        deadline = time.time() + 10
        while len(shim.calls) < len(want) and time.time() < deadline:
            assert shim.poll() == 0
            time.sleep(0.005)
        assert [g[:3] for g in shim.calls] == [x[:3] for x in want]
        assert [g[3] for g in shim.calls] == [x[3] for x in want]
    finally:
        shim.close()
