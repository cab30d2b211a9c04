"""GPU parity against the committed golden fixtures: reference-produced sFlow
records and IP-list answers, oracle-produced Netflow records."""
import struct

import numpy as np
import pytest

import golden_io
import xenoeye_amd
from xenoeye_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def eng():
    e = xenoeye_amd.Engine(device=0)
    yield e
    e.close()


def test_sflow_records_equal_reference(eng):
    pkts, counts, rcs, flows = golden_io.sflow_reference()
    eng.add_mo("all", "")
    eng.commit()
    b = synth.Batch()
    for p in pkts:
        b.add(p, b"\0\0\0\0", kind=1)
    out = eng.prepare(b)
    out.run()
    st = out.dropped()
    got = out.flow_info()[st != 2]
    assert got.shape == flows.shape
    bad = np.nonzero((got != flows).any(axis=1))[0]
    assert bad.size == 0, "records %s differ from the reference's" % bad[:8]
    assert int(out.counters()[0, 0]) == len(flows)
    assert (out.masks()[st != 2, 0] == 1).all()


def test_device_list_lookup_equals_reference(eng):
    texts, p4, p6, want4, want6 = golden_io.iplist_reference()
    for i, t in enumerate(texts):
        eng.add_iplist("l%d" % i, t)
    eng.commit()
    for i in range(len(texts)):
        assert np.array_equal(eng.iplist_match("l%d" % i, 4, p4), want4[i]), i
        assert np.array_equal(eng.iplist_match("l%d" % i, 6, p6), want6[i]), i


def test_list_filters_equal_reference_answers(eng):
    """the same answers through the classification kernels: one object per
    list and direction over records that carry the probe addresses"""
    texts, p4, p6, want4, want6 = golden_io.iplist_reference()
    for i, t in enumerate(texts):
        eng.add_iplist("l%d" % i, t)
    for i in range(len(texts)):
        eng.add_mo("s%d" % i, "src net l%d" % i)
    for i in range(len(texts)):
        eng.add_mo("d%d" % i, "dst net6 l%d" % i)
    eng.commit()
    n = min(len(p4), len(p6))
    t = [(8, 4), (28, 16), (4, 1)]
    rec = np.zeros((n, 21), dtype=np.uint8)
    rec[:, 0:4] = p4[:n]
    rec[:, 4:20] = p6[:n]
    rec[:, 20] = 6
    exp = bytes([192, 0, 2, 9])
    b = synth.Batch()
    b.add(synth.ipfix_template_packet(400, t, 5), exp)
    for i in range(0, n, 60):
        body = rec[i:i + 60].tobytes()
        body = struct.pack(">HH", 400, 4 + len(body)) + body
        b.add(synth.ipfix_header(16 + len(body), 5) + body, exp)
    out = eng.prepare(b)
    out.run()
    m = out.masks()[:, 0]
    # every data set also yields one trailing partial record (the set length
    # includes its 4-byte header): keep the records that carry all fields
    fi = out.flow_info()
    proto = [f for f in xenoeye_amd.fields() if f["name"] == "protocol"][0]
    has_proto = fi[:, ((proto["off"] + proto["max"] + 3) & ~3) + 4]   # value, int size, int has
    sel = np.nonzero(has_proto)[0]
    assert len(sel) == n
    L = len(texts)
    for i in range(L):
        assert np.array_equal(((m[sel] >> np.uint64(i)) & np.uint64(1)).astype(np.uint8), want4[i][:n]), i
        assert np.array_equal(((m[sel] >> np.uint64(L + i)) & np.uint64(1)).astype(np.uint8), want6[i][:n]), i


def test_netflow_regression_pins(eng):
    for case in golden_io.netflow_oracle():
        e = xenoeye_amd.Engine(device=0)
        for n, t in case["iplists"].items():
            e.add_iplist(n, t)
        for d in case["devices"]:
            e.add_device(**d)
        for f, p in case["mos"]:
            e.add_mo("mo", f, p)
        e.commit()
        out = e.prepare(case["batch"])
        out.run()
        assert out.records == len(case["masks"]), case["name"]
        assert np.array_equal(out.masks(), case["masks"]), case["name"]
        assert np.array_equal(out.dropped(), case["dropped"]), case["name"]
        assert np.array_equal(out.counters(), case["counters"]), case["name"]
        keep = case["dropped"] == 0
        assert np.array_equal(out.flow_info()[keep], case["flows"][keep]), case["name"]
        out.free()
        e.close()
