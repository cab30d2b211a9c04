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


@pytest.mark.parametrize("root_bits", [16, 24])
def test_device_list_lookup_equals_reference(eng, root_bits):
    """(with the list tables' roots over 16 and over 24 address bits)"""
    texts, p4, p6, want4, want6 = golden_io.iplist_reference()
    eng.set_list_root_bits(root_bits)
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
    eng.set_list_root_bits(24)       # (the narrow root is what every other test runs with)
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


def test_geo_functions_equal_reference_answers(eng):
    """asn(), asd() and the geo functions (filter.c:753-834) over records that
    carry the probe addresses of the geo fixture, against what the reference's
    own IP_BTRIE_LOOKUP (ip-btrie.h:90-115, instantiated as geoip.c:33-88 does)
    found in database images made by its IP_BTRIE_ADD"""
    fix = golden_io.geo_reference()
    for name, (kind, fam) in {"as4": (1, 4), "as6": (1, 6), "geo4": (0, 4), "geo6": (0, 6)}.items():
        eng.set_geodb(kind, fam, fix[name][0])
    geo_off = [0, 3, 6, 41, 105, 156, 171, 191]

    def cstr(row, off):
        b = bytes(row[off:])
        return b[:b.index(0)].decode() if 0 in b else b.decode()

    def asn(name):
        _, _, found, pay = fix[name]
        return found.astype(bool), pay[:, :4].copy().view("<u4")[:, 0].astype(np.int64)

    def text(name, off):
        _, _, found, pay = fix[name]
        return found.astype(bool), [cstr(r, off) if f else None for r, f in zip(pay, found)]

    f4, n4 = asn("as4")
    f6, n6 = asn("as6")
    _, d4 = text("as4", 4)
    _, cc4 = text("geo4", geo_off[1])
    _, city4 = text("geo4", geo_off[4])
    _, cont6 = text("geo6", geo_off[0])
    g4 = fix["geo4"][2].astype(bool)
    g6 = fix["geo6"][2].astype(bool)
    n = min(len(fix["as4"][1]), len(fix["as6"][1]), len(fix["geo4"][1]), len(fix["geo6"][1]))
    # one record per probe: IPv4 source = as4 probe, IPv4 destination = geo4 probe,
    # IPv6 source = as6 probe, IPv6 destination = geo6 probe
    mos = [("asn(src host) 64000-64400", lambda i: f4[i] and 64000 <= n4[i] <= 64400),
           ("asn(src host) 100-700 or 64740", lambda i: f4[i] and (100 <= n4[i] <= 700 or n4[i] == 64740)),
           ("asn(src host) 7", lambda i: f4[i] and n4[i] == 7),
           ("asd(src host) 'ACME'", lambda i: f4[i] and d4[i] == "ACME"),
           ("asd(src host) 'Example Net' or 'x'", lambda i: f4[i] and d4[i] in ("Example Net", "x")),
           ("country_code(dst host) 'DE'", lambda i: g4[i] and cc4[i] == "DE"),
           ("city(dst host) 'San Jose' or 'Tokyo'", lambda i: g4[i] and city4[i] in ("San Jose", "Tokyo")),
           ("asn(src host6) 65000-65004", lambda i: f6[i] and 65000 <= n6[i] <= 65004),
           ("not asn(src host6) 7", lambda i: not (f6[i] and n6[i] == 7)),
           ("continent(dst host6) 'EU'", lambda i: g6[i] and cont6[i] == "EU")]
    for filt, _ in mos:
        m = eng.add_mo("mo", filt)
        assert not eng.mo_host_only(m), filt
    eng.commit()
    t = [(8, 4), (12, 4), (27, 16), (28, 16), (4, 1)]
    rec = np.zeros((n, 41), dtype=np.uint8)
    rec[:, 0:4] = fix["as4"][1][:n]
    rec[:, 4:8] = fix["geo4"][1][:n]
    rec[:, 8:24] = fix["as6"][1][:n]
    rec[:, 24:40] = fix["geo6"][1][:n]
    rec[:, 40] = 6
    exp = bytes([192, 0, 2, 9])
    b = synth.Batch()
    b.add(synth.ipfix_template_packet(400, t, 5), exp)
    for i in range(0, n, 30):
        body = rec[i:i + 30].tobytes()
        body = struct.pack(">HH", 400, 4 + len(body)) + body
        b.add(synth.ipfix_header(16 + len(body), 5) + body, exp)
    for rnd in range(2):         # host walk, then device walk
        out = eng.prepare(b)
        out.run()
        m = out.masks()[:, 0]
        fi = out.flow_info()
        proto = [f for f in xenoeye_amd.fields() if f["name"] == "protocol"][0]
        has_proto = fi[:, ((proto["off"] + proto["max"] + 3) & ~3) + 4]
        sel = np.nonzero(has_proto)[0]
        assert len(sel) == n
        for k, (filt, want) in enumerate(mos):
            got = ((m[sel] >> np.uint64(k)) & np.uint64(1)).astype(bool)
            exp_bits = np.array([bool(want(i)) for i in range(n)])
            bad = np.nonzero(got != exp_bits)[0]
            assert bad.size == 0, (filt, rnd, bad[:5])
            assert exp_bits.any() and not exp_bits.all(), filt
        out.free()
