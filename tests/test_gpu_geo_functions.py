"""asn(), asd() and the geo functions (filter.c:753-834) on the device.  The
reference tree holds no database file, so the databases here are synthetic
images in the reference's own layout (geoip.h:30-55: an array of trie nodes
{uint32 next[2]; int is_leaf; payload}) and the results are compared with the
CPU oracle's restatement of geoip_lookup4/6 / as_lookup4/6: parity unpinned
beyond that."""
import socket
import struct

import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from helpers import compare, configure, run_oracle

pytestmark = pytest.mark.gpu

GEO_SIZES = [3, 3, 35, 64, 51, 15, 20, 23]


def as_payload(asn, asd):
    return struct.pack("<I", asn) + asd.encode().ljust(200, b"\0")


def geo_payload(*fields):
    return b"".join(f.encode().ljust(n, b"\0") for f, n in zip(fields, GEO_SIZES)).ljust(216, b"\0")


def build_db(entries, node_bytes):
    """IP_BTRIE_ADD (ip-btrie.h:4-52) over (address bytes, mask length, payload)"""
    nodes = [[0, 0, 0, b""]]
    for addr, mask, payload in entries:
        node = 0
        for i in range(mask):
            bit = (addr[i // 8] >> (7 - i % 8)) & 1
            if not nodes[node][bit]:
                nodes.append([0, 0, 0, b""])
                nodes[node][bit] = len(nodes) - 1
            node = nodes[node][bit]
        nodes[node][2] = 1
        nodes[node][3] = payload
    out = bytearray()
    for n0, n1, leaf, pay in nodes:
        out += struct.pack("<IIi", n0, n1, leaf) + pay.ljust(node_bytes - 12, b"\0")
    return bytes(out)


def _databases(w=None):
    """prefixes that cover part of the traffic of workload_c2, nested ones among them"""
    rng = synth.Rng(2024)
    as4, geo4, as6, geo6 = [], [], [], []
    names = ["Example Net", "ACME", "Universitaet", "x"]
    cities = [("EU", "DE", "Germany", "Berlin", "Berlin", "10115", "52.5", "13.4"),
              ("EU", "RU", "Russia", "Moscow", "Moscow", "101000", "55.7", "37.6"),
              ("NA", "US", "United States", "California", "San Jose", "95101", "37.3", "-121.9"),
              ("AS", "JP", "Japan", "Tokyo", "Tokyo", "100-0001", "35.6", "139.7")]
    # the zero address is what an absent field reads as
    as4.append((bytes(4), 8, as_payload(7, "Zero Net")))
    geo4.append((bytes(4), 8, geo_payload("ZZ", "ZZ", "Nowhere", "", "", "", "0", "0")))
    as6.append((bytes(16), 16, as_payload(7, "Zero Net")))
    for k in range(400):
        a = bytes(rng.bytes(4))
        if a[0] == 0:        # nothing below the zero prefix: the walk of 0.0.0.0 must end on it
            a = b"\x01" + a[1:]
        m = int(rng.below(1, 17)[0]) + 4
        as4.append((a, m, as_payload(64000 + k % 40 * 37, names[k % 4])))
        geo4.append((a, min(m + 2, 32), geo_payload(*cities[k % 4])))
        if k % 5 == 0:      # a longer prefix under it
            as4.append((a, m + 6, as_payload(100 + k, names[(k + 1) % 4])))
    for k in range(200):
        a = bytes([0x20, int(rng.below(1, 16)[0])]) + bytes(rng.bytes(14))
        m = int(rng.below(1, 30)[0]) + 10
        as6.append((a, m, as_payload(65000 + k % 10, names[k % 4])))
        geo6.append((a, m, geo_payload(*cities[k % 4])))
    return dict(as4=build_db(as4, 216), as6=build_db(as6, 216), geo4=build_db(geo4, 228), geo6=build_db(geo6, 228))


MOS = [("asn(src host) 64000-64400", -1),
       ("(asn(dst host) 100-500) or (asn(dst host) 64740)", -1),
       ("asd(src host) 'ACME'", -1),
       ("asd(dst host) 'Example Net' or 'x'", 2),
       ("country_code(src host) 'DE'", -1),
       ("city(dst host) 'San Jose' and not country(dst host) 'Japan'", -1),
       ("asn(src host6) 65000-65004", -1),
       ("continent(dst host6) 'EU'", -1),
       ("asn(src host) 7", -1),                 # matches records without an IPv4 source
       ("not asn(src host6) 7", -1),
       ("lat(src host) '52.5' and proto 6", 4),
       ("asn(src host) 64000-70000 and dst port 0-1023", -1),
       ("zip(src host) '101000'", -1),
       ("asn(dev-ip) 0-100000", -1)]


def _set(eng, orc, dbs, which=("as4", "as6", "geo4", "geo6")):
    for name in which:
        kind, fam = (1 if name.startswith("as") else 0), (6 if name.endswith("6") else 4)
        eng.set_geodb(kind, fam, dbs[name])
        orc.geodb_set(kind, fam, dbs[name])


def test_functions_on_netflow_and_sflow(orc):
    w = synth.workload_c2(20000, prefixes=300)
    dbs = _databases(w)
    eng = xenoeye_amd.Engine(device=0)
    _set(eng, orc, dbs)
    configure(eng, orc, w["iplists"], w["mos"][:4] + MOS)
    for m in range(4, 4 + len(MOS)):
        assert not eng.mo_host_only(m), MOS[m - 4]
    for rnd in range(2):              # the second batch is walked on the device
        orc.log_clear()
        b = eng.prepare(w["batch"])
        b.run()
        run_oracle(orc, w["batch"])
        compare(b, orc)
        c = b.counters()
        assert (c[4:, 0] > 0).sum() >= len(MOS) - 2, c[:, 0]
        b.free()
    sf = synth.workload_c3(4000)["batch"]
    orc.log_clear()
    b = eng.prepare(sf)
    b.run()
    run_oracle(orc, sf)
    compare(b, orc)
    eng.close()


def test_without_a_database_the_object_is_host_only(orc):
    w = synth.workload_c1(500)
    dbs = _databases(w)
    eng = xenoeye_amd.Engine(device=0)
    _set(eng, orc, dbs, which=("as4",))
    configure(eng, orc, w["iplists"], [("asn(src host) 1-70000", -1), ("country(src host) 'Germany'", -1),
                                       ("asn(src host6) 1", -1), ("asn(src port) 1", -1), ("proto 6", -1)])
    assert [eng.mo_host_only(m) for m in range(5)] == [False, True, True, True, False]
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"])
    dm, om = b.masks(), orc.masks()
    # objects 0 and 4 are decided on the device; the host-only ones carry candidate bits
    assert np.array_equal(dm[:, 0] & 0x11, om[:, 0] & 0x11)
    assert ((dm[:, 0] >> 1) & 7 == 7).all()
    eng.close()


def test_mark_filter_with_a_function(orc):
    w = synth.workload_c5(4000, prefixes=100, n_exporters=4)
    dbs = _databases(w)
    eng = xenoeye_amd.Engine(device=0)
    _set(eng, orc, dbs)
    devices = [dict(ip="10.5.0.1", sampling_rate=10, marks=["asn(src host) 64000-70000", "country_code(dst host) 'US'"]),
               dict(ip="10.5.0.3", skip_unmarked=True, marks=["asd(src host) 'ACME'"])]
    configure(eng, orc, w["iplists"], w["mos"] + [("asn(dst host) 64000-64400", -1)], devices=devices)
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"])
    compare(b, orc)
    eng.close()


def test_bad_images_are_refused():
    eng = xenoeye_amd.Engine(device=0)
    good = build_db([(bytes([10, 0, 0, 0]), 8, as_payload(1, "a"))], 216)
    with pytest.raises(xenoeye_amd.XeError):
        eng.set_geodb(1, 4, good[:-1])
    bad = bytearray(good)
    bad[0:4] = struct.pack("<I", 999)        # a child outside the image
    with pytest.raises(xenoeye_amd.XeError):
        eng.set_geodb(1, 4, bytes(bad))
    bad = bytearray(good)
    bad[216 + 4:216 + 8] = struct.pack("<I", 1)   # a node that is its own child
    with pytest.raises(xenoeye_amd.XeError):
        eng.set_geodb(1, 4, bytes(bad))
    eng.set_geodb(1, 4, good)
    eng.close()


def test_functions_when_every_object_runs_on_its_own(orc):
    """more distinct terms than the shared classification holds: the programs are
    interpreted one by one, the function terms through the same list walk"""
    w = synth.workload_c2(4000, prefixes=100)
    dbs = _databases(w)
    eng = xenoeye_amd.Engine(device=0)
    _set(eng, orc, dbs)
    many = [("dst port %d-%d" % (150 * i, 150 * i + 100), -1) for i in range(300)]
    configure(eng, orc, w["iplists"], MOS + many)
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"])
    compare(b, orc)
    eng.close()


def test_reload_replaces_the_databases(orc, tmp_path):
    """geoip_reload() (geoip.c:131-178): new files into a configured handle; the objects stay, the
    lists made from the databases are rebuilt, the answers follow the new images"""
    w = synth.workload_c2(8000, prefixes=200)
    dbs = _databases(w)
    eng = xenoeye_amd.Engine(device=0)
    _set(eng, orc, dbs)
    configure(eng, orc, w["iplists"], w["mos"][:4] + MOS)
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"])
    compare(b, orc)
    before = b.counters().copy()
    b.free()
    # the same prefixes with other owners and places: AS numbers moved out of the objects' ranges,
    # countries swapped
    rng = synth.Rng(77)
    as4 = [(bytes(4), 8, as_payload(7, "Zero Net"))]
    geo4 = [(bytes(4), 8, geo_payload("ZZ", "ZZ", "Nowhere", "", "", "", "0", "0"))]
    for k in range(300):
        a = bytes(rng.bytes(4))
        if a[0] == 0:
            a = b"\x01" + a[1:]
        m = int(rng.below(1, 12)[0]) + 2
        as4.append((a, m, as_payload(64100 + k % 7, "ACME" if k % 2 else "x")))
        geo4.append((a, m, geo_payload("EU", "DE", "Germany", "Berlin", "Berlin", "10115", "52.5", "13.4")))
    new = dict(dbs)
    new["as4"] = build_db(as4, 216)
    new["geo4"] = build_db(geo4, 228)
    for name in ("as4", "geo4", "as6", "geo6"):
        (tmp_path / (name + ".db")).write_bytes(new[name])
    assert eng.reload_geodb(str(tmp_path)) == 4
    for name in ("as4", "as6", "geo4", "geo6"):
        orc.geodb_set(1 if name.startswith("as") else 0, 6 if name.endswith("6") else 4, new[name])
    orc.log_clear()
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"])
    compare(b, orc)
    after = b.counters()
    assert (before[4:, 0] != after[4:, 0]).sum() >= 4
    b.free()
    # a database cannot be added or dropped once objects exist
    with pytest.raises(xenoeye_amd.XeError):
        eng.set_geodb(0, 4, b"")
    eng.close()
