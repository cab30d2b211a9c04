"""GPU parity of the filter terms that are lowered at commit instead of being
evaluated as written: div_l() (libm log / pow over the int quotient, cut into
ranges of the quotient), portstr() / ppstr() (texts of port values, turned
into sets / conditions over the ports with a caller-supplied name table), and
of object trees that contain objects only the caller's filter_match() decides."""
import os

import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from helpers import configure, run_oracle, compare
from oracle_binding import port_names

pytestmark = pytest.mark.gpu

EXP = bytes([192, 0, 2, 1])
FIELDS = [(8, 4), (12, 4), (7, 2), (11, 2), (4, 1), (1, 8), (2, 8)]   # 5-tuple, 8-byte octets / packets


@pytest.fixture()
def eng():
    e = xenoeye_amd.Engine(device=0)
    yield e
    e.close()


def _batch(rng, n, ports=None, octets=None, packets=None, per_packet=40):
    recs = synth.flow_columns(rng, n, FIELDS)     # [n, 29]: src 0-4, dst 4-8, ports 8-12, proto 12, octets 13-21, packets 21-29
    if ports is not None:
        recs[:, 8:10] = synth._be(ports[0], 2)
        recs[:, 10:12] = synth._be(ports[1], 2)
    if octets is not None:
        recs[:, 13:21] = synth._be(octets, 8)
    if packets is not None:
        recs[:, 21:29] = synth._be(packets, 8)
    b = synth.Batch()
    b.add(synth.v9_template_packet(256, FIELDS, 77), EXP)
    b.extend(synth.v9_data_packets(256, recs, per_packet, 77), EXP)
    return b


def test_div_l(eng, orc):
    rng = synth.Rng(0xD17)
    n = 6000
    # quotients around every power of 2, 3 and 10, zero divisors, quotients beyond int
    q = np.concatenate([2 ** np.arange(0, 40, dtype=np.uint64), 3 ** np.arange(0, 25, dtype=np.uint64),
                        10 ** np.arange(0, 12, dtype=np.uint64)])
    q = np.concatenate([q, q + np.uint64(1), np.maximum(q, np.uint64(1)) - np.uint64(1)])
    packets = (rng.u64(n) % np.uint64(5)).astype(np.uint64)           # 0 = division by zero
    packets[packets == 0] = np.where(rng.u64(int((packets == 0).sum())) & np.uint64(1), 0, 1)
    octets = q[(rng.u64(n) % np.uint64(len(q))).astype(np.int64)] * np.maximum(packets, np.uint64(1))
    octets[::7] = rng.u64(len(octets[::7]))                             # anything, incl. quotients with bit 31 set
    mos = [("div_l(octets, packets, 2) 64-128", -1), ("div_l(octets, packets, 2) 1 or 2 or 4 or 1024-65536", -1),
           ("div_l(bits, packets, 10) 100 or 1000", -1), ("div_l(octets, packets, 3) 0-100", -1),
           ("not div_l(octets, packets, 10) 0", -1), ("div_l(octets, packets, 1) 1", -1),
           ("div_l(octets, packets, 0) 0-5", -1), ("div_l(octets, packets, -2) 0", -1),
           ("div_l(packets, octets, 7) 0-49", -1), ("div_l(octets, packets, 10) -5-0 or 1000000000-2000000000", -1),
           ("div(octets, packets) 64-127", -1), ("div_l(octets, packets, 2) 64", 10), ("div_r(octets, packets, 10) 1000-1990", -1)]
    configure(eng, orc, None, mos)
    bb = _batch(rng, n, octets=octets, packets=packets)
    b2 = eng.prepare(bb)
    b2.run()
    run_oracle(orc, bb)
    compare(b2, orc)
    c = b2.counters()[:, 0]
    assert all(0 < int(x) < n for x in c[[0, 1, 2, 3, 4, 10, 12]]), c
    # div_l with base 2 over 64..128 is the quotient range 64..255 (pow(2, floor(log2 q)))
    assert int(c[11]) == int(c[10])


def test_portstr_and_ppstr(eng, orc):
    names = port_names()
    named = np.array([p for p in range(65536) if names[p]], dtype=np.uint64)
    unnamed = np.array([p for p in range(65536) if not names[p]], dtype=np.uint64)
    rng = synth.Rng(0x9075)
    n = 8000

    def pick(pool, k):
        return pool[(rng.u64(k) % np.uint64(len(pool))).astype(np.int64)]

    sp = np.concatenate([pick(named, n // 4), pick(unnamed, n // 4), pick(named, n // 4), pick(unnamed, n // 4)])
    dp = np.concatenate([pick(named, n // 4), pick(named, n // 4), pick(unnamed, n // 4), pick(unnamed, n // 4)])
    # make the interesting pairs frequent
    hot_n = [int(x) for x in named[:6]] + [80, 443, 53]
    hot_u = [int(x) for x in unnamed[:6]] + [int(unnamed[-1])]
    for i in range(0, n, 3):
        sp[i] = (hot_n + hot_u)[(i // 3) % (len(hot_n) + len(hot_u))]
    for i in range(1, n, 5):
        dp[i] = (hot_u + hot_n)[(i // 5) % (len(hot_n) + len(hot_u))]
    sp[::41] = dp[::41]                                                 # equal ports

    eng.set_port_names(names)
    orc.set_port_names(names)

    def ptxt(p):
        return orc.port_to_str(int(p))

    def pptxt(a, b):
        return orc.ports_pair_to_str(int(a), int(b))

    u0, u1, n0, n1 = hot_u[0], hot_u[1], hot_n[0], hot_n[1]
    lo_u, hi_u = min(u0, u1), max(u0, u1)
    mos = [("portstr(src port) '%s'" % ptxt(n0), -1),
           ("portstr(dst port) '%s' or '%s' or 'no such text'" % (ptxt(n1).upper(), ptxt(u0)), -1),
           ("portstr(src port) '%s'" % ptxt(u1), -1),
           ("not portstr(dst port) '%s'" % ptxt(80), -1),
           ("portstr(proto) '6'", -1),                                  # any 1/2/4/8-byte field is read
           ("ppstr(src port, dst port) '%s'" % pptxt(n0, n1), -1),       # both named
           ("ppstr(src port, dst port) '%s'" % pptxt(lo_u, hi_u), -1),   # 'X ->': any larger unnamed dst
           ("ppstr(src port, dst port) '%s'" % pptxt(hi_u, lo_u), -1),   # '-> Y'
           ("ppstr(src port, dst port) '%s' or '%s'" % (pptxt(n0, unnamed[-1]), pptxt(unnamed[-1], n0)), -1),
           ("ppstr(src port, dst port) '%s'" % pptxt(u0, u0), -1),       # equal, unnamed
           ("ppstr(dst port, src port) '%s'" % pptxt(n1, n1).lower(), -1),
           ("ppstr(src port, dst port) '%s'" % pptxt(lo_u, max(int(x) for x in named)), -1),
           ("ppstr(src port, dst port) '80 -> 80' or '-> 0' or '65535 ->'", -1),
           ("proto 6 and ((portstr(dst port) '%s') or (ppstr(src port, dst port) '%s'))" % (ptxt(443), pptxt(53, 53)), -1),
           ("portstr(src port) '%s'" % ptxt(n0), 3)]
    configure(eng, orc, None, mos)
    assert not any(eng.mo_host_only(i) for i in range(len(mos)))
    # the pairs the operands were written for
    for k, (a, b_) in enumerate([(n0, n1), (lo_u, hi_u), (hi_u, lo_u), (n0, int(unnamed[-1])), (int(unnamed[-1]), n0),
                                 (u0, u0), (n1, n1), (80, 80), (53, 53), (lo_u, int(named.max()))]):
        sp[100 + k::211] = a
        dp[100 + k::211] = b_
    bb = _batch(rng, n, ports=(sp, dp))
    b = eng.prepare(bb)
    b.run()
    run_oracle(orc, bb)
    compare(b, orc)
    c = b.counters()[:, 0]
    assert all(int(x) > 0 for x in c[[0, 1, 2, 3, 5, 6, 7, 9]]), c


def test_portstr_without_a_name_table_is_left_to_the_caller(eng):
    eng.add_mo("a", "portstr(src port) 'http (80)'")
    eng.add_mo("b", "proto 6")
    assert eng.mo_host_only(0) and not eng.mo_host_only(1)
    assert "port-name table" in eng.mo_host_only_reason(0)
    with pytest.raises(xenoeye_amd.XeError):          # a devices.conf mark must be decidable on the device
        eng.add_device(ip="10.0.0.1", marks=["portstr(src port) 'http (80)'"])


def _write_tree(root, tree, parent=""):
    for name, (filt, kids) in tree.items():
        d = os.path.join(root, parent, name)
        os.makedirs(d)
        with open(os.path.join(d, "mo.conf"), "w") as f:
            f.write('{\n\t/* sample */\n\t"filter": "%s"\n}\n' % filt)
        _write_tree(root, kids, os.path.join(parent, name))


def test_object_tree_with_one_string_object_loads(eng, orc, tmp_path):
    """the shape of the reference's sample tree (lxc/mo: ingress / egress objects with per-protocol
    children and a tfstr() grandchild) plus one object that compares a string: everything loads,
    exactly that object is flagged, and the others classify as the oracle says"""
    tree = {
        "mynet-ingress": ("dst net mynet", {
            "tcp": ("proto 6", {"syn": ("tfstr(tcp-flags) 'SYN'", {})}),
            "udp": ("proto 17", {"dns": ("src port 53", {}), "ntp": ("src port 123", {})}),
            "icmp": ("proto 1", {}),
            "frag": ("(proto 6 or 17) and (src port 0 or dst port 0)", {}),
            "web": ("vashost 'example.com'", {"tls": ("dst port 443", {})}),
        }),
        "mynet-egress": ("src net mynet", {"tcp": ("proto 6", {})}),
    }
    _write_tree(str(tmp_path / "mo"), tree)
    w = synth.workload_c1(4000, per_packet=32)
    for name, text in w["iplists"].items():
        eng.add_iplist(name, text)
        orc.iplist_add(name, text)
    eng.load_mos(str(tmp_path / "mo"))
    n = eng.mo_count()
    assert n == 12
    flagged = [eng.mo_name(i) for i in range(n) if eng.mo_host_only(i)]
    assert len(flagged) == 1 and flagged[0].endswith("web"), flagged
    gated = [eng.mo_name(i) for i in range(n) if eng.mo_host_gated(i)]
    assert len(gated) == 2 and gated[1].endswith("tls")
    eng.commit()
    # the oracle evaluates the string object itself: an absent vashost equals every operand
    # (strncmp over a size of 0, filter.c:641), so it takes what its parent takes - which is
    # also what the device marks as candidates
    for i in range(n):
        dump = eng.mo_dump(i)
        orc.mo_add(_filter_of(tree, eng.mo_name(i)), eng.mo_parent(i))
        assert dump
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"])
    compare(b, orc)


def _filter_of(tree, name, prefix=""):
    """filter text of the object whose name is its directory names joined with '_' (monit-objects.c:162-167)"""
    for k, (filt, kids) in tree.items():
        full = (prefix + "_" + k) if prefix else k
        if full == name:
            return filt
        r = _filter_of(kids, name, full)
        if r is not None:
            return r
    return None
