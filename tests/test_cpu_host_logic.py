"""Host-side logic of the product library, no GPU needed: the C ABI loads and
exports what include/xe_gpu.h declares, the filter compiler produces the same
postfix programs as the oracle, the flattened IP lists give the oracle's
answers, the datagram pre-pass finds the oracle's record counts, and the
configuration readers walk the reference's sample files."""
import json
import os
import re
import struct

import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from host_binding import Host, walk_flat
from helpers import configure, run_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KNOWN = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))
EXP = bytes([192, 0, 2, 1])


@pytest.fixture()
def host():
    h = Host()
    yield h
    h.close()


def test_library_exports_every_declared_symbol():
    L = xenoeye_amd.load_library()
    text = open(os.path.join(ROOT, "include", "xe_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(xe_[a-z0-9_]+)\s*\(", text))
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(L, n), "libxenoeye_gpu.so does not export %s" % n
    assert L.xe_gpu_build_arch().decode() == "gfx950"
    # the test hooks of xe_hosttest.c are a library of their own, not part of the product library
    assert not hasattr(L, "xe_ht_create")


def test_shim_exports_the_reference_entry_points():
    import ctypes
    so = os.path.join(ROOT, "xenoeye_amd", "libxenoeye_gpu_shim.so")
    assert os.path.exists(so), "build() makes the shim"
    L = ctypes.CDLL(so)
    text = open(os.path.join(ROOT, "include", "xenoeye_gpu_shim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b((?:xenoeye_gpu_shim_|netflow_process|sflow_process)[a-z0-9_]*)\s*\(", text))
    assert {"netflow_process_init", "netflow_process", "sflow_process", "xenoeye_gpu_shim_init"} <= names
    for n in sorted(names):
        assert hasattr(L, n), "libxenoeye_gpu_shim.so does not export %s" % n
    # struct flow_packet_info as flow-info.h:35-46 lays it out on x86-64
    import shim_binding
    F = shim_binding.FlowPacketInfo
    assert (F.src_addr_ipv4.offset, F.source_id.offset, F.epoch.offset, F.time_ns.offset, F.rawpacket.offset,
            F.sampling_rate.offset, ctypes.sizeof(F)) == (16, 20, 24, 32, 40, 65576, 65584)


def test_no_gpu_means_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(xenoeye_amd.XeError):
        xenoeye_amd.Engine(device=0)


def test_field_table_equals_oracle(oracle):
    mine = xenoeye_amd.fields()
    theirs = oracle.fields()
    assert len(mine) == len(theirs) == 66
    for a, b in zip(mine, theirs):
        assert (a["name"], a["id"], a["min"], a["max"], a["off"]) == (b["name"], b["id"], b["min"], b["max"], b["off"])


FILTER_CORPUS = [c["filter"] for c in KNOWN["filter_dump"]] + [
    # the distinct filters of the reference's sample object tree (lxc/mo/**/mo.conf)
    "src net mynet", "dst net mynet", "src net6 mynet", "dst net6 mynet", "proto 6", "proto 17", "proto 1",
    "src port 53", "src port 123", "(proto 6 or 17) and (src port 0 or dst port 0)",
    # more shapes
    "not proto 6", "not (proto 6)", "not (not proto 6)", "proto 6 and not proto 17 or proto 1",
    "host 1.2.3.4 or 5.6.7.8 or mynet or 10.0.0.0/8", "src host mynet and dst host6 mynet",
    "tos 0-7 and tcp-flags 0x12 and ifidx 1 or 2 or 3-5", "src as 64512-65534 and dst as 0",
    "min-ttl 1 and max-ttl 255 and frag-id 7 and endreason 1 and d1qvlan 5 and d1qcvlan 6",
    "bgp-nh 10.0.0.1 and dir 0 and fwdst 64 and smplr 1 and vrf 3", "src vrf 3 or dst vrf 4",
    "mac 00:11:22:33:44:55 or aa:bb:cc:dd:ee:ff", "src mac zz:11:22:33:44:55", "dev-ip6 1.2.3.4",
    "PROTO 6 AND SRC PORT 80", "proto\t6\nand\r\nport 80", "port -5", "port 5-", "port 010", "port 0x10-0x20",
    "div(octets, packets) 0-100 or 200", "div_r(ioctets, roctets, 10) 5", "min(tos, proto) 3",
    "min(src tos, dst tos) 1", "min(host, port) 0", "div(bits, ipackets) 1 or rpackets 5",
    "proto 6 or (proto 17 and (port 53 or (port 123 and not port 124)))",
    "((((proto 6))))", "proto 6 or 17 or proto 1",
]

FILTER_ERRORS = KNOWN["filter_errors"] + [
    "proto 6 or and", "not not proto 6", "port 80 81", "host", "src dst port 1", "div(octets, packets, 3) 1", "min(octets, packets) 1",
    "min(tos) 1", "proto (6)", "net 10.0.0.0/a", "proto 6 and (port 80", ")", "proto 6 or (", "'abc'",
]


def test_filter_compiler_matches_oracle(host, orc):
    text = "10.0.0.0/8\n2001:db8::/32\n"
    host.add_iplist("mynet", text)
    orc.iplist_add("mynet", text)
    nerr = 0
    for f in FILTER_CORPUS:
        rc, dump, err = host.filter_dump(f)
        try:
            want = orc.filter_dump(orc.filter_parse(f))
        except ValueError:
            # e.g. the token after a value-list "or" is swallowed (filter-parser.c:60-75)
            assert rc == -4, (f, rc, dump)
            nerr += 1
            continue
        assert rc in (0, -5), (f, rc, err)
        assert dump == want, f
    assert nerr <= 6
    for f in FILTER_ERRORS:
        rc, dump, err = host.filter_dump(f)
        assert rc == -4, (f, rc, dump)
        with pytest.raises(ValueError):
            orc.filter_parse(f)


def _random_filter(rng, depth=0):
    fields = [("proto", lambda: str(rng.integers(0, 256))), ("port", lambda: "%d-%d" % tuple(sorted(rng.integers(0, 65536, 2)))),
              ("net", lambda: "%d.%d.0.0/%d" % (rng.integers(0, 256), rng.integers(0, 256), rng.integers(0, 33))),
              ("host", lambda: "mynet"), ("net6", lambda: "2001:db8:%x::/%d" % (rng.integers(0, 65536), rng.integers(0, 129))),
              ("tcp-flags", lambda: hex(int(rng.integers(0, 64)))), ("as", lambda: str(rng.integers(0, 70000))),
              ("mac", lambda: ":".join("%02x" % rng.integers(0, 256) for _ in range(6))), ("vlan", lambda: str(rng.integers(0, 4096)))]
    r = rng.integers(0, 10)
    if depth < 3 and r < 3:
        return "(%s) %s (%s)" % (_random_filter(rng, depth + 1), ["and", "or"][rng.integers(0, 2)], _random_filter(rng, depth + 1))
    if depth < 3 and r == 3:
        return "not " + _random_filter(rng, depth + 1)
    if r == 4:
        return "%s %s %s" % (_random_filter(rng, depth + 1), ["and", "or"][rng.integers(0, 2)], _random_filter(rng, depth + 1)) if depth < 3 else "proto 6"
    name, gen = fields[rng.integers(0, len(fields))]
    d = ["", "src ", "dst "][rng.integers(0, 3)]
    vals = " or ".join(gen() for _ in range(int(rng.integers(1, 4))))
    return "%s%s %s" % (d, name, vals)


def test_filter_compiler_matches_oracle_on_random_filters(host, orc):
    text = "10.0.0.0/8\n2001:db8::/32\n"
    host.add_iplist("mynet", text)
    orc.iplist_add("mynet", text)
    rng = np.random.default_rng(3)
    for _ in range(400):
        f = _random_filter(rng)
        rc, dump, err = host.filter_dump(f)
        try:
            want = orc.filter_dump(orc.filter_parse(f))
        except ValueError:
            assert rc == -4, f
            continue
        assert rc in (0, -5), (f, rc, err)
        assert dump == want, f


def test_filters_the_device_cannot_decide_are_reported(host, orc):
    """they compile (same postfix program as the oracle's) and are flagged: objects with such a filter
    stay with the caller's own filter_match()"""
    for f in ["vashost 'example.com'", "class0 'web'", "proto tcp", "sni 'x'", "port aa:bb:cc:dd:ee:ff", "net 5",
              "src vashost 'a' or 'b'", "mfreq(src port, dst port) 80 or 443", "asn(src host) 64512-65000",
              "asd(dst host) 'EXAMPLE-AS'", "country(dst host) 'NL' or 'BE'", "CITY(src host6) 'x'",
              "country_code(dst host) 'x'", "tfstr(vashost) 'SYN'",
              # no port-name table on this side
              "portstr(src port) 'http (80)'", "ppstr(src port, dst port) '-> 80'",
              "proto 6 and (portstr(dst port) 'https (443)' or 'x') or not sni 'y'"]:
        rc, dump, err = host.filter_dump(f)
        assert rc == -5 and err, (f, rc, err)
        if f not in ("proto tcp", "port aa:bb:cc:dd:ee:ff", "net 5"):   # operands the reference never initialises
            assert dump == orc.filter_dump(orc.filter_parse(f)), f
    for f in ["div_l(octets, packets, 10) 100", "div_l(bits, packets, 2) 1-1024 or 4096"]:
        rc, dump, err = host.filter_dump(f)
        assert rc == 0 and dump == orc.filter_dump(orc.filter_parse(f)), (f, rc, err)
    for f in ["mfreq(src port) 1", "mfreq src port", "portstr(octets) 'x'", "ppstr(src port) 'x'", "asn() 1",
              "country(host, host) 'x'", "country_code(host) 'x'", "portstr(port) 'x'", "lat", "mfreq(src port, dst port)"]:
        rc, dump, err = host.filter_dump(f)
        assert rc == -4, (f, rc, dump)
        with pytest.raises(ValueError):
            orc.filter_parse(f)


def test_div_l_steps_equal_the_oracle(host, orc):
    """the pieces the quotient line is cut into at commit against xdiv() (filter.h:396-418) evaluated by
    the oracle with the same libm: every piece end, and the neighbourhood of every power of k"""
    import ctypes
    L = host.L
    L.xe_div_l_pieces.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_int32)] * 3 + [ctypes.c_int]
    orc.L.xo_xdiv.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
    for k in [2, 3, 7, 10, 16, 100, 1000, 46341, 1, 0, -1, -2, -10]:
        lo, hi, val = [(ctypes.c_int32 * 80)() for _ in range(3)]
        n = L.xe_div_l_pieces(k, lo, hi, val, 80)
        assert n >= (3 if k > 1 else 1), (k, n)
        assert lo[0] == -2 ** 31 and hi[n - 1] == 2 ** 31 - 1
        probes = set()
        for i in range(n):
            if i:
                assert lo[i] == hi[i - 1] + 1
            for q in (lo[i], hi[i], (lo[i] + hi[i]) // 2):
                probes.add((q, val[i]))
        for q, want in probes:
            # the reference divides u64 values and truncates the quotient to int: q / 1 with q as u64
            got = orc.L.xo_xdiv(q & 0xFFFFFFFF if q >= 0 else (q + 2 ** 32), 1, 1, k)
            assert got == want, (k, q, got, want)
        if k > 1:
            j = 1
            while j < 2 ** 31:
                for q in (j - 1, j, j + 1):
                    if 0 < q < 2 ** 31:
                        piece = [i for i in range(n) if lo[i] <= q <= hi[i]][0]
                        assert orc.L.xo_xdiv(q, 1, 1, k) == val[piece], (k, q)
                j *= k


def test_flattened_lists_match_oracle(host, orc):
    rng = synth.Rng(1234)
    lists = {
        "nested": "10.0.0.0/8\n10.1.0.0/16\n10.1.2.0/24\n10.1.2.3/32\n0.0.0.0/1\n",
        "zero": "0.0.0.0/0\n::/0\n",
        "weird": "7.7.7.7/33\n9.9.9.9/40\n8.0.0.0/-1\n2001:db8::/200\n",
        "empty": "# nothing\n",
        "rand": synth.iplist_text(synth.prefix_list_v4(rng, 4000, np.arange(1, 33), nested_frac=0.3),
                                  synth.prefix_list_v6(rng, 2000, lo=3, hi=128)),
    }
    for n, t in lists.items():
        host.add_iplist(n, t)
        orc.iplist_add(n, t)
    words, devs = host.flatten()
    # ... and with the roots over 24 address bits (one group of lists here: 2 x 64 MB of tables)
    words24, devs24 = host.flatten(root_bits=24)
    assert all(d[0].rbits == 16 and d[1].rbits == 16 for d in devs)
    assert all(d[0].rbits == 24 and d[1].rbits == 24 for d in devs24 if d[0] or d[1])
    probe4 = [struct.pack(">I", x) for x in [0, 0xFFFFFFFF, 0x0A010203, 0x0A010204, 0x0A010300, 0x0A020000,
                                              0x07070707, 0x07070706, 0x09090909, 0x08000001, 0x7FFFFFFF, 0x80000000,
                                              # below 10.1.2.0/24 towards 10.1.2.3/32: leaves with the first bit
                                              # (stays at the /24), with a later one (ends at a node without a prefix)
                                              0x0A010280, 0x0A010240, 0x0A010202, 0x0A010207]]
    r4 = (rng.u64(4000) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    probe4 += [struct.pack(">I", int(x)) for x in r4]
    v4 = synth.prefix_list_v4(synth.Rng(1234), 4000, np.arange(1, 33), nested_frac=0.3)
    inside = synth.addrs_in_prefixes_v4(rng, 3000, *v4)
    probe4 += [struct.pack(">I", int(x)) for x in inside]
    # the same addresses with one bit flipped: walks that leave a path part of the way down
    flip = (np.uint32(1) << (rng.u64(3000) % np.uint64(28)).astype(np.uint32))
    probe4 += [struct.pack(">I", int(x)) for x in (inside.astype(np.uint32) ^ flip)]
    probe6 = [bytes(16), bytes([255] * 16), bytes.fromhex("20010db8000000000000000000000001")]
    probe6 += [bytes(r) for r in rng.bytes(1500 * 16).reshape(-1, 16)]
    for (name, _), (root4, root6, nodes4, nodes6, bit, chains4, chains6) in zip(lists.items(), devs):
        for a in probe4:
            assert walk_flat(words, root4, nodes4, a, bit, chains4) == orc.match4(name, a), (name, a.hex())
        for a in probe6:
            assert walk_flat(words, root6, nodes6, a, bit, chains6) == orc.match6(name, a), (name, a.hex())
    for (name, _), (root4, root6, nodes4, nodes6, bit, chains4, chains6) in zip(lists.items(), devs24):
        for a in probe4[::3]:
            assert walk_flat(words24, root4, nodes4, a, bit, chains4) == orc.match4(name, a), (name, a.hex())
        for a in probe6[::3]:
            assert walk_flat(words24, root6, nodes6, a, bit, chains6) == orc.match6(name, a), (name, a.hex())


def _count_cases():
    """batches whose record counts the pre-pass must get right (the same
    shapes the GPU parity tests decode)"""
    cases = []
    rng = synth.Rng(77)
    t = synth.TMPL_V9_C1
    rec = synth.flow_columns(rng, 6, t)
    for count in (0, 1, 2, 3, 6, 7, 9, 100):
        for nrec in (0, 1, 2, 6):
            for extra in (b"", b"\xc0\xa8\x01\x01", b"\0" * 30):
                b = synth.Batch()
                b.add(synth.v9_template_packet(256, t, 77), EXP)
                body = struct.pack(">HH", 256, 4 + 21 * nrec) + rec[:nrec].tobytes() + extra
                b.add(synth.v9_header(count, 77) + body, EXP)
                cases.append(b)
    # templates with 1-byte first field: partial iterations keep consuming
    t2 = [(4, 1), (8, 4)]
    r2 = synth.flow_columns(rng, 3, t2)
    for count in (1, 5, 12):
        b = synth.Batch()
        b.add(synth.v9_template_packet(300, t2, 1), EXP)
        b.add(synth.v9_header(count, 1) + struct.pack(">HH", 300, 4 + 15) + r2.tobytes(), EXP)
        cases.append(b)
    # ipfix: padding, short sets, tiny templates
    for t3 in ([(8, 4), (7, 2), (4, 1)], [(4, 1)], synth.TMPL_IPFIX_V4):
        rl = sum(l for _, l in t3)
        r3 = synth.flow_columns(rng, 4, t3)
        for nrec in (0, 1, 4):
            for pad in (0, 1, 2, 3, 5):
                for lie in (0, -3, 7):
                    b = synth.Batch()
                    b.add(synth.ipfix_template_packet(256, t3, 9), EXP)
                    body = struct.pack(">HH", 256, max(0, 4 + rl * nrec + pad + lie)) + r3[:nrec].tobytes() + b"\0" * pad
                    b.add(synth.ipfix_header(16 + len(body), 9) + body, EXP)
                    cases.append(b)
    return cases


def test_prepass_record_counts_match_oracle(host, orc):
    orc.mo_add("")
    for b in _count_cases():
        h = Host()
        orc.reset()
        orc.mo_add("")
        run_oracle(orc, b, keep_flows=False)
        plan = h.prepass(b)
        h.close()
        assert plan["nrec"] == orc.nrecords(), [p.hex() for p in b.pkts]


def test_prepass_counts_on_mutated_datagrams(orc):
    """random corruption of headers, set ids, lengths and counts"""
    rng = np.random.default_rng(5)
    w1 = synth.workload_c1(640)
    w2 = synth.workload_c2(600, prefixes=10)
    base = w1["batch"].pkts[:6] + w2["batch"].pkts[:140:7] + [synth.v5_packet(synth.Rng(1), 5)]
    for trial in range(300):
        b = synth.Batch()
        b.add(synth.v9_template_packet(256, synth.TMPL_V9_C1, 77), EXP)
        for e in range(3):
            b.add(synth.ipfix_template_packet(256, synth.TMPL_IPFIX_V4, e + 1), bytes([10, 255, 0, e + 1]))
            b.add(synth.ipfix_template_packet(257, synth.TMPL_IPFIX_V6, e + 1), bytes([10, 255, 0, e + 1]))
        for _ in range(6):
            p = bytearray(base[int(rng.integers(0, len(base)))])
            for _ in range(int(rng.integers(0, 4))):
                pos = int(rng.integers(0, min(len(p), 48)))
                p[pos] = int(rng.integers(0, 256))
            if rng.integers(0, 5) == 0:
                p = p[:int(rng.integers(0, len(p)))]
            src = EXP if p[:2] == b"\0\x09" else bytes([10, 255, 0, int(rng.integers(1, 4))])
            b.add(bytes(p), src)
        h = Host()
        orc.reset()
        orc.mo_add("")
        run_oracle(orc, b, keep_flows=False)
        plan = h.prepass(b)
        h.close()
        assert plan["nrec"] == orc.nrecords(), trial


def test_fixed_window_field_parser_matches_oracle(host, orc):
    """`fields` strings of an mo.conf "fwm" entry (parse_field(), filter-parser.c:396-484)"""
    good = ["src host", "dst host desc", "proto", "octets", "bits desc", "packets asc", "src port", "dst port",
            "src host6", "dst mac", "src as", "tcp-flags", "src tos", "src ifidx", "dst vlan", "SRC HOST DESC", "dev-ip",
            "ioctets", "rpackets", "src net", "dst net6 asc",
            # the function forms (filter-parser.c:413-442)
            "div(octets, packets)", "div_r(octets, packets, 100) desc", "div_l(bits, packets, 2)", "min(src port, dst port)",
            "MIN(src as, dst as) asc", "tfstr(tcp-flags)", "portstr(src port)", "portstr(dst port) desc",
            "ppstr(src port, dst port)"]
    for f in good:
        rc, got, err = host.field_parse(f)
        assert rc == 0, (f, err)
        assert got == orc.field_parse(f), f
    assert host.field_parse("bits")[1][3] == 8 and host.field_parse("src host6")[1][2] == 16
    for f in ["", "host", "port", "tos", "src", "src nothing", "octets extra", "src host desc desc", "dev-id", "dev-mark",
              "src host asc desc", "proto 6", "(proto)", "div(octets)", "div(src port, packets)", "div_r(octets, packets)",
              "min(src port)", "min(port, dst port)", "tfstr tcp-flags", "portstr(src port) 80", "ppstr(src port)",
              "div(octets, packets", "min(src port, dst port) desc asc"]:
        rc, got, err = host.field_parse(f)
        assert rc == -4, (f, rc, got)
        with pytest.raises(ValueError):
            orc.field_parse(f)
    for f in ["mfreq(src port, dst port)", "asn(src host)", "asd(dst host)", "country(src host)", "vashost", "sni",
              "tfstr(sni)"]:
        rc, got, err = host.field_parse(f)
        assert rc == -5 and err, (f, rc)


def test_prepass_with_threads_equals_single_thread():
    """the pre-pass sharded over threads by exporter gives the same plan as
    the single-threaded walk: set order, record indices, exporter table"""
    w5 = synth.workload_c5(300000, prefixes=50)
    w2 = synth.workload_c2(150000, prefixes=50)
    # templates that change in mid-stream, and v5 datagrams in between
    b = synth.Batch()
    for i, (p, s, k) in enumerate(zip(w2["batch"].pkts, w2["batch"].src, w2["batch"].kind)):
        b.add(p, s, k)
        if i % 997 == 500:
            e = (i // 997) % 64
            b.add(synth.ipfix_template_packet(256, synth.TMPL_IPFIX_V6 if (i // 997) % 2 else synth.TMPL_IPFIX_V4, e + 1),
                  bytes([10, 255, 0, e + 1]))
        if i % 1501 == 7:
            b.add(synth.v5_packet(synth.Rng(i), 9), bytes([10, 77, 0, i % 5]))
    # sFlow datagrams (dealt to the threads by position) between the others; some
    # IPFIX datagrams come from an agent's address
    w3 = synth.workload_c3(42000, seed=31)
    mixed = synth.Batch()
    for i, (p, s) in enumerate(zip(w3["batch"].pkts, w3["batch"].src)):
        mixed.add(p, s, 1)
        if i % 2 == 0 and i // 2 < len(w2["batch"].pkts):
            j = i // 2
            mixed.add(w2["batch"].pkts[j], s if j % 30 == 7 else w2["batch"].src[j], 0)
    for batch in (w5["batch"], b, w3["batch"], mixed):
        assert len(batch) >= 4096
        plans = []
        for T in (1, 2, 7, 16):
            h = Host()
            plans.append(h.prepass(batch, threads=T))
            h.close()
        for p in plans[1:]:
            assert p["nrec"] == plans[0]["nrec"] and p["skipped"] == plans[0]["skipped"]
            assert p["nsf"] == plans[0]["nsf"]
            assert p["exporters"] == plans[0]["exporters"]
            assert len(p["sets"]) == len(plans[0]["sets"])
            for a, c in zip(p["sets"], plans[0]["sets"]):
                # device template slots may be numbered in another order
                assert {k: v for k, v in a.items() if k != "tmpl"} == {k: v for k, v in c.items() if k != "tmpl"}


def test_prepass_threads_under_thread_sanitizer(tmp_path):
    """the sharded pre-pass built with -fsanitize=thread (host code only): no
    data race reported, and templates that change in mid-stream give the same
    plan as the one-thread walk"""
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prepass_threads_driver.c")
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xenoeye_amd", "csrc")
    exe = str(tmp_path / "pp_tsan")
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-w", "-fsanitize=thread", "-pthread", "-I", csrc, src] + \
          [os.path.join(csrc, f) for f in ("xe_prepass.c", "xe_fields.c", "xe_filter.c", "xe_iplist.c", "xe_conf.c")] + ["-o", exe, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "tsan" in (r.stderr or "").lower():
        pytest.skip("no ThreadSanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "same plans" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]


def test_prepass_exporters_and_device_lookup(host):
    # id-only fallthrough of devices.c:224: entry 0 has ip+id, ip differs, id equals -> still found
    host.L.xe_ht_add_device(host.h, 1, int.from_bytes(bytes([10, 9, 9, 9]), "little"), 1,
                            int.from_bytes(struct.pack(">I", 77), "little"), 7, 0)
    host.L.xe_ht_add_device(host.h, 1, int.from_bytes(EXP, "little"), 0, 0, 9, 0)
    w = synth.workload_c1(128)
    w["batch"].add(synth.v9_template_packet(256, synth.TMPL_V9_C1, 78), EXP)
    w["batch"].extend(synth.v9_data_packets(256, synth.flow_columns(synth.Rng(1), 3, synth.TMPL_V9_C1), 3, 78), EXP)
    plan = host.prepass(w["batch"])
    ex = plan["exporters"]
    assert [e["rate"] for e in ex] == [7, 9]
    assert [e["device"] for e in ex] == [0, 1]
    assert ex[0]["dev_id"] == struct.pack(">I", 77) and ex[0]["dev_ip"] == EXP


def test_mo_directory_walk(tmp_path, host):
    def mk(path, text):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "mo.conf"), "w") as f:
            f.write(text)
    root = str(tmp_path / "mo")
    mk(os.path.join(root, "a"), '{ /* c */ "filter": "proto 6", "fwm": [ {"name": "x", "fields": ["packets"]} ] }')
    mk(os.path.join(root, "a", "b"), '{"filter": "dst port 80", // trailing\n "debug": {"dump-flows": "none"}}')
    mk(os.path.join(root, "a", "b", "c"), '{"mavg": [], "filter": "src port 1-1023"}')
    mk(os.path.join(root, "nofilter"), '{"fwm": []}')
    mk(os.path.join(root, "broken"), '{"filter": "proto 6"')
    mk(os.path.join(root, "broken", "child"), '{"filter": "proto 17"}')
    os.makedirs(os.path.join(root, ".hidden"))
    os.makedirs(os.path.join(root, "noconf"))
    with open(os.path.join(root, "file.txt"), "w") as f:
        f.write("x")
    rc, rows = host.list_mos(root)
    assert rc == 0
    got = {r[0]: r for r in rows}
    assert set(got) == {"a", "a_b", "a_b_c", "nofilter"}
    names = [r[0] for r in rows]
    assert got["a"][1] == -1 and got["a"][2] == "proto 6"
    assert got["a_b"][1] == names.index("a") and got["a_b"][2] == "dst port 80"
    assert got["a_b_c"][1] == names.index("a_b")
    assert got["nofilter"][2] == ""
    assert names.index("a") < names.index("a_b") < names.index("a_b_c")


@pytest.mark.skipif(not os.path.isdir("/root/reference/lxc/mo"), reason="reference checkout absent")
def test_reference_sample_object_tree_loads(host):
    rc, rows = host.list_mos("/root/reference/lxc/mo")
    assert rc == 0
    assert len(rows) == 18
    byname = {r[0]: r for r in rows}
    names = [r[0] for r in rows]
    assert byname["mynet-ingress"][2] == "dst net mynet"
    assert byname["mynet-ingress_frag"][2] == "(proto 6 or 17) and (src port 0 or dst port 0)"
    assert byname["mynet-ingress_frag"][1] == names.index("mynet-ingress")
    # every filter of the tree parses with the product's compiler once the list exists
    host.add_iplist("mynet", "10.0.0.0/8\n")
    unsupported = []
    for name, parent, filt in rows:
        rc, dump, err = host.filter_dump(filt)
        if rc != 0:
            unsupported.append((name, filt, rc))
    assert unsupported == []


def test_tfstr_compiles_like_the_oracle(host, orc):
    """tfstr() (filter-parser-funcs.c:526-580): same dump text, same errors"""
    good = ["tfstr(tcp-flags) 'SYN'", "tfstr(tcp-flags) 'syn' or 'ACK+SYN' or 'nonsense'",
            "not tfstr(tcp-flags) 'RST' and proto 6", "tfstr(proto) 'FIN' or dst port 80",
            "tfstr(src port) 'SYN'", "tfstr(src host) 'CWR+ECE+URG+ACK+PSH+RST+SYN+FIN'",
            "proto 6 and ((tfstr(tcp-flags) 'SYN') or (tfstr(tcp-flags) 'ACK+SYN'))"]
    for f in good:
        rc, dump, err = host.filter_dump(f)
        assert rc == 0, (f, rc, err)
        assert dump == orc.filter_dump(orc.filter_parse(f)), f
    for f in ["tfstr tcp-flags 'SYN'", "tfstr(port) 'SYN'", "tfstr(dev-id) 'SYN'", "tfstr(tcp-flags 'SYN'",
              "tfstr(tcp-flags)", "tfstr(octets) 'SYN'",
              # the token after a value list's `or` is dropped (filter-parser.c:79)
              "proto 6 and (tfstr(tcp-flags) 'SYN' or tfstr(tcp-flags) 'ACK+SYN')"]:
        rc, dump, err = host.filter_dump(f)
        assert rc == -4, (f, rc, dump)
        with pytest.raises(ValueError):
            orc.filter_parse(f)
    # operands without text have no defined meaning in the reference: refused
    for f in ["tfstr(tcp-flags) 2", "tfstr(tcp-flags) SYN", "tfstr(vashost) 'SYN'"]:
        rc, dump, err = host.filter_dump(f)
        assert rc == -5 and err, (f, rc)


def test_devices_conf_reader(tmp_path, host):
    p = tmp_path / "devices.conf"
    p.write_text('''[
        { "ip": "127.0.0.1", "id": 0, "sampling-rate": 1 },
        // comment
        { "ip": "1.2.3.4", "sampling-rate": 1000, "mark": ["proto 6", "dst port 80"], "skip-unmarked": true },
        { "id": "77", "mark": "src net 10.0.0.0/8", "unknown-key": {"a": [1, 2, {"b": null}]} }
    ]''')
    rc, rows = host.list_devices(str(p))
    assert rc == 0
    assert rows[0][:6] == ["1", str(int.from_bytes(bytes([127, 0, 0, 1]), "little")), "1", "0", "1", "0"]
    assert rows[1][:6] == ["1", str(int.from_bytes(bytes([1, 2, 3, 4]), "little")), "0", "0", "1000", "1"]
    assert rows[1][6:] == ["proto 6", "dst port 80"]
    assert rows[2][:6] == ["0", "0", "1", "77", "0", "0"] and rows[2][6:] == ["src net 10.0.0.0/8"]
    if os.path.exists("/root/reference/devices.conf"):
        rc, rows = host.list_devices("/root/reference/devices.conf")
        assert rc == 0 and len(rows) == 2 and rows[1][4] == "1000"


def test_empty_sets_cannot_blow_a_batch_up(host, orc):
    """count = 65535 and a datagram full of empty data sets of a known template: every 4 bytes stand
    for 65535 empty records (netflow.c:272).  Such a datagram is left out and counted; its neighbours
    are walked as ever, and the batch does not fail."""
    b = synth.Batch()
    b.add(synth.v9_template_packet(256, synth.TMPL_V9_C1, 77), EXP)
    good = synth.workload_c1(128, per_packet=64)["batch"]
    b.add(good.pkts[1], EXP)
    hostile = synth.v9_header(65535, 77) + struct.pack(">HH", 256, 0) * 340
    for _ in range(3):
        b.add(hostile, EXP)
    b.add(good.pkts[2], EXP)
    # fewer empty sets stay under the bound and are walked like the reference walks them
    mild = synth.v9_header(1000, 77) + struct.pack(">HH", 256, 0) * 10
    b.add(mild, EXP)
    for threads in (1, 4):
        os.environ["XE_PREPASS_MIN_DATAGRAMS"] = "1"
        try:
            plan = host.prepass(b, threads=threads)
        finally:
            del os.environ["XE_PREPASS_MIN_DATAGRAMS"]
        assert plan["oversized"] == 3 and plan["skipped"] >= 3
        assert plan["nrec"] == 64 + 64 + 10 * 1000
        # per datagram: the records it owns
        first = plan["dg_first_nf"]
        assert first == [0, 0, 64, 64, 64, 64, 128, 10128]
        assert plan["dg_first_sf"] == [0] * 8
    # the oracle walks the mild datagram the same way
    orc.mo_add("", -1)
    orc.log_config(keep_flows=False)
    orc.netflow(b.pkts[0], EXP)
    orc.netflow(mild, EXP)
    assert orc.nrecords() == 10000


def test_shim_idle_flush_wakes_the_capture_thread(tmp_path):
    """scapture_thread() (scapture.c:25-61) blocks in recvfrom() and has no idle
    branch: three datagrams, then silence.  The shim's wake-up datagram must bring
    the capture thread back so that the due batch is delivered - on that thread,
    once - with the flow-path library replaced by stand-ins (no GPU), also under
    ThreadSanitizer."""
    import subprocess
    src = [os.path.join(ROOT, "tests", "idle_flush_driver.c"), os.path.join(ROOT, "shim", "xenoeye_gpu_shim.c")]
    for extra in ([], ["-fsanitize=thread", "-g"]):
        exe = str(tmp_path / ("idle" + str(len(extra))))
        subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread"] + extra
                       + ["-I", os.path.join(ROOT, "include")] + src + ["-o", exe], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "ThreadSanitizer" not in r.stderr, r.stderr
        assert r.stdout.split() == ["calls=3", "on_capture_thread=1", "wakeups=1", "before_due=0"], r.stdout


def test_product_library_has_no_diagnostic_switches():
    """the kernel ablation / phase-timing switches exist only in builds made with -DXE_DIAG
    (tools/build_variant.sh): the shipped library does not even hold their names"""
    blob = open(os.path.join(ROOT, "xenoeye_amd", "libxenoeye_gpu.so"), "rb").read()
    for name in (b"XE_ABLATE", b"XE_FAST_BLOCKS", b"XE_FAST_R", b"XE_COUNTER_COPIES", b"XE_FORCE_GENERAL"):
        assert name not in blob, name
