"""The CPU oracle against the reference's own translation units (compiled
unmodified into oracle/_ref) and against recorded known answers."""
import json
import os
import socket
import struct
import tempfile

import numpy as np
import pytest

from oracle_binding import Oracle, RefGeo, RefIplist, RefSflow, RefUtils, have_ref, port_names, FLOW_INFO_SIZE
from xenoeye_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
KNOWN = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))

needs_ref_iplist = pytest.mark.skipif(not have_ref("libref_iplist.so"), reason="oracle/_ref not built")
needs_ref_sflow = pytest.mark.skipif(not have_ref("libref_sflow.so"), reason="oracle/_ref not built")
needs_ref_utils = pytest.mark.skipif(not have_ref("libref_utils.so"), reason="oracle/_ref not built")
needs_ref_geo = pytest.mark.skipif(not have_ref("libref_geo.so"), reason="oracle/_ref not built")

_uniq = [0]


def _ref_lists(texts):
    """load list texts into the reference's registry under fresh names"""
    ref = RefIplist()
    d = tempfile.mkdtemp(prefix="xe_lists_")
    names = []
    for t in texts:
        _uniq[0] += 1
        n = "l%d_%d" % (os.getpid(), _uniq[0])
        with open(os.path.join(d, n), "wb") as f:
            f.write(t if isinstance(t, bytes) else t.encode())
        names.append(n)
    assert ref.load_dir(d) == 1
    return ref, names


@needs_ref_iplist
def test_iplist_loader_quirks_match_reference(orc):
    texts = [
        "10.0.0.0/8\n10.1.0.0/16\n",                        # nested: deepest node, not longest prefix
        "192.168.0.0/16\n172.16.0.0/12",                    # last line without newline is dropped
        "# comment\n\n  8.8.8.8  \n1.2.3.0/24 # trailing\n2001:db8::/32\nfe80::1\n::/0\n",   # v6 default /32
        "0.0.0.0/0\n10.0.0.0/x\n300.1.1.1/8\n7.7.7.7/33\n1.1.1.1/0\n",
        "",
        "\n",
        "9.9.9.9/32\n9.9.9.8/31\n9.9.9.0/24\n9.9.0.0/16\n9.0.0.0/8\n",
    ]
    ref, names = _ref_lists(texts)
    rng = np.random.default_rng(7)
    probes4 = [socket.inet_aton(a) for a in ("10.1.2.3", "10.128.0.1", "10.0.0.1", "10.2.0.1", "192.168.1.1",
                                             "172.16.5.5", "8.8.8.8", "8.8.8.9", "1.2.3.77", "9.9.9.9", "9.9.9.8",
                                             "9.9.9.1", "9.9.1.1", "9.1.1.1", "0.0.0.0", "255.255.255.255",
                                             "1.1.1.1", "7.7.7.7")]
    probes4 += [bytes(rng.integers(0, 256, 4, dtype=np.uint8)) for _ in range(300)]
    probes6 = [socket.inet_pton(socket.AF_INET6, a) for a in ("2001:db8::1", "2001:db9::1", "fe80::1", "fe80::2",
                                                                "::", "ffff::")]
    probes6 += [bytes(rng.integers(0, 256, 16, dtype=np.uint8)) for _ in range(100)]
    for t, n in zip(texts, names):
        orc.iplist_add(n, t)
        h = ref.get(n)
        assert h, n
        for a in probes4:
            assert orc.match4(n, a) == ref.match4(h, a), (t, a)
        for a in probes6:
            assert orc.match6(n, a) == ref.match6(h, a), (t, a)


@needs_ref_iplist
def test_iplist_random_prefixes_match_reference(orc):
    rng = synth.Rng(99)
    v4 = synth.prefix_list_v4(rng, 3000, np.arange(0, 33), nested_frac=0.3)
    v6 = synth.prefix_list_v6(rng, 1500, lo=1, hi=128)
    text = synth.iplist_text(v4, v6)
    ref, (name,) = _ref_lists([text])
    orc.iplist_add(name, text)
    h = ref.get(name)
    a4 = np.concatenate([synth.addrs_in_prefixes_v4(rng, 3000, *v4), (rng.u64(3000) & np.uint64(0xFFFFFFFF)).astype(np.uint32)])
    for x in a4.tolist():
        b = struct.pack(">I", x)
        assert orc.match4(name, b) == ref.match4(h, b)
    a6 = np.concatenate([synth.addrs_in_prefixes_v6(rng, 1500, *v6), rng.bytes(1500 * 16).reshape(-1, 16)])
    for row in a6:
        b = bytes(row)
        assert orc.match6(name, b) == ref.match6(h, b)


def test_iplist_known_answers(orc):
    for case in KNOWN["iplist"]:
        orc.reset()
        orc.iplist_add("l", case["text"])
        for addr, want in case["v4"].items():
            assert orc.match4("l", socket.inet_aton(addr)) == want, (case["text"], addr)
        for addr, want in case.get("v6", {}).items():
            assert orc.match6("l", socket.inet_pton(socket.AF_INET6, addr)) == want, (case["text"], addr)


@needs_ref_utils
def test_mac_addr_read_matches_reference(orc):
    import ctypes as C
    ref = RefUtils()
    cases = ["00:11:22:33:44:55", "aa:BB:cc:DD:ee:ff", "zz:11:22:33:44:55", "00-11-22-33-44-55", "0:1:2:3:4:5",
             "00:11:22:33:44:5", "00:11:22:33:44:555", "1234567890123456-", "12:45:78:01:34:67", "::::::::::::::::-",
             "gg:hh:ii:jj:kk:ll", "10.0.0.0/8xxxxxxx:"]
    for s in cases:
        a = C.create_string_buffer(6)
        b = C.create_string_buffer(6)
        ra = ref.L.mac_addr_read(s.encode(), a)
        rb = orc.L.xo_mac_addr_read(s.encode(), b)
        assert ra == rb, s
        if ra:
            assert a.raw == b.raw, s


def _oracle_flags_text(orc, v):
    import ctypes as C
    out = C.create_string_buffer(40)
    orc.L.xo_tcp_flags_str.argtypes = [C.c_uint8, C.c_char_p]
    orc.L.xo_tcp_flags_str.restype = None
    orc.L.xo_tcp_flags_str(v, out)
    return out.value.decode()


@needs_ref_utils
def test_tcp_flags_text_matches_reference(orc):
    """the rule the oracle states for tfstr() against the reference's 256-entry table"""
    import ctypes as C
    ref = RefUtils()
    ref.L.tcp_flags_to_str.argtypes = [C.c_uint8]
    ref.L.tcp_flags_to_str.restype = C.c_char_p
    for v in range(256):
        assert _oracle_flags_text(orc, v) == ref.L.tcp_flags_to_str(v).decode(), v


@needs_ref_utils
def test_port_texts_match_reference(orc):
    """port_to_str() / ports_pair_to_str() as the oracle restates them (utils.c:28-77) over the names
    a binding derives from port_to_str(), against the reference's own functions: every port, and
    pairs that cover all branches (both named, one named either side, none, equal, both orders)"""
    ref = RefUtils()
    names = ref.port_names()
    named = [p for p in range(65536) if names[p]]
    assert 1000 < len(named) < 60000 and names[80] and names[443]
    orc.set_port_names(names)
    for p in range(65536):
        assert orc.port_to_str(p) == ref.port_to_str(p), p
    rng = np.random.default_rng(7)
    unnamed = [p for p in range(65536) if not names[p]]
    pools = {"n": named, "u": unnamed}
    pairs = [(p, p) for p in (0, 1, 80, 65535, unnamed[0], unnamed[-1])]
    for a in "nu":
        for b in "nu":
            xs = rng.choice(pools[a], 600)
            ys = rng.choice(pools[b], 600)
            pairs += list(zip(xs.tolist(), ys.tolist()))
    for p1, p2 in pairs:
        if (p1 > p2) and not names[p1] and names[p2]:
            # utils.c:70-71 has no `return` on this branch and falls off the end: same text either way
            pass
        assert orc.ports_pair_to_str(p1, p2) == ref.ports_pair_to_str(p1, p2), (p1, p2)


def test_tcp_flags_text_known_answers(orc):
    # filter.c:836-858 compares these texts; values as the reference's table holds them
    for v, want in {0: "", 1: "FIN", 2: "SYN", 0x12: "ACK+SYN", 0x18: "ACK+PSH", 0x20: "URG", 0x40: "ECE",
                    0x80: "CWR", 0xFF: "CWR+ECE+URG+ACK+PSH+RST+SYN+FIN"}.items():
        assert _oracle_flags_text(orc, v) == want


def test_tfstr_match_reads_the_reference_byte(orc):
    """filter.c:842-847: byte 1 of a 2-byte field, byte 0 otherwise; no presence test"""
    fields = {f["name"]: f for f in orc.fields()}
    flow = bytearray(FLOW_INFO_SIZE)
    h_syn = orc.filter_parse("tfstr(tcp-flags) 'syn' or 'ACK+SYN'")
    h_none = orc.filter_parse("not tfstr(tcp-flags) 'FIN' or 'SYN' or 'ACK+SYN' or 'RST'")   # '' is not a token
    h_proto = orc.filter_parse("tfstr(proto) 'RST+SYN'")
    h_port = orc.filter_parse("tfstr(dst port) 'FIN'")
    assert not orc.filter_match(h_syn, bytes(flow)) and orc.filter_match(h_none, bytes(flow))
    off = fields["tcp_flags"]["off"]
    flow[off + 1] = 0x02
    assert orc.filter_match(h_syn, bytes(flow)) and not orc.filter_match(h_none, bytes(flow))
    flow[off + 1] = 0x12
    assert orc.filter_match(h_syn, bytes(flow))
    flow[off], flow[off + 1] = 0x02, 0x00           # the sFlow placement of the flags byte
    assert not orc.filter_match(h_syn, bytes(flow)) and orc.filter_match(h_none, bytes(flow))
    flow[fields["protocol"]["off"]] = 0x06
    assert orc.filter_match(h_proto, bytes(flow))
    po = fields["l4_dst_port"]["off"]
    flow[po], flow[po + 1] = 0x01, 0x00
    assert not orc.filter_match(h_port, bytes(flow))
    flow[po], flow[po + 1] = 0x00, 0x01
    assert orc.filter_match(h_port, bytes(flow))


def _flow_sets(flows, fields):
    """per record: {field: (size, bytes)} of the fields that are set"""
    out = []
    for r in flows:
        d = {}
        for f in fields:
            if int.from_bytes(r[f["off_has"]:f["off_has"] + 4].tobytes(), "little"):
                d[f["name"]] = (int.from_bytes(r[f["off_size"]:f["off_size"] + 4].tobytes(), "little"),
                                bytes(r[f["off"]:f["off"] + f["max"]]))
        out.append(d)
    return out


@needs_ref_sflow
def test_sflow_walk_matches_reference_on_synthetic_mix(orc):
    ref = RefSflow()
    w = synth.workload_c3(350)
    orc.mo_add("")
    orc.log_config(keep_flows=True)
    total = 0
    for pkt, src in zip(w["batch"].pkts, w["batch"].src):
        n0 = orc.nrecords()
        rc = orc.sflow(pkt, src)
        rrc, recs = ref.process(pkt)
        assert rc == rrc
        mine = orc.flows()[n0:]
        assert mine.shape == recs.shape
        assert np.array_equal(mine, recs)
        total += len(recs)
    assert total > 300


@needs_ref_sflow
def test_sflow_walk_matches_reference_on_special_cases(orc):
    ref = RefSflow()
    rng = synth.Rng(5)
    agent = bytes([10, 0, 0, 1])
    pkts = []
    for kind in ["ipip", "6in6", "4in6", "frag", "arp", "v6udp", "v4icmp", "v4tcp", "v6tcp", "v4udp"]:
        for vm in (0, 1, 2):
            for cutlen in (128, 96, 70, 58, 54, 40, 20, 14):
                hdr = (synth.sflow_header_sample(rng, kind, vm) + rng.bytes(128).tobytes())[:128]
                el = synth.sflow_raw_element(hdr, 777, header_len=cutlen)
                els = [el, synth.sflow_ipv4_element(rng, proto=int(rng.below(1, 3)[0]) * 5 + 1), (1001, bytes(16))]
                s1 = synth.sflow_flow_sample(1, 33, 0x40000005, 0x80000006, els, expanded=False)
                s2 = synth.sflow_flow_sample(2, 34, 5, 6, [el], expanded=True)
                pkts.append(synth.sflow_datagram(agent, 3, [s1, s2]))
    pkts.append(synth.sflow_datagram(None, 3, [s1], agent6=bytes(range(1, 17))))
    pkts.append(pkts[0][:100])
    pkts.append(pkts[1][:37])
    pkts.append(b"")
    orc.mo_add("")
    orc.log_config(keep_flows=True)
    for pkt in pkts:
        n0 = orc.nrecords()
        rc = orc.sflow(pkt, agent)
        rrc, recs = ref.process(pkt)
        assert rc == rrc
        assert np.array_equal(orc.flows()[n0:], recs)


@needs_ref_sflow
def test_sflow_walk_matches_reference_on_mutated_datagrams(orc):
    """byte-level mutations of valid datagrams: lengths, tags and counts go
    wrong in every way; both walkers must deliver the same records"""
    ref = RefSflow()
    w = synth.workload_c3(140)
    rng = np.random.default_rng(11)
    orc.mo_add("")
    orc.log_config(keep_flows=True)
    checked = 0
    for pkt in w["batch"].pkts:
        for _ in range(25):
            b = bytearray(pkt)
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(0, min(len(b), 200)))
                b[pos] = int(rng.integers(0, 256)) if rng.integers(0, 2) else (b[pos] ^ (1 << int(rng.integers(0, 8))))
            if rng.integers(0, 4) == 0:
                b = b[:int(rng.integers(0, len(b)))]
            n0 = orc.nrecords()
            rc = orc.sflow(bytes(b), b"\0\0\0\0")
            rrc, recs = ref.process(bytes(b))
            assert rc == rrc
            assert np.array_equal(orc.flows()[n0:], recs)
            checked += 1
        orc.log_clear()
    assert checked == 500


def test_filter_dump_known_answers(orc):
    orc.iplist_add("mynet", "10.0.0.0/8\n")
    for case in KNOWN["filter_dump"]:
        h = orc.filter_parse(case["filter"])
        assert orc.filter_dump(h) == case["dump"], case["filter"]


def test_filter_parse_errors(orc):
    for text in KNOWN["filter_errors"]:
        with pytest.raises(ValueError):
            orc.filter_parse(text)


def test_v9_decode_known_answers(orc):
    """the record counts and field presence the reference showed on these
    datagrams (template = 7 fields / 21 bytes)"""
    exp = bytes([192, 0, 2, 1])
    fields = orc.fields()
    for case in KNOWN["v9_decode"]:
        orc.reset()
        orc.mo_add("")
        orc.log_config(keep_flows=True)
        orc.netflow(synth.v9_template_packet(256, synth.TMPL_V9_C1, 77), exp)
        body = bytes.fromhex(case["body_hex"])
        assert orc.netflow(synth.v9_header(case["count"], 77) + body, exp) == 1
        sets = _flow_sets(orc.flows(), fields)
        assert len(sets) == case["records"], case["name"]
        for i, want in enumerate(case["present"]):
            got = sorted(k for k in sets[i] if k not in ("dev_ip", "dev_id"))
            assert got == sorted(want), (case["name"], i)
        for i, vals in case.get("values", {}).items():
            for k, hexv in vals.items():
                assert sets[int(i)][k][1].hex() == hexv, (case["name"], i, k)


def test_struct_layout_known_offsets(orc):
    off = {f["name"]: f["off"] for f in orc.fields()}
    for name, want in KNOWN["flow_info_offsets"].items():
        if name in off:
            assert off[name] == want, name
    assert orc.L.xo_off_sampling_rate() == KNOWN["flow_info_offsets"]["sampling_rate"]
    assert orc.L.xo_off_payload_ptr() == KNOWN["flow_info_offsets"]["payload_ptr"]


@pytest.mark.skipif(not os.path.exists("/root/reference/netflow.def"), reason="reference checkout absent")
def test_field_table_matches_reference_definition(orc):
    import re
    want = []
    for line in open("/root/reference/netflow.def"):
        m = re.match(r'FIELD\((\w+),\s*"[^"]*",\s*NF_FIELD_(\w+),\s*(\d+),\s*(\d+),\s*(\w+)\)', line)
        if m:
            mx = 32 if m.group(5) == "CLASS_NAME_MAX" else int(m.group(5))
            want.append((m.group(1), {"IP_ADDR": 0, "INT": 1, "MAC": 2, "STRING": 3}[m.group(2)], int(m.group(3)),
                         int(m.group(4)), mx))
    got = [(f["name"], f["type"], f["id"], f["min"], f["max"]) for f in orc.fields()[:63]]
    assert got == want


# ------------------------------------------------------------------ geo / AS databases
GEO_KINDS = {"as4": (1, 4), "as6": (1, 6), "geo4": (0, 4), "geo6": (0, 6)}


@needs_ref_geo
def test_geo_node_layout_matches_reference():
    """struct btrie_node_geo / btrie_node_as as the reference's compiler lays them out (geoip.h:43-55):
    the sizes and offsets the oracle and the library use for the database images"""
    ref = RefGeo()
    assert (ref.node_size(0), ref.node_size(1)) == (228, 216)
    assert (ref.payload_offset(0), ref.payload_offset(1)) == (12, 12)
    assert ref.field_layout() == [(0, 3), (3, 3), (6, 35), (41, 64), (105, 51), (156, 15), (171, 20), (191, 23)]


@needs_ref_geo
def test_geo_lookup_matches_reference(orc):
    """database images made by the reference's IP_BTRIE_ADD; every probe answered by its
    IP_BTRIE_LOOKUP (instantiated as geoip.c:33-88 does) and by the oracle"""
    import golden_io
    sys_path_make = os.path.join(HERE, "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_fixtures", os.path.join(sys_path_make, "make_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    ref = RefGeo()
    ents = dict(zip(("as4", "as6", "geo4", "geo6"), mk.geo_entries()))
    fix = golden_io.geo_reference()
    rng = np.random.default_rng(11)
    for name, (kind, fam) in GEO_KINDS.items():
        img = ref.build(kind, ents[name])
        assert img == fix[name][0], "the committed image of %s is not what the reference builds today" % name
        orc.geodb_set(kind, fam, img)
        n = 4 if fam == 4 else 16
        probes = [bytes(p) for p in fix[name][1]] + [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for _ in range(2000)]
        nfound = 0
        for a in probes:
            want = ref.lookup(kind, fam, img, a)
            got = orc.geodb_lookup(kind, fam, a)
            assert got == want, (name, a.hex())
            nfound += want is not None
        assert nfound > 200


def test_geo_fixture_pins_the_oracle(orc):
    """the same without the reference at hand: its recorded answers"""
    import golden_io
    for name, (img, probes, found, payload) in golden_io.geo_reference().items():
        kind, fam = GEO_KINDS[name]
        orc.geodb_set(kind, fam, img)
        for a, f, pay in zip(probes, found, payload):
            got = orc.geodb_lookup(kind, fam, bytes(a))
            assert (got is not None) == bool(f), (name, bytes(a).hex())
            if f:
                assert got == bytes(pay)


@needs_ref_utils
def test_port_names_fixture_matches_reference():
    assert port_names() == RefUtils().port_names()


# ------------------------------------------------------------------ keyword tables of the filter compiler
def _keyword_rows():
    import ctypes as C
    import host_binding
    L = host_binding.load_hosttest()
    kw = C.c_char_p()
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    fields, aggr = [], []
    i = 0
    while L.xe_filter_keyword(i, C.byref(kw), C.byref(a), C.byref(b), C.byref(c)):
        fields.append((kw.value.decode(), a.value, b.value, c.value))
        i += 1
    i = 0
    while L.xe_filter_aggr_keyword(i, C.byref(kw), C.byref(a), C.byref(b)):
        aggr.append((kw.value.decode(), a.value, b.value))
        i += 1
    return fields, aggr


@pytest.mark.skipif(not os.path.exists("/root/reference/filter.def"), reason="reference checkout absent")
def test_filter_keyword_tables_match_reference_definitions(orc):
    """the library's filter keywords with their operand type and source / destination flow fields
    against the reference's X-macro tables filter.def and filter-ag.def, row by row"""
    import re
    index = {f["name"]: f["idx"] for f in orc.fields()}
    types = {"ADDR4": 0, "ADDR6": 1, "RANGE": 2, "MAC": 3, "STRING": 4}
    want = []
    for line in open("/root/reference/filter.def"):
        m = re.match(r'FIELD\((\w+),\s*"([^"]*)",\s*(\w+),\s*(\w+),\s*(\w+)\)', line)
        if m:
            want.append((m.group(2), types[m.group(3)], index[m.group(4)], index[m.group(5)]))
    want_ag = []
    for line in open("/root/reference/filter-ag.def"):
        m = re.match(r'FIELD\((\w+),\s*"([^"]*)",\s*(\w+),\s*(\d+)\)', line)
        if m:
            want_ag.append((m.group(2), index[m.group(3)], int(m.group(4))))
    fields, aggr = _keyword_rows()
    assert len(want) == 36 and len(want_ag) == 7
    assert fields == want
    assert aggr == want_ag


# ------------------------------------------------------------------ the shim's view of the reference's structures
@pytest.mark.skipif(not os.path.exists("/root/reference/flow-info.h"), reason="reference checkout absent")
def test_shim_compiles_against_the_reference_flow_info_header(tmp_path):
    """shim/xenoeye_gpu_shim.c built with the reference's own flow-info.h in front of it (its struct
    flow_packet_info then replaces the shim header's copy), plus compile-time checks that the copy in
    include/xenoeye_gpu_shim.h and XE_FLOW_INFO_SIZE agree with the reference's structures"""
    import subprocess
    root = os.path.dirname(HERE)
    check = tmp_path / "layout_check.c"
    check.write_text("""
#include <stddef.h>
#include <stdint.h>
#include <sys/socket.h>
#include "utils.h"
#include "flow-info.h"                 /* the reference's */
#define flow_packet_info shim_flow_packet_info
#undef flow_info_h_included
#undef MAX_NF_PACKET_SIZE
#include "xenoeye_gpu_shim.h"          /* the shim's own copy, under another name */
#undef flow_packet_info
#include "xe_gpu.h"
#define SAME(M) _Static_assert(offsetof(struct flow_packet_info, M) == offsetof(struct shim_flow_packet_info, M), #M)
SAME(src_addr); SAME(src_addr_ipv4); SAME(source_id); SAME(epoch); SAME(time_ns); SAME(rawpacket); SAME(sampling_rate);
_Static_assert(sizeof(struct flow_packet_info) == sizeof(struct shim_flow_packet_info), "struct flow_packet_info");
_Static_assert(sizeof(struct flow_info) == XE_FLOW_INFO_SIZE, "struct flow_info");
int main(void) { return 0; }
""")
    r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-I/root/reference", "-I", os.path.join(root, "include"),
                        "-c", str(check), "-o", str(tmp_path / "layout_check.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["gcc", "-std=gnu11", "-O2", "-fPIC", "-Wall", "-I/root/reference", "-include", "utils.h", "-include",
                        "flow-info.h", "-I", os.path.join(root, "include"), "-c",
                        os.path.join(root, "shim", "xenoeye_gpu_shim.c"), "-o", str(tmp_path / "shim.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
