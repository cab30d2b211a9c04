"""GPU parity: sFlow v5 datagram walk + raw header extraction + classification
through the C ABI against the CPU oracle."""
import struct

import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from helpers import configure, run_oracle, compare

pytestmark = pytest.mark.gpu

AGENT = bytes([10, 254, 0, 1])


@pytest.fixture()
def eng():
    e = xenoeye_amd.Engine(device=0)
    yield e
    e.close()


def both(eng, orc, w):
    configure(eng, orc, w.get("iplists"), w["mos"], w.get("devices", ()))
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=True)
    compare(b, orc)
    return b


def one(samples, **kw):
    b = synth.Batch()
    b.add(synth.sflow_datagram(AGENT, 7, samples, **kw), AGENT, kind=1)
    return b


def test_c3_mix(eng, orc):
    b = both(eng, orc, synth.workload_c3(700))
    assert (b.dropped() == 0).sum() > 600


def test_vlan_tcp_syn_is_left_aligned(eng, orc):
    """a 1-byte TCP flags value lands in byte 0 of the 2-byte field, so the
    range filter sees value << 8; the VLAN field keeps the whole TCI"""
    rng = synth.Rng(21)
    hdr = synth._eth(rng, 0x0800, vlan=0x625f) + synth._ipv4(rng, 6) + synth._tcp(rng, flags=0x02)
    hdr = (hdr + bytes(128))[:128]
    s = synth.sflow_flow_sample(1, 400, 7, 9, [synth.sflow_raw_element(hdr, 1500)])
    w = dict(batch=one([s]), mos=[("tcp-flags 512", -1), ("tcp-flags 2", -1), ("vlan 25183", -1), ("ifidx 7", -1)])
    b = both(eng, orc, w)
    assert b.masks()[0, 0] == 0b1101
    fi = b.flow_info()[0]
    assert int.from_bytes(fi[2304:2308].tobytes(), "little") == 400
    assert bytes(fi[0:8]) == (1500).to_bytes(8, "big")


def test_tfstr_reads_byte_one_of_the_flags_field(eng, orc):
    """sFlow puts the flags byte at byte 0 of the field, tfstr() reads byte 1
    (filter.c:842-847): the text is always empty, so no name ever matches"""
    w = synth.workload_c3(500)
    w["mos"] = [("tfstr(tcp-flags) 'SYN'", -1), ("not tfstr(tcp-flags) 'SYN' or 'ACK' or 'ACK+PSH' or 'ACK+SYN'", -1),
                ("tfstr(proto) 'RST+SYN'", -1), ("tfstr(src vlan) 'SYN' or 'FIN' or 'ACK'", -1)]
    b = both(eng, orc, w)
    c = b.counters()
    assert c[0, 0] == 0 and c[1, 0] == (b.dropped() == 0).sum() and c[2, 0] > 0


def test_ipv6_destination_is_the_source(eng, orc):
    rng = synth.Rng(22)
    src6 = bytes.fromhex("20010db80000000000000000000000aa")
    dst6 = bytes.fromhex("fd0000000000000000000000000000bb")
    hdr = synth._eth(rng, 0x86dd) + synth._ipv6(rng, 17, src6, dst6) + synth._udp(rng)
    hdr = (hdr + bytes(128))[:128]
    s = synth.sflow_flow_sample(1, 10, 1, 2, [synth.sflow_raw_element(hdr, 200)])
    w = dict(batch=one([s]), mos=[("dst host6 2001:db8::aa", -1), ("dst host6 fd00::bb", -1), ("proto 17", -1)])
    b = both(eng, orc, w)
    assert b.masks()[0, 0] == 0b101


def test_tunnels_fragments_and_non_ip(eng, orc):
    rng = synth.Rng(23)
    samples = []
    for i, kind in enumerate(["ipip", "6in6", "4in6", "frag", "arp", "v6udp", "v4icmp", "v4tcp"]):
        for vm in (0, 1, 2):
            hdr = synth.sflow_header_sample(rng, kind, vm)
            hdr = (hdr + rng.bytes(128).tobytes())[:128]
            samples.append(synth.sflow_flow_sample(i, 100 + i, 1, 2, [synth.sflow_raw_element(hdr, 600)],
                                                   expanded=bool(i % 2)))
    b = synth.Batch()
    for k in range(0, len(samples), 6):
        b.add(synth.sflow_datagram(AGENT, 7, samples[k:k + 6]), AGENT, kind=1)
    w = dict(batch=b, mos=[("", -1), ("proto 6", -1), ("proto 17 or 1", -1), ("not proto 0-255", -1),
                           ("d1qcvlan 0-65535", -1), ("min-ttl 1-255", -1)])
    both(eng, orc, w)


def test_truncated_headers_and_exact_ends(eng, orc):
    """every bounds test is ptr + header >= end: a header ending exactly at
    the declared header length fails"""
    rng = synth.Rng(24)
    full = synth._eth(rng, 0x0800) + synth._ipv4(rng, 6) + synth._tcp(rng)
    samples = []
    for cut in (10, 14, 15, 33, 34, 35, 53, 54, 55, 60):
        samples.append(synth.sflow_flow_sample(cut, 5, 1, 2, [synth.sflow_raw_element(full[:cut] + bytes(8), 100,
                                                                                   header_len=cut)]))
    b = synth.Batch()
    for k in range(0, len(samples), 5):
        b.add(synth.sflow_datagram(AGENT, 7, samples[k:k + 5]), AGENT, kind=1)
    both(eng, orc, dict(batch=b, mos=[("", -1), ("proto 6", -1), ("port 0-65535", -1)]))


def test_header_length_beyond_datagram(eng, orc):
    """the declared header length is not clamped to the datagram: bytes past
    its end read as zero"""
    rng = synth.Rng(25)
    hdr = synth._eth(rng, 0x0800) + synth._ipv4(rng, 6)       # TCP header missing
    el = (1, struct.pack(">IIII", 1, 900, 0, 128) + hdr)      # claims 128 bytes, carries 34
    s = synth.sflow_flow_sample(1, 5, 1, 2, [el])
    both(eng, orc, dict(batch=one([s]), mos=[("", -1), ("src port 0", -1), ("tcp-flags 0", -1)]))


def test_multiple_elements_share_the_flow(eng, orc):
    rng = synth.Rng(26)
    h1 = (synth._eth(rng, 0x0800, vlan=100) + synth._ipv4(rng, 17) + synth._udp(rng) + bytes(64))[:96]
    h2 = (synth._eth(rng, 0x86dd) + synth._ipv6(rng, 6) + synth._tcp(rng) + bytes(64))[:96]
    els = [synth.sflow_raw_element(h1, 300), (1001, struct.pack(">IIII", 10, 0, 20, 0)),
           synth.sflow_ipv4_element(rng, proto=1), (9999, b"\0" * 12), synth.sflow_raw_element(h2, 400)]
    s = synth.sflow_flow_sample(1, 64, 3, 4, els)
    b = both(eng, orc, dict(batch=one([s]), mos=[("", -1), ("vlan 100", -1), ("proto 6", -1), ("net6 ::/0", -1)]))
    assert (b.dropped() == 0).sum() == 3


def test_counter_samples_ipv6_agent_and_garbage(eng, orc):
    rng = synth.Rng(27)
    hdr = (synth._eth(rng, 0x0800) + synth._ipv4(rng, 6) + synth._tcp(rng) + bytes(64))[:64]
    flow = synth.sflow_flow_sample(1, 5, 1, 2, [synth.sflow_raw_element(hdr, 100)])
    counters = struct.pack(">II", 2, 12) + b"\0" * 12
    unknown = struct.pack(">II", 77, 8) + b"\0" * 8
    b = synth.Batch()
    b.add(synth.sflow_datagram(AGENT, 7, [counters, flow, unknown, flow]), AGENT, kind=1)
    b.add(synth.sflow_datagram(None, 9, [flow], agent6=bytes(range(16))), AGENT, kind=1)
    b.add(synth.sflow_datagram(AGENT, 7, [flow, flow])[:-20], AGENT, kind=1)        # second sample cut short
    b.add(struct.pack(">II", 4, 1) + b"\0" * 40, AGENT, kind=1)                      # wrong version
    b.add(b"\0\0", AGENT, kind=1)                                                    # too short
    bad_proto = synth.sflow_flow_sample(1, 5, 1, 2, [synth.sflow_raw_element(hdr, 100, proto=5)])
    b.add(synth.sflow_datagram(AGENT, 7, [flow, bad_proto, flow]), AGENT, kind=1)    # aborts at the bad element
    both(eng, orc, dict(batch=b, mos=[("", -1), ("dev-ip 10.254.0.1", -1), ("dev-id 9", -1),
                                      ("dev-ip6 0.1.2.3", -1)]))


def test_sflow_and_netflow_in_one_batch(eng, orc):
    w = synth.workload_c1(300)
    w3 = synth.workload_c3(70)
    for p, s, k in zip(w3["batch"].pkts, w3["batch"].src, w3["batch"].kind):
        w["batch"].add(p, s, k)
    w["mos"] = [("src net mynet", -1), ("proto 6", -1), ("vlan 0-4095", -1)]
    # netflow records come first in the device's record order, sflow after them
    configure(eng, orc, w["iplists"], w["mos"])
    b = eng.prepare(w["batch"])
    b.run()
    nf = synth.Batch()
    sf = synth.Batch()
    for p, s, k in zip(w["batch"].pkts, w["batch"].src, w["batch"].kind):
        (sf if k else nf).add(p, s, k)
    run_oracle(orc, nf)
    run_oracle(orc, sf)
    compare(b, orc)


def test_large_batches_take_the_threaded_header_walk(eng, orc):
    """more than 4096 datagrams: the header walk is spread over host threads;
    sFlow alone, then sFlow and IPFIX datagrams interleaved (an agent and an
    exporter share an address), with every thread count"""
    w = synth.workload_c3(42000, seed=31)
    assert len(w["batch"].pkts) >= 4096
    configure(eng, orc, w.get("iplists"), w["mos"], w.get("devices", ()))
    w2 = synth.workload_c2(60000, prefixes=100, seed=32)
    mixed = synth.Batch()
    sf, nf = w["batch"], w2["batch"]
    j = 0
    for i, (p, s) in enumerate(zip(sf.pkts, sf.src)):
        mixed.add(p, s, 1)
        if i % 3 == 0 and j < len(nf.pkts):
            # every 30th IPFIX datagram comes from the address of an sFlow agent
            mixed.add(nf.pkts[j], sf.src[i] if j % 30 == 7 else nf.src[j], 0)
            j += 1
    assert len(mixed.pkts) >= 4096 and j > 1000
    # the device numbers a batch's Netflow records first, then its sFlow elements
    # (each in arrival order): the oracle is fed in that order
    mixed_sorted = synth.Batch()
    for kind in (0, 1):
        for p, s, k in zip(mixed.pkts, mixed.src, mixed.kind):
            if k == kind:
                mixed_sorted.add(p, s, k)
    for batch, for_oracle in ((w["batch"], w["batch"]), (mixed, mixed_sorted)):
        ref = None
        for threads in (1, 3, 8):
            eng.set_prepass_threads(threads)
            orc.log_clear()
            b = eng.prepare(batch)
            b.run()
            if ref is None:
                run_oracle(orc, for_oracle, keep_flows=True)
                compare(b, orc)
                ref = (b.masks().copy(), b.flow_info().copy(), b.counters().copy(), b.dropped().copy())
            else:
                assert np.array_equal(b.masks(), ref[0]) and np.array_equal(b.counters(), ref[2])
                assert np.array_equal(b.dropped(), ref[3]) and np.array_equal(b.flow_info(), ref[1])
            b.free()
