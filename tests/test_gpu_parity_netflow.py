"""GPU parity: Netflow v5/v9/IPFIX decode + classification through the C ABI
against the CPU oracle, bit for bit (struct flow_info bytes, match bitmask,
dropped flags, counters)."""
import struct

import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from helpers import configure, run_oracle, compare

pytestmark = pytest.mark.gpu

EXP = bytes([192, 0, 2, 1])


@pytest.fixture()
def eng():
    e = xenoeye_amd.Engine(device=0)
    yield e
    e.close()


def both(eng, orc, w, flows=True):
    configure(eng, orc, w.get("iplists"), w["mos"], w.get("devices", ()))
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=flows)
    compare(b, orc, flows=flows)
    return b


def test_c1_small(eng, orc):
    b = both(eng, orc, synth.workload_c1(4096))
    assert b.records == 4096
    assert 0 < int(b.counters()[0, 0]) < 4096


def test_c1_ragged_last_datagram(eng, orc):
    both(eng, orc, synth.workload_c1(1000, per_packet=24))


def test_c2_small(eng, orc):
    w = synth.workload_c2(6000, prefixes=500)
    b = both(eng, orc, w)
    c = b.counters()
    assert (c[:, 0] > 0).all(), "every object should see some records"


def _v9_case(count, body, template=synth.TMPL_V9_C1):
    b = synth.Batch()
    b.add(synth.v9_template_packet(256, template, 77), EXP)
    b.add(synth.v9_header(count, 77) + body, EXP)
    return b


def test_v9_count_exceeds_records(eng, orc):
    """count=4 over a set of two records: the 3rd iteration sees only the 4
    bytes after the set, the 4th nothing; both are still classified."""
    rec = synth.flow_columns(synth.Rng(1), 2, synth.TMPL_V9_C1)
    body = struct.pack(">HH", 256, 4 + 42) + rec.tobytes()
    w = dict(batch=_v9_case(4, body), mos=[("", -1), ("not proto 6", -1), ("src host 0.0.0.0", -1)])
    b = both(eng, orc, w)
    assert b.records == 4


def test_v9_trailing_bytes_become_a_field(eng, orc):
    rec = synth.flow_columns(synth.Rng(2), 1, synth.TMPL_V9_C1)
    body = struct.pack(">HH", 256, 4 + 21) + rec.tobytes() + bytes([0xC0, 0xA8, 1, 1])
    w = dict(batch=_v9_case(2, body), mos=[("", -1), ("src host 192.168.1.1", -1)])
    b = both(eng, orc, w)
    assert b.records == 2
    assert b.masks()[1, 0] == 3


def test_v9_second_set_is_lost(eng, orc):
    rec = synth.flow_columns(synth.Rng(3), 2, synth.TMPL_V9_C1)
    one = struct.pack(">HH", 256, 4 + 21)
    body = one + rec[0].tobytes() + one + rec[1].tobytes()
    b = both(eng, orc, dict(batch=_v9_case(1, body), mos=[("", -1)]))
    assert b.records == 1


def test_v9_unknown_template_and_options(eng, orc):
    rec = synth.flow_columns(synth.Rng(4), 3, synth.TMPL_V9_C1)
    bt = synth.Batch()
    bt.add(synth.v9_header(3, 77) + struct.pack(">HH", 300, 4 + 63) + rec.tobytes(), EXP)  # no template yet
    bt.add(synth.v9_template_packet(256, synth.TMPL_V9_C1, 77), EXP)
    bt.add(synth.v9_header(3, 77) + struct.pack(">HH", 1, 8) + b"\0" * 4
           + struct.pack(">HH", 256, 4 + 63) + rec.tobytes(), EXP)                          # options set first
    bt.add(synth.v9_header(3, 77) + struct.pack(">HH", 256, 4 + 63) + rec.tobytes(), EXP)
    bt.add(synth.v9_header(3, 78) + struct.pack(">HH", 256, 4 + 63) + rec.tobytes(), EXP)  # other source id
    b = both(eng, orc, dict(batch=bt, mos=[("", -1)]))
    assert b.records == 3


def test_v9_template_change_mid_batch(eng, orc):
    t2 = [(12, 4), (8, 4), (4, 1), (7, 2), (11, 2), (2, 4), (1, 4)]
    bt = synth.Batch()
    bt.add(synth.v9_template_packet(256, synth.TMPL_V9_C1, 77), EXP)
    r1 = synth.flow_columns(synth.Rng(5), 5, synth.TMPL_V9_C1)
    bt.extend(synth.v9_data_packets(256, r1, 5, 77), EXP)
    bt.add(synth.v9_template_packet(256, t2, 77), EXP)
    r2 = synth.flow_columns(synth.Rng(6), 5, t2)
    bt.extend(synth.v9_data_packets(256, r2, 5, 77), EXP)
    both(eng, orc, dict(batch=bt, mos=[("proto 6", -1), ("src port 0-1023", -1)]))


def test_v9_odd_templates(eng, orc):
    """invalid lengths, unknown elements, a repeated element, short in_bytes,
    a string field, zero-length fields"""
    t = [(1, 3), (8, 4), (999, 5), (1, 8), (1, 4), (82, 7), (6, 2), (6, 1), (10, 2), (56, 6), (2, 0), (27, 16), (28, 15)]
    n = 9
    rec = synth.flow_columns(synth.Rng(7), n, t)
    bt = synth.Batch()
    bt.add(synth.v9_template_packet(400, t, 5), EXP)
    bt.extend(synth.v9_data_packets(400, rec, 4, 5), EXP)
    both(eng, orc, dict(batch=bt, mos=[("", -1), ("mac 00:00:00:00:00:00", -1), ("tcp-flags 0-255", -1),
                                       ("div(octets, packets) 0-100000", -1), ("ifidx 1-48", -1)]))


def test_v9_long_records(eng, orc):
    """records of 200-256 bytes (many elements the field table does not know):
    staged whole while the block's LDS holds them, decoded per lane beyond"""
    rng = synth.Rng(5)
    for pad_fields in (44, 52, 57, 58, 70):
        tmpl = [(8, 4), (12, 4), (7, 2), (11, 2), (4, 1)] + [(30000 + i, 4) for i in range(pad_fields)] + [(1, 4), (2, 4)]
        rl = sum(l for _, l in tmpl)
        nrec = 5 * (1400 // rl) + 3
        rec = rng.bytes(nrec * rl).reshape(nrec, rl)
        b = synth.Batch()
        b.add(synth.v9_template_packet(300, tmpl, 9), EXP)
        b.extend(synth.v9_data_packets(300, rec, 1400 // rl, 9), EXP)
        e = xenoeye_amd.Engine(device=0)
        orc.reset()
        w = dict(batch=b, mos=[("proto 6", -1), ("src net 10.0.0.0/8 or dst port 0-1023", -1), ("", -1)] +
                 [("dst port %d" % i, -1) for i in range(40)])
        both(e, orc, w)
        e.close()


@pytest.mark.parametrize("nobj", [8, 64])
def test_long_records_beside_the_counter_copies(orc, nobj):
    """records of 209-241 bytes with few objects: a block keeps several copies
    of its counters in LDS then, and the staging area of the longest records
    must leave room for all of them (host walk, then device walk)"""
    rng = synth.Rng(6)
    for pad_fields in (47, 50, 53, 54, 55):
        tmpl = [(8, 4), (12, 4), (7, 2), (11, 2), (4, 1)] + [(30000 + i, 4) for i in range(pad_fields)] + [(1, 4), (2, 4)]
        rl = sum(l for _, l in tmpl)
        nrec = 60 * (1400 // rl) + 2
        rec = rng.bytes(nrec * rl).reshape(nrec, rl)
        b = synth.Batch()
        b.add(synth.v9_template_packet(300, tmpl, 9), EXP)
        b.extend(synth.v9_data_packets(300, rec, 1400 // rl, 9), EXP)
        e = xenoeye_amd.Engine(device=0)
        orc.reset()
        mos = [("proto 6", -1), ("src net 10.0.0.0/8 or dst port 0-1023", -1), ("", -1)] + \
            [("dst port %d-%d" % (900 * i, 900 * i + 700), -1) for i in range(nobj - 3)]
        configure(e, orc, None, mos)
        for rnd in range(2):
            orc.log_clear()
            out = e.prepare(b)
            out.run()
            run_oracle(orc, b)
            compare(out, orc)
            out.free()
        e.close()


def test_v9_two_templates_in_one_flowset(eng, orc):
    """only the first template of a template flowset is registered; the second
    one's header is then read as a flowset header"""
    t1, t2 = synth.TMPL_V9_C1, [(8, 4), (12, 4)]
    b1 = struct.pack(">HH", 256, len(t1)) + b"".join(struct.pack(">HH", a, l) for a, l in t1)
    b2 = struct.pack(">HH", 257, len(t2)) + b"".join(struct.pack(">HH", a, l) for a, l in t2)
    fs = struct.pack(">HH", 0, 4 + len(b1) + len(b2)) + b1 + b2
    bt = synth.Batch()
    bt.add(synth.v9_header(2, 77) + fs, EXP)
    bt.add(synth.v9_header(2, 77) + fs, EXP)      # second time: template 257's bytes get "decoded"
    rec = synth.flow_columns(synth.Rng(8), 2, t1)
    bt.extend(synth.v9_data_packets(256, rec, 2, 77), EXP)
    both(eng, orc, dict(batch=bt, mos=[("", -1)]))


def test_ipfix_padding_makes_an_extra_record(eng, orc):
    t = [(8, 4), (7, 2), (4, 1)]
    rec = synth.flow_columns(synth.Rng(9), 2, t)
    body = struct.pack(">HH", 256, 4 + 14 + 1) + rec.tobytes() + b"\0"
    bt = synth.Batch()
    bt.add(synth.ipfix_template_packet(256, t, 9), EXP)
    bt.add(synth.ipfix_header(16 + len(body), 9) + body, EXP)
    b = both(eng, orc, dict(batch=bt, mos=[("", -1), ("src host 0.0.0.0", -1)]))
    assert b.records == 3


def test_ipfix_variable_length_and_enterprise(eng, orc):
    t = [(8, 4), (82, 65535), (2001, 65535), (100, 4, 12345), (1, 65535), (4, 1)]
    bt = synth.Batch()
    bt.add(synth.ipfix_template_packet(300, t, 9), EXP)
    recs = b""
    rng = synth.Rng(10)
    for i in range(7):
        name = b"eth%d" % i
        host = b"host-%d.example" % i if i % 3 else b"x" * 300    # 300 > 256: rejected; 3-byte length form
        ib = struct.pack(">I", 1000 + i) if i % 2 else struct.pack(">Q", 7 << 33)
        recs += rng.bytes(4).tobytes() + bytes([len(name)]) + name
        recs += (bytes([len(host)]) if len(host) < 255 else b"\xff" + struct.pack(">H", len(host))) + host
        recs += b"ABCD" + bytes([len(ib)]) + ib + bytes([6])
    body = struct.pack(">HH", 300, 4 + len(recs)) + recs
    bt.add(synth.ipfix_header(16 + len(body), 9) + body, EXP)
    b = both(eng, orc, dict(batch=bt, mos=[("", -1), ("proto 6", -1), ("div(octets, packets) 0", -1)]))
    assert b.records == 7


def test_ipfix_multiple_sets_and_unknown_ids(eng, orc):
    t4, t6 = synth.TMPL_IPFIX_V4, synth.TMPL_IPFIX_V6
    rng = synth.Rng(11)
    r4 = synth.flow_columns(rng, 3, t4).tobytes()
    r6 = synth.flow_columns(rng, 2, t6).tobytes()
    sets = synth.ipfix_template_set(256, t4) + synth.ipfix_template_set(257, t6)
    sets += struct.pack(">HH", 256, 4 + len(r4)) + r4
    sets += struct.pack(">HH", 3, 8) + b"\0" * 4          # options template: skipped
    sets += struct.pack(">HH", 100, 6) + b"\0\0"          # reserved id: skipped
    sets += struct.pack(">HH", 257, 4 + len(r6)) + r6
    sets += struct.pack(">HH", 999, 8) + b"\0" * 4        # unknown template: rest abandoned
    sets += struct.pack(">HH", 256, 4 + len(r4)) + r4
    bt = synth.Batch()
    bt.add(synth.ipfix_header(16 + len(sets), 3) + sets, EXP)
    b = both(eng, orc, dict(batch=bt, mos=[("", -1), ("proto 17", -1)]))
    assert b.records == 5


def test_v5(eng, orc):
    bt = synth.Batch()
    rng = synth.Rng(12)
    for n in (1, 30, 7):
        bt.add(synth.v5_packet(rng, n, engine_id=n), EXP)
    bt.add(synth.v5_packet(rng, 3)[:-1], EXP)            # wrong length: ignored
    bt.add(struct.pack(">H", 7) + b"\0" * 40, EXP)       # unknown version
    b = both(eng, orc, dict(batch=bt, mos=[("", -1), ("proto 6", -1), ("dev-id 30", -1), ("src as 0-32767", -1)]))
    assert b.records == 38


def test_nested_objects_and_quirky_filters(eng, orc):
    w = synth.workload_c1(3000)
    w["mos"] = [("src net mynet", -1), ("proto 6", 0), ("dst port 0-1023", 1), ("proto 17", 0),
                ("(proto 6 or 17) and (src port 0 or dst port 0)", -1), ("not src net mynet", -1),
                ("src port 80 or not dst port 80", -1), ("dst net mynet", 5), ("", 7),
                ("min(src port, dst port) 0-1023", -1), ("div_r(octets, packets, 100) 100-500", -1),
                ("dev-ip 192.0.2.1", -1), ("dev-ip 192.0.2.2", -1), ("dev-id 77", -1),
                ("host 10.1.2.3 or 10.0.0.0/8", -1)]
    both(eng, orc, w)


def test_tfstr_objects(eng, orc):
    """tfstr() lowered to a set of byte values: the shipped sample tree's
    `tfstr(tcp-flags) 'SYN'`, case-insensitive texts, other fields, and the
    same terms inside device mark filters (per-lane kernel)"""
    w = synth.workload_c2(4000, prefixes=100)
    w["mos"] = [("proto 6", -1), ("tfstr(tcp-flags) 'SYN'", 0), ("tfstr(tcp-flags) 'ack+syn' or 'ACK' or 'bogus'", 0),
                ("not tfstr(tcp-flags) 'ACK+PSH'", -1), ("tfstr(proto) 'RST+SYN'", -1),
                ("tfstr(dst port) 'FIN' or 'SYN' or 'SYN+FIN'", -1), ("tfstr(src host) 'ACK+PSH+SYN'", -1),
                ("tfstr(src host6) 'URG'", -1)]
    b = both(eng, orc, w)
    c = b.counters()
    assert c[4, 0] > 0, "proto 6 reads as RST+SYN"


def test_tfstr_in_device_marks(eng, orc):
    w2 = synth.workload_c1(3000)
    w2["mos"] = [("proto 6", -1), ("tfstr(tcp-flags) 'SYN'", 0), ("not tfstr(proto) 'RST+SYN'", -1)]
    w2["devices"] = [dict(ip="192.0.2.1", sampling_rate=10, marks=("tfstr(proto) 'RST+SYN'", "tfstr(dst port) 'SYN'"))]
    w2["mos"] = list(w2["mos"]) + [("dev-mark 1", -1), ("dev-mark 2", -1)]
    b = both(eng, orc, w2)
    assert b.per_lane_records == 0      # mark filters run in the kernel that takes a tile of one template


def test_devices_marks_and_sampling(eng, orc):
    w = synth.workload_c1(2000)
    w["devices"] = [dict(ip="192.0.2.1", dev_id=77, sampling_rate=100, skip_unmarked=True,
                         marks=("proto 6", "src port 0-1023", "dst net mynet"))]
    w["mos"] = [("src net mynet", -1), ("dev-mark 1", -1), ("dev-mark 2-3", -1), ("", -1)]
    b = both(eng, orc, w)
    assert b.dropped().sum() > 0
    assert b.per_lane_records == 0


def test_devices_sampling_rate_only(eng, orc):
    """devices.conf entries that only set a sampling rate: records carry a
    device mark of 0 and counters are scaled; exporters without an entry and
    one with mark filters share the batch"""
    w = synth.workload_c2(3000, prefixes=200)
    w["devices"] = [dict(ip="10.255.0.1", sampling_rate=1000), dict(ip="10.255.0.2", dev_id=2, sampling_rate=64),
                    dict(ip="10.255.0.3", sampling_rate=10, marks=("proto 6",)),
                    dict(ip="10.255.0.4", sampling_rate=5, skip_unmarked=True)]
    b = both(eng, orc, w)
    fi = b.flow_info()
    rates = np.unique(fi[:, 2304:2308].copy().view("<u4"))
    assert set(rates.tolist()) >= {1, 10, 64, 1000}
    assert b.per_lane_records == 0      # four kinds of entries share the tiles


def test_devices_mark_read_by_an_object(eng, orc):
    w = synth.workload_c2(2000, prefixes=100)
    w["devices"] = [dict(ip="10.255.0.1", sampling_rate=1000), dict(ip="10.255.0.5", sampling_rate=2, marks=("proto 17",))]
    w["mos"] = list(w["mos"]) + [("dev-mark 0", -1), ("dev-mark 1", -1), ("not dev-mark 0", -1)]
    b = both(eng, orc, w)
    assert b.per_lane_records == 0


def test_devices_id_only_fallthrough(eng, orc):
    """an entry with ip and id whose ip does not match is still matched on the
    id alone (devices.c:224 is `if`, not `else if`)"""
    w = synth.workload_c1(500)
    w["devices"] = [dict(ip="10.9.9.9", dev_id=77, sampling_rate=7), dict(ip="192.0.2.1", sampling_rate=9)]
    b = both(eng, orc, w)
    fi = b.flow_info(0, 1)
    assert int.from_bytes(fi[0, 2304:2308].tobytes(), "little") == 7


def test_many_objects_multiword_mask(eng, orc):
    w = synth.workload_c1(1500)
    mos = []
    for i in range(150):
        if i % 10 == 0:
            mos.append(("src net mynet", -1))
        else:
            mos.append(("dst port %d-%d" % (i * 400, i * 400 + 5000), (i // 10) * 10 if i % 3 else -1))
    w["mos"] = mos
    b = both(eng, orc, w, flows=False)
    assert b.mask_words == 3


def test_empty_batch(eng, orc):
    configure(eng, orc, mos=[("", -1)])
    b = eng.prepare(synth.Batch())
    b.run()
    assert b.records == 0
    assert b.masks().shape == (0, 1)
    assert int(b.counters().sum()) == 0


def test_rerun_is_idempotent(eng, orc):
    w = synth.workload_c1(2048)
    b = both(eng, orc, w)
    m1, c1 = b.masks(), b.counters()
    b.run()
    b.run()
    assert np.array_equal(b.masks(), m1)
    assert np.array_equal(b.counters(), c1)


def test_templates_persist_across_batches(eng, orc):
    w = synth.workload_c1(640)
    configure(eng, orc, w["iplists"], w["mos"])
    first = synth.Batch()
    first.add(w["batch"].pkts[0], EXP)
    b0 = eng.prepare(first)
    b0.run()
    rest = synth.Batch()
    rest.extend(w["batch"].pkts[1:], EXP)
    b1 = eng.prepare(rest)
    b1.run()
    run_oracle(orc, w["batch"])
    compare(b1, orc)
