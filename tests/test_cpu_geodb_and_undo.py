"""Host-side pieces of the round's additions, without a GPU: the IP lists made
of geo / AS databases answer like the oracle's database lookups, and the log
of template changes puts the template store back exactly."""
import numpy as np
import pytest

from xenoeye_amd import synth
from host_binding import Host, walk_flat
from test_gpu_geo_functions import _databases, build_db, as_payload

FLOW_INFO = 2312
OFF = {"src host": 92, "dst host": 140, "src host6": 248, "dst host6": 272}     # flow-info.h offsets
FN_GEO, FN_ASN, FN_ASD = 9, 10, 11


@pytest.fixture()
def host():
    h = Host()
    yield h
    h.close()


def _flow(field, addr):
    f = bytearray(FLOW_INFO)
    f[OFF[field]:OFF[field] + len(addr)] = addr
    return bytes(f)


def _probe_addresses(rng, image, node_bytes, nbytes, n):
    """random addresses, the zero address, and addresses that follow the database's trie some way"""
    out = [bytes(nbytes), bytes([0xFF] * nbytes)]
    nodes = len(image) // node_bytes
    for _ in range(n):
        a = bytearray(int(x) for x in rng.integers(0, 256, nbytes))
        if rng.integers(0, 3):
            # walk down from the root along existing children for a random number of bits
            node = 0
            for i in range(8 * nbytes):
                n0 = int.from_bytes(image[node * node_bytes:node * node_bytes + 4], "little")
                n1 = int.from_bytes(image[node * node_bytes + 4:node * node_bytes + 8], "little")
                leaf = int.from_bytes(image[node * node_bytes + 8:node * node_bytes + 12], "little")
                if (not n0 and not n1) or (leaf and rng.integers(0, 3)):
                    break
                b = 1 if (n1 and (not n0 or rng.integers(0, 2))) else 0
                a[i // 8] = (a[i // 8] & ~(1 << (7 - i % 8))) | (b << (7 - i % 8))
                node = n1 if b else n0
                assert node < nodes
        out.append(bytes(a))
    return out


@pytest.mark.parametrize("case", [
    ("asn(%s) 64000-64400 or 100-300", FN_ASN, 0, dict(ranges=[(64000, 64400), (100, 300)])),
    ("asn(%s) 7", FN_ASN, 0, dict(ranges=[(7, 7)])),
    ("asd(%s) 'ACME' or 'x'", FN_ASD, 0, dict(strs=["ACME", "x"])),
    ("country_code(%s) 'DE'", FN_GEO, 1, dict(strs=["DE"])),
    ("city(%s) 'San Jose' or 'Moscow'", FN_GEO, 4, dict(strs=["San Jose", "Moscow"])),
    ("long(%s) '13.4'", FN_GEO, 7, dict(strs=["13.4"])),
])
def test_lists_made_of_a_database_answer_like_its_lookup(host, orc, case):
    text, func, gfield, ops = case
    dbs = _databases(None)
    rng = np.random.default_rng(11)
    host.add_iplist("user", "10.0.0.0/8\n")
    for name, image in dbs.items():
        kind, fam = (1 if name.startswith("as") else 0), (6 if name.endswith("6") else 4)
        assert host.geodb_set(kind, fam, image) == 0
        orc.geodb_set(kind, fam, image)
    hits = 0
    for fam, fld, nbytes in ((4, "src host", 4), (6, "dst host6", 16)):
        if func == FN_GEO and fam == 6 and gfield == 7:
            continue
        li = host.geodb_list(func, gfield, fam, **ops)
        assert li >= 1
        words, dev = host.flatten()
        root4, root6, nodes4, nodes6, bit, chains4, chains6 = dev[li]
        fh = orc.filter_parse(text % fld)
        image = dbs[("as" if func != FN_GEO else "geo") + str(fam)]
        for a in _probe_addresses(rng, image, 216 if func != FN_GEO else 228, nbytes, 1500):
            want = orc.filter_match(fh, _flow(fld, a))
            got = walk_flat(words, root4 if fam == 4 else root6, nodes4 if fam == 4 else nodes6, a, bit,
                            chains4 if fam == 4 else chains6)
            assert got == want, (text % fld, a.hex(), got, want)
            hits += want
    assert hits > 0, "no probe address matched: the test would prove nothing"


def test_bad_database_images(host):
    good = build_db([(bytes([10, 0, 0, 0]), 8, as_payload(1, "a"))], 216)
    assert host.geodb_set(1, 4, good) == 0
    assert host.geodb_set(1, 4, good[:-4]) == -4
    bad = bytearray(good)
    bad[4:8] = (5000).to_bytes(4, "little")
    assert host.geodb_set(1, 4, bytes(bad)) == -4
    bad = bytearray(good)
    bad[2 * 216:2 * 216 + 4] = (1).to_bytes(4, "little")       # a child in front of its parent
    assert host.geodb_set(1, 4, bytes(bad)) == -4
    assert host.geodb_set(1, 6, b"") == 0
    assert host.geodb_list(FN_ASN, 0, 6, ranges=[(1, 1)]) < 0    # no IPv6 database


def test_template_changes_of_a_batch_can_be_taken_back(host):
    fa = synth.TMPL_V9_C1
    fb = [(12, 4), (8, 4), (11, 2), (7, 2), (4, 1), (1, 4), (2, 4), (10, 4)]
    rng = synth.Rng(3)
    e1 = bytes([192, 0, 2, 1])
    rec_a = synth.flow_columns(rng, 64 * 6, fa)
    rec_b = synth.flow_columns(rng, 50 * 4, fb)
    pk_a = synth.v9_data_packets(256, rec_a, 64, 77)
    pk_b = synth.v9_data_packets(256, rec_b, 50, 77)
    pk_c = synth.v9_data_packets(300, rec_b, 50, 77)

    first = synth.Batch()
    first.add(synth.v9_template_packet(256, fa, 77), e1)
    first.extend(pk_a[:2], e1)
    base = host.prepass(first, threads=1)
    assert base["nrec"] == 128
    old_tmpl = base["sets"][0]["tmpl"]

    second = synth.Batch()
    second.add(synth.v9_template_packet(256, fb, 77), e1)      # replaces template 256
    second.add(synth.v9_template_packet(300, fb, 77), e1)      # a template nobody had
    second.add(synth.v9_template_packet(256, fb, 77), e1)      # the same again: no change
    second.extend(pk_b[:2], e1)
    second.extend(pk_c[:2], e1)
    for threads in (1, 4):
        host.undo_begin()
        got = host.prepass(second, threads=threads)
        assert got["nrec"] == 200 and host.undo_count() == 2
        assert got["sets"][0]["stride"] == 25
        host.undo_rollback()
        assert host.undo_count() == 0
        host.undo_end()
        # the store is where it was: the old layout decodes, template 300 is unknown again
        third = synth.Batch()
        third.extend(pk_a[2:4], e1)
        third.extend(pk_c[2:3], e1)
        again = host.prepass(third, threads=threads)
        assert again["nrec"] == 128
        assert all(s["tmpl"] == old_tmpl and s["stride"] == 21 for s in again["sets"])
    # and without the rollback the change stays
    host.undo_begin()
    host.prepass(second, threads=1)
    host.undo_end()
    kept = host.prepass(third, threads=1)
    assert kept["nrec"] > 128 and kept["sets"][0]["stride"] == 25
