"""GPU parity at larger shapes: 256 objects over big lists (multi-word masks,
listed and table-resolved objects mixed), and size-independent properties of
the benchmark-size batch."""
import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from helpers import configure, run_oracle, compare

pytestmark = pytest.mark.gpu


@pytest.fixture()
def eng():
    e = xenoeye_amd.Engine(device=0)
    yield e
    e.close()


def test_c4_shape_256_objects(eng, orc):
    w = synth.workload_c4(20000, prefixes=30000)
    assert len(w["mos"]) == 256
    configure(eng, orc, w["iplists"], w["mos"])
    b = eng.prepare(w["batch"])
    b.run()
    assert b.mask_words == 4
    run_oracle(orc, w["batch"], keep_flows=True)
    compare(b, orc)
    c = b.counters()
    assert (c[:, 0] > 0).sum() > 200     # nearly every object sees traffic


@pytest.mark.timeout(600)
def test_c4_full_scale_lists(eng, orc):
    """8 lists of 1 000 000 prefixes each (routing-table like lengths), 256
    objects: tables of several hundred MB, every walk depth in use"""
    w = synth.workload_c4(30000, prefixes=1000000)
    configure(eng, orc, w["iplists"], w["mos"])
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    compare(b, orc, flows=False)
    assert (b.counters()[:, 0] > 0).sum() > 200


def test_more_terms_and_lookups_than_the_shared_classification_holds(eng, orc):
    """over 256 distinct terms, and lists in more than 8 (group, field) pairs:
    every object program is then run on its own, same results"""
    w = synth.workload_c1(3000)
    w["mos"] = [("src net mynet", -1)] + [("dst port %d-%d" % (200 * i, 200 * i + 150), 0 if i % 7 == 0 else -1)
                                          for i in range(300)]
    configure(eng, orc, w["iplists"], w["mos"])
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=True)
    compare(b, orc)
    assert (b.counters()[:, 0] > 0).sum() > 250


def test_seventy_lists(eng, orc):
    w = synth.workload_c1(3000)
    rng = synth.Rng(404)
    lists = dict(w["iplists"])
    mos = [("src net mynet", -1)]
    for k in range(70):
        v4 = synth.prefix_list_v4(rng, 40, np.arange(4, 20), nested_frac=0.2)
        lists["l%02d" % k] = synth.iplist_text(v4, None)
        mos.append(("%s net l%02d" % (["src", "dst", ""][k % 3], k), -1))
    configure(eng, orc, lists, mos)
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    compare(b, orc, flows=False)
    assert (b.counters()[:, 0] > 0).sum() > 60


def test_c4_256_distinct_lists(eng, orc):
    """every object tests a list of its own: 32 lookups (four words of lookup
    bits per record), 256 list-only terms numbered behind the others"""
    w = synth.workload_c4(12000, prefixes=1500, n_lists=256, variants=1)
    mos = list(w["mos"])
    mos += [("dst net big%d" % k, -1) for k in (3, 77, 200)] + [("proto 6", 3), ("net big9 or big250", 5)]
    configure(eng, orc, w["iplists"], mos)
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    compare(b, orc, flows=False)
    assert (b.counters()[:, 0] > 0).sum() > 100


def test_c4_nested_objects(eng, orc):
    """children under table-resolved and listed parents, across mask words"""
    w = synth.workload_c4(6000, prefixes=5000, n_lists=4, variants=4)
    mos = list(w["mos"])
    n0 = len(mos)
    for i in range(n0):
        mos.append(("proto 6", i))                 # child of every object
    for i in range(n0):
        mos.append(("dst port 0-1023 or src port 0-1023", n0 + i))   # grandchild
    mos += [("", -1), ("not net big0", -1), ("src net big1 or dst net big2", 2 * n0)]
    configure(eng, orc, w["iplists"], mos)
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    compare(b, orc, flows=False)


def test_benchmark_size_properties(eng, orc):
    """4 Mi IPFIX records: counters = column sums over the match masks, a rerun
    gives the same bytes, and a prefix of the batch equals the oracle."""
    w = synth.workload_c2(1 << 22)
    configure(eng, orc, w["iplists"], w["mos"])
    b = eng.prepare(w["batch"])
    b.run()
    masks = b.masks()[:, 0]
    c = b.counters()
    names = [f["name"] for f in xenoeye_amd.fields()]
    bytes_be = b.column(names.index("in_bytes"))
    pkts_be = b.column(names.index("in_pkts"))
    nbytes = bytes_be.view(">u8").reshape(-1).astype(np.uint64)
    npkts = pkts_be.view(">u8").reshape(-1).astype(np.uint64)
    for m in range(len(w["mos"])):
        sel = ((masks >> np.uint64(m)) & np.uint64(1)).astype(bool)
        assert int(sel.sum()) == int(c[m, 0]), m
        assert int(nbytes[sel].sum(dtype=np.uint64)) == int(c[m, 1]), m
        assert int(npkts[sel].sum(dtype=np.uint64)) == int(c[m, 2]), m
    assert 0.2 < (masks != 0).mean() < 0.9
    # idempotence
    b.run()
    assert np.array_equal(b.masks()[:, 0], masks)
    assert np.array_equal(b.counters(), c)
    # the first datagrams against the oracle (templates come first in the batch)
    nd = 4000
    sub = synth.Batch()
    for p, s in zip(w["batch"].pkts[:nd], w["batch"].src[:nd]):
        sub.add(p, s)
    run_oracle(orc, sub, keep_flows=False)
    n = orc.nrecords()
    assert n > 90000
    assert np.array_equal(masks[:n], orc.masks()[:, 0])


def _random_filter(rng, depth=0):
    """boolean structure of every shape: conjunctions, negated disjunctions,
    nesting beyond what the clause form takes (those stay interpreted)"""
    atoms = ["proto %d" % int(rng.integers(0, 3) * 5 + int(rng.integers(0, 2)) * 6 + 1),
             "src port %d-%d" % tuple(sorted(rng.integers(0, 65536, 2))),
             "dst port 53 or 80 or 443", "net big0", "src net big1", "dst net big2 or big3",
             "src as %d-%d" % tuple(sorted(rng.integers(0, 400000, 2))), "dst as 64512-65534",
             "tos 0", "dst net 10.0.0.0/8 or big1", "src port 0"]
    r = rng.integers(0, 10)
    if depth < 4 and r < 4:
        return "(%s) %s (%s)" % (_random_filter(rng, depth + 1), ["and", "or"][int(rng.integers(0, 2))],
                                 _random_filter(rng, depth + 1))
    if depth < 4 and r < 6:
        return "not (%s)" % _random_filter(rng, depth + 1)
    return atoms[int(rng.integers(0, len(atoms)))]


def test_random_boolean_structures(eng, orc):
    rng = np.random.default_rng(2024)
    w = synth.workload_c4(8000, prefixes=3000, n_lists=4, variants=1)
    mos = [(_random_filter(rng), -1) for _ in range(150)]
    # some of them nested under earlier ones
    for i in range(40):
        mos.append((_random_filter(rng), int(rng.integers(0, 150 + i))))
    configure(eng, orc, w["iplists"], mos)
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    compare(b, orc, flows=False)
    c = b.counters()
    assert 20 < (c[:, 0] > 0).sum() < 190


def test_c5_shards_sum_to_the_whole(eng, orc):
    """512 exporters (half Netflow v9, half IPFIX), 17 objects: the per-object
    counters of 8 exporter shards, each generated and processed on its own,
    add up (wrapping u64) to the counters of the unsharded batch, and a prefix
    of the unsharded batch equals the oracle record for record."""
    from xenoeye_amd import shard
    world = 8
    w = synth.workload_c5(600000, prefixes=2000)
    configure(eng, orc, w["iplists"], w["mos"])
    b = eng.prepare(w["batch"])
    b.run()
    assert b.records == w["nrec"]
    whole = b.counters().copy()
    assert (whole[:, 0] > 0).sum() >= 9      # the v4 objects and `mynet` all see traffic
    total = np.zeros_like(whole)
    nrec = 0
    for r in range(world):
        part = synth.workload_c5(600000, prefixes=2000, rank=r, world=world)
        pb = eng.prepare(part["batch"])
        pb.run()
        nrec += pb.records
        total = total + pb.counters()        # uint64: wraps like the reference's adds
        # the same shard cut out of the whole batch gives the same counters
        cut = eng.prepare(shard.select(w["batch"].packed(), r, world))
        cut.run()
        assert np.array_equal(cut.counters(), pb.counters()), r
        pb.free()
        cut.free()
    assert nrec == w["nrec"]
    assert np.array_equal(total, whole)
    nd = 512 + 3000
    sub = synth.Batch()
    for p, a in zip(w["batch"].pkts[:nd], w["batch"].src[:nd]):
        sub.add(p, a)
    run_oracle(orc, sub, keep_flows=False)
    n = orc.nrecords()
    assert n > 100000
    assert np.array_equal(b.masks()[:n], orc.masks())
