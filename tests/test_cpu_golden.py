"""The oracle (and the product's host-side list flattening) against the
committed golden fixtures.  Runs anywhere: needs neither the reference checkout
nor a GPU."""
import numpy as np

import golden_io
from helpers import configure, run_oracle
from host_binding import Host, walk_flat


def test_oracle_sflow_equals_reference_records(orc):
    pkts, counts, rcs, flows = golden_io.sflow_reference()
    orc.mo_add("")
    orc.log_config(keep_flows=True)
    pos = 0
    for p, n, rc in zip(pkts, counts, rcs):
        n0 = orc.nrecords()
        assert orc.sflow(p, b"\0\0\0\0") == rc
        mine = orc.flows()[n0:]
        assert len(mine) == n
        assert np.array_equal(mine, flows[pos:pos + n])
        pos += n
    assert pos == len(flows) > 300


def test_oracle_iplists_equal_reference_answers(orc):
    texts, p4, p6, want4, want6 = golden_io.iplist_reference()
    for i, t in enumerate(texts):
        orc.iplist_add("l%d" % i, t)
        got4 = [orc.match4("l%d" % i, bytes(a)) for a in p4]
        got6 = [orc.match6("l%d" % i, bytes(a)) for a in p6]
        assert got4 == want4[i].tolist(), i
        assert got6 == want6[i].tolist(), i


def test_flattened_lists_equal_reference_answers():
    texts, p4, p6, want4, want6 = golden_io.iplist_reference()
    h = Host()
    for i, t in enumerate(texts):
        h.add_iplist("l%d" % i, t)
    words, devs = h.flatten()
    for i, (root4, root6, nodes4, nodes6, bit, chains4, chains6) in enumerate(devs):
        got4 = [walk_flat(words, root4, nodes4, bytes(a), bit, chains4) for a in p4]
        got6 = [walk_flat(words, root6, nodes6, bytes(a), bit, chains6) for a in p6]
        assert got4 == want4[i].tolist(), i
        assert got6 == want6[i].tolist(), i
    h.close()


def test_oracle_netflow_regression_pins(orc):
    for case in golden_io.netflow_oracle():
        orc.reset()
        configure(None, orc, case["iplists"], case["mos"], case["devices"])
        run_oracle(orc, case["batch"], keep_flows=True)
        assert np.array_equal(orc.masks(), case["masks"]), case["name"]
        assert np.array_equal(orc.flows(), case["flows"]), case["name"]
        assert np.array_equal(orc.counters(), case["counters"]), case["name"]
        assert np.array_equal(orc.dropped(), case["dropped"]), case["name"]


def test_recorded_edge_case_counts():
    """record counts the reference was observed to deliver for the decode
    edge cases (count beyond the set's records, trailing bytes, lost second
    set, padding that becomes a record)"""
    case = [c for c in golden_io.netflow_oracle() if c["name"] == "edges"][0]
    # v9: 4 + 2 + 1; ipfix 2 real + 1 from the padding; v5: 7
    assert len(case["masks"]) == 4 + 2 + 1 + 3 + 7
