"""Shared helpers of the parity tests: drive the oracle and the HIP engine
with the same configuration and datagrams and compare everything they emit."""
import numpy as np


def configure(eng, orc, iplists=None, mos=(), devices=(), tables=()):
    for name, text in (iplists or {}).items():
        if eng is not None:
            eng.add_iplist(name, text)
        orc.iplist_add(name, text)
    for d in devices:
        if eng is not None:
            eng.add_device(**d)
        orc.device_add(**d)
    for filt, parent in mos:
        if eng is not None:
            eng.add_mo("mo", filt, parent)
        orc.mo_add(filt, parent)
    for mo, fields in tables:
        if eng is not None:
            eng.add_fwm(mo, fields)
        orc.fwm_add(mo, fields)
    if eng is not None:
        eng.commit()


def run_oracle(orc, batch, keep_flows=True):
    orc.log_config(keep_flows=keep_flows)
    for p, s, k in zip(batch.pkts, batch.src, batch.kind):
        if k == 1:
            orc.sflow(p, s)
        else:
            orc.netflow(p, s)


def compare(batch_out, orc, flows=True, sflow_states=None):
    """batch_out: PreparedBatch after run(); orc: oracle after run_oracle()."""
    n = orc.nrecords()
    dd, od = batch_out.dropped(), orc.dropped()
    # sFlow samples that never reach classification have a slot on the device
    # (status 2) but no entry in the oracle's log
    sel = dd != 2
    assert int(sel.sum()) == n, "record count: device %d, oracle %d" % (int(sel.sum()), n)
    dd = dd[sel]
    dm, om = batch_out.masks(), orc.masks()
    assert not dm[~sel].any(), "an unclassified sample has match bits"
    dm = dm[sel]
    assert dm.shape == om.shape
    bad = np.nonzero((dm != om).any(axis=1))[0]
    assert bad.size == 0, "match bitmask differs at records %s: device / oracle words %s" % (
        bad[:10], [("%x" % int(dm[r, w]), "%x" % int(om[r, w])) for r in bad[:4] for w in range(dm.shape[1])
                   if dm[r, w] != om[r, w]])
    assert np.array_equal(dd, od), "dropped flags differ"
    assert np.array_equal(batch_out.counters(), orc.counters()), "counters differ"
    if flows and n:
        dall = batch_out.flow_info()
        # the record-after-record read-out hands out the same bytes, whatever it skips
        picks = np.unique(np.linspace(0, dall.shape[0] - 1, num=min(dall.shape[0], 1500)).astype(np.int64))
        assert np.array_equal(batch_out.flow_records(picks), dall[picks]), "cursor read-out differs from the block read-out"
        df, of = dall[sel], orc.flows()
        # a dropped record never reaches consumers; the reference leaves its
        # dev_mark unset, compare only delivered records byte for byte
        keep = od == 0
        bad = np.nonzero((df[keep] != of[keep]).any(axis=1))[0]
        if bad.size:
            r = np.nonzero(keep)[0][bad[0]]
            diff = np.nonzero(df[r] != of[r])[0]
            raise AssertionError("flow_info differs at record %d, byte offsets %s" % (r, diff[:16]))
