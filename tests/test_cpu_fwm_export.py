"""The oracle's window export against a restatement in Python of the sort key
of fwm_sort_and_dump() (monit-objects-fwm.c:536-565) on a small batch, and the
mo.conf reader's "limit" / "time" keys.  (tkvdb, whose cursor order the
reference relies on, is not in this image: the order is taken to be ascending
by key bytes, parity unpinned.)"""
import struct

from xenoeye_amd import synth
from helpers import run_oracle


def _sort_key(fields, key, sums):
    out, kp, vi = b"", 0, 0
    for aggr, size, desc in fields:
        if aggr:
            part = struct.pack(">Q", sums[vi]); vi += 1
        else:
            part = key[kp:kp + size]; kp += size
        out += bytes(255 - x for x in part) if desc else part
    return out


def test_oracle_export_order(orc):
    w = synth.workload_c2(4000, prefixes=100, seed=2)
    orc_tables = [(["packets desc", "proto", "dst port"], [(1, 8, 1), (0, 1, 0), (0, 2, 0)]),
                  (["src host desc", "octets", "proto desc"], [(0, 4, 1), (1, 8, 0), (0, 1, 1)])]
    for name, text in w["iplists"].items():
        orc.iplist_add(name, text)
    orc.mo_add("", -1)
    ids = [orc.fwm_add(0, f) for f, _ in orc_tables]
    run_oracle(orc, w["batch"], keep_flows=False)
    for t, (_, desc) in zip(ids, orc_tables):
        rows = orc.fwm_read(t)
        want = sorted(rows.items(), key=lambda kv: _sort_key(desc, kv[0], kv[1]))
        assert len(want) > 50
        got, others, hit = orc.fwm_export(t, 0)
        assert got == want and not hit and not any(others)
        got, others, hit = orc.fwm_export(t, 10)
        assert got == want[:10] and hit
        nv = len(want[0][1])
        assert others == tuple(sum(v[j] for _, v in want[10:]) & (2 ** 64 - 1) for j in range(nv))
        got, others, hit = orc.fwm_export(t, len(want))
        assert got == want and hit and not any(others)
