"""One handle, batches of very different sizes and kinds one after the other:
the handle hands the device buffers of freed batches to later ones, and what
an earlier batch left in them must not show in any result."""
import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from helpers import compare, configure, run_oracle

pytestmark = pytest.mark.gpu


def test_batches_of_changing_size_on_one_handle(orc):
    w = synth.workload_c2(30000, prefixes=500, seed=21)
    eng = xenoeye_amd.Engine(device=0)
    configure(eng, orc, w["iplists"], w["mos"])
    sf = synth.workload_c3(4000, seed=8)["batch"]
    plan = [w["batch"],
            synth.workload_c2(300, prefixes=500, seed=22)["batch"],
            sf,
            synth.workload_c2(12000, prefixes=500, seed=23)["batch"],
            synth.workload_c2(40, prefixes=500, seed=24)["batch"],
            synth.workload_c2(29000, prefixes=500, seed=25)["batch"],
            synth.Batch()]
    kept = None
    for i, batch in enumerate(plan):
        orc.log_clear()
        b = eng.prepare(batch)
        b.run()
        run_oracle(orc, batch)
        if len(batch.pkts):
            compare(b, orc)
        else:
            assert b.records == 0
        # one batch stays alive over the next one (its buffers are not handed on)
        if kept is not None:
            km, kb = kept
            assert np.array_equal(kb.masks(), km)
            kb.free()
            kept = None
        if i == 2:
            kept = (b.masks().copy(), b)
        else:
            b.free()
    eng.close()


def test_overlapped_batches_equal_the_oracle(orc):
    """Overlap mode: prepare() does not wait for the datagram bytes, the next batch is prepared while
    the one before still runs, results are fetched one batch late - every batch must still equal the
    oracle (Netflow from page-locked buffers read in place, from ordinary memory, and sFlow, whose walk
    waits for the bytes)."""
    w = synth.workload_c2(20000, prefixes=500, seed=31)
    eng = xenoeye_amd.Engine(device=0)
    configure(eng, orc, w["iplists"], w["mos"])
    eng.set_overlap(True)
    batches = [w["batch"],
               synth.workload_c2(15000, prefixes=500, seed=32)["batch"],
               synth.workload_c3(3000, seed=9)["batch"],
               synth.workload_c2(18000, prefixes=500, seed=33)["batch"],
               synth.workload_c2(500, prefixes=500, seed=34)["batch"]]
    prev = None
    for i, batch in enumerate(batches + [None]):
        cur = None
        if batch is not None:
            packed = batch.packed()
            if i != 1:      # batch 1 comes from ordinary memory
                packed = (eng.pinned(packed[0]),) + tuple(packed[1:])
            cur = (eng.prepare(packed), batch, packed)
            cur[0].run()
        if prev is not None:
            orc.log_clear()
            run_oracle(orc, prev[1])
            compare(prev[0], orc)
            prev[0].input_done()
            prev[0].free()
        prev = cur
    eng.set_overlap(False)
    b = eng.prepare(batches[0])
    b.run()
    orc.log_clear()
    run_oracle(orc, batches[0])
    compare(b, orc)
    b.free()
    eng.close()


def test_uploads_sent_ahead_equal_the_oracle(orc):
    """xe_gpu_upload_begin / xe_gpu_batch_prepare_uploaded: the bytes of the next batch are on their way
    while this one is prepared; the walk reads the device copy.  Netflow, sFlow, an upload that is given
    up, and a batch whose byte count does not match its upload."""
    w = synth.workload_c2(20000, prefixes=500, seed=41)
    eng = xenoeye_amd.Engine(device=0)
    configure(eng, orc, w["iplists"], w["mos"])
    batches = [w["batch"],
               synth.workload_c2(9000, prefixes=500, seed=42)["batch"],
               synth.workload_c3(2500, seed=10)["batch"],
               synth.workload_c2(14000, prefixes=500, seed=43)["batch"]]
    packed = []
    for b in batches:
        p = b.packed()
        packed.append((eng.pinned(p[0]),) + tuple(p[1:]))
    up = eng.upload_begin(packed[0][0])
    prev = None
    for i in range(len(batches)):
        nxt = eng.upload_begin(packed[i + 1][0]) if i + 1 < len(batches) else None
        cur = eng.prepare(packed[i], upload=up)
        up = nxt
        cur.run()
        if prev is not None:
            orc.log_clear()
            run_oracle(orc, prev[1])
            compare(prev[0], orc)
            prev[0].free()
        prev = (cur, batches[i])
    orc.log_clear()
    run_oracle(orc, prev[1])
    compare(prev[0], orc)
    prev[0].free()
    # an upload nobody prepares, and one of the wrong size
    eng.upload_free(eng.upload_begin(packed[1][0]))
    bad = eng.upload_begin(packed[1][0])
    with pytest.raises(xenoeye_amd.XeError):
        eng.prepare(packed[0], upload=bad)      # (the call takes the upload over even when it fails)
    eng.close()
