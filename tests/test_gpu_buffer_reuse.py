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
