"""Readers of the golden fixtures in tests/golden (data only; see
tests/golden/make_fixtures.py for how each file was produced)."""
import ast
import os

import numpy as np

from xenoeye_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def datagrams(data, offsets):
    return [bytes(data[int(offsets[i]):int(offsets[i + 1])]) for i in range(len(offsets) - 1)]


def sflow_reference():
    z = load("sflow_reference.npz")
    return datagrams(z["data"], z["offsets"]), z["counts"], z["rcs"], z["flows"]


def iplist_reference():
    z = load("iplist_reference.npz")
    texts = [bytes(t).decode() for t in z["texts"]]
    return texts, z["probes4"], z["probes6"], z["want4"], z["want6"]


def geo_reference():
    """{name: (image bytes, probes, found, payload)} for as4 / as6 / geo4 / geo6: answers of the reference's
    IP_BTRIE_LOOKUP over images its IP_BTRIE_ADD built"""
    z = load("geo_reference.npz")
    return {n: (bytes(z[n + "_image"]), z[n + "_probes"], z[n + "_found"], z[n + "_payload"]) for n in ("as4", "as6", "geo4", "geo6")}


def netflow_oracle():
    z = load("netflow_oracle.npz")
    for case in [bytes(c).decode() for c in z["cases"]]:
        g = lambda k: z[case + "_" + k]  # noqa: E731
        mos = []
        for m in g("mos"):
            p, f = bytes(m).decode().split("\t", 1)
            mos.append((f, int(p)))
        lists = {}
        for m in g("lists"):
            if bytes(m):
                n, t = bytes(m).decode().split("\t", 1)
                lists[n] = t
        devices = [ast.literal_eval(bytes(d).decode()) for d in g("devices") if bytes(d)]
        b = synth.Batch()
        for p, s in zip(datagrams(g("data"), g("offsets")), g("src")):
            b.add(p, int(s).to_bytes(4, "little"))
        yield dict(name=case, iplists=lists, mos=mos, devices=devices, batch=b, masks=g("masks"), flows=g("flows"),
                   counters=g("counters"), dropped=g("dropped"))
