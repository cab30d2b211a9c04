"""GPU parity under random configurations and damaged input: random object
trees (all term kinds, lists, functions), random devices.conf entries, random
v9 / IPFIX templates (unknown, repeated and wrongly sized elements,
variable-length ones), datagrams with bytes, lengths and counts corrupted or
cut short, v5 and sFlow mixed in.  Every case is compared with the CPU oracle
record for record (match masks, dropped flags, counters, struct flow_info)."""
import struct

import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from helpers import run_oracle, compare
from oracle_binding import port_names

NAMES = port_names()
NAMED = [p for p in range(65536) if NAMES[p]]

pytestmark = pytest.mark.gpu

FIELD_IDS = [(8, 4), (12, 4), (7, 2), (11, 2), (4, 1), (6, 1), (6, 2), (1, 4), (1, 8), (2, 4), (2, 8), (10, 2), (10, 4),
             (14, 4), (16, 2), (16, 4), (17, 4), (27, 16), (28, 16), (5, 1), (32, 2), (56, 6), (80, 6), (58, 2), (59, 2),
             (21, 4), (22, 4), (9, 1), (13, 1), (15, 4), (52, 1), (53, 1)]


def _port_text(p):
    return "%s (%d)" % (NAMES[p], p) if NAMES[p] else "%d" % p


def _pair_text(rng, a, b):
    """one of the shapes ports_pair_to_str() (utils.c:39-77) produces, not always the one of (a, b)"""
    r = int(rng.integers(0, 5))
    na = "%s(%d)" % (NAMES[a], a) if NAMES[a] else "%d" % a
    nb = "%s(%d)" % (NAMES[b], b) if NAMES[b] else "%d" % b
    return ["%s -> %s" % (na, nb), "%s ->" % na, "-> %s" % nb, "%d -> %d" % (a, a), "%s -> %s" % (nb, na)][r]


def _some_port(rng):
    r = int(rng.integers(0, 3))
    if r == 0:
        return NAMED[int(rng.integers(0, len(NAMED)))]
    if r == 1:
        return [53, 80, 123, 443, 8080, 22, 25, 0][int(rng.integers(0, 8))]     # the generator's frequent ports
    return int(rng.integers(0, 65536))


AS_NAMES = ["Example Net", "ACME", "x", ""]
GEO = [("EU", "DE", "Germany", "Berlin", "Berlin", "10115", "52.5", "13.4"),
       ("EU", "RU", "Russia", "Moscow", "Moscow", "101000", "55.7", "37.6"),
       ("NA", "US", "United States", "California", "San Jose", "95101", "37.3", "-121.9"),
       ("", "", "", "", "", "", "", "")]
GEO_KW = ["continent", "country_code", "country", "state", "city", "zip", "lat", "long"]


def _geo_atom(rng):
    """asn() / asd() / geo function over an address field (filter.c:753-834)"""
    fld = ["src host", "dst host", "src host6", "dst host6", "dev-ip"][int(rng.integers(0, 5))]
    r = int(rng.integers(0, 3))
    if r == 0:
        return "asn(%s) %d-%d" % (fld, *sorted(int(x) for x in rng.integers(0, 12, 2)))
    # (an empty quoted string does not lex, filter-lexer.c: the texts asked for are never empty)
    if r == 1:
        return "asd(%s) '%s' or '%s'" % (fld, AS_NAMES[int(rng.integers(0, 3))], AS_NAMES[int(rng.integers(0, 3))])
    g = int(rng.integers(0, 8))
    return "%s(%s) '%s'" % (GEO_KW[g], fld, GEO[int(rng.integers(0, 3))][g])


def _geo_databases(rng):
    """small databases in the reference's layout: short prefixes, so that random addresses fall into
    them, nested ones, the zero address (what an absent field reads as) among them"""
    from test_gpu_geo_functions import as_payload, geo_payload, build_db
    out = {}
    for fam, nbytes in ((4, 4), (6, 16)):
        asl, gel = [], []
        for k in range(int(rng.integers(1, 40))):
            a = bytes(int(x) for x in rng.integers(0, 256, nbytes))
            if fam == 6:
                a = bytes([0x20 | (a[0] & 0x0f)]) + a[1:]
            if rng.integers(0, 8) == 0:
                a = bytes(nbytes)
            m = int(rng.integers(0, 14 if fam == 4 else 24))
            asl.append((a, m, as_payload(int(rng.integers(0, 12)), AS_NAMES[int(rng.integers(0, 4))])))
            gel.append((a, int(rng.integers(0, 14 if fam == 4 else 24)), geo_payload(*GEO[int(rng.integers(0, 4))])))
        out[(1, fam)] = build_db(asl, 216)
        out[(0, fam)] = build_db(gel, 228)
    return out


def _atom(rng, nlists):
    k = int(rng.integers(0, 22))
    if k >= 20:
        return _geo_atom(rng)
    if k == 17:
        return "div_l(octets, packets, %d) %d-%d" % (int(rng.integers(-1, 12)), *sorted(int(x) for x in rng.integers(0, 2000, 2)))
    if k == 18:
        return "portstr(%s port) '%s' or '%s'" % (["src", "dst"][int(rng.integers(0, 2))], _port_text(_some_port(rng)),
                                                 _port_text(_some_port(rng)).upper())
    if k == 19:
        return "ppstr(src port, dst port) '%s'" % _pair_text(rng, _some_port(rng), _some_port(rng))
    d = ["", "src ", "dst "][int(rng.integers(0, 3))]
    d2 = ["src ", "dst "][int(rng.integers(0, 2))]
    if k == 0:
        return "proto %d" % [1, 6, 17, 47][int(rng.integers(0, 4))]
    if k == 1:
        return "%sport %d-%d" % (d, *sorted(int(x) for x in rng.integers(0, 65536, 2)))
    if k == 2:
        return "%sport 53 or 80 or 443 or %d" % (d, int(rng.integers(0, 65536)))
    if k == 3:
        return "%snet l%d" % (d, int(rng.integers(0, nlists)))
    if k == 4:
        return "%snet l%d or l%d or %d.0.0.0/%d" % (d, int(rng.integers(0, nlists)), int(rng.integers(0, nlists)),
                                                     int(rng.integers(1, 224)), int(rng.integers(1, 9)))
    if k == 5:
        return "%snet6 l%d" % (d, int(rng.integers(0, nlists)))
    if k == 6:
        return "%shost %d.%d.0.0/%d" % (d, int(rng.integers(0, 256)), int(rng.integers(0, 256)), int(rng.integers(0, 20)))
    if k == 7:
        return "%sas %d-%d" % (d2, *sorted(int(x) for x in rng.integers(0, 500000, 2)))
    if k == 8:
        return "tcp-flags %d or %d" % (int(rng.integers(0, 64)), int(rng.integers(0, 64)) << 8)
    if k == 9:
        return "tfstr(tcp-flags) '%s'" % ["SYN", "ACK", "ACK+SYN", "ACK+PSH", "FIN", "RST"][int(rng.integers(0, 6))]
    if k == 10:
        return "min(src port, dst port) %d-%d" % tuple(sorted(int(x) for x in rng.integers(0, 4000, 2)))
    if k == 11:
        return "div_r(octets, packets, %d) %d-%d" % (int(rng.integers(1, 200)), *sorted(int(x) for x in rng.integers(0, 1600, 2)))
    if k == 12:
        return "div(octets, packets) %d-%d" % tuple(sorted(int(x) for x in rng.integers(0, 1600, 2)))
    if k == 13:
        return "%smac %s" % (d2, ":".join("%02x" % int(x) for x in rng.integers(0, 4, 6)))
    if k == 14:
        return "%sifidx %d-%d" % (d2, *sorted(int(x) for x in rng.integers(0, 60, 2)))
    if k == 15:
        return "tos %d" % int(rng.integers(0, 4))
    if rng.integers(0, 2):
        return "dev-ip 10.%d.0.0/16" % int(rng.integers(0, 4))
    return "dev-id %d or %d" % (int(rng.integers(0, 6)), int(rng.integers(0, 6)))


def _filter(rng, nlists, depth=0):
    r = int(rng.integers(0, 10))
    if depth < 3 and r < 3:
        return "(%s) %s (%s)" % (_filter(rng, nlists, depth + 1), ["and", "or"][int(rng.integers(0, 2))],
                                 _filter(rng, nlists, depth + 1))
    if depth < 3 and r == 3:
        return "not (%s)" % _filter(rng, nlists, depth + 1)
    if r == 4 and depth == 0:
        return ""
    return _atom(rng, nlists)


def _template(rng):
    n = int(rng.integers(1, 14))
    t = []
    for _ in range(n):
        r = int(rng.integers(0, 12))
        if r == 0:
            t.append((int(rng.integers(300, 400)), int(rng.integers(1, 9))))        # unknown element
        elif r == 1:
            fid, ln = FIELD_IDS[int(rng.integers(0, len(FIELD_IDS)))]
            t.append((fid, int(rng.integers(1, 18))))                               # odd size
        else:
            t.append(FIELD_IDS[int(rng.integers(0, len(FIELD_IDS)))])
    if int(rng.integers(0, 4)) == 0:
        t.append(t[int(rng.integers(0, len(t)))])                                   # repeated element
    return t


def _damage(rng, p):
    p = bytearray(p)
    r = int(rng.integers(0, 10))
    if r < 5:
        return bytes(p)
    if r < 8:
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, min(len(p), 40)))
            p[pos] = int(rng.integers(0, 256))
        return bytes(p)
    if r == 8:
        return bytes(p[:int(rng.integers(0, len(p) + 1))])
    return bytes(p) + bytes(int(x) for x in rng.integers(0, 256, int(rng.integers(1, 9))))


def _case(seed):
    rng = np.random.default_rng(seed)
    srng = synth.Rng(seed)
    nlists = int(rng.integers(1, 5)) if seed % 3 else int(rng.integers(9, 30))
    lists = {}
    for k in range(nlists):
        v4 = synth.prefix_list_v4(srng, int(rng.integers(1, 60)), np.arange(1, 33), nested_frac=0.3)
        v6 = synth.prefix_list_v6(srng, int(rng.integers(1, 30)), lo=3, hi=128, top=0x2000)
        lists["l%d" % k] = synth.iplist_text(v4, v6 if rng.integers(0, 3) else None)
    nmo = int(rng.integers(1, 40)) if seed % 4 else int(rng.integers(60, 140))
    mos = []
    for i in range(nmo):
        parent = int(rng.integers(0, i)) if (i and rng.integers(0, 4) == 0) else -1
        mos.append((_filter(rng, nlists), parent))
    devices = []
    if seed % 2:
        for e in range(int(rng.integers(1, 4))):
            marks = tuple(_atom(rng, nlists) for _ in range(int(rng.integers(0, 3)))) if rng.integers(0, 2) else ()
            marks = tuple(m for m in marks if "dev-" not in m)
            devices.append(dict(ip="10.%d.0.%d" % (e, e + 1), sampling_rate=int(rng.integers(1, 2000)),
                                skip_unmarked=bool(rng.integers(0, 2)) and bool(marks), marks=marks))
    b = synth.Batch()
    exps = [bytes([10, e, 0, e + 1]) for e in range(4)]
    for e, addr in enumerate(exps):
        sets = []
        for tid in (256, 257, 300):
            t = _template(rng)
            nrec = int(rng.integers(1, 120))
            rec = synth.flow_columns(srng, nrec, t)
            rl = rec.shape[1]
            if (e + tid) % 2:
                per = max(1, min(nrec, 1400 // max(rl, 1)))
                b.add(synth.v9_template_packet(tid, t, e + 1), addr)
                for p in synth.v9_data_packets(tid, rec, per, e + 1):
                    b.add(_damage(rng, p), addr)
            else:
                per = max(1, min(nrec, 400 // max(rl, 1)))
                b.add(synth.ipfix_template_packet(tid, t, e + 1), addr)
                sets += [p[16:] for p in synth.ipfix_data_packets(tid, rec, per, e + 1)]
        # IPFIX: one to five sets a datagram, the data sets of the exporter's
        # templates mixed, now and then an options template set or a set of an
        # id that means nothing between them
        order = rng.permutation(len(sets))
        at = 0
        while at < len(sets):
            k = int(rng.integers(1, 6)) if rng.integers(0, 3) else 1
            body = b""
            for j in order[at:at + k]:
                r = int(rng.integers(0, 8))
                if r == 0:
                    body += struct.pack(">HHHHHHH", 3, 14, 400, 1, 1, 149, 4)
                elif r == 1:
                    body += struct.pack(">HH", int(rng.integers(4, 256)), 8) + bytes(4)
                body += sets[int(j)]
            at += k
            b.add(_damage(rng, synth.ipfix_header(16 + len(body), e + 1) + body), addr)
        if rng.integers(0, 2):
            b.add(_damage(rng, synth.v5_packet(srng, int(rng.integers(1, 20)))), addr)
    if seed % 5 == 0:
        sf = synth.workload_c3(60, seed=seed)
        for p, a, k in zip(sf["batch"].pkts, sf["batch"].src, sf["batch"].kind):
            b.add(_damage(rng, p), a, k)
    # fixed-window tables over some objects
    keyf = ["src host", "dst host", "src host6", "dst host6", "proto", "src port", "dst port", "tcp-flags", "src as",
            "dst as", "src mac", "dst mac", "src ifidx", "dst ifidx", "src tos", "src vlan", "dev-ip", "src mask",
            "div(octets, packets)", "div_r(bits, packets, 8) desc", "div_l(octets, packets, 2)", "min(src port, dst port)",
            "tfstr(tcp-flags)", "portstr(dst port)", "ppstr(src port, dst port)", "min(src as, dst ifidx)"]
    valf = ["octets", "bits", "packets", "bits desc", "packets asc"]
    tables = []
    for _ in range(int(rng.integers(0, 4))):
        nk = int(rng.integers(0, 4))
        ks = [keyf[int(x)] for x in rng.integers(0, len(keyf), nk)]
        vs = [valf[int(x)] for x in rng.integers(0, len(valf), int(rng.integers(1, 4)))]
        fl = ks + vs
        rng.shuffle(fl)
        tables.append((int(rng.integers(0, nmo)), [str(x) for x in fl]))
    return dict(batch=b, iplists=lists, mos=mos, devices=devices, tables=tables, geodb=_geo_databases(rng))


import os
CASES_PER_CHUNK = int(os.environ.get("XE_FUZZ_CASES", "25"))
SEED0 = int(os.environ.get("XE_FUZZ_SEED0", "0"))


@pytest.mark.parametrize("chunk", range(10))
def test_random_configurations_and_damaged_datagrams(chunk, orc, monkeypatch):
    if chunk % 2:
        # every other chunk: the header walk of these small batches is spread over
        # host threads as well (by default only batches of 4096 datagrams are)
        monkeypatch.setenv("XE_PREPASS_MIN_DATAGRAMS", "1")
        monkeypatch.setenv("XE_PREPASS_THREADS", str(2 + chunk // 2))
    compared = records = one_sided = walked = 0
    for seed in range(SEED0 + chunk * CASES_PER_CHUNK, SEED0 + (chunk + 1) * CASES_PER_CHUNK):
        w = _case(seed)
        eng = xenoeye_amd.Engine(device=0)
        try:
            orc.reset()
            ok, skipped = _configure_both(eng, orc, w, seed)
            one_sided += skipped
            if not ok:
                continue
            b = eng.prepare(w["batch"])
            b.run()
            if b.unsupported_sets:
                # a damaged template that declares more elements than a device
                # template holds: its sets are left out, which the oracle does not do
                continue
            run_oracle(orc, w["batch"], keep_flows=True)
            try:
                compare(b, orc)
            except AssertionError as e:
                raise AssertionError("seed %d: %s" % (seed, e))
            for t in range(len(w["tables"])):
                got, dropped = eng.fwm_read(t)
                assert dropped == 0 and got == orc.fwm_read(t), "seed %d: fixed-window table %d differs" % (seed, t)
            compared += 1
            records += orc.nrecords()
            # the same datagrams once more: the templates are known now, so the
            # device walks the headers of the datagrams of the usual shape itself
            b.free()
            orc.log_clear()
            b = eng.prepare(w["batch"])
            b.run()
            if not b.unsupported_sets:
                run_oracle(orc, w["batch"], keep_flows=True)
                try:
                    compare(b, orc)
                except AssertionError as e:
                    raise AssertionError("seed %d, second pass: %s" % (seed, e))
                walked += b.device_walked()[0]
        finally:
            eng.close()
    assert compared >= CASES_PER_CHUNK * 3 // 4 and records > 200 * CASES_PER_CHUNK, (compared, records)
    assert one_sided <= CASES_PER_CHUNK // 5, "%d configurations only one side accepts" % one_sided
    assert walked > 0, "no datagram of the second passes was walked on the device"


def _configure_both(eng, orc, w, seed):
    """the same configuration into the engine and the oracle, piece by piece: both must accept a
    piece or both reject it (then the case ends: the reference would not start with it).  The one
    allowed difference is a piece this library declares not evaluable on the device
    (XE_EUNSUPPORTED); those are counted.  Returns (configured, one-sided pieces)."""
    def pair(what, text, on_engine, on_oracle):
        code = 0
        try:
            on_engine()
        except xenoeye_amd.XeError as e:
            code = e.code
        refused = False
        try:
            on_oracle()
        except ValueError:
            refused = True
        if code == -5 and not refused:
            return "unsupported"
        assert (code != 0) == refused, "seed %d: %s %r: engine code %d, oracle %s" % (
            seed, what, text, code, "rejects" if refused else "accepts")
        return "rejected" if refused else "ok"

    eng.set_port_names(NAMES)
    orc.set_port_names(NAMES)
    for (kind, fam), image in w["geodb"].items():
        eng.set_geodb(kind, fam, image)
        orc.geodb_set(kind, fam, image)
    for name, text in w["iplists"].items():
        eng.add_iplist(name, text)
        orc.iplist_add(name, text)
    for d in w["devices"]:
        r = pair("device", d, lambda: eng.add_device(**d), lambda: orc.device_add(**d))
        if r != "ok":
            return False, int(r == "unsupported")
    for i, (filt, parent) in enumerate(w["mos"]):
        r = pair("filter", filt, lambda: eng.add_mo("mo", filt, parent), lambda: orc.mo_add(filt, parent))
        if r != "ok":
            return False, int(r == "unsupported")
        # nothing the generator writes needs the caller's filter_match()
        assert not eng.mo_host_only(i), "seed %d: %r is left to the host: %s" % (seed, filt, eng.mo_host_only_reason(i))
    for mo, fields in w["tables"]:
        r = pair("table", fields, lambda: eng.add_fwm(mo, fields), lambda: orc.fwm_add(mo, fields))
        if r != "ok":
            return False, int(r == "unsupported")
    eng.commit()
    return True, 0
