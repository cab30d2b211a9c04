"""Deterministic synthetic export-packet generators (Netflow v9, IPFIX, v5,
sFlow v5) and IP-list text generators.

Everything is derived from splitmix64 streams so the same seed gives the same
bytes everywhere.  The generators only build *inputs*; they know nothing about
how the packets are decoded.
"""
import struct

import numpy as np

MASK64 = (1 << 64) - 1

# (element id, encoded length) lists of the workload templates
TMPL_V9_C1 = [(8, 4), (12, 4), (7, 2), (11, 2), (4, 1), (1, 4), (2, 4)]                      # 21 B
TMPL_IPFIX_V4 = [(8, 4), (12, 4), (7, 2), (11, 2), (4, 1), (6, 1), (1, 8), (2, 8),
                 (10, 4), (14, 4), (16, 4), (17, 4)]                                        # 46 B
TMPL_IPFIX_V6 = [(27, 16), (28, 16), (7, 2), (11, 2), (4, 1), (6, 1), (1, 8), (2, 8),
                 (10, 4), (14, 4)]                                                          # 62 B

SEED_C1 = 0x5EED0001
SEED_C2 = 0x5EED0002
SEED_C3 = 0x5EED0003
SEED_C4 = 0x5EED0004
SEED_C5 = 0x5EED0005


def splitmix64_stream(seed, n):
    """n outputs of splitmix64 started at `seed`, vectorised."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & MASK64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class Rng:
    """Chunked splitmix64 source with helpers; one instance per stream."""

    def __init__(self, seed):
        self.seed = seed & MASK64
        self.pos = 0

    def u64(self, n):
        with np.errstate(over="ignore"):
            start = (self.seed + self.pos * 0x9E3779B97F4A7C15) & MASK64
        out = splitmix64_stream(start, n)
        self.pos += n
        return out

    def below(self, n, bound):
        return (self.u64(n) % np.uint64(bound)).astype(np.int64)

    def bytes(self, n):
        return self.u64((n + 7) // 8).view(np.uint8)[:n].copy()


def _be(arr, width):
    """uint array -> big-endian bytes [n, width]."""
    a = np.asarray(arr, dtype=np.uint64)
    out = np.empty((a.shape[0], width), dtype=np.uint8)
    for i in range(width):
        out[:, i] = (a >> np.uint64(8 * (width - 1 - i))) & np.uint64(0xFF)
    return out


# ------------------------------------------------------------------ ip lists
def prefix_list_v4(rng, n, lens, nested_frac=0.0):
    """n IPv4 prefixes; `lens` = array of candidate lengths sampled uniformly
    (repeat a length to weight it).  A `nested_frac` share are more-specifics
    of an earlier prefix.  Returns uint32 network (host order) and length arrays."""
    lens = np.asarray(lens)
    ln = lens[rng.below(n, len(lens))]
    addr = (rng.u64(n) & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    k = int(n * nested_frac)
    if k:
        parent = rng.below(k, n - k)
        extra = 1 + rng.below(k, 6)
        pl = ln[parent]
        nl = np.minimum(pl + extra, 32)
        keep = np.where(pl > 0, (np.uint64(0xFFFFFFFF) << (32 - pl).astype(np.uint64)) & np.uint64(0xFFFFFFFF),
                        np.uint64(0))
        addr[n - k:] = (addr[parent] & keep) | (addr[n - k:] & ~keep & np.uint64(0xFFFFFFFF))
        ln[n - k:] = nl
    mask = np.where(ln > 0, (np.uint64(0xFFFFFFFF) << (32 - ln).astype(np.uint64)) & np.uint64(0xFFFFFFFF),
                    np.uint64(0))
    return (addr & mask).astype(np.uint32), ln.astype(np.int64)


def prefix_list_v6(rng, n, lo=29, hi=64, top=0x2000):
    """n IPv6 prefixes of length lo..hi inside top::/3-ish space.  Returns
    [n,16] uint8 network bytes and lengths."""
    ln = lo + rng.below(n, hi - lo + 1)
    raw = rng.bytes(n * 16).reshape(n, 16)
    raw[:, 0] = (top >> 8) | (raw[:, 0] & 0x1F)
    out = np.zeros_like(raw)
    for i in range(16):
        bits = np.clip(ln - 8 * i, 0, 8)
        m = (0xFF00 >> bits) & 0xFF
        out[:, i] = raw[:, i] & m.astype(np.uint8)
    return out, ln.astype(np.int64)


def iplist_text(v4=None, v6=None):
    """Text of an IP-list file (one prefix per line, '\\n'-terminated)."""
    import ipaddress
    lines = []
    if v4 is not None:
        a, l = v4
        for x, n in zip(a.tolist(), l.tolist()):
            lines.append("%d.%d.%d.%d/%d" % (x >> 24, (x >> 16) & 255, (x >> 8) & 255, x & 255, n))
    if v6 is not None:
        a, l = v6
        for row, n in zip(a, l.tolist()):
            lines.append("%s/%d" % (ipaddress.IPv6Address(bytes(row)).compressed, n))
    return "\n".join(lines) + "\n"


def addrs_in_prefixes_v4(rng, n, nets, lens):
    """n addresses (uint32 host order) each uniform inside a random prefix."""
    pick = rng.below(n, len(nets))
    l = lens[pick]
    host = (rng.u64(n) & np.uint64(0xFFFFFFFF))
    hm = np.where(l < 32, (np.uint64(1) << (32 - l).astype(np.uint64)) - np.uint64(1), np.uint64(0))
    return (nets[pick].astype(np.uint64) | (host & hm)).astype(np.uint32)


def addrs_in_prefixes_v6(rng, n, nets, lens):
    pick = rng.below(n, len(nets))
    rnd = rng.bytes(n * 16).reshape(n, 16)
    out = nets[pick].copy()
    l = lens[pick]
    for i in range(16):
        bits = np.clip(l - 8 * i, 0, 8)
        m = ((0xFF00 >> bits) & 0xFF).astype(np.uint8)
        out[:, i] = (out[:, i] & m) | (rnd[:, i] & ~m)
    return out


# ------------------------------------------------------------------ netflow v9 / ipfix
def v9_header(count, source_id, seq=0, uptime=1000, secs=1700000000):
    return struct.pack(">HHIIII", 9, count, uptime, secs, seq, source_id)


def v9_template_flowset(template_id, fields):
    body = struct.pack(">HH", template_id, len(fields)) + b"".join(struct.pack(">HH", t, l) for t, l in fields)
    return struct.pack(">HH", 0, 4 + len(body)) + body


def v9_template_packet(template_id, fields, source_id):
    return v9_header(1, source_id) + v9_template_flowset(template_id, fields)


def ipfix_header(length, domain, seq=0, secs=1700000000):
    return struct.pack(">HHIII", 10, length, secs, seq, domain)


def ipfix_template_set(template_id, fields):
    """fields: (id, len) or (id, len, enterprise_number)."""
    body = struct.pack(">HH", template_id, len(fields))
    for f in fields:
        if len(f) == 3:
            body += struct.pack(">HHI", f[0] | 0x8000, f[1], f[2])
        else:
            body += struct.pack(">HH", f[0], f[1])
    return struct.pack(">HH", 2, 4 + len(body)) + body


def ipfix_template_packet(template_id, fields, domain):
    s = ipfix_template_set(template_id, fields)
    return ipfix_header(16 + len(s), domain) + s


def _records_matrix(columns):
    """columns: list of [n, w] uint8 arrays -> [n, sum w]."""
    return np.concatenate(columns, axis=1)


def flow_columns(rng, n, fields, src4=None, dst4=None, src6=None, dst6=None, with_as=False):
    """Random record bytes for a template given as (id, len) list."""
    cols = []
    for fid, flen in fields:
        if fid == 8:
            v = src4 if src4 is not None else (rng.u64(n) & np.uint64(0xFFFFFFFF))
            cols.append(_be(v, 4))
        elif fid == 12:
            v = dst4 if dst4 is not None else (rng.u64(n) & np.uint64(0xFFFFFFFF))
            cols.append(_be(v, 4))
        elif fid == 27:
            cols.append(src6 if src6 is not None else rng.bytes(n * 16).reshape(n, 16))
        elif fid == 28:
            cols.append(dst6 if dst6 is not None else rng.bytes(n * 16).reshape(n, 16))
        elif fid in (7, 11):
            # a quarter of the ports are well-known ones so port filters hit
            p = rng.below(n, 65536)
            wk = np.array([53, 80, 123, 443, 8080, 22, 25, 0], dtype=np.int64)[rng.below(n, 8)]
            p = np.where(rng.below(n, 4) == 0, wk, p)
            cols.append(_be(p, flen))
        elif fid == 4:
            r = rng.below(n, 100)
            cols.append(_be(np.where(r < 70, 6, np.where(r < 95, 17, 1)), flen))
        elif fid == 6:
            cols.append(_be(rng.below(n, 64), flen))
        elif fid == 2:
            cols.append(_be(1 + rng.below(n, 64), flen))
        elif fid == 1:
            cols.append(_be((40 + rng.below(n, 1461)) * (1 + rng.below(n, 64)), flen))
        elif fid in (10, 14):
            cols.append(_be(1 + rng.below(n, 48), flen))
        elif fid in (16, 17):
            cols.append(_be(np.where(rng.below(n, 4) == 0, 64512 + rng.below(n, 1023), rng.below(n, 400000)), flen))
        else:
            cols.append(rng.bytes(n * flen).reshape(n, flen))
    return _records_matrix(cols)


def v9_data_packets(template_id, rec_bytes, per_packet, source_id):
    """Split [n, reclen] record bytes into v9 datagrams with one data set each,
    header count == records in the datagram."""
    n, rl = rec_bytes.shape
    pkts = []
    for s in range(0, n, per_packet):
        chunk = rec_bytes[s:s + per_packet]
        c = chunk.shape[0]
        pkts.append(v9_header(c, source_id, seq=s) + struct.pack(">HH", template_id, 4 + c * rl) + chunk.tobytes())
    return pkts


def ipfix_data_packets(template_id, rec_bytes, per_packet, domain):
    n, rl = rec_bytes.shape
    pkts = []
    for s in range(0, n, per_packet):
        chunk = rec_bytes[s:s + per_packet]
        c = chunk.shape[0]
        pkts.append(ipfix_header(16 + 4 + c * rl, domain, seq=s) + struct.pack(">HH", template_id, 4 + c * rl)
                    + chunk.tobytes())
    return pkts


def v5_packet(rng, nflows, engine_id=0):
    hdr = struct.pack(">HHIIIIBBH", 5, nflows, 1000, 1700000000, 0, 0, 0, engine_id, 0)
    return hdr + rng.bytes(48 * nflows).tobytes()


class Batch:
    """A list of datagrams with their exporter address, in arrival order."""

    def __init__(self):
        self.pkts = []      # bytes
        self.src = []       # 4-byte exporter IPv4 (network order bytes)
        self.kind = []      # 0 = netflow (v5/v9/ipfix), 1 = sflow

    def add(self, pkt, src, kind=0):
        self.pkts.append(bytes(pkt))
        self.src.append(bytes(src))
        self.kind.append(kind)

    def extend(self, pkts, src, kind=0):
        for p in pkts:
            self.add(p, src, kind)

    def __len__(self):
        return len(self.pkts)

    def packed(self):
        """(bytes uint8[total], offsets uint64[n+1], src uint32[n] (network-order
        bytes reinterpreted little-endian), kind uint8[n])"""
        lens = np.fromiter((len(p) for p in self.pkts), dtype=np.uint64, count=len(self.pkts))
        offs = np.zeros(len(self.pkts) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offs[1:])
        data = np.frombuffer(b"".join(self.pkts), dtype=np.uint8).copy()
        src = np.frombuffer(b"".join(self.src), dtype="<u4").copy()
        return data, offs, src, np.array(self.kind, dtype=np.uint8)


# ------------------------------------------------------------------ workloads
MYNET_C1 = ["10.0.0.0/8", "10.1.0.0/16", "172.16.0.0/12", "192.168.0.0/16", "192.168.10.0/24",
            "100.64.0.0/10", "198.18.0.0/15", "203.0.113.0/24", "198.51.100.0/24", "192.0.2.0/24",
            "45.10.0.0/14", "45.11.128.0/17", "77.88.0.0/18", "91.200.16.0/22", "185.10.20.0/23",
            "8.8.8.0/24"]


def _parse_v4_prefixes(lst):
    nets, lens = [], []
    for s in lst:
        a, l = s.split("/")
        o = [int(x) for x in a.split(".")]
        nets.append((o[0] << 24) | (o[1] << 16) | (o[2] << 8) | o[3])
        lens.append(int(l))
    return np.array(nets, dtype=np.uint32), np.array(lens, dtype=np.int64)


def workload_c1(nrec, seed=SEED_C1, per_packet=64):
    """Netflow v9, one exporter 192.0.2.1 / source id 77, template 256 of 7
    fields (21 B/record), 64 records per datagram, 1 object `src net mynet`."""
    rng = Rng(seed)
    nets, lens = _parse_v4_prefixes(MYNET_C1)
    inside = addrs_in_prefixes_v4(rng, nrec, nets, lens)
    anyaddr = (rng.u64(nrec) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    src = np.where(rng.below(nrec, 2) == 0, inside, anyaddr)
    rec = flow_columns(rng, nrec, TMPL_V9_C1, src4=src)
    b = Batch()
    exporter = bytes([192, 0, 2, 1])
    b.add(v9_template_packet(256, TMPL_V9_C1, 77), exporter)
    b.extend(v9_data_packets(256, rec, per_packet, 77), exporter)
    iplists = {"mynet": "\n".join(MYNET_C1) + "\n"}
    mos = [("src net mynet", -1)]
    return dict(batch=b, iplists=iplists, mos=mos, devices=[], nrec=nrec)


def workload_c2(nrec, seed=SEED_C2, n_exporters=64, n_lists=8, prefixes=10000, exporter_net=255):
    """IPFIX, 64 exporters 10.<exporter_net>.0.1-64 (observation domain = index), 70 % of
    the data sets use the IPv4 template 256 (46 B/record, 30 per datagram) and
    30 % the IPv6 template 257 (62 B/record, 22 per datagram).  16 objects:
    8 x `net listK` + 8 x `net6 listK`; every list holds `prefixes` IPv4
    (/12-/28, 10 % nested) and `prefixes` IPv6 (/29-/64) prefixes; a sixth of
    the addresses of each side is drawn from inside some list, the rest is
    uniform (the short prefixes alone cover a good part of the IPv4 space, so
    about three quarters of the records match some object)."""
    rng = Rng(seed)
    lists4, lists6, iplists = [], [], {}
    for k in range(n_lists):
        lr = Rng(seed ^ (0x1000 + k))
        v4 = prefix_list_v4(lr, prefixes, np.arange(12, 29), nested_frac=0.1)
        v6 = prefix_list_v6(lr, prefixes)
        lists4.append(v4)
        lists6.append(v6)
        iplists["list%d" % k] = iplist_text(v4, v6)
    all4 = (np.concatenate([a for a, _ in lists4]), np.concatenate([l for _, l in lists4]))
    all6 = (np.concatenate([a for a, _ in lists6]), np.concatenate([l for _, l in lists6]))

    n4 = int(nrec * 0.7) // 30 * 30
    n6 = (nrec - n4) // 22 * 22
    b = Batch()
    per_exp = []
    for e in range(n_exporters):
        per_exp.append(bytes([10, exporter_net & 255, (e + 1) >> 8, (e + 1) & 255]))
        b.add(ipfix_template_packet(256, TMPL_IPFIX_V4, e + 1), per_exp[e])
        b.add(ipfix_template_packet(257, TMPL_IPFIX_V6, e + 1), per_exp[e])

    def mix4(n):
        # the list prefixes cover well under 1 % of the space, so uniform
        # addresses practically never match: ~1/6 of each side inside gives
        # about 30 % of the records a hit on src or dst
        inside = addrs_in_prefixes_v4(rng, n, *all4)
        rnd = (rng.u64(n) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        return np.where(rng.below(n, 6) == 0, inside, rnd)

    def mix6(n):
        inside = addrs_in_prefixes_v6(rng, n, *all6)
        rnd = rng.bytes(n * 16).reshape(n, 16)
        rnd[:, 0] = 0x20 | (rnd[:, 0] & 0x1F)
        sel = (rng.below(n, 6) == 0)[:, None]
        return np.where(sel, inside, rnd)

    rec4 = flow_columns(rng, n4, TMPL_IPFIX_V4, src4=mix4(n4), dst4=mix4(n4))
    rec6 = flow_columns(rng, n6, TMPL_IPFIX_V6, src6=mix6(n6), dst6=mix6(n6))
    p4 = [rec4[i:i + 30] for i in range(0, n4, 30)]
    p6 = [rec6[i:i + 22] for i in range(0, n6, 22)]
    # interleave the two kinds of datagram deterministically (7:3) and deal
    # them round-robin over the exporters
    order = []
    i4 = i6 = 0
    pattern = [0, 0, 1, 0, 0, 1, 0, 0, 1, 0]
    k = 0
    while i4 < len(p4) or i6 < len(p6):
        want6 = pattern[k % 10] == 1
        k += 1
        if (want6 and i6 < len(p6)) or i4 >= len(p4):
            order.append((1, i6)); i6 += 1
        else:
            order.append((0, i4)); i4 += 1
    for j, (is6, idx) in enumerate(order):
        e = j % n_exporters
        if is6:
            chunk = p6[idx]
            tid = 257
        else:
            chunk = p4[idx]
            tid = 256
        body = chunk.tobytes()
        pkt = ipfix_header(16 + 4 + len(body), e + 1, seq=j) + struct.pack(">HH", tid, 4 + len(body)) + body
        b.add(pkt, per_exp[e])
    mos = [("net list%d" % k, -1) for k in range(n_lists)] + [("net6 list%d" % k, -1) for k in range(n_lists)]
    return dict(batch=b, iplists=iplists, mos=mos, devices=[], nrec=n4 + n6)


def bgp_like_lengths():
    """candidate IPv4 prefix lengths weighted like a full routing table:
    about 60 % /24, 10 % /23, 10 % /22, the rest /8-/21 and a few /25-/32"""
    lens = [24] * 60 + [23] * 10 + [22] * 10
    lens += [8, 10, 11, 12, 13, 14, 15, 16, 16, 17, 18, 19, 19, 20, 20, 21, 21]
    lens += [25, 28, 32]
    return np.array(lens)


def workload_c4(nrec, seed=SEED_C4, n_lists=8, prefixes=1000000, variants=32, per_packet=50):
    """Netflow v9 records of the C1 shape plus src/dst AS (29 B/record) against
    n_lists x variants objects over `n_lists` big IPv4 lists with a routing
    table like length histogram (5 % more-specifics).  About a third of the
    source addresses and a third of the destinations fall inside some list."""
    rng = Rng(seed)
    tmpl = TMPL_V9_C1 + [(16, 4), (17, 4)]
    lists, iplists = [], {}
    for k in range(n_lists):
        lr = Rng(seed ^ (0x2000 + k))
        v4 = prefix_list_v4(lr, prefixes, bgp_like_lengths(), nested_frac=0.05)
        lists.append(v4)
        iplists["big%d" % k] = iplist_text(v4, None)
    all4 = (np.concatenate([a for a, _ in lists]), np.concatenate([l for _, l in lists]))

    def mix(n):
        inside = addrs_in_prefixes_v4(rng, n, *all4)
        rnd = (rng.u64(n) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        return np.where(rng.below(n, 3) == 0, inside, rnd)

    rec = flow_columns(rng, nrec, tmpl, src4=mix(nrec), dst4=mix(nrec), with_as=True)
    b = Batch()
    exporter = bytes([192, 0, 2, 4])
    b.add(v9_template_packet(256, tmpl, 4), exporter)
    b.extend(v9_data_packets(256, rec, per_packet, 4), exporter)
    mos = []
    for k in range(n_lists):
        for v in range(variants):
            kind = v % 4
            if kind == 0:
                f = "src net big%d and (dst port 80 or 443 or %d-%d) and src as 64512-65534" % (k, 8000 + v, 8999 + v)
            elif kind == 1:
                f = "dst net big%d and not proto %d" % (k, 17 if v < 16 else 6)
            elif kind == 2:
                f = "net big%d and dst port %d-%d" % (k, 1000 * v, 1000 * v + 999)
            else:
                f = "src net big%d and dst as %d-%d" % (k, 1000 * v, 1000 * v + 20000)
            mos.append((f, -1))
    return dict(batch=b, iplists=iplists, mos=mos, devices=[], nrec=nrec)


def workload_c5(nrec, seed=SEED_C5, n_exporters=512, prefixes=10000, n_lists=8, rank=0, world=1, owner=None):
    """Sharded replay: `n_exporters` exporters 10.5.x.y, the first half Netflow v9
    with the 21-byte template of workload_c1 (source id = index), the second
    half IPFIX with the IPv4 template of workload_c2 (observation domain =
    index); the 16 objects of workload_c2 (same lists) plus `src net mynet`.
    Every exporter's records come from its own stream (seed ^ index), so a
    rank can generate just its share: with `world` > 1 only the exporters with
    owner(address array, world) == rank are produced (default owner:
    shard.exporter_rank).  Datagrams of the exporters are interleaved round
    robin, templates first."""
    from . import shard
    if owner is None:
        owner = shard.exporter_rank
    lists4, iplists = [], {}
    for k in range(n_lists):
        lr = Rng(SEED_C2 ^ (0x1000 + k))
        v4 = prefix_list_v4(lr, prefixes, np.arange(12, 29), nested_frac=0.1)
        v6 = prefix_list_v6(lr, prefixes)
        lists4.append(v4)
        iplists["list%d" % k] = iplist_text(v4, v6)
    iplists["mynet"] = "\n".join(MYNET_C1) + "\n"
    all4 = (np.concatenate([a for a, _ in lists4]), np.concatenate([l for _, l in lists4]))
    mynets, mylens = _parse_v4_prefixes(MYNET_C1)

    addrs = [bytes([10, 5, (e + 1) >> 8, (e + 1) & 255]) for e in range(n_exporters)]
    own = owner(np.frombuffer(b"".join(addrs), dtype="<u4"), world) if world > 1 else np.zeros(n_exporters, dtype=np.int64)
    per_exp = max(1, nrec // n_exporters)
    streams, head, total = [], Batch(), 0
    for e in range(n_exporters):
        if own[e] != rank:
            continue
        rng = Rng(seed ^ (e + 1))
        if e < n_exporters // 2:
            n = max(64, per_exp // 64 * 64)
            inside = addrs_in_prefixes_v4(rng, n, mynets, mylens)
            rnd = (rng.u64(n) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            rec = flow_columns(rng, n, TMPL_V9_C1, src4=np.where(rng.below(n, 2) == 0, inside, rnd))
            head.add(v9_template_packet(256, TMPL_V9_C1, e + 1), addrs[e])
            pk = v9_data_packets(256, rec, 64, e + 1)
        else:
            n = max(30, per_exp // 30 * 30)

            def mix4(m):
                inside = addrs_in_prefixes_v4(rng, m, *all4)
                rnd = (rng.u64(m) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
                return np.where(rng.below(m, 6) == 0, inside, rnd)
            rec = flow_columns(rng, n, TMPL_IPFIX_V4, src4=mix4(n), dst4=mix4(n))
            head.add(ipfix_template_packet(256, TMPL_IPFIX_V4, e + 1), addrs[e])
            pk = ipfix_data_packets(256, rec, 30, e + 1)
        total += n
        streams.append((addrs[e], pk))
    b = head
    depth = max((len(pk) for _, pk in streams), default=0)
    for j in range(depth):
        for a, pk in streams:
            if j < len(pk):
                b.add(pk[j], a)
    mos = [("net list%d" % k, -1) for k in range(n_lists)] + [("net6 list%d" % k, -1) for k in range(n_lists)]
    mos.append(("src net mynet", -1))
    return dict(batch=b, iplists=iplists, mos=mos, devices=[], nrec=total)


# ------------------------------------------------------------------ sflow v5
def _eth(rng, ethertype, vlan=None, qinq=None):
    h = rng.bytes(12).tobytes()
    if qinq is not None:
        h += struct.pack(">HH", 0x88a8, qinq)
    if vlan is not None:
        h += struct.pack(">HH", 0x8100, vlan)
    return h + struct.pack(">H", ethertype)


def _ipv4(rng, proto, src=None, dst=None, frag=0, ihl=5, total=100):
    s = src if src is not None else rng.bytes(4).tobytes()
    d = dst if dst is not None else rng.bytes(4).tobytes()
    h = struct.pack(">BBHHHBBH", 0x40 | ihl, int(rng.below(1, 256)[0]), total, int(rng.below(1, 65536)[0]),
                    frag, int(1 + rng.below(1, 255)[0]), proto, 0) + s + d
    return h + b"\0" * (4 * (ihl - 5))


def _ipv6(rng, nxt, src=None, dst=None):
    s = src if src is not None else rng.bytes(16).tobytes()
    d = dst if dst is not None else rng.bytes(16).tobytes()
    return struct.pack(">IHBB", 0x60000000, 60, nxt, int(1 + rng.below(1, 255)[0])) + s + d


def _tcp(rng, sport=None, dport=None, flags=None):
    sp = int(rng.below(1, 65536)[0]) if sport is None else sport
    dp = int(rng.below(1, 65536)[0]) if dport is None else dport
    fl = int(rng.below(1, 64)[0]) if flags is None else flags
    return struct.pack(">HHIIBBHHH", sp, dp, 1, 2, 0x50, fl, 1000, 0, 0)


def _udp(rng, sport=None, dport=None):
    sp = int(rng.below(1, 65536)[0]) if sport is None else sport
    dp = int(rng.below(1, 65536)[0]) if dport is None else dport
    return struct.pack(">HHHH", sp, dp, 20, 0)


def _icmp(rng):
    return struct.pack(">BBHI", int(rng.below(1, 20)[0]), 0, 0, 0)


def sflow_header_sample(rng, kind, vlan_mode):
    """One sampled frame header (bytes) of the C3 mix."""
    vlan = qinq = None
    if vlan_mode == 1:
        vlan = int(rng.below(1, 4096)[0]) | (int(rng.below(1, 8)[0]) << 13)
    elif vlan_mode == 2:
        qinq = int(rng.below(1, 4096)[0])
        vlan = int(rng.below(1, 4096)[0])
    if kind == "v4tcp":
        dport = 443 if int(rng.below(1, 3)[0]) == 0 else None
        f = _eth(rng, 0x0800, vlan, qinq) + _ipv4(rng, 6) + _tcp(rng, dport=dport)
    elif kind == "v4udp":
        f = _eth(rng, 0x0800, vlan, qinq) + _ipv4(rng, 17) + _udp(rng)
    elif kind == "v6tcp":
        f = _eth(rng, 0x86dd, vlan, qinq) + _ipv6(rng, 6) + _tcp(rng)
    elif kind == "v6udp":
        f = _eth(rng, 0x86dd, vlan, qinq) + _ipv6(rng, 17) + _udp(rng)
    elif kind == "v4icmp":
        f = _eth(rng, 0x0800, vlan, qinq) + _ipv4(rng, 1) + _icmp(rng)
    elif kind == "frag":
        f = _eth(rng, 0x0800, vlan, qinq) + _ipv4(rng, 17, frag=0x2000 | int(rng.below(1, 100)[0])) + _udp(rng)
    elif kind == "ipip":
        f = _eth(rng, 0x0800, vlan, qinq) + _ipv4(rng, 4) + _ipv4(rng, 6) + _tcp(rng)
    elif kind == "6in6":
        f = _eth(rng, 0x86dd, vlan, qinq) + _ipv6(rng, 41) + _ipv6(rng, 17) + _udp(rng)
    elif kind == "4in6":
        f = _eth(rng, 0x86dd, vlan, qinq) + _ipv6(rng, 4) + _ipv4(rng, 6) + _tcp(rng)
    elif kind == "arp":
        f = _eth(rng, 0x0806, vlan, qinq) + rng.bytes(28).tobytes()
    else:
        raise ValueError(kind)
    return f


def sflow_flow_sample(seq, rate, in_if, out_if, elements, expanded=False, source_id=1):
    """elements: list of (tag, body bytes)."""
    body = struct.pack(">I", seq)
    body += struct.pack(">II", 0, source_id) if expanded else struct.pack(">I", source_id)
    body += struct.pack(">III", rate, 1000, 0)
    if expanded:
        body += struct.pack(">IIII", 0, in_if, 0, out_if)
    else:
        body += struct.pack(">II", in_if, out_if)
    body += struct.pack(">I", len(elements))
    for tag, eb in elements:
        body += struct.pack(">II", tag, len(eb)) + eb
    return struct.pack(">II", 3 if expanded else 1, len(body)) + body


def sflow_raw_element(header, frame_len, header_len=None, proto=1):
    hl = len(header) if header_len is None else header_len
    pad = (-len(header)) % 4
    return (1, struct.pack(">IIII", proto, frame_len, 0, hl) + header + b"\0" * pad)


def sflow_ipv4_element(rng, proto=6):
    return (3, struct.pack(">II", 100, proto) + rng.bytes(4).tobytes() + rng.bytes(4).tobytes()
            + struct.pack(">IIII", int(rng.below(1, 65536)[0]), int(rng.below(1, 65536)[0]),
                          int(rng.below(1, 64)[0]), int(rng.below(1, 256)[0])))


def sflow_datagram(agent4, agent_id, samples, seq=0, agent6=None):
    if agent6 is not None:
        hdr = struct.pack(">II", 5, 2) + bytes(agent6)
    else:
        hdr = struct.pack(">II", 5, 1) + bytes(agent4)
    hdr += struct.pack(">IIII", agent_id, seq, 12345, len(samples))
    return hdr + b"".join(samples)


C3_MIX = (["v4tcp"] * 60 + ["v4udp"] * 15 + ["v6tcp"] * 10 + ["v4icmp"] * 5 + ["frag"] * 5 + ["arp"] * 5)


def workload_c3(nsamples, seed=SEED_C3, n_agents=32, per_packet=7):
    """sFlow v5, 32 agents, 7 compact flow samples per datagram, one raw
    Ethernet header element of 128 bytes each."""
    rng = Rng(seed)
    b = Batch()
    kinds = rng.below(nsamples, len(C3_MIX))
    vl = rng.below(nsamples, 100)
    j = 0
    for s in range(0, nsamples, per_packet):
        agent = (s // per_packet) % n_agents
        samples = []
        for i in range(s, min(s + per_packet, nsamples)):
            vm = 2 if vl[i] < 5 else (1 if vl[i] < 55 else 0)
            hdr = sflow_header_sample(rng, C3_MIX[kinds[i]], vm)
            hdr = (hdr + rng.bytes(128).tobytes())[:128]
            el = sflow_raw_element(hdr, int(64 + rng.below(1, 1455)[0]))
            samples.append(sflow_flow_sample(i, 512, 1 + i % 48, 1 + (i * 7) % 48, [el]))
        src = bytes([10, 254, 0, agent + 1])
        b.add(sflow_datagram(src, agent + 1, samples, seq=j), src, kind=1)
        j += 1
    lr = Rng(seed ^ 0x1000)
    v6 = prefix_list_v6(lr, 1000)
    iplists = {"list0": iplist_text(prefix_list_v4(lr, 1000, np.arange(8, 25)), v6)}
    mos = [("vlan 100-199", -1), ("proto 6 and dst port 443", -1), ("net6 list0", -1),
           ("mac 00:11:22:33:44:55", -1), ("proto 17", -1), ("net list0", -1),
           ("tcp-flags 512", -1), ("not proto 6", -1)]
    return dict(batch=b, iplists=iplists, mos=mos, devices=[], nrec=nsamples)


# ---------------------------------------------------------------- capture files
def udp_frame(payload, src, dst=b"\x0a\x00\x00\x01", sport=40000, dport=2055, ip_options=b"",
              ip_len=None, udp_len=None, proto=17, trailer=b""):
    """Ethernet + IPv4 + UDP around one datagram (checksums left zero: the
    collector does not look at them)."""
    ihl = 5 + len(ip_options) // 4
    if ip_len is None:
        ip_len = 4 * ihl + 8 + len(payload)
    if udp_len is None:
        udp_len = 8 + len(payload)
    eth = b"\x02\x00\x00\x00\x00\x01" + b"\x02\x00\x00\x00\x00\x02" + b"\x08\x00"
    ip = struct.pack(">BBHHHBBH", 0x40 | ihl, 0, ip_len & 0xFFFF, 0, 0, 64, proto, 0) + bytes(src) + bytes(dst) + ip_options
    udp = struct.pack(">HHHH", sport, dport, udp_len & 0xFFFF, 0)
    return eth + ip + udp + bytes(payload) + trailer


def write_pcap(path, frames, byte_order="<", nanosecond=False, linktype=1, snaplen=262144):
    """classic pcap file out of raw frames (bytes, or (bytes, captured_length))"""
    magic = 0xa1b23c4d if nanosecond else 0xa1b2c3d4
    with open(path, "wb") as f:
        f.write(struct.pack(byte_order + "IHHiIII", magic, 2, 4, 0, 0, snaplen, linktype))
        for i, fr in enumerate(frames):
            cap = None
            if isinstance(fr, tuple):
                fr, cap = fr
            cap = len(fr) if cap is None else cap
            f.write(struct.pack(byte_order + "IIII", 1700000000 + i // 1000, i % 1000, cap, len(fr)))
            f.write(fr[:cap])


def batch_to_pcap(path, batch, dport=2055, **kw):
    write_pcap(path, [udp_frame(p, s, dport=dport) for p, s in zip(batch.pkts, batch.src)], **kw)


def write_workload_dir(w, path, max_datagrams=None):
    """A workload as files, for programs that are not Python (tools/shimbench/shim_bench.c):
    batch.bin = u64 n, u64 offsets[n + 1], u32 src[n], u8 kind[n], the datagram bytes;
    mos.txt = one object per line (parent index, tab, filter); iplists/<name>.  Returns the
    number of datagrams written (the first max_datagrams of the batch: templates come first)."""
    import os
    os.makedirs(os.path.join(path, "iplists"), exist_ok=True)
    for n, t in (w.get("iplists") or {}).items():
        with open(os.path.join(path, "iplists", n), "w") as f:
            f.write(t)
    with open(os.path.join(path, "mos.txt"), "w") as f:
        for filt, parent in w["mos"]:
            f.write("%d\t%s\n" % (parent, filt))
    data, offs, src, kinds = w["batch"].packed()[:4]
    n = len(src) if max_datagrams is None else min(len(src), int(max_datagrams))
    with open(os.path.join(path, "batch.bin"), "wb") as f:
        f.write(np.uint64(n).tobytes())
        f.write(np.asarray(offs[:n + 1], dtype=np.uint64).tobytes())
        f.write(np.asarray(src[:n], dtype=np.uint32).tobytes())
        f.write(np.asarray(kinds[:n], dtype=np.uint8).tobytes())
        f.write(np.asarray(data[:int(offs[n])], dtype=np.uint8).tobytes())
    return n
