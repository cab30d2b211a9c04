"""Capture-file replay (xe_pcap_*): frames are unwrapped by the rules of the
reference's pcap_packet() (pcapture.c:111-180).  Host code only."""
import struct

import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth


def _read_all(path, **kw):
    r = xenoeye_amd.PcapReader(path, **kw)
    out = []
    for data, offs, src, _ in r:
        for i in range(len(src)):
            out.append((bytes(data[int(offs[i]):int(offs[i + 1])]), src[i:i + 1].tobytes()))
    st = r.stats()
    r.close()
    return out, st


@pytest.mark.parametrize("order,nano", [("<", False), (">", False), ("<", True), (">", True)])
def test_round_trip_of_a_synthetic_capture(tmp_path, order, nano):
    w = synth.workload_c1(2000)
    path = str(tmp_path / "c.pcap")
    synth.batch_to_pcap(path, w["batch"], byte_order=order, nanosecond=nano)
    got, st = _read_all(path)
    assert [g[0] for g in got] == w["batch"].pkts
    assert [g[1] for g in got] == w["batch"].src
    assert st["taken"] == len(w["batch"].pkts) == st["frames"] and st["ignored"] == 0 and st["damaged"] == 0


def test_batches_split_by_bytes_and_count_keep_every_datagram(tmp_path):
    w = synth.workload_c1(3000)
    path = str(tmp_path / "c.pcap")
    synth.batch_to_pcap(path, w["batch"])
    total = sum(len(p) for p in w["batch"].pkts)
    biggest = max(len(p) for p in w["batch"].pkts)
    for kw in (dict(max_bytes=biggest), dict(max_bytes=3 * biggest + 7), dict(max_datagrams=1),
               dict(max_datagrams=7), dict(max_bytes=total, max_datagrams=len(w["batch"].pkts))):
        got, _ = _read_all(path, **kw)
        assert [g[0] for g in got] == w["batch"].pkts, kw


def test_frame_rules_of_the_reference(tmp_path):
    src = b"\xc0\x00\x02\x07"
    pay = bytes(range(40))
    frames = [
        synth.udp_frame(pay, src),                                    # 0 taken
        synth.udp_frame(pay[:20], src),                               # caplen 62 < 64: ignored
        synth.udp_frame(pay, src, proto=6),                           # not UDP: ignored
        synth.udp_frame(pay, src, ip_options=b"\x01" * 8),            # 1 taken, IHL 7
        synth.udp_frame(pay, src, trailer=b"\xEE" * 10),              # 2 taken: Ethernet padding is not payload
        synth.udp_frame(pay, src, udp_len=30),                        # 3 taken: 30 bytes (the field is compared as it stands)
        synth.udp_frame(pay, src, ip_len=20 + 8 + 24),                # 4 taken: 24 bytes
        synth.udp_frame(pay, src, ip_len=20 + 8),                     # length 0: damaged
        synth.udp_frame(pay, src, ip_len=20 + 4),                     # negative: damaged
        synth.udp_frame(pay, src, ip_len=20 + 8 + 400, udp_len=408),  # beyond the capture: damaged
        (synth.udp_frame(pay + bytes(100), src), 14 + 20 + 8 + 60),   # cut by the snap length: damaged
        synth.udp_frame(pay, src, dport=6343),                        # 5 taken (no port selection)
    ]
    # IHL below 5
    bad = bytearray(synth.udp_frame(pay, src))
    bad[14] = 0x44
    frames.append(bytes(bad))                                         # ignored
    path = str(tmp_path / "r.pcap")
    synth.write_pcap(path, frames)
    got, st = _read_all(path)
    assert [g[0] for g in got] == [pay, pay, pay, pay[:30], pay[:24], pay]
    assert all(g[1] == src for g in got)
    assert st == dict(frames=13, taken=6, ignored=3, damaged=4)

    got, st = _read_all(path, udp_port=2055)
    assert len(got) == 5 and st["ignored"] == 4
    got, st = _read_all(path, udp_port=6343)
    assert [g[0] for g in got] == [pay]


def test_truncated_and_foreign_files(tmp_path):
    w = synth.workload_c1(500)
    path = str(tmp_path / "c.pcap")
    synth.batch_to_pcap(path, w["batch"])
    raw = open(path, "rb").read()
    cut = str(tmp_path / "cut.pcap")
    open(cut, "wb").write(raw[:len(raw) - 17])
    got, st = _read_all(cut)
    assert [g[0] for g in got] == w["batch"].pkts[:-1] and st["damaged"] == 1

    for name, content, code in (
            ("empty.pcap", b"", -4),
            ("text.pcap", b"this is not a capture file at all........", -4),
            ("ng.pcap", struct.pack("<III", 0x0A0D0D0A, 28, 0x1A2B3C4D) + bytes(16), -5),
            ("sll.pcap", struct.pack("<IHHiIII", 0xa1b2c3d4, 2, 4, 0, 0, 65535, 113), -5)):
        p = str(tmp_path / name)
        open(p, "wb").write(content)
        with pytest.raises(xenoeye_amd.XeError) as e:
            xenoeye_amd.PcapReader(p)
        assert e.value.code == code, name
        assert name.split(".")[0] in str(e.value) or "pcap" in str(e.value)
    with pytest.raises(xenoeye_amd.XeError) as e:
        xenoeye_amd.PcapReader(str(tmp_path / "missing.pcap"))
    assert e.value.code == -7

    # a record header that announces more bytes than any frame has
    broken = str(tmp_path / "broken.pcap")
    open(broken, "wb").write(raw[:24] + struct.pack("<IIII", 0, 0, 0x7FFFFFFF, 0x7FFFFFFF) + bytes(64))
    r = xenoeye_amd.PcapReader(broken)
    with pytest.raises(xenoeye_amd.XeError) as e:
        r.read()
    assert e.value.code == -4

    # a datagram larger than the caller's buffer
    r = xenoeye_amd.PcapReader(path, max_bytes=16)
    with pytest.raises(xenoeye_amd.XeError) as e:
        r.read()
    assert e.value.code == -1


def test_replayed_batch_is_what_the_oracle_decodes(tmp_path):
    """the datagrams a capture yields, fed to the CPU oracle, give the records
    of the original batch"""
    from oracle_binding import Oracle
    w = synth.workload_c1(1500)
    path = str(tmp_path / "c.pcap")
    synth.batch_to_pcap(path, w["batch"])

    def run(pairs):
        orc = Oracle()
        for name, text in w["iplists"].items():
            orc.iplist_add(name, text)
        for filt, parent in w["mos"]:
            orc.mo_add(filt, parent)
        orc.log_config(keep_flows=True)
        for p, s in pairs:
            orc.netflow(p, s)
        return orc.masks().copy(), orc.flows().copy(), orc.counters().copy()

    a = run(zip(w["batch"].pkts, w["batch"].src))
    got, _ = _read_all(path)
    b = run(got)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_damaged_captures_under_sanitizers(tmp_path):
    """the reader, built with -fsanitize=address,undefined, over captures with
    random bytes and length fields changed or cut short: no invalid access"""
    import os
    import random
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    exe = str(tmp_path / "pcap_drv")
    cmd = ["gcc", "-std=gnu11", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(root, "xenoeye_amd", "csrc"), os.path.join(here, "pcap_reader_driver.c"),
           os.path.join(root, "xenoeye_amd", "csrc", "xe_pcap.c"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "san" in (r.stderr or "").lower():
        pytest.skip("no sanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    w = synth.workload_c2(2000, prefixes=50)
    base = str(tmp_path / "base.pcap")
    synth.batch_to_pcap(base, w["batch"])
    raw = open(base, "rb").read()
    rnd = random.Random(5)
    files = [base]
    for i in range(80):
        b = bytearray(raw[:rnd.randrange(24, len(raw))] if i % 3 == 0 else raw)
        for _ in range(rnd.randrange(1, 40)):
            b[rnd.randrange(0, len(b))] = rnd.randrange(256)
        if i % 5 == 0:
            off = 24
            for _ in range(rnd.randrange(1, 20)):
                if off + 16 > len(b):
                    break
                cap = struct.unpack_from("<I", b, off + 8)[0]
                if rnd.random() < 0.3:
                    struct.pack_into("<I", b, off + 8, rnd.choice([0, 1, 63, 64, 65, cap + 1, max(cap, 1) - 1, 262144,
                                                                      262145, 0xFFFFFFFF]))
                    break
                off += 16 + cap
        path = str(tmp_path / ("f%03d.pcap" % i))
        open(path, "wb").write(bytes(b))
        files.append(path)
    r = subprocess.run([exe] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert r.stdout.startswith("datagrams ")
