"""Replay of a capture file through the device path.

    python -m xenoeye_amd.replay --pcap flows.pcap --mo-dir /var/lib/xenoeye/mo \\
        [--iplists DIR] [--devices devices.conf] [--port 2055] [--sflow]

reads the frames the way the reference's pcap capture does (pcapture.c:111-180),
runs every batch on the GPU and prints per monitoring object the records,
octets and packets it received (what monit_object_process_nf() would have been
handed, monit-objects.c:923-1023) and, per fixed-window table of mo.conf, the
number of rows (with --rows N also the first rows in the order of the
reference's window export, monit-objects-fwm.c:496-589).
"""
import argparse
import ctypes as C
import json
import sys
import time

import numpy as np

from . import Engine, PcapReader, XeError


def replay(eng, path, udp_port=0, sflow=False, batch_bytes=64 << 20, batch_datagrams=1 << 18, on_batch=None):
    """Every datagram of the capture `path` through the committed engine `eng`.
    Returns (records, counters u64[nmo][3], reader statistics).  on_batch, if
    given, is called with every PreparedBatch after its run."""
    L = eng.L
    reader = PcapReader(path, udp_port=udp_port, max_bytes=batch_bytes, max_datagrams=batch_datagrams)
    # datagrams are appended straight into page-locked memory
    p = L.xe_gpu_host_alloc(eng._h, batch_bytes)
    if not p:
        raise XeError(-2, "xe_gpu_host_alloc(%d) failed" % batch_bytes)
    data = np.ctypeslib.as_array((C.c_uint8 * batch_bytes).from_address(p))
    offs = np.zeros(batch_datagrams + 1, dtype=np.uint64)
    src = np.zeros(batch_datagrams, dtype=np.uint32)
    kinds = np.full(batch_datagrams, 1 if sflow else 0, dtype=np.uint8)
    records, counters = 0, None
    try:
        while True:
            n = L.xe_pcap_read(reader._h, reader.udp_port, p, batch_bytes, offs.ctypes.data, src.ctypes.data,
                               batch_datagrams)
            if n < 0:
                raise XeError(n, L.xe_pcap_error(reader._h).decode(errors="replace"))
            if n == 0:
                break
            b = eng.prepare((data[:int(offs[n])], offs[:n + 1], src[:n], kinds[:n]))
            b.run()
            c = b.counters()
            counters = c.copy() if counters is None else counters + c
            records += b.records
            if on_batch is not None:
                on_batch(b)
            b.free()
        stats = reader.stats()
    finally:
        reader.close()
        L.xe_gpu_host_free(eng._h, p)
    if counters is None:
        counters = np.zeros((eng.mo_count(), 3), dtype=np.uint64)
    return records, counters, stats


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--pcap", required=True)
    ap.add_argument("--mo-dir", required=True)
    ap.add_argument("--iplists")
    ap.add_argument("--devices")
    ap.add_argument("--port", type=int, default=0, help="UDP destination port (0 = every UDP datagram)")
    ap.add_argument("--sflow", action="store_true", help="the datagrams are sFlow v5")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--rows", type=int, default=0,
                    help="print this many rows of every fixed-window table, in the order and with the "
                         "limit of the reference's window export (keys in hex)")
    a = ap.parse_args(argv)

    eng = Engine(device=a.device)
    if a.iplists:
        eng.load_iplists(a.iplists)
    if a.devices:
        eng.load_devices(a.devices)
    ntab, skipped = eng.load_mos_with_tables(a.mo_dir)
    eng.commit()
    t0 = time.perf_counter()
    records, counters, stats = replay(eng, a.pcap, udp_port=a.port, sflow=a.sflow)
    dt = time.perf_counter() - t0
    out = dict(stats, records=records, seconds=round(dt, 3), tables=ntab, table_entries_skipped=skipped, objects=[])
    for i in range(eng.mo_count()):
        out["objects"].append(dict(name=eng.mo_name(i), records=int(counters[i, 0]), octets=int(counters[i, 1]),
                                   packets=int(counters[i, 2])))
    out["fixed_window"] = []
    for t, (mo, name) in enumerate(eng.fwm_tables()):
        rows, dropped = eng.fwm_read(t)
        entry = dict(object=eng.mo_name(mo), name=name, rows=len(rows), dropped=dropped)
        if a.rows > 0:
            first, others, hit = eng.fwm_export(t)      # the entry's own "limit"
            entry["first_rows"] = [dict(key=k.hex(), sums=list(v)) for k, v in first[:a.rows]]
            if hit:
                entry["others"] = list(others)
        out["fixed_window"].append(entry)
    json.dump(out, sys.stdout, indent=1)
    sys.stdout.write("\n")
    eng.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
