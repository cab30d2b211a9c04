#!/usr/bin/env python3
"""Flow-path benchmark: device-resident decode + classify throughput.

One step = one pass of the HIP path (counter reset + decode/classify kernels)
over one batch of synthetic export packets that already sits in HBM.  The
headline workload (C2) is IPFIX from 64 exporters, 70 % IPv4 records (46 B) /
30 % IPv6 records (62 B), 16 monitoring objects over 8 IP lists of 10 000 IPv4
+ 10 000 IPv6 prefixes each.  With --gpus N every rank gets its own exporters
(weak scaling, no data-path collective); rank 0 prints one JSON line.

Besides the throughput the line carries
  roofline      algorithmic bytes per launch / mean kernel time (HIP events on
                the launch stream) against the 8 TB/s HBM peak, the copy
                bandwidth measured in this run as a second denominator, and
                for the list-heavy workloads the measured random-gather rate;
  workloads     the same measurement for the other configurations (C1, C3,
                C4 at 8 x 1 M prefixes, C5) from the same run;
  cpu_baseline  the CPU oracle (a scalar restatement of the reference's
                algorithm) on a bounded sample of the same workload, one
                thread; cpu_baseline_nproc: one process per host core, each
                with its own exporters (the reference's thread-per-capture
                model);
  sustained     the headline step repeated for a few seconds (the K timed
                steps last milliseconds).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0
KERNELS = ["xe_nf_fast_kernel", "xe_nf_general_kernel", "xe_sf_kernel"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--records", type=int, default=1 << 22, help="records per GPU per step")
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c4", "c5", "c2m"])
    ap.add_argument("--c4-prefixes", type=int, default=1000000, help="prefixes per list of the c4 workload")
    ap.add_argument("--c4-lists", type=int, default=8, help="distinct lists of the c4 workload (objects = 256)")
    ap.add_argument("--cpu-sample", type=int, default=8000000, help="records timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="processes of the all-cores CPU baseline (-1: the box's CPU share, 0 = skip)")
    ap.add_argument("--replicate", type=int, default=1, help="repeat the generated datagrams this many times")
    ap.add_argument("--others", type=int, default=1, help="also measure the other configurations (0 = skip)")
    ap.add_argument("--other-steps", type=int, default=100)
    ap.add_argument("--sustain-seconds", type=float, default=3.0, help="repeat the headline step this long (0 = skip)")
    ap.add_argument("--pipeline", type=int, default=4,
                    help="capture threads (one handle each) of the sustained host-inclusive measurement, 0/1 = skip")
    ap.add_argument("--pipeline-rounds", type=int, default=20)
    ap.add_argument("--pipeline-walkers", type=int, default=0,
                    help="header-walking threads per capture thread of the pipelined measurement (0: 16 / threads)")
    ap.add_argument("--shim-records", type=int, default=1 << 20,
                    help="records of the workload sent through the drop-in library's entry points (0 = skip)")
    ap.add_argument("--fwm", type=int, default=1, help="also time the fixed-window table kernel (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (gloo: rehearsal of the multi-rank path on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--traffic", type=float, default=None,
                    help="HBM bytes per launch from a rocprofv3 --pmc pass (reported as roofline.traffic)")
    ap.add_argument("--root-bits", type=int, default=0, choices=[0, 16, 24],
                    help="width of the list tables' roots (0 = the library's own choice per table)")
    ap.add_argument("--walk-reps", type=int, default=4,
                    help="prepare() calls the device header walk is timed over (0 = skip)")
    ap.add_argument("--spawn-check", action="store_true",
                    help="ranks only join the process group, add up their ranks and report (no GPU work): "
                         "rehearsal of the --gpus N launch path")
    ap.add_argument("--overlap-leg", type=int, default=1,
                    help="time one host thread with consecutive batches overlapped on one handle (0 = skip)")
    ap.add_argument("--quick", action="store_true",
                    help="the headline measurement only (no CPU legs, other configurations, sustained / pipelined / table legs)")
    a = ap.parse_args()
    if a.quick:
        a.cpu_sample = 0
        a.others = 0
        a.sustain_seconds = 0.0
        a.pipeline = 0
        a.fwm = 0
        a.shim_records = 0
        a.overlap_leg = 0
    return a


def build_workload(name, nrec, rank, c4_prefixes=1000000, world=1, c4_lists=8):
    from xenoeye_amd import synth
    if name == "c5":
        # one replay of 512 exporters cut by exporter hash: this rank generates its share only
        return synth.workload_c5(nrec * world, rank=rank, world=world), \
            "C5 sharded replay, 512 exporters (256 netflow-v9 21B + 256 ipfix 46B records), 17 objects, %d ranks" % world
    if name == "c4":
        return synth.workload_c4(nrec, seed=synth.SEED_C4 ^ (rank << 32), prefixes=c4_prefixes,
                                 n_lists=c4_lists, variants=256 // c4_lists), \
            "C4 netflow-v9 29B records, 256 objects = %d lists x %d prefixes x %d variants (net + port/AS/proto)" % (
                c4_lists, c4_prefixes, 256 // c4_lists)
    if name == "c3":
        return synth.workload_c3(nrec, seed=synth.SEED_C3 ^ (rank << 32)), \
            "C3 sflow-v5 32 agents 7 samples/datagram 128B raw headers 8 objects"
    if name == "c1":
        return synth.workload_c1(nrec, seed=synth.SEED_C1 ^ (rank << 32)), \
            "C1 netflow-v9 1 exporter 21B records 1 object (src net mynet)"
    # every rank owns its own 64 exporters (10.255.0.x on rank 0, 10.254.0.x on rank 1 ...)
    w = synth.workload_c2(nrec, seed=synth.SEED_C2 ^ (rank << 32), exporter_net=255 - rank)
    if name == "c2m":
        # the same traffic with a devices.conf entry per exporter: a sampling rate, two mark filters
        # (device_rules_check(), devices.c:286-314), and an object that reads the mark
        w["devices"] = [dict(ip="10.%d.%d.%d" % ((255 - rank) & 255, (e + 1) >> 8, (e + 1) & 255), sampling_rate=100 + e,
                             marks=("proto 6", "dst port 0-1023")) for e in range(64)]
        w["mos"] = list(w["mos"]) + [("dev-mark 2", -1)]
        return w, "C2 + devices.conf: 64 entries with a sampling rate and two mark filters each, 17 objects (one reads the mark)"
    return w, "C2 ipfix 64 exporters 70% v4(46B)/30% v6(62B) 16 objects x 10k+10k-prefix lists"


def replicate(batch, times):
    if times <= 1:
        return
    n0 = len(batch.pkts)
    for _ in range(times - 1):
        batch.pkts.extend(batch.pkts[:n0])
        batch.src.extend(batch.src[:n0])
        batch.kind.extend(batch.kind[:n0])


ROOT_BITS = 0


def make_engine(w, device):
    import xenoeye_amd
    eng = xenoeye_amd.Engine(device=device)
    if ROOT_BITS:
        eng.set_list_root_bits(ROOT_BITS)
    for name, text in w["iplists"].items():
        eng.add_iplist(name, text)
    for filt, parent in w["mos"]:
        eng.add_mo("mo", filt, parent)
    for d in w.get("devices", ()):
        eng.add_device(**d)
    eng.commit()
    return eng


def oracle_pass(w, sample, part=None):
    """Time the oracle on about `sample` records: whole passes over the batch's
    datagrams (templates stay learnt between passes, as on a live collector).
    part = (index, of): only the datagrams of the exporters of that share."""
    from oracle_binding import Oracle
    orc = Oracle()
    for name, text in w["iplists"].items():
        orc.iplist_add(name, text)
    for filt, parent in w["mos"]:
        orc.mo_add(filt, parent)
    orc.log_config(keep_flows=False)
    pkts, srcs = w["batch"].pkts, [int.from_bytes(s, "little") for s in w["batch"].src]
    kinds = w["batch"].kind
    if part is not None:
        from xenoeye_amd import shard
        import numpy as np
        own = shard.exporter_rank(np.array(srcs, dtype=np.uint64), part[1])
        keep = [i for i in range(len(pkts)) if own[i] == part[0]]
        pkts, srcs, kinds = [pkts[i] for i in keep], [srcs[i] for i in keep], [kinds[i] for i in keep]
    if not pkts:
        return 0.0, 0, 0, 0.0
    per_dgram = max(1.0, w["nrec"] / max(1, len(w["batch"].pkts)))
    todo = int(sample / per_dgram) + 1
    L = orc.L
    nd = 0
    nrec = 0
    t0 = time.perf_counter()
    while nd < todo:
        k = min(len(pkts), todo - nd)
        for p, s, kd in zip(pkts[:k], srcs[:k], kinds[:k]):
            if kd == 1:
                L.xo_sflow_process(p, len(p), s)
            else:
                L.xo_netflow_process(p, len(p), s)
        nd += k
        nrec += orc.nrecords()
        L.xo_log_clear()
    dt = time.perf_counter() - t0
    orc.reset()
    return nrec / dt, nrec, nd, dt


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """the CPU share of this job: the box exports it through the thread-count variables"""
    for v in ("OMP_NUM_THREADS", "MAX_JOBS"):
        try:
            n = int(os.environ.get(v, ""))
            if n > 0:
                return min(n, os.cpu_count() or n)
        except ValueError:
            pass
    return min(16, os.cpu_count() or 1)


def cpu_baseline_nproc(w, sample, procs):
    """one process per core, each with the datagrams of its own exporters (the
    reference scales by pinning exporters to capture threads): records of all
    processes / wall time from the common start to the last finish.  Must run
    before this process touches the GPU (the workers are forked)."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    barrier = ctx.Barrier(procs)
    q = ctx.Queue()

    def run(i):
        try:
            barrier.wait(timeout=120)
            t0 = time.time()
            _, nrec, _, _ = oracle_pass(w, sample // procs, (i, procs))
            q.put((nrec, t0, time.time()))
        except Exception:
            q.put((0, time.time(), time.time()))

    ps = [ctx.Process(target=run, args=(i,)) for i in range(procs)]
    for p in ps:
        p.start()
    res = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join()
    nrec = sum(r[0] for r in res)
    wall = max(r[2] for r in res) - min(r[1] for r in res)
    busy = sum(1 for r in res if r[0])
    return nrec / wall if wall > 0 else 0.0, nrec, wall, busy


def measured_traffic(workload, nrec):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3
    --pmc passes (profiles/traffic.json), when they were taken at this size"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f).get(workload)
        if t and int(t["records"]) == int(nrec):
            return float(t["bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def copy_bandwidth(torch, device, nbytes=1 << 30, reps=10):
    """device-to-device copy of 1 GiB, read + write bytes per second: what the
    memory system of this box delivers to a plain streaming kernel"""
    a = torch.empty(nbytes, dtype=torch.uint8, device=device)
    b = torch.empty(nbytes, dtype=torch.uint8, device=device)
    a.zero_()
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def gather_rate(torch, device, table_bytes, n=1 << 24, reps=5):
    """independent random 4-byte gathers per second from a table as large as the
    workload's list tables: the ceiling of one table visit per record, if no
    visit had to wait for the one before"""
    words = max(1024, int(table_bytes) // 4)
    table = torch.arange(words, dtype=torch.int32, device=device)
    idx = torch.randint(0, words, (n,), dtype=torch.int64, device=device)
    out = torch.empty(n, dtype=torch.int32, device=device)
    for _ in range(2):
        torch.index_select(table, 0, idx, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.index_select(table, 0, idx, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del table, idx, out
    return n / (ms * 1e-3)


def time_steps(torch, batch, stream, steps, warmup, barrier=None):
    """(seconds of exactly `steps` steps between two synchronisations, mean step ms by events,
    mean ms of each kernel)"""
    import numpy as np
    for _ in range(warmup):
        batch.run(stream)
    (barrier or torch.cuda.synchronize)()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(steps, 64))]
    t0 = time.perf_counter()
    for i in range(steps):
        if i < len(evs):
            evs[i][0].record()
        batch.run(stream)
        if i < len(evs):
            evs[i][1].record()
    (barrier or torch.cuda.synchronize)()
    dt = time.perf_counter() - t0
    step_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    # the kernels alone: separate launches outside the timed region, events of the library on the launch stream
    kms = []
    for _ in range(min(steps, 20)):
        batch.run(stream)
        kms.append(batch.kernel_ms())
    kms = np.array(kms, dtype=np.float64).mean(axis=0)
    return dt, step_ms, kms


def fresh_batch_kernel_ms(eng, packed, stream, cycles=4):
    """the dominant kernel's time on a batch that has just been prepared: the timed steps run one batch
    again and again, whose list tables then sit in the L2 and whose datagrams partly in the Infinity
    Cache; here every run follows the copy and the clearing of a new batch's buffers"""
    import numpy as np
    ms = []
    for _ in range(cycles):
        b = eng.prepare(packed)
        b.timing(True)
        b.run(stream)
        ms.append(float(np.max(b.kernel_ms())))
        b.free()
    return float(np.mean(ms))


def roofline_of(batch, kms, workload, nrec, traffic=None):
    import numpy as np
    kidx = int(np.argmax(kms))
    kernel_ms = float(kms[kidx])
    alg = batch.algorithmic_bytes
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic if traffic is not None else measured_traffic(workload, nrec),
            "kernel": KERNELS[kidx], "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_record": alg / max(nrec, 1)}


def measure_other(torch, name, args, device, stream):
    """one of the other configurations, device-resident, same method as the headline"""
    t0 = time.perf_counter()
    # the sFlow generator builds every sampled header one by one: a smaller set of
    # distinct datagrams, repeated (every copy is decoded and classified like the first)
    nrec = {"c1": 1 << 22, "c2": 1 << 22, "c2m": 1 << 22, "c3": 200000, "c4": 1 << 21, "c5": 1 << 22}[name]
    rep = {"c3": 16}.get(name, 1)
    w, wl_name = build_workload(name, nrec, 0, args.c4_prefixes, 1, args.c4_lists)
    replicate(w["batch"], rep)
    eng = make_engine(w, device)
    packed = w["batch"].packed()
    eng.prepare(packed).free()     # the batch that brings the templates, untimed
    batch = eng.prepare(packed)
    batch.timing(True)
    dt, step_ms, kms = time_steps(torch, batch, stream, args.other_steps, 3)
    n = batch.records
    walk_ms = measure_walk(eng, packed, args.walk_reps)
    out = {"workload": wl_name, "records_per_step": n, "steps": args.other_steps,
           "headers_walked_on_device": dict(zip(("netflow_datagrams", "netflow_records", "sflow_records"), batch.device_walked())),
           "ms_per_step": dt / args.other_steps * 1e3, "records_per_s": n * args.other_steps / dt,
           "payload_GBps": batch.payload_bytes * args.other_steps / dt / 1e9,
           "roofline": roofline_of(batch, kms, name, n)}
    out["per_lane_kernel_records"] = batch.per_lane_records
    fresh = fresh_batch_kernel_ms(eng, packed, stream, 3)
    out["roofline"]["kernel_ms_fresh_batch"] = fresh
    out["roofline"]["frac_fresh_batch"] = out["roofline"]["frac"] * out["roofline"]["kernel_ms"] / fresh
    out["device_path"] = device_path(out["roofline"], walk_ms, n)
    if name in ("c4", "c2", "c5"):
        out["list_walks"] = list_walks(torch, batch, stream, w, device, out["roofline"]["kernel_ms"])
    if rep > 1:
        out["distinct_records"] = n // rep
    batch.free()
    eng.close()
    out["setup_s"] = time.perf_counter() - t0 - dt
    return out


def list_walks(torch, batch, stream, w, device, kernel_ms):
    """the IP-list walks of one launch against what the memory system gives random gathers: loads
    counted by the kernel itself (one more run with its counters on), the rate of independent 4-byte
    gathers from a table as large as the workload's list tables measured now, and the share of the
    kernel's time those gathers alone would take at that rate"""
    tb = list_table_bytes(w)
    g = gather_rate(torch, device, tb)
    root, deep = batch.count_gathers(stream)
    n = max(batch.records, 1)
    return {"table_bytes": tb, "random_gathers_per_s": g, "root_loads_per_record": root / n,
            "node_and_chain_loads_per_record": deep / n,
            "gathers_per_s_in_kernel": deep / (kernel_ms * 1e-3),
            "frac_of_gather_bound": (deep / g) / (kernel_ms * 1e-3) if g else None,
            "note": "root tables (256 KB each) stay in the L2; node and chain-record loads are the gathers that leave it"}


def list_table_bytes(w):
    """size of the flattened list tables of a workload (the host-side flattening alone, no device)"""
    try:
        from host_binding import Host
        h = Host()
        for name, text in w["iplists"].items():
            h.add_iplist(name, text)
        words, _ = h.flatten()
        h.close()
        return int(len(words)) * 4
    except Exception:
        return 0


def measure_walk(eng, packed, reps):
    """mean device time (ms) of the header walk of a prepare() over `packed`: every launch that
    turns datagram bytes resident in HBM into the run / tile tables (Netflow) or the per-datagram
    sample tables (sFlow) the decode kernels start from; HIP events on the walk's stream"""
    if reps <= 0:
        return None
    eng.set_walk_timing(True)
    ms = []
    try:
        # (the first prepare with the events on creates them and measures that too: 0.53 against
        # 0.25 ms on C3 - not part of the mean)
        eng.prepare(packed).free()
        for _ in range(reps):
            b = eng.prepare(packed)
            t = b.walk_ms()
            b.free()
            if t >= 0:
                ms.append(t)
    finally:
        eng.set_walk_timing(False)
    return float(sum(ms) / len(ms)) if ms else None


def device_path(roof, walk_ms, nrec):
    """the whole device path of one batch: header walk + dominant decode / classify kernel"""
    if walk_ms is None:
        return None
    ms = walk_ms + roof["kernel_ms"]
    ach = roof["algorithmic_bytes_per_launch"] / (ms * 1e-3) / 1e9
    out = {"walk_ms": walk_ms, "kernel_ms": roof["kernel_ms"], "device_path_ms": ms,
           "records_per_s_device_path": nrec / (ms * 1e-3), "achieved": ach, "unit": "GB/s",
           "frac": ach / HBM_PEAK_GBPS}
    if roof.get("kernel_ms_fresh_batch"):
        # a batch the GPU sees for the first time: the walk is measured that way already
        out["device_path_ms_fresh_batch"] = walk_ms + roof["kernel_ms_fresh_batch"]
    return out


def launch_ranks(n):
    """`bench.py --gpus N` from a plain shell: start one child process per GPU (the reference
    starts one worker thread per capture entry, xenoeye.c:584-625), each with the rank variables
    torch.distributed reads; rank 0's line is this process's output.  The parent never touches
    the GPU and replaces no process image."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out.decode(errors="replace"))
    sys.stdout.flush()
    return rc


def spawn_check(args):
    """every rank joins the group and adds up the ranks: what --gpus N starts, without GPU work"""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group(backend="gloo")
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
        total = int(t.item())
    else:
        total = 1
    if rank == 0:
        print(json.dumps({"spawn_check": True, "n_gpus": world, "sum_of_ranks_plus_1": total,
                          "asked": args.gpus}), flush=True)


def c5_sharded(torch, dist, args, rank, world, device, stream, dev):
    """the 512-exporter replay cut by exporter hash over the ranks: every rank generates and runs its
    share, the per-object counters merge by wrapping addition, and rank 0 checks the merged table
    against its own run of the whole replay (same seeds)"""
    from xenoeye_amd import shard, synth
    total = 1 << 21
    w = synth.workload_c5(total, rank=rank, world=world)
    eng = make_engine(w, device)
    packed = w["batch"].packed()
    eng.prepare(packed).free()
    batch = eng.prepare(packed)
    batch.timing(True)
    steps = max(10, args.other_steps // 2)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()
    dt, step_ms, kms = time_steps(torch, batch, stream, steps, 3, barrier)
    ctr = shard.merge_counters(batch.counters(), dist, dev)
    dt_all, tot = shard.reduce_scalars([dt, batch.records], ["max", "sum"], dist, dev)
    out = None
    if rank == 0:
        whole = synth.workload_c5(total)
        e1 = make_engine(whole, device)
        p1 = whole["batch"].packed()
        e1.prepare(p1).free()
        b1 = e1.prepare(p1)
        b1.run(stream)
        c1 = b1.counters()
        out = {"workload": "C5 sharded replay, 512 exporters cut by exporter hash over %d ranks" % world,
               "records_all_ranks_per_step": int(tot), "records_one_rank_whole_replay": int(b1.records),
               "steps": steps, "ms_per_step": dt_all / steps * 1e3, "records_per_s": tot * steps / dt_all,
               "merged_counters_equal_one_rank": bool(int(tot) == int(b1.records) and (ctr == c1).all())}
        b1.free()
        e1.close()
    batch.free()
    eng.close()
    return out


def shim_leg(w, records, passes=3):
    """datagrams through the reference's entry points of the drop-in library (netflow_process /
    sflow_process -> monit_object_process_nf) on one capture thread: tools/shimbench/shim_bench,
    a program of its own, run while this process has not touched the GPU.  A bounded sample of
    the workload: the first datagrams, about `records` records of them, `passes` times."""
    import shutil
    import subprocess
    import tempfile
    from xenoeye_amd import synth
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "shimbench", "shim_bench")
    if not os.path.exists(exe):
        return {"error": "tools/shimbench/shim_bench is not built (__graft_entry__.build())"}
    if w.get("devices"):
        return {"error": "the program loads no devices.conf"}
    nd = len(w["batch"].pkts)
    nd = max(1, min(nd, int(nd * min(1.0, records / max(1, w["nrec"])))))
    d = tempfile.mkdtemp(prefix="xe_shimbench_")
    try:
        nd = synth.write_workload_dir(w, d, nd)
        r = subprocess.run([exe, d, str(passes), "1"], capture_output=True, text=True, timeout=600)
        if r.returncode != 0 or not r.stdout.strip():
            return {"error": (r.stderr or "no output").strip()[-400:]}
        out = json.loads(r.stdout.strip().splitlines()[-1])
        out["unit"] = "records/s through netflow_process()/sflow_process() and the exit calls, one capture thread"
        out["sample"] = "%d datagrams of the workload, %d passes after one warm-up pass" % (nd, passes)
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    if args.spawn_check:
        spawn_check(args)
        return
    global ROOT_BITS
    ROOT_BITS = args.root_bits
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    t_start = time.perf_counter()
    w, wl_name = build_workload(args.workload, args.records, rank, args.c4_prefixes, world, args.c4_lists)

    # the CPU legs come first: the all-cores one forks its workers, which a process may only do
    # while it has not touched the GPU.  Rank 0 of a one-GPU run only.
    cpu = {}
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        rps, n_cpu, nd, secs = oracle_pass(w, args.cpu_sample)
        cpu["cpu_baseline"] = {"value": rps, "unit": "records/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
                               "sample": "%d datagrams (%d records) of the same batch, %.1f s, "
                                         "oracle/xe_oracle.c single thread" % (nd, n_cpu, secs)}
        procs = host_cores() if args.cpu_procs < 0 else args.cpu_procs
        if procs > 1:
            try:
                rps_n, n_all, wall, busy = cpu_baseline_nproc(w, args.cpu_sample * min(procs, 4), procs)
                cpu["cpu_baseline_nproc"] = {
                    "value": rps_n, "unit": "records/s", "cores": procs, "kind": "port", "cpu": cpu_model(),
                    "sample": "%d records in %.1f s wall: %d processes, the exporters dealt to them by address "
                              "hash (%d had exporters), each its own oracle" % (n_all, wall, procs, busy)}
            except Exception as ex:
                cpu["cpu_baseline_nproc"] = {"error": repr(ex)}
    if rank == 0 and world == 1 and args.shim_records > 0:
        try:
            cpu["drop_in"] = shim_leg(w, args.shim_records)
        except Exception as ex:
            cpu["drop_in"] = {"error": repr(ex)}

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the flow path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    import xenoeye_amd
    eng = make_engine(w, local_rank)
    replicate(w["batch"], args.replicate)
    packed = w["batch"].packed()
    # the first batch of a handle brings the templates (its exporters are walked by the host) and
    # sizes the buffers: untimed, as on a collector that has been running for a minute
    eng.prepare(packed).free()
    t_prep0 = time.perf_counter()
    batch = eng.prepare(packed)
    t_prep = time.perf_counter() - t_prep0
    walked = dict(zip(("netflow_datagrams", "netflow_records", "sflow_records"), batch.device_walked()))
    nrec = batch.records
    stream = torch.cuda.current_stream().cuda_stream
    batch.timing(True)     # HIP events on the launch stream around each kernel

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dt, step_ms, kms = time_steps(torch, batch, stream, args.steps, args.warmup, barrier)
    # the header walk that built this batch's run / tile tables, from datagram bytes resident in HBM
    walk_ms = measure_walk(eng, packed, args.walk_reps) if rank == 0 else None
    fresh_ms = fresh_batch_kernel_ms(eng, packed, stream) if rank == 0 else None

    # the headline step repeated for seconds: the K timed steps above last milliseconds, which no
    # activity sampler sees; same batch, same launches
    sustained = None
    if args.sustain_seconds > 0 and world == 1:
        per = max(dt / args.steps, 1e-5)
        n_s = max(args.steps, int(args.sustain_seconds / per))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_s):
            batch.run(stream)
            if (i & 1023) == 1023:
                torch.cuda.synchronize()      # bounded queue depth
        torch.cuda.synchronize()
        ds = time.perf_counter() - t1
        sustained = {"steps": n_s, "seconds": ds, "records_per_s": nrec * n_s / ds, "ms_per_step": ds / n_s * 1e3}

    # end-to-end including host pre-pass + H2D + D2H of masks and counters (one pass, reported only)
    t_e2e = None
    for _ in range(3):           # the fastest of three passes: one pass alone is at the mercy of the box
        t1 = time.perf_counter()
        b2 = eng.prepare(packed)
        b2.run(stream)
        m = b2.masks()
        c = b2.counters()
        t_ = time.perf_counter() - t1
        t_e2e = t_ if t_e2e is None else min(t_e2e, t_)
        b2.free()
        del m
    # the same from a page-locked datagram buffer (xe_gpu_host_alloc), as a capture thread would fill it
    # and into a page-locked result buffer that the thread keeps between batches
    pinned = (eng.pinned(packed[0]),) + tuple(packed[1:])
    mbuf = eng.pinned_array((batch.records * batch.mask_words,), np.uint64)
    def pinned_pass():
        best = None
        for _ in range(3):
            t1 = time.perf_counter()
            b3 = eng.prepare(pinned)
            b3.run(stream)
            m = b3.masks(out=mbuf)
            c3 = b3.counters()
            t_ = time.perf_counter() - t1
            best = t_ if best is None else min(best, t_)
            assert (c3 == c).all()
            b3.free()
            del m
        return best
    t_e2e_pinned = pinned_pass()
    # the same with one host thread for whatever header walk is left to the host
    eng.set_prepass_threads(1)
    t_e2e_pinned_1 = pinned_pass()
    eng.set_prepass_threads(0)

    # ONE host thread, ONE handle, consecutive batches overlapped: the bytes of batch k + 1 travel while
    # batch k's kernel runs and its masks come back (two page-locked input buffers, as a capture thread
    # that fills one while the other is on the link would hold them)
    overlapped = None
    if world == 1 and args.overlap_leg:
        try:
            eng.set_overlap(True)
            ins = [pinned, (eng.pinned(packed[0]),) + tuple(packed[1:])]
            mb2 = [mbuf, eng.pinned_array((batch.records * batch.mask_words,), np.uint64)]
            rounds = max(8, 2 * args.pipeline_rounds)

            def overlapped_pass(rounds):
                prev = None
                done = 0
                t1 = time.perf_counter()
                for k in range(rounds):
                    cur = eng.prepare(ins[k & 1])
                    cur.run(stream)
                    if prev is not None:
                        prev.masks(out=mb2[(k - 1) & 1])
                        prev.counters()
                        done += prev.records
                        prev.free()
                    prev = cur
                prev.masks(out=mb2[(rounds - 1) & 1])
                cl = prev.counters()
                done += prev.records
                prev.free()
                return time.perf_counter() - t1, done, cl

            overlapped_pass(2)
            dt_o, done_o, c_o = overlapped_pass(rounds)
            assert (c_o == c).all()
            overlapped = {"rounds": rounds, "records": int(done_o), "seconds": dt_o, "records_per_s": done_o / dt_o,
                          "ms_per_batch": dt_o / rounds * 1e3}
            # one step further: the bytes of batch k + 1 are sent before batch k is prepared, the walk
            # reads the device copy (three input buffers; nothing but uploads and results on the link)
            ins.append((eng.pinned(packed[0]),) + tuple(packed[1:]))

            def ahead_pass(rounds):
                prev = None
                done = 0
                t1 = time.perf_counter()
                up = eng.upload_begin(ins[0][0])
                for k in range(rounds):
                    nxt = eng.upload_begin(ins[(k + 1) % 3][0]) if k + 1 < rounds else None
                    cur = eng.prepare(ins[k % 3], upload=up)
                    up = nxt
                    cur.run(stream)
                    if prev is not None:
                        prev.masks(out=mb2[(k - 1) & 1])
                        prev.counters()
                        done += prev.records
                        prev.free()
                    prev = cur
                prev.masks(out=mb2[(rounds - 1) & 1])
                cl = prev.counters()
                done += prev.records
                prev.free()
                return time.perf_counter() - t1, done, cl

            ahead_pass(3)
            dt_a, done_a, c_a = ahead_pass(rounds)
            assert (c_a == c).all()
            overlapped["uploads_sent_ahead"] = {"rounds": rounds, "records": int(done_a), "seconds": dt_a,
                                                "records_per_s": done_a / dt_a, "ms_per_batch": dt_a / rounds * 1e3}
        except Exception as ex:
            overlapped = {"error": repr(ex)}
        finally:
            eng.set_overlap(False)

    # sustained host-inclusive rate: `args.pipeline` capture threads, each with a handle of its own
    # (the arrangement a collector with several capture threads would use), take turns on this GPU: while one walks headers on the host,
    # another copies or runs.  Every thread owns a share of the exporters and a page-locked buffer.
    piped = None
    if args.pipeline > 1 and world == 1:
        try:
            import threading
            from xenoeye_amd import shard as _sh
            nthr, rounds = args.pipeline, args.pipeline_rounds
            engines, inputs = [], []
            for t in range(nthr):
                e = xenoeye_amd.Engine(device=local_rank)
                if ROOT_BITS:
                    e.set_list_root_bits(ROOT_BITS)
                for name, text in w["iplists"].items():
                    e.add_iplist(name, text)
                for filt, parent in w["mos"]:
                    e.add_mo("mo", filt, parent)
                e.set_prepass_threads(args.pipeline_walkers if args.pipeline_walkers > 0 else max(1, 16 // nthr))
                e.commit()
                part = _sh.select(packed, t, nthr)
                inputs.append((e.pinned(part[0]),) + tuple(part[1:]))
                engines.append(e)
            done = [0] * nthr
            mbufs = [None] * nthr

            def worker(t):
                st = torch.cuda.Stream(device=local_rank)
                # capture threads do not run in lock step: started together they would all copy
                # at once and then all fetch at once, leaving the link idle in between
                time.sleep(t * t_e2e_pinned / nthr / nthr)
                for _ in range(rounds):
                    bb = engines[t].prepare(inputs[t])
                    bb.run(st.cuda_stream)
                    bb.masks(out=mbufs[t])
                    bb.counters()
                    done[t] += bb.records
                    bb.free()

            for t in range(nthr):          # first batch of every handle untimed (templates, buffers)
                bb = engines[t].prepare(inputs[t]); bb.run(stream); bb.counters()
                mbufs[t] = engines[t].pinned_array((bb.records * bb.mask_words,), np.uint64)
                bb.free()
            ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthr)]
            t1 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dt_p = time.perf_counter() - t1
            piped = {"threads": nthr, "rounds": rounds, "records": int(sum(done)), "seconds": dt_p,
                     "records_per_s": sum(done) / dt_p}
            for e in engines:
                e.close()

        except Exception as ex:   # an optional measurement must not lose the benchmark line
            piped = {"error": repr(ex)}
    # fixed-window tables (the consumer of the match masks, monit-objects.c:923-1023) on a handle of
    # their own: the same batch with two tables, timed with and without the table kernel
    fixed_window = None
    if args.fwm and world == 1 and args.workload in ("c1", "c2", "c5"):
        try:
            e = xenoeye_amd.Engine(device=local_rank)
            for name, text in w["iplists"].items():
                e.add_iplist(name, text)
            for filt, parent in w["mos"]:
                e.add_mo("mo", filt, parent)
            allmo = e.add_mo("mo", "", -1)
            t0_ = e.add_fwm(allmo, ["proto", "dst port", "octets", "packets"], 1 << 19)
            t1_ = e.add_fwm(0, ["src host", "dst host", "bits desc"], 1 << 21)
            e.commit()
            fb = e.prepare(pinned)
            fb.timing(True)
            fb.run(stream)
            torch.cuda.synchronize()
            rows0, dr0 = e.fwm_read(t0_)
            rows1, dr1 = e.fwm_read(t1_)
            ms, ms_later = [], []
            for _ in range(5):
                e.fwm_clear(t0_)
                e.fwm_clear(t1_)
                fb.run(stream)
                ms.append(fb.fwm_ms())          # first batch of a window: every key is new
                fb.run(stream)
                ms_later.append(fb.fwm_ms())    # later batches: the keys are in the table
            fms = float(np.mean(ms_later))
            fms_first = float(np.mean(ms))
            cc = fb.counters()
            # bytes the table kernel must move: per record its mask word and rate, per (record, table
            # that takes it) the key fields and 8 bytes per summed field, read and added
            alg_fwm = fb.records * (8 * fb.mask_words + 4) + int(cc[allmo, 0]) * (1 + 2 + 2 * 8 + 2 * 8) \
                + int(cc[0, 0]) * (4 + 4 + 8 + 8)
            fixed_window = {"tables": [{"object": "all records", "fields": ["proto", "dst port", "octets", "packets"],
                                        "rows": len(rows0), "dropped": dr0},
                                       {"object": "object 0", "fields": ["src host", "dst host", "bits desc"],
                                        "rows": len(rows1), "dropped": dr1}],
                            "kernel": "xe_fwm_kernel", "kernel_ms": fms, "kernel_ms_first_batch_of_window": fms_first,
                            "records_per_s": fb.records / (fms * 1e-3),
                            "algorithmic_bytes_per_launch": alg_fwm,
                            "achieved_GBps": alg_fwm / (fms * 1e-3) / 1e9, "frac_of_hbm_peak": alg_fwm / (fms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            e.close()

        except Exception as ex:
            fixed_window = {"error": repr(ex)}

    from xenoeye_amd import shard
    dev = torch.device("cuda", local_rank) if args.backend == "nccl" else None
    dt_all, tot, pay = shard.reduce_scalars([dt, nrec, batch.payload_bytes], ["max", "sum", "sum"],
                                            dist if world > 1 else None, dev)
    # per-object counters of the shards merge by wrapping u64 addition
    ctr = shard.merge_counters(c, dist if world > 1 else None, dev)
    value = tot * args.steps / dt_all
    c5n = None
    if world > 1 and args.others:
        try:
            c5n = c5_sharded(torch, dist, args, rank, world, local_rank, stream, dev)
        except Exception as ex:
            c5n = {"error": repr(ex)}

    if rank == 0:
        roof = roofline_of(batch, kms, args.workload, nrec, args.traffic)
        roof["step_ms_events"] = step_ms
        roof["kernel_ms_fresh_batch"] = fresh_ms
        roof["frac_fresh_batch"] = roof["frac"] * roof["kernel_ms"] / fresh_ms if fresh_ms else None
        dpath = device_path(roof, walk_ms, nrec)
        # second denominator: what a plain device copy reaches on this box, measured now
        try:
            roof["peak_measured"] = copy_bandwidth(torch, torch.device("cuda", local_rank))
            roof["frac_of_peak_measured"] = roof["achieved"] / roof["peak_measured"]
        except Exception as ex:
            roof["peak_measured"] = None
            roof["peak_measured_error"] = repr(ex)
        if args.workload in ("c2", "c4", "c5"):
            try:
                roof["list_walks"] = list_walks(torch, batch, stream, w, torch.device("cuda", local_rank), roof["kernel_ms"])
            except Exception as ex:
                roof["list_walks"] = {"error": repr(ex)}
        out = {
            "metric": "device-resident flow records/sec decoded+classified",
            "value": value,
            "unit": "records/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_all / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": wl_name, "records_per_gpu_per_step": nrec,
                       "datagrams_per_gpu": len(w["batch"]), "objects": len(w["mos"]),
                       "sharding": "exporters per rank, no data-path collective"},
            "payload_GBps": pay * args.steps / dt_all / 1e9,
            "merged_counters_object0": [int(x) for x in ctr[0]] if len(ctr) else [],
            "merged_records_per_object": [int(x) for x in ctr[:, 0]] if len(ctr) else [],
            "records_all_steps": int(tot) * args.steps,
            "roofline": roof,
            "device_path": dpath,
            "sustained": sustained,
            "host_inclusive": {"prepare_s": t_prep, "prepare_run_fetch_s": t_e2e,
                               "records_per_s": nrec / t_e2e,
                               "page_locked_input_s": t_e2e_pinned,
                               "page_locked_input_records_per_s": nrec / t_e2e_pinned,
                               "page_locked_input_one_host_thread_records_per_s": nrec / t_e2e_pinned_1,
                               "headers_walked_on_device": walked,
                               "one_thread_one_handle_overlapped": overlapped,
                               "pipelined": piped},
            "fixed_window": fixed_window,
        }
        batch.free()
        eng.close()
        if args.others and world == 1:
            others = {}
            for name in ("c1", "c2", "c3", "c4", "c5", "c2m"):
                if name == args.workload:
                    continue
                try:
                    others[name] = measure_other(torch, name, args, local_rank, stream)
                except Exception as ex:
                    others[name] = {"error": repr(ex)}
            out["workloads"] = others
        if c5n is not None:
            out["workloads"] = {"c5": c5n}
        out.update(cpu)
        out["bench_wall_s"] = time.perf_counter() - t_start
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
