"""Exporter sharding over ranks (world_size 2, gloo, CPU): every rank works on
its own exporters, counters merge by addition, and the merged result equals the
single-process result.  The oracle stands in for the per-rank engine here; the
GPU engine is checked against the oracle elsewhere."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from xenoeye_amd import shard, synth  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_run(w, packed):
    from oracle_binding import Oracle
    orc = Oracle()
    for name, text in w["iplists"].items():
        orc.iplist_add(name, text)
    for filt, parent in w["mos"]:
        orc.mo_add(filt, parent)
    orc.log_config(keep_flows=False)
    data, offs, src, kinds = packed
    for i in range(len(offs) - 1):
        p = bytes(data[int(offs[i]):int(offs[i + 1])])
        orc.netflow(p, int(src[i]).to_bytes(4, "little"))
    return orc.nrecords(), orc.counters().copy(), orc.masks().copy()


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.workload_c2(6000, prefixes=200)
    mine = shard.select(w["batch"].packed(), rank, world)
    nrec, ctr, _ = _oracle_run(w, mine)
    merged = shard.merge_counters(ctr, dist)
    tot, tmax = shard.reduce_scalars([nrec, rank + 1.0], ["sum", "max"], dist)
    q.put((rank, nrec, len(mine[1]) - 1, merged.tolist(), tot, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_select_partitions_exporters():
    w = synth.workload_c2(3000, prefixes=50)
    packed = w["batch"].packed()
    data, offs, src, kinds = packed
    parts = [shard.select(packed, r, 3) for r in range(3)]
    assert sum(len(p[1]) - 1 for p in parts) == len(offs) - 1
    seen = set()
    for r, (d, o, s, k) in enumerate(parts):
        assert not (set(s.tolist()) & seen)
        seen |= set(s.tolist())
        assert np.all(shard.exporter_rank(s, 3) == r)
        # arrival order and bytes are kept
        idx = np.nonzero(shard.exporter_rank(src, 3) == r)[0]
        for j, i in enumerate(idx[:50]):
            assert bytes(d[int(o[j]):int(o[j + 1])]) == bytes(data[int(offs[i]):int(offs[i + 1])])
    assert len(seen) == 64
    one = shard.select(packed, 0, 1)
    assert one[0] is not None and len(one[1]) == len(offs)


def test_c5_generated_per_rank_equals_cut_from_the_whole():
    """the 512-exporter replay: a rank that generates only its exporters gets
    the datagrams that select() cuts out of the whole batch, and the oracle's
    counters over the shards add up to its counters over the whole"""
    world = 3
    w = synth.workload_c5(60000, prefixes=100)
    full = w["batch"].packed()
    nrec, ctr, _ = _oracle_run(w, full)
    assert nrec == w["nrec"]
    total = np.zeros_like(ctr)
    seen = 0
    for r in range(world):
        part = synth.workload_c5(60000, prefixes=100, rank=r, world=world)
        mine = part["batch"].packed()
        cut = shard.select(full, r, world)
        assert sorted(mine[2].tolist()) == sorted(cut[2].tolist())
        n_r, c_r, _ = _oracle_run(part, mine)
        n_c, c_c, _ = _oracle_run(w, cut)
        assert n_r == n_c == part["nrec"] and np.array_equal(c_r, c_c)
        total = total + c_r
        seen += n_r
    assert seen == nrec and np.array_equal(total, ctr)
    assert (ctr[:, 0] > 0).sum() >= 9


def test_wrapping_merge_without_process_group():
    c = np.array([[2**64 - 1, 5, 7]], dtype=np.uint64)
    assert np.array_equal(shard.merge_counters(c), c)


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    w = synth.workload_c2(6000, prefixes=200)
    nrec, ctr, _ = _oracle_run(w, w["batch"].packed())
    assert res[0][1] + res[1][1] == nrec == res[0][4] == res[1][4]
    assert res[0][1] > 0 and res[1][1] > 0
    assert res[0][3] == res[1][3] == ctr.tolist()
    assert res[0][5] == 2.0


@pytest.mark.timeout(300)
def test_bench_gpus_n_starts_n_ranks():
    """`bench.py --gpus 2` from a plain shell starts two ranks that find each other
    (the rehearsal mode stops before any GPU work) and prints one line for the job"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--spawn-check"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert lines == [{"spawn_check": True, "n_gpus": 2, "sum_of_ranks_plus_1": 3, "asked": 2}]
