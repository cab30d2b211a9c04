"""The entry point that allocates nothing (xe_gpu_iplist_match_noalloc: the
list probe staged in a static device array) against the allocating entry point
it is a variant of (xe_gpu_iplist_match), bit for bit, through the C ABI: two
calls in a row, the second with a smaller input (the array keeps the first
call's contents), both address families, the size limit."""
import numpy as np
import pytest

import golden_io
import xenoeye_amd

pytestmark = pytest.mark.gpu

MAX_ADDRS = 65536   # XE_GPU_NOALLOC_MAX_ADDRS


@pytest.fixture()
def eng():
    e = xenoeye_amd.Engine(device=0)
    yield e
    e.close()


def _random_addrs(rng, texts, family, n):
    """half of the probes inside a prefix of some list, half anywhere"""
    width = 4 if family == 4 else 16
    out = rng.integers(0, 256, size=(n, width), dtype=np.uint8)
    nets = []
    for t in texts:
        for line in (t.decode() if isinstance(t, bytes) else t).splitlines():
            line = line.split("#")[0].strip()
            if not line or ((":" in line) != (family == 6)):
                continue
            nets.append(line.split("/")[0])
    import socket
    fam = socket.AF_INET if family == 4 else socket.AF_INET6
    keep = []
    for a in nets:
        try:
            keep.append(np.frombuffer(socket.inet_pton(fam, a), dtype=np.uint8))
        except OSError:
            pass
    if keep:
        pick = rng.integers(0, len(keep), size=n // 2)
        for i, k in enumerate(pick):
            head = int(rng.integers(1, width))
            out[i, :head] = keep[k][:head]
    return out


def test_no_alloc_probe_equals_allocating_probe(eng):
    texts, p4, p6, want4, want6 = golden_io.iplist_reference()
    for i, t in enumerate(texts):
        eng.add_iplist("l%d" % i, t)
    eng.commit()
    rng = np.random.default_rng(0x5EED0A11)
    big4, big6 = _random_addrs(rng, texts, 4, 50000), _random_addrs(rng, texts, 6, 40000)
    small4, small6 = big4[7:7 + 333][::-1].copy(), big6[11:11 + 97][::-1].copy()
    for i in range(len(texts)):
        name = "l%d" % i
        # first call large, second smaller: nothing of the first may show through
        for fam, big, small in ((4, big4, small4), (6, big6, small6)):
            for addrs in (big, small):
                want = eng.iplist_match(name, fam, addrs)
                got = eng.iplist_match_noalloc(name, fam, addrs)
                assert got.shape == want.shape
                assert np.array_equal(got, want), (name, fam, len(addrs))
        # and the reference's own answers for its probe set
        assert np.array_equal(eng.iplist_match_noalloc(name, 4, p4), want4[i]), i
        assert np.array_equal(eng.iplist_match_noalloc(name, 6, p6), want6[i]), i


def test_no_alloc_probe_limits(eng):
    eng.add_iplist("one", b"10.0.0.0/8\n2001:db8::/32\n")
    eng.commit()
    full = np.zeros((MAX_ADDRS, 4), dtype=np.uint8)
    full[:, 0] = 10
    full[1::2, 0] = 11
    got = eng.iplist_match_noalloc("one", 4, full)
    assert np.array_equal(got, eng.iplist_match("one", 4, full))
    assert int(got.sum()) == MAX_ADDRS // 2
    # one more than the static area holds: refused, with a reason
    over = np.zeros((MAX_ADDRS + 1, 4), dtype=np.uint8)
    with pytest.raises(xenoeye_amd.XeError) as ei:
        eng.iplist_match_noalloc("one", 4, over)
    assert ei.value.code == -1 and "static staging area" in str(ei.value)
    # empty input
    assert eng.iplist_match_noalloc("one", 6, np.zeros((0, 16), dtype=np.uint8)).size == 0
    # a second engine on the same GPU shares the array: sequential calls stay right
    e2 = xenoeye_amd.Engine(device=0)
    try:
        e2.add_iplist("one", b"11.0.0.0/8\n")
        e2.commit()
        got2 = e2.iplist_match_noalloc("one", 4, full[:1000])
        assert np.array_equal(got2, e2.iplist_match("one", 4, full[:1000]))
        assert int(got2.sum()) == 500
        assert np.array_equal(eng.iplist_match_noalloc("one", 4, full[:10]), got[:10])
    finally:
        e2.close()
