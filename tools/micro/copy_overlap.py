#!/usr/bin/env python3
"""Do a host-to-device and a device-to-host copy overlap on this box?  Times 226 MB in, 33.5 MB out
(one C2 batch and its match masks) alone, back to back on one stream, and together on two streams."""
import time
import torch

dev = torch.device("cuda", 0)
hin = torch.empty(226 << 20, dtype=torch.uint8).pin_memory()
din = torch.empty(226 << 20, dtype=torch.uint8, device=dev)
dout = torch.empty(int(33.5 * (1 << 20)), dtype=torch.uint8, device=dev)
hout = torch.empty(int(33.5 * (1 << 20)), dtype=torch.uint8).pin_memory()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def h2d():
    with torch.cuda.stream(s1):
        din.copy_(hin, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        hout.copy_(dout, non_blocking=True)


def serial():
    with torch.cuda.stream(s1):
        din.copy_(hin, non_blocking=True)
        hout.copy_(dout, non_blocking=True)


def both():
    h2d()
    d2h()


print("h2d alone %.3f ms, d2h alone %.3f ms, one stream %.3f ms, two streams %.3f ms"
      % (timed(h2d), timed(d2h), timed(serial), timed(both)))
