#!/usr/bin/env python3
"""Host cost of ONE library call, three ways, on the smallest kernel of the step (vfm_elbo_finalize_f32: one workgroup):
through torch.ops (the TORCH_LIBRARY shim), through ctypes with the arguments converted per call, and through ctypes with
the arguments converted once (ops.Prepared).  Prints microseconds of host time per call (queue never full: one sync per
200 calls) -- what a step made of ten such calls pays before any kernel time."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import _lib, ops

dev = torch.device("cuda")
partials = torch.zeros(_lib.PARTIALS_LEN, dtype=torch.float64, device=dev)
partials[7] = 4
scal = torch.tensor([0.5, 0.1, 0.9], device=dev)
loss = torch.zeros(3, device=dev)
lib = _lib.load()
o = _lib.ops()
spec = ops.Spec(T=10, F=2, d=8, group_hi=(5, 10), group_n=(5.0, 5.0), likelihood=0, nb_train=100)
p = ops._problem(spec, 4, 4, 64)


def timed(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn()
        if i % 200 == 199:
            torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def via_ops():
    o.elbo_finalize(partials, scal, loss, 100, 4, 0, 1)


def via_ctypes():
    lib.vfm_elbo_finalize_f32(C.byref(p), _lib.ptr(partials), _lib.ptr(scal), _lib.ptr(loss), _lib.current_stream_ptr(dev))


args = (C.byref(p), _lib.ptr(partials), _lib.ptr(scal), _lib.ptr(loss), _lib.current_stream_ptr(dev))


def via_prepared():
    lib.vfm_elbo_finalize_f32(*args)


def empty():
    pass


print("torch.ops shim      %.2f us" % timed(via_ops))
print("ctypes, per-call    %.2f us" % timed(via_ctypes))
print("ctypes, prepared    %.2f us" % timed(via_prepared))
print("python call only    %.2f us" % timed(empty))
