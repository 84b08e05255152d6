"""Does a memory/latency-bound kernel hide beside an MFMA-bound one when a captured graph forks?  cross_bwd (B = 4096, d = 416,
L = 3) beside dense_bwd(512 -> 256): one graph with the two launches in series on one stream, one with cross_bwd on a second
stream between a fork and a join, REPS pairs per graph.  Prints the time per pair."""
import ctypes
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recalgorithm_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
B, d, L = 4096, 416, 3
REPS = 20
p = lambda t: ctypes.c_void_p(t.data_ptr())
x0 = torch.randn(B, d, device=dev)
w, b = torch.randn(L, d, device=dev) * 0.05, torch.randn(L, d, device=dev) * 0.05
g = torch.randn(B, d, device=dev)
dx0, dw, db = torch.empty_like(x0), torch.empty_like(w), torch.empty_like(b)
ws = torch.empty(lib.recalgo_cross_bwd_workspace_bytes(B, d, L), dtype=torch.uint8, device=dev)
xd = torch.randn(B, 512, device=dev).clamp_(min=0)
wd = torch.randn(512, 256, device=dev) / 512 ** 0.5
gd = torch.randn(B, 256, device=dev) * (torch.rand(B, 256, device=dev) > 0.5)
dwd, dbd = torch.empty_like(wd), torch.empty(256, device=dev)


def cross(stream):
    _lib.check(lib.recalgo_cross_bwd(p(x0), d, p(w), p(b), p(g), d, None, B, d, L, p(dx0), p(dw), p(db), p(ws), 1,
                                     ctypes.c_void_p(stream.cuda_stream)), "cross_bwd")


def dense():
    ops.dense_bwd(xd, gd, None, wd, dwd, dbd, defer=True, premask=xd)
    ops._dense_pending.clear()


side = torch.cuda.Stream(device=dev)


def serial():
    for _ in range(REPS):
        cross(torch.cuda.current_stream())
        dense()


def forked():
    main = torch.cuda.current_stream()
    for _ in range(REPS):
        side.wait_stream(main)
        cross(side)
        dense()
        main.wait_stream(side)


def only(fn):
    def run():
        for _ in range(REPS):
            fn()
    return run


for name, fn in (("cross_bwd alone", only(lambda: cross(torch.cuda.current_stream()))), ("dense_bwd alone", only(dense)),
                 ("serial", serial), ("forked", forked)):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:18s} {e0.elapsed_time(e1) * 1e3 / (20 * REPS):8.2f} us per pair")
