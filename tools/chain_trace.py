"""Phase timing INSIDE tt_mlp_chain_wide (tools only; tt_mlp_chain_wide_set_trace): per stage, for the slowest workgroup and for
workgroup 0 -- wait at the barrier, K loop of the first column block (+ reduction), rest of the stage.  10 ns ticks -> us."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import _lib, ops  # noqa: E402

g = torch.Generator().manual_seed(0)


def lin(n, k, act=1, side_k=0):
    return ops.ChainLinear(torch.randn(n, k, generator=g) * k ** -0.5, torch.zeros(n), act=act, side_k=side_k)


def trace(name, R, stages_fn, groups=None):
    x, st = stages_fn(R)
    L = _lib.lib()
    for _ in range(3):
        ops.mlp_chain(x, st, wide=True, groups=groups)
    torch.cuda.synchronize()
    buf = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
    L.tt_mlp_chain_wide_set_trace(ctypes.c_void_p(buf.data_ptr()))
    ops.mlp_chain(x, st, wide=True, groups=groups)
    torch.cuda.synchronize()
    L.tt_mlp_chain_wide_set_trace(ctypes.c_void_p(0))
    t = buf.view(256, 64).cpu()
    live = t[:, 0] > 0
    t = t[live]
    nwg = t.shape[0]
    t0 = t[:, 0].min()
    ns = len(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.mlp_chain(x, st, wide=True, groups=groups)
    e1.record()
    torch.cuda.synchronize()
    print(f"== {name}: R={R}, {ns} stages, {nwg} workgroups, {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch back to back; "
          f"entry skew {(t[:, 0].max() - t0).item() / 100:.2f} us")
    # stamps: [0] entry; per stage: entry, after barrier, after first block's K loop (only if the workgroup has a block), stage end
    # -> variable count; decode per workgroup by replaying which stamps exist is fragile: print raw relative stamps of wg 0 and the
    # per-stage END (max over workgroups) instead
    for w in (0, nwg - 1):
        row = t[w]
        n = int((row > 0).sum())
        print(f"  wg {w}: " + " ".join(f"{(row[i] - t0).item() / 100:.2f}" for i in range(n)))
    print(f"  last stamp, max over workgroups: {(t.max() - t0).item() / 100:.2f} us")


def merge(R):
    x = torch.randn(R, 1024, generator=g).cuda()
    wp, ct = torch.randn(R, 2, generator=g).cuda(), torch.randn(R, 4, generator=g).cuda()
    m = [lin(512, 1024), lin(512, 512), lin(256, 514, side_k=2), lin(64, 256), lin(2, 64, 0), lin(256, 516, side_k=4), lin(64, 256),
         lin(4, 64, 0)]
    h, o2, o4 = torch.empty(R, 512, device="cuda"), torch.empty(R, 2, device="cuda"), torch.empty(R, 4, device="cuda")
    return x, [{"lin": m[0], "src": -1}, {"lin": m[1], "src": 0, "out": (h, 0)}, {"lin": m[2], "src": 1, "side": wp},
               {"lin": m[5], "src": 1, "side": ct}, {"lin": m[3], "src": 2}, {"lin": m[6], "src": 3},
               {"lin": m[4], "src": 4, "res": (wp, 0), "out": (o2, 0)}, {"lin": m[7], "src": 5, "res": (ct, 0), "out": (o4, 0)}]


def ffn(R):
    x = torch.randn(R, 256, generator=g).cuda()
    y = torch.empty(R, 256, device="cuda")
    return x, [{"lin": lin(1024, 256, 3), "src": -1}, {"lin": lin(256, 1024, 0), "src": 0, "res": (x, 0), "out": (y, 0)}]


def query(R):
    x = torch.randn(R, 1552, generator=g).cuda()
    off, aw = torch.empty(R, 512, device="cuda"), torch.empty(R, 256, device="cuda")
    return x, [{"lin": lin(512, 1544, 3), "src": -1}, {"lin": lin(256, 512, 3), "src": 0},
               {"lin": lin(512, 256, 0), "src": 1, "out": (off, 0)}, {"lin": lin(256, 256, 0), "src": 1, "out": (aw, 0)}]


def flat(R):
    x = torch.randn(R, 2304, generator=g).cuda()
    y = torch.empty(R, 256, device="cuda")
    return x, [{"lin": lin(512, 2304), "src": -1}, {"lin": lin(256, 512, 0), "src": 0, "out": (y, 0)}]


def single(R):
    x = torch.randn(R, 2048, generator=g).cuda()
    y = torch.empty(R, 1152, device="cuda")
    return x, [{"lin": lin(1152, 2048, 0), "src": -1, "out": (y, 0)}]


trace("merge", 4, merge)
trace("flat update", 1, flat)
trace("G (one stage)", 1, single, groups=36)
trace("ffn", 480, ffn)
trace("query", 480, query)
