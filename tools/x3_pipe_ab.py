#!/usr/bin/env python
"""A/B of the bf16x3 conv tiles: the hand-pipelined kernels of the long-K layers (product, csrc/conv_x3_pipe.hip) against the
compiler-scheduled 8-wave tiles everywhere (TT_X3_PIPE=0, conv_igemm_glds.hip X3 body).  Each arm runs in its own process (the knob is read once); the
outputs of the two arms must be BIT-IDENTICAL (same operand split, same K order, same term order), and both are checked
against the exact-f32 kernel.  Usage:  python tools/x3_pipe_ab.py [rounds]     (needs a GPU)"""
import json
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

# (N, H, W, Cin, Cout, k, stride, residual+relu)
SHAPES_128 = [
    (64, 112, 224, 384, 128, 3, 1, False),    # UNet: M = 1,605,632, K = 3456
    (64, 56, 112, 128, 128, 3, 1, True),      # M = 401,408, K = 1152
    (8, 84, 84, 128, 128, 3, 1, True),        # SECOND block: M = 56,448
    (64, 56, 112, 512, 128, 1, 1, False),     # 1x1, K = 512 (below the pipe threshold)
]

SHAPES = [
    (64, 112, 224, 256, 256, 3, 1, False),    # dominant layer: M = 1,605,632
    (64, 28, 56, 512, 512, 3, 1, True),       # DepthNet 512 -> 512 (10 calls per forward), 784 tiles: tail split
    (64, 56, 112, 256, 256, 3, 1, True),      # M = 401,408
    (64, 56, 112, 256, 256, 3, 2, False),     # stride 2
    (64, 28, 56, 1024, 256, 1, 1, True),      # 1x1, K = 1024
    (64, 112, 224, 64, 256, 1, 1, False),     # 1x1, K = 64 (two K tiles)
    (8, 112, 224, 256, 1280, 1, 1, False),    # decoder value_proj-like
    (3, 37, 53, 96, 256, 3, 1, True),         # ragged M (5883 rows), odd image size, 3 channel chunks
]


def worker(out_path):
    from thinktwice_amd import ops, weights
    res = []
    for (N, H, W, Cin, Cout, k, stride, rr) in (SHAPES_128 if os.environ.get("TT_AB_SET") == "128" else SHAPES):
        g = torch.Generator(device="cuda").manual_seed(1234)
        x = torch.randn(N, H, W, Cin, device="cuda", generator=g)
        w = torch.randn(Cout, k, k, Cin, device="cuda", generator=g) * (Cin * k * k) ** -0.5
        pad = k // 2
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        r = torch.randn(N, OH, OW, Cout, device="cuda", generator=g) if rr else None
        wx = weights.split_pairs_x3(w)
        conv = lambda: ops.conv2d(x, w, stride=stride, pad=pad, act=1 if rr else 0, res1=r, w_x3=wx)
        y = conv()
        kern = ops._last_conv_kernel()
        torch.cuda.synchronize()
        times = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                y = conv()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / 10)
        # repeatability inside the arm (a race in the pipeline would show as run-to-run differences)
        y2 = conv()
        same = bool(torch.equal(y, y2))
        yf = ops.conv2d(x, w, stride=stride, pad=pad, act=1 if rr else 0, res1=r)       # exact f32 kernel
        err = float((y - yf).abs().max() / yf.abs().max())
        M = N * OH * OW
        fl = 2.0 * M * Cout * k * k * Cin
        bits = y.contiguous().view(torch.int32)
        res.append(dict(shape=[N, H, W, Cin, Cout, k, stride, rr], M=M, ms=min(times), ms_all=times, tf=fl / min(times) / 1e9,
                        repeat_equal=same, rel_err_vs_f32=err, kernel=kern,
                        xor=int(bits.flatten()[:: 1].to(torch.int64).sum().item()),
                        sample=bits.flatten()[::max(1, bits.numel() // 4096)][:4096].cpu().tolist()))
    json.dump(res, open(out_path, "w"))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "worker":
        worker(sys.argv[2])
        return
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    arms = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]      # TT_X3_PIPE: 0 = 8-wave tiles only, 1 = product
    shapes = SHAPES_128 if os.environ.get("TT_AB_SET") == "128" else SHAPES                 # TT_AB_SET=128: shapes that take the 128-wide tile
    outs = {a: [] for a in arms}
    for rd in range(rounds):
        for arm in arms:
            f = tempfile.mktemp(suffix=".json")
            env = dict(os.environ, TT_X3_PIPE=str(arm))
            subprocess.run([sys.executable, __file__, "worker", f], check=True, env=env)
            outs[arm].append(json.load(open(f)))
    ok = True
    print(f"{'shape':44s} " + " ".join(f"{'arm' + str(a) + ' ms':>10s} {'TF/s':>7s}" for a in arms) + "  bit-equal  repeatable  err_vs_f32")
    for i, sh in enumerate(shapes):
        best = {a: min(o[i]["ms"] for o in outs[a]) for a in arms}
        r0 = outs[arms[0]][0][i]
        eq = all(outs[a][0][i]["xor"] == r0["xor"] and outs[a][0][i]["sample"] == r0["sample"] for a in arms)
        rep = all(o[i]["repeat_equal"] for a in arms for o in outs[a])
        err = max(outs[a][0][i]["rel_err_vs_f32"] for a in arms)
        ok = ok and eq and rep and err < 1e-4
        fl = r0["tf"] * r0["ms"]
        print(f"{str(sh):44s} " + " ".join(f"{best[a]:10.3f} {fl / best[a]:7.1f}" for a in arms) + f"  {str(eq):5s} rep={rep}  {err:.1e}  " +
              " | ".join(outs[a][0][i]["kernel"] for a in arms))
    print("ALL OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
