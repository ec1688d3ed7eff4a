#!/usr/bin/env python
"""A/B of the h2 conv kernels: the hand-pipelined one-wave-per-SIMD kernel of the long-K 128-wide layers (product,
conv_h2_pipe_kernel) against the compiler-scheduled eight-wave kernel everywhere (TT_H2_PIPE=0).  Each arm runs in its own
process (the knob is read once); the outputs must be BIT-IDENTICAL (same operands, K order and term order).
Usage:  python tools/h2_pipe_ab.py [rounds]     (needs a GPU)"""
import json
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

# (N, H, W, Cin, Cout, k, stride, residual+relu, f32 output)
SHAPES = [
    (64, 112, 224, 256, 256, 3, 1, False, True),     # fpn0: M = 1,605,632
    (64, 56, 112, 256, 256, 3, 1, False, False),     # fpn1
    (64, 56, 112, 256, 256, 3, 2, True, False),      # down: stride 2 (+ residual)
    (64, 28, 56, 256, 256, 3, 1, False, True),       # fpn2 / paf1
    (8, 56, 112, 256, 256, 3, 1, False, True),       # batch-1 tick
    (3, 37, 53, 128, 384, 3, 1, True, False),        # ragged M (5883 rows), odd image, three column tiles, 18 K tiles
    (2, 9, 11, 1152, 128, 1, 1, False, True),        # 1x1, K = 1152, M = 198 (one partial row tile)
    (2, 40, 48, 192, 256, 3, 1, True, True),         # Cin = 192: three channel chunks; f32 output with a half residual
]


def worker(out_path):
    from thinktwice_amd import ops, weights
    res = []
    for (N, H, W, Cin, Cout, k, stride, rr, f32o) in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(4321)
        x = torch.randn(N, H, W, Cin, device="cuda", generator=g).half()
        w = torch.randn(Cout, k, k, Cin, device="cuda", generator=g) * (Cin * k * k) ** -0.5
        pad = k // 2
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        r = torch.randn(N, OH, OW, Cout, device="cuda", generator=g).half() if rr else None
        wh = weights.split_pairs_h2(w)
        conv = lambda: ops.conv2d(x, w.half(), stride=stride, pad=pad, act=1 if rr else 0, res1=r, w_h2=wh,
                                  out_dtype=torch.float32 if f32o else None)
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
        same = bool(torch.equal(y, conv()))                    # repeatability inside the arm
        ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, stride, pad)
        if rr:
            ref = torch.relu(ref + r.float().permute(0, 3, 1, 2))
        err = float((y.float().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max())
        M = N * OH * OW
        bits = y.contiguous().view(torch.int32 if f32o else torch.int16)
        res.append(dict(shape=[N, H, W, Cin, Cout, k, stride, rr, f32o], M=M, ms=min(times),
                        tf=2.0 * M * Cout * k * k * Cin / min(times) / 1e9, repeat_equal=same, rel_err=err, kernel=kern,
                        xor=int(bits.flatten().to(torch.int64).sum().item()),
                        sample=bits.flatten()[::max(1, bits.numel() // 4096)][:4096].cpu().tolist()))
    json.dump(res, open(out_path, "w"))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "worker":
        worker(sys.argv[2])
        return
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    ok = True
    for rnd in range(rounds):
        out = {}
        for arm in ("0", "1"):
            with tempfile.NamedTemporaryFile(suffix=".json", delete=False) as f:
                path = f.name
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "worker", path], env=dict(os.environ, TT_H2_PIPE=arm),
                               capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                print(f"arm {arm} FAILED\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
                sys.exit(1)
            out[arm] = json.load(open(path))
            os.unlink(path)
        for a, b in zip(out["0"], out["1"]):
            equal = a["xor"] == b["xor"] and a["sample"] == b["sample"]
            good = equal and a["repeat_equal"] and b["repeat_equal"] and max(a["rel_err"], b["rel_err"]) < (2e-5 if a["shape"][8] else 1e-3)
            ok = ok and good
            print(f"{str(a['shape']):48s} M={a['M']:8d}  8-wave {a['ms']:7.3f} ms {a['tf']:6.1f} TF/s | pipe {b['ms']:7.3f} ms {b['tf']:6.1f} TF/s "
                  f"({a['ms'] / b['ms']:.3f}x)  bit-identical {equal}  repeatable {a['repeat_equal'] and b['repeat_equal']}  "
                  f"err {b['rel_err']:.1e}  [{a['kernel']} | {b['kernel']}]{'' if good else '   <-- FAIL'}")
    print("ALL OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
