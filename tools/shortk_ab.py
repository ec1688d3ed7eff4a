#!/usr/bin/env python
"""A/B of the bf16x3 kernels on the SHORT-K 1x1 / GEMM layers of the forward (21 ms of the step at 0.30 of HBM):
each arm = a set of environment knobs, run in its own process (the knobs are read once); per shape the best of 3 x 10 launches and a
hash of the output -- every arm must produce BIT-IDENTICAL results (same operand split, K order and term order).
    python tools/shortk_ab.py [arm ...]         (needs a GPU; arms: see ARMS)"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

# (N, H, W, Cin, Cout, residual + relu, calls per forward)
SHAPES = [
    (64, 28, 56, 256, 1024, True, 7),      # M = 100,352  N = 1024 K = 256   ResNet layer3 conv3
    (64, 56, 112, 128, 512, True, 4),      # M = 401,408  N = 512  K = 128   layer2 conv3
    (64, 28, 56, 1024, 256, False, 6),     # M = 100,352  N = 256  K = 1024  layer3 conv1
    (32, 112, 224, 256, 1280, False, 1),   # M = 802,816  N = 1280 K = 256   decoder value GEMM, level 0
    (64, 112, 224, 64, 256, True, 4),      # M = 1,605,632 N = 256 K = 64    layer1 conv3
    (64, 112, 224, 256, 256, False, 1),    # M = 1,605,632 N = 256 K = 256
    (64, 56, 112, 512, 128, False, 3),     # M = 401,408  N = 128  K = 512   layer2 conv1
    (64, 56, 112, 512, 256, False, 2),     # M = 401,408  N = 256  K = 512
    (64, 14, 28, 512, 2048, True, 3),      # M = 25,088   N = 2048 K = 512   layer4 conv3
    (64, 56, 112, 256, 512, False, 1),     # M = 401,408  N = 512  K = 256
    (8, 112, 224, 256, 1280, False, 1),    # M = 200,704  N = 1280 K = 256   value GEMM, level 1
    (64, 112, 224, 256, 64, False, 2),     # M = 1,605,632 N = 64  K = 256   layer1 conv1
]

ARMS = {
    "base": {},
    "t64": {"TT_GLDS_X3_TILE": "64"},
    "t128": {"TT_GLDS_X3_TILE": "128"},
    "stg20": {"TT_GLDS_STAGGER_US": "20"},
    "stg40": {"TT_GLDS_STAGGER_US": "40"},
    "stg80": {"TT_GLDS_STAGGER_US": "80"},
    "pers": {"TT_X3_PERSIST": "1"},                                      # persistent 256 x 128 tiles, eight waves (default geometry)
    "pers4": {"TT_X3_PERSIST": "1", "TT_X3_PERSIST_WAVES": "4"},         # ... four waves, one per SIMD
}


def worker(out_path):
    from thinktwice_amd import ops, weights
    res = []
    for (N, H, W, Cin, Cout, rr, calls) in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(99)
        x = torch.randn(N, H, W, Cin, device="cuda", generator=g)
        w = torch.randn(Cout, 1, 1, Cin, device="cuda", generator=g) * Cin ** -0.5
        r = torch.randn(N, H, W, Cout, device="cuda", generator=g) if rr else None
        wx = weights.split_pairs_x3(w)
        out = torch.empty(N, H, W, Cout, device="cuda")
        conv = lambda: ops.conv2d(x, w, act=1 if rr else 0, res1=r, w_x3=wx, out=out)
        y = conv()
        kern = ops._last_conv_kernel()
        torch.cuda.synchronize()
        times = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                conv()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / 10)
        M = N * H * W
        nbytes = 4 * (M * Cin + M * Cout * (2 if rr else 1))
        h = hashlib.sha1(y[: min(N, 4)].contiguous().cpu().numpy().tobytes()).hexdigest()[:12]
        res.append({"shape": f"M={M} N={Cout} K={Cin}{' +res' if rr else ''}", "calls": calls, "ms": min(times), "tb_s": nbytes / min(times) / 1e9,
                    "tf": 2.0 * M * Cout * Cin / min(times) / 1e9, "kernel": kern, "hash": h})
        del x, w, r, out, y
        torch.cuda.empty_cache()
    json.dump(res, open(out_path, "w"))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        return worker(sys.argv[2])
    arms = sys.argv[1:] or list(ARMS)
    results = {}
    for arm in arms:
        env = dict(os.environ)
        env.update(ARMS[arm])
        with tempfile.NamedTemporaryFile(suffix=".json", delete=False) as f:
            path = f.name
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", path], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"arm {arm} failed:\n{r.stderr[-2000:]}")
            continue
        results[arm] = json.load(open(path))
        os.unlink(path)
    base = results.get("base") or next(iter(results.values()))
    print(f"{'shape':34s} " + " ".join(f"{a:>9s}" for a in results) + "   (ms per launch; TB/s of in + out (+ residual) for the first arm)")
    tot = {a: 0.0 for a in results}
    for i, b in enumerate(base):
        row = f"{b['shape']:34s} "
        for a, rs in results.items():
            row += f"{rs[i]['ms']:9.4f}"
            tot[a] += rs[i]["ms"] * rs[i]["calls"]
            if rs[i]["hash"] != b["hash"]:
                row += "!"
        print(row + f"   {b['tb_s']:.2f} TB/s  x{b['calls']}")
    print(f"{'per forward (calls weighted)':34s} " + " ".join(f"{tot[a]:9.3f}" for a in results))
    for a, rs in results.items():
        print(f"  {a}: " + "; ".join(sorted({r['kernel'].split('<')[0] + '<' + r['kernel'].split('<')[1][:40] for r in rs})))
    bad = [(a, rs[i]["shape"]) for a, rs in results.items() for i in range(len(base)) if rs[i]["hash"] != base[i]["hash"]]
    print("BIT-IDENTICAL across arms" if not bad else f"MISMATCH: {bad}")


if __name__ == "__main__":
    main()
