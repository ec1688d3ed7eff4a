"""Phase timing inside tt_dec_gru / tt_dec_flatten (tools only; tt_dec_set_trace): wall-clock stamps of workgroup 0, us since entry."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import _lib, config, decoder_fused as DF, ops, params  # noqa: E402

cfg = config.model_config(final_dim=(128, 256))
sd = params.init_params(cfg, seed=0, parts=("fusion", "decoder"))
L = _lib.lib()


def traced(fn, name):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = torch.zeros(64, dtype=torch.int64, device="cuda")
    L.tt_dec_set_trace(ctypes.c_void_p(buf.data_ptr()))
    fn()
    torch.cuda.synchronize()
    L.tt_dec_set_trace(ctypes.c_void_p(0))
    t = buf.cpu()
    clk = (t[63] - t[62]).item() if t[63] > 0 else 0
    t[62] = t[63] = 0
    n = int((t > 0).sum())
    rel = [(t[i] - t[0]).item() / 100 for i in range(n)]
    print(f"== {name}: {n} stamps, total {rel[-1]:.1f} us")
    print("   deltas: " + " ".join(f"{rel[i] - rel[i - 1]:.1f}" for i in range(1, n)))
    if clk:
        print(f"   s_memtime ticks over the kernel: {clk} -> {clk / rel[-1]:.0f} per us")


for B in (1, 8):
    w = DF.prep_gru(sd, "decoder.decoder_layers.0.prediction_module.spatial_gru", "cuda")
    inp6, state, fut = torch.randn(B, 4, 6).cuda(), torch.randn(B, 441, 32).cuda(), torch.empty(B, 4, 441, 32).cuda()
    traced(lambda: ops.dec_gru(w, inp6, state, fut), f"gru B={B} (entry, map load, then per step: class sums + 8 convs)")
    fw = DF.prep_flatten(sd, "cuda")
    maps = torch.randn(B * 4, 441, 32).cuda()
    traced(lambda: ops.dec_flatten(fw, maps), f"flatten head maps={B * 4} (load, conv21_10, MLP10: conv1, conv2, pool, fc1+fc2, gate; conv10_4)")


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


bw = DF.prep_bev_update(sd, "decoder.decoder_layers.0", "cuda")
for B in (1, 8):
    bev, G, o = torch.randn(B, 441, 32).cuda(), torch.randn(B, 1152).cuda(), torch.empty(B, 441, 32).cuda()
    print(f"bev_update B={B}: {timeit(lambda: ops.dec_bev_update(bw, bev, G, o)):.1f} us per launch (back to back)")
