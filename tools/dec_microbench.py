"""Timing of the composite-decoder kernels in isolation (tools only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from thinktwice_amd import config, decoder_fused as DF, ops, params  # noqa: E402


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
    return e0.elapsed_time(e1) / iters * 1e3     # us


def chain(R, dims, acts=None):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(R, (dims[0] + 15) // 16 * 16, generator=g).cuda()
    lins = [ops.ChainLinear(torch.randn(dims[i + 1], dims[i], generator=g) * dims[i] ** -0.5, torch.zeros(dims[i + 1]),
                            act=(acts[i] if acts else 1)) for i in range(len(dims) - 1)]
    out = torch.empty(R, dims[-1], device="cuda")
    st = [{"lin": l, "src": i - 1} for i, l in enumerate(lins)]
    st[-1]["out"] = (out, 0)
    return timeit(lambda: ops.mlp_chain(x, st))


print("chain R=8    256->256            ", chain(8, [256, 256]))
print("chain R=8    256->256->256->1    ", chain(8, [256, 256, 256, 1]))
print("chain R=32   1024->512->512      ", chain(32, [1024, 512, 512]))
print("chain R=32   2304->512->256      ", chain(32, [2304, 512, 256]))
print("chain R=3840 256->256            ", chain(3840, [256, 256]))
print("chain R=3840 1544->512           ", chain(3840, [1544, 512]))
print("chain R=3840 1544->512->256->512 ", chain(3840, [1544, 512, 256, 512]))
print("chain R=3840 256->1024->256      ", chain(3840, [256, 1024, 256]))
print("chain R=128  256->1024->256      ", chain(128, [256, 1024, 256]))
cfg = config.model_config(final_dim=(128, 256))
sd = params.init_params(cfg, seed=0, parts=("fusion", "decoder"))
for B in (1, 8):
    w = DF.prep_gru(sd, "decoder.decoder_layers.0.prediction_module.spatial_gru", "cuda")
    inp6, state, fut = torch.randn(B, 4, 6).cuda(), torch.randn(B, 441, 32).cuda(), torch.empty(B, 4, 441, 32).cuda()
    print(f"gru B={B}", timeit(lambda: ops.dec_gru(w, inp6, state, fut), 20))
    fw = DF.prep_flatten(sd, "cuda")
    maps = torch.randn(B * 4, 441, 32).cuda().abs()
    print(f"flatten maps={B * 4}", timeit(lambda: ops.dec_flatten(fw, maps), 20))
    bw = DF.prep_bev_update(sd, "decoder.decoder_layers.0", "cuda")
    bev, G, o = torch.randn(B, 441, 32).cuda(), torch.randn(B, 1152).cuda(), torch.empty(B, 441, 32).cuda()
    print(f"bev_update B={B}", timeit(lambda: ops.dec_bev_update(bw, bev, G, o), 20))
