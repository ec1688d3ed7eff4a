"""Host-side logic that needs no GPU: the closed-loop previous-sweep cache driver and the img_metas constants that
a captured graph takes as inputs."""
import os

import pytest
import torch


class _FakeModel:
    """Records what PrevSweepCache passes; the 'key BEV' of tick t is a tensor filled with t."""

    def __init__(self):
        self.calls = []
        self.t = 0

    def forward_inference(self, batch, channel_last_out=False, prev_bev=None):
        self.calls.append(None if prev_bev is None else float(prev_bev.flatten()[0]))
        out = {"_key_bev_cl": torch.full((1, 2, 2, 4), float(self.t))[..., :3], "tick": self.t}   # non-contiguous view
        self.t += 1
        return out


def test_prev_sweep_cache_ring_semantics():
    from thinktwice_amd.encoder_decoder import PrevSweepCache
    m = _FakeModel()
    drv = PrevSweepCache(m, lag=3)
    for _ in range(7):
        drv.tick({})
    # ticks 0..2: cache cold -> full two-sweep forward; tick t >= 3 gets the key BEV of tick t-3
    assert m.calls == [None, None, None, 0.0, 1.0, 2.0, 3.0]
    assert all(t.is_contiguous() for t in drv.ring)            # stored copies, not views of a live buffer
    drv.reset()
    drv.tick({})
    assert m.calls[-1] is None


def test_host_constants_shapes_and_key_frame_selection():
    from thinktwice_amd import synth
    from thinktwice_amd.lss import LSS
    metas = synth.make_img_metas(3)
    c = LSS.host_constants(metas, 4)
    assert c["gm"].shape == (12, 2, 4, 4) and c["mlp_in"].shape == (12, 22)
    assert c["lidar2img"].shape == (3, 4, 4, 4) and c["ida_mat"].shape == (3, 4, 4, 4)
    # geometry pairs are [inv(ida), sensor2ego @ inv(intrin)] of the KEY frame (last sweep index)
    ida_key = torch.as_tensor(metas[0][-1]["ida_mats"], dtype=torch.float32)
    assert torch.allclose(c["gm"][:4, 0], ida_key.inverse(), atol=1e-6)
    assert torch.equal(c["ida_mat"][0], ida_key)


def test_adamw_kernel_op_order_emulation_meets_the_gpu_test_thresholds():
    """The GPU test of tt_adamw_step (tests/test_optim.py) compares against torch.optim.AdamW with thresholds of a few
    f32 ulps.  This CPU test replays the kernel's exact f32 operation order (csrc/optim.hip: no FMA contraction, f32
    hyper-parameters, bias corrections in double) and checks that those thresholds hold for it, so that a failure on
    hardware means a kernel bug rather than a tolerance chosen blind."""
    import math
    n = 200_003
    g = torch.Generator().manual_seed(11)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * s for s in (1.0, 0.01, 0.3)]
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7)
    f32 = lambda x: torch.tensor(x, dtype=torch.float32)          # noqa: E731
    lr, b1, b2, eps, wd, one = f32(1e-4), f32(0.9), f32(0.999), f32(1e-8), f32(1e-7), f32(1.0)
    max_norm = 100.0 * (n / 1_000_003) ** 0.5                      # same clip pattern as the 1M-element GPU test
    p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    for i, gr in enumerate(grads):
        ref_p.grad = gr.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_([ref_p], max_norm)
        opt.step()
        norm = torch.sqrt((gr.double() ** 2).sum()).float()        # the kernel sums f32 partials in double
        coef = f32(max_norm) / (norm + f32(1e-6))
        gs = torch.minimum(coef, one)
        want = min(1.0, max_norm / (float(ref_norm) + 1e-6))
        assert abs(float(gs) - want) < 1e-4 * want
        step = i + 1
        bc1 = f32(1.0 - float(b1) ** step)
        bc2s = f32(math.sqrt(1.0 - float(b2) ** step))
        gi = gr * gs
        pi = p * (one - lr * wd)
        mi = m + (gi - m) * (one - b1)
        vi = v * b2 + gi * gi * (one - b2)
        pi = pi - (lr / bc1) * (mi / (torch.sqrt(vi) / bc2s + eps))
        p, m, v = pi, mi, vi
        assert float((p - ref_p.detach()).abs().max()) < 2e-6
    st = opt.state[ref_p]
    assert float((m - st["exp_avg"]).abs().max()) < 2e-6
    assert float((v - st["exp_avg_sq"]).abs().max()) < 1e-4 * float(st["exp_avg_sq"].max())
    assert 1e-4 < float((p - p0).abs().max()) < 1e-3


def test_config_fromfile_merges_base_files(tmp_path):
    """Mini Config.fromfile (train.py:116): `_base_` files first, child keys override, dicts merge, `_delete_` replaces."""
    from thinktwice_amd.cfgfile import Config
    (tmp_path / "base").mkdir()
    (tmp_path / "base" / "rt.py").write_text("log = dict(interval=50, hooks=[1, 2])\nmodel = dict(type='X', a=1, sub=dict(p=1, q=2))\n")
    (tmp_path / "c.py").write_text("import os\n_base_ = ['./base/rt.py']\nn = 4\n"
                                   "model = dict(a=n, sub=dict(q=3), enc=dict(type='LSS'))\nlog = dict(_delete_=True, interval=7)\n")
    cfg = Config.fromfile(str(tmp_path / "c.py"))
    assert cfg.model.type == "X" and cfg.model.a == 4 and cfg.model.sub == {"p": 1, "q": 3} and cfg.model.enc.type == "LSS"
    assert cfg.log == {"interval": 7} and "os" not in cfg and "_base_" not in cfg


def test_reference_config_file_builds_the_model_verbatim():
    """B2: `build_model(Config.fromfile('configs/thinktwice.py').model)` -- the reference's own config dict, unmodified
    -- constructs this package's EncoderDecoder, and its hyper-parameters equal the restated thinktwice_amd/config.py."""
    import os
    import pytest
    path = "/root/reference/open_loop_training/configs/thinktwice.py"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present (build container only)")
    import torch
    from thinktwice_amd import config as C, model as tm  # noqa: F401  (registers the modules)
    from thinktwice_amd.cfgfile import Config
    from thinktwice_amd.registry import build_model
    cfg = Config.fromfile(path)
    m = build_model(cfg.model)
    assert isinstance(m, torch.nn.Module) and type(m).__name__ == "EncoderDecoder"
    ours = C.model_config()

    def same(a, b, path=""):
        if isinstance(a, dict):
            for k in a:
                if k in b:
                    same(a[k], b[k], path + k + ".")
        else:
            la = list(a) if isinstance(a, (list, tuple)) else a
            lb = list(b) if isinstance(b, (list, tuple)) else b
            assert la == lb or str(la) == str(lb), (path, a, b)
    same(ours["img_encoder"], cfg.model.img_encoder)
    same(ours["lidar_encoder"], cfg.model.lidar_encoder)
    same(ours["cfg"], cfg.model.train_cfg)
    assert cfg.optimizer == {"type": "AdamW", "lr": 1e-4, "weight_decay": 1e-7}


def test_lidar_voxelize_has_no_host_fallback_and_picks_the_configured_cap():
    """mmcv's max_voxels cap (configs/thinktwice.py:161-165: 120000 training / 160000 inference) is implemented on the device
    (tt_lidar_voxelize_capped, tests/test_lidar.py); the host side only selects the cap -- and a host tensor is refused, never
    voxelised on the CPU."""
    import torch
    from thinktwice_amd import _lib, config, lidarnet
    cfg = config.model_config()["lidar_encoder"]
    assert tuple(cfg["pts_voxel_layer"]["max_voxels"]) == (120000, 160000)
    net = lidarnet.LidarNet.__new__(lidarnet.LidarNet)
    net.vl = cfg["pts_voxel_layer"]
    net.training = False
    with pytest.raises(_lib.TTError):
        net.forward(torch.zeros(1, 1000, 5))


def test_ctypes_mirrors_match_the_c_header_layout(tmp_path):
    """The Python host talks to the C ABI through ctypes mirrors of four structs of include/thinktwice_hip.h.  Compile the
    header with gcc, print sizeof / offsetof of every field, and compare with the mirrors: a field appended on one side
    only (tt_conv_desc grew twice in round 3) would otherwise shift every later pointer silently."""
    import ctypes
    import re
    import subprocess
    from thinktwice_amd import control, ops
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    header = open(os.path.join(root, "include", "thinktwice_hip.h")).read()
    mirrors = {"tt_conv_desc": ops._ConvDesc, "tt_chain_stage": ops._ChainStage, "tt_action_cfg": control.ActionCfg,
               "tt_action_state": control.ActionState}
    rename = {"in_": "in"}                      # `in` is a Python keyword
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "thinktwice_hip.h"', "int main(void) {"]
    for cname, cls in mirrors.items():
        assert re.search(r"typedef struct %s\b" % cname, header), cname
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, rename.get(fname, fname)))
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        c, f, v = ln.split()
        got[(c, f)] = int(v)
    for cname, cls in mirrors.items():
        assert got[(cname, "sizeof")] == ctypes.sizeof(cls), (cname, got[(cname, "sizeof")], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_product_package_never_imports_the_oracle():
    """oracle/ is the checker: only tests/, __graft_entry__.smoke() (smoke_forward.py) and bench.py's cpu_baseline leg
    (bench_cpu_baseline.py) may import it -- nothing inside the product package."""
    import ast
    pkg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "thinktwice_amd")
    bad = []
    for f in sorted(os.listdir(pkg)):
        if not f.endswith(".py"):
            continue
        for node in ast.walk(ast.parse(open(os.path.join(pkg, f)).read())):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            bad += [(f, n) for n in names if n == "oracle" or n.startswith("oracle.")]
    assert not bad, bad


def test_bench_counts_three_mfmas_per_product_for_every_bf16x3_kernel():
    """the label test looked at the GATHER template flag; the X3 flag is the last one."""
    from thinktwice_amd.bench_forward import mfma_per_product as f
    assert f("conv_igemm_glds_kernel<float, 256, 4, 2, 128, 23, false, true>", "bf16x3") == 3       # dense x3 tile
    assert f("conv_igemm_glds_kernel<float, 256, 4, 2, 128, 23, false, true> + tail", "bf16x3") == 3
    assert f("conv_igemm_glds_kernel<float, 64, 8, 1, 128, 2, true, true>", "bf16x3") == 3           # gathered x3
    assert f("conv_igemm_glds_kernel<float, 128, 4, 2, 64, 3, false, false>", "bf16x3") == 1         # exact f32 layer
    assert f("conv_igemm_glds_kernel<float, 128, 4, 2, 64, 3, true, false>", "bf16x3") == 1          # GATHER alone is not x3
    assert f("conv_x3_pipe_kernel<4, 1>", "bf16x3") == 3
    assert f("sp_conv_runs_kernel<2, 4, 2>", "bf16x3") == 3
    assert f("conv_small_kernel", "bf16x3") == 1
    assert f("conv_igemm_glds_kernel<float, 256, 4, 2, 128, 23, false, true>", "f32") == 1


def _plan_file_bytes(nstreams=2, buffers=(("weights", 64), ("arena", 256)), blob=b"\0" * 32, relocs=((8, 1, 16),),
                     ops=None, outputs=((b"out", 1, 0, 1, (4,), (1,)),)):
    """A plan file in tt_plan_save's format (little-endian i64 / f64 fields; thinktwice_amd/csrc/plan.cpp)."""
    import struct
    q = lambda v: struct.pack("<q", v)
    st = lambda b: q(len(b)) + b
    if ops is None:   # one call (tt_fill_u32(dst, n_words, pattern) on stream 0), one dependency (stream 1 waits for 0)
        ops = [("call", b"tt_fill_u32", 0, [(2, 1, 0, 0.0), (0, 0, 4, 0.0), (0, 0, 7, 0.0)]), ("sync", 1, 0)]
    out = q(0x314E414C50545454) + q(nstreams) + q(len(buffers))
    for name, nbytes in buffers:
        out += st(name.encode()) + q(nbytes)
    out += q(len(blob)) + blob + q(len(relocs))
    for r in relocs:
        out += q(r[0]) + q(r[1]) + q(r[2])
    out += q(len(ops))
    for o in ops:
        if o[0] == "sync":
            out += q(1) + q(o[1]) + q(o[2])
        else:
            out += q(0) + st(o[1]) + q(o[2]) + q(len(o[3]))
            for kind, buf, val, f in o[3]:
                out += q(kind) + q(buf) + q(val) + struct.pack("<d", f)
    out += q(len(outputs))
    for name, buf, off, ndim, shape, stride in outputs:
        out += st(name) + q(buf) + q(off) + q(ndim)
        out += b"".join(q(shape[i] if i < len(shape) else 0) for i in range(8))
        out += b"".join(q(stride[i] if i < len(stride) else 0) for i in range(8))
    return out


def test_plan_load_rejects_malformed_files(tmp_path):
    """tt_plan_load must apply the builder's checks to a whole file -- a relocation outside the blob would be a
    heap write in tt_plan_bind, a stream index outside the plan a wild read of streams[] in tt_plan_run."""
    import ctypes
    from thinktwice_amd import _lib
    L = _lib.lib()
    L.tt_plan_load.restype = ctypes.c_void_p
    L.tt_last_error.restype = ctypes.c_char_p

    def load(data):
        f = tmp_path / "p.plan"
        f.write_bytes(data)
        h = L.tt_plan_load(str(f).encode())
        if h:
            L.tt_plan_destroy(ctypes.c_void_p(h))
        return bool(h), L.tt_last_error().decode()

    ok, _ = load(_plan_file_bytes())
    assert ok, "the well-formed plan must load"
    call = lambda args, stream=0: [("call", b"tt_fill_u32", stream, args)]
    good = [(2, 1, 0, 0.0), (0, 0, 4, 0.0), (0, 0, 7, 0.0)]
    bad = {
        "stream count": _plan_file_bytes(nstreams=-1),
        "stream count (huge)": _plan_file_bytes(nstreams=1 << 20),
        "outside the blob": _plan_file_bytes(relocs=((28, 1, 0),)),            # 28 + 8 > 32
        "outside the blob (negative)": _plan_file_bytes(relocs=((-8, 1, 0),)),
        "points outside buffer": _plan_file_bytes(relocs=((0, 5, 0),)),
        "points outside buffer (offset)": _plan_file_bytes(relocs=((0, 1, 1 << 20),)),
        "runs on stream": _plan_file_bytes(ops=call(good, stream=9)),
        "stream dependency": _plan_file_bytes(ops=[("sync", 0, 7)]),
        "stream dependency (negative)": _plan_file_bytes(ops=[("sync", -1, 0)]),
        "argument kind": _plan_file_bytes(ops=call([(9, 1, 0, 0.0)] + good[1:])),
        "device pointer outside buffer": _plan_file_bytes(ops=call([(2, 40, 0, 0.0)] + good[1:])),
        "device pointer outside buffer (offset)": _plan_file_bytes(ops=call([(2, 1, 1 << 30, 0.0)] + good[1:])),
        "blob argument": _plan_file_bytes(ops=call([(3, 0, 4096, 0.0)] + good[1:])),
        "argument count mismatch": _plan_file_bytes(ops=call(good[:2])),
        "unknown entry": _plan_file_bytes(ops=[("call", b"tt_no_such_entry", 0, good)]),
        "decoder marker": _plan_file_bytes(outputs=((b"__decoder_first_op", -1, 99, 0, (), ()),)),
        "output outside buffer": _plan_file_bytes(outputs=((b"o", 7, 0, 1, (4,), (1,)),)),
        "truncated": _plan_file_bytes()[:-40],
    }
    for what, data in bad.items():
        ok, err = load(data)
        assert not ok, f"{what}: a malformed plan loaded"
        assert what.split(" (")[0] in err, (what, err)


def test_plan_arena_compaction_reuses_memory_by_liveness_and_stream():
    """tt_plan_compact_arena (round 4): temporaries of the recorded forward are re-placed by liveness.  Memory moves on only
    between allocations whose every use is on one stream; buffers seen from two streams and outputs keep their own; pointers
    inside argument blobs count as uses; a pointer outside every declared allocation leaves the plan untouched."""
    import ctypes
    from thinktwice_amd import _lib
    L = _lib.lib()
    L.tt_plan_create.restype = ctypes.c_void_p
    L.tt_plan_compact_arena.restype = ctypes.c_longlong
    L.tt_plan_add_blob.restype = ctypes.c_longlong
    ll, ci = ctypes.c_longlong, ctypes.c_int

    def build(allocs, calls, outputs=(), bad_ptr=None):
        """calls: (stream, [arena offsets used as the dst of a tt_fill_u32])."""
        p = ctypes.c_void_p(L.tt_plan_create())
        assert L.tt_plan_set_buffer(p, 0, ll(64), b"weights") == 0 and L.tt_plan_set_buffer(p, 1, ll(1 << 20), b"arena") == 0
        for off, n in allocs:
            assert L.tt_plan_add_arena_alloc(p, ll(off), ll(n)) == 0
        for stream, offs in calls:
            for off in offs:
                kinds, bufs = (ci * 3)(2, 0, 0), (ci * 3)(1, 0, 0)
                iv, fv = (ll * 3)(off, 1, 0), (ctypes.c_double * 3)()
                assert L.tt_plan_add_call(p, b"tt_fill_u32", ci(3), kinds, bufs, iv, fv, ci(stream)) == 0
        for name, off in outputs:
            sh = (ll * 8)(4)
            assert L.tt_plan_add_output(p, name, ci(1), ll(off), ci(1), sh, sh) == 0
        if bad_ptr is not None:
            kinds, bufs = (ci * 3)(2, 0, 0), (ci * 3)(1, 0, 0)
            assert L.tt_plan_add_call(p, b"tt_fill_u32", ci(3), kinds, bufs, (ll * 3)(bad_ptr, 1, 0), (ctypes.c_double * 3)(), ci(0)) == 0
        return p

    def arg_offsets(p, tmp):
        """arena offsets of the calls after compaction, read back through save + the file format."""
        import struct
        assert L.tt_plan_save(p, tmp.encode()) == 0
        raw = open(tmp, "rb").read()
        offs, i = [], 0
        needle = struct.pack("<q", len(b"tt_fill_u32")) + b"tt_fill_u32"
        while True:
            i = raw.find(needle, i)
            if i < 0:
                break
            j = i + len(needle) + 16            # stream, nargs
            kind, buf, val = struct.unpack_from("<qqq", raw, j)
            assert kind == 2 and buf == 1
            offs.append(val)
            i = j
        return offs

    import tempfile
    tmp = tempfile.mktemp(suffix=".plan")
    A, B, C, D = (0, 1024), (1024, 1024), (2048, 512), (4096, 1024)
    # stream 0: A used by ops 0-1, then B (op 2), then C (op 3), D is an output written by op 4: B reuses A's block, C fits too
    p = build([A, B, C, D], [(0, [0, 512]), (0, [1024]), (0, [2048]), (0, [4096])], outputs=[(b"out", 4096)])
    nb = L.tt_plan_compact_arena(p, ci(1), ll(256))
    offs = arg_offsets(p, tmp)
    assert offs[0] == 0 and offs[1] == 512                      # A at 0 (interior pointer keeps its distance)
    assert offs[2] == 0                                         # B took A's block (A died at op 1, B born at op 2)
    assert offs[3] in (0, 1024) and offs[3] != offs[4]          # C reuses B's (or a remainder); never the output's
    assert nb <= 1024 + 1024 + 256 and nb < 5120                # smaller than the bump layout
    L.tt_plan_destroy(p)
    # two streams: B lives on stream 1 -> may not take A's block (A's uses are on stream 0); E touched by both streams keeps its own
    E = (5120, 1024)
    p = build([A, B, E], [(0, [0]), (1, [1024]), (0, [5120]), (1, [5120 + 4])])
    assert L.tt_plan_compact_arena(p, ci(1), ll(256)) == 3 * 1024
    offs = arg_offsets(p, tmp)
    assert len({offs[0], offs[1], offs[2]}) == 3 and offs[3] == offs[2] + 4
    L.tt_plan_destroy(p)
    # a pointer in no declared allocation: refused, plan unchanged
    p = build([A], [(0, [0])], bad_ptr=3000)
    assert L.tt_plan_compact_arena(p, ci(1), ll(256)) < 0
    assert arg_offsets(p, tmp) == [0, 3000]
    L.tt_plan_destroy(p)
    os.remove(tmp)


def test_wide_chain_workspace_and_argument_checks_are_host_side():
    """tt_mlp_chain_wide_workspace_bytes is pure host arithmetic: one fragment-major f32 scratch per stage output a later stage
    reads, rows padded to the workgroup's row blocks (1, 2 or 4 x 32), columns to 32, each rounded to 256 B; and the launch entry
    refuses a missing / short workspace and an over-wide grid before touching the device."""
    import ctypes
    from thinktwice_amd import _lib, ops
    L = _lib.lib()
    L.tt_mlp_chain_wide_workspace_bytes.restype = ctypes.c_longlong

    def stages(dims, srcs):
        arr = (ops._ChainStage * (len(dims) - 1))()
        for i in range(len(dims) - 1):
            arr[i].w = 0x1000
            arr[i].K, arr[i].Kp, arr[i].N = dims[i], (dims[i] + 15) // 16 * 16, dims[i + 1]
            arr[i].in_sel = srcs[i]
        return arr

    merge = stages([1024, 512, 512, 256], [-1, 0, 1])                      # outputs of stages 0 and 1 are read later
    ws = lambda R, arr, n: int(L.tt_mlp_chain_wide_workspace_bytes(ctypes.c_longlong(R), ctypes.c_int(n), arr))
    assert ws(4, merge, 3) == 2 * 32 * 512 * 4                             # one row block
    assert ws(33, merge, 3) == 2 * 64 * 512 * 4                            # two row blocks in one workgroup
    assert ws(480, merge, 3) == 2 * 512 * 512 * 4                          # 15 row blocks -> 4 groups of 4
    narrow = stages([256, 4, 8], [-1, 0])                                  # a 4-wide intermediate still takes a 32-column block
    assert ws(1, narrow, 2) == 32 * 32 * 4
    assert ws(1, stages([256, 256], [-1]), 1) == 256                       # nothing kept: the minimum
    assert ws(0, merge, 3) == -1 and ws(4, merge, 0) == -1
    x = ctypes.c_void_p(0x2000)
    call = lambda R, g, wsp, nbytes: L.tt_mlp_chain_wide(x, ctypes.c_longlong(R), ctypes.c_int(1024), ctypes.c_int(3), merge,
                                                         ctypes.c_int(g), ctypes.c_void_p(wsp), ctypes.c_longlong(nbytes), None)
    assert call(4, 16, 0, 1 << 20) != 0 and b"workspace" in L.tt_last_error()
    assert call(4, 16, 0x4000, 1024) != 0 and b"workspace" in L.tt_last_error()
    assert call(480, 65, 0x4000, 1 << 24) != 0 and b"co-resident" in L.tt_last_error()


def test_splitk_query_answers_for_the_bf16x3_tile_when_the_operand_is_there():
    """tt_conv2d_splitk_slices (host arithmetic only): a few-row long-K layer WITH a bf16x3 operand is planned on the 64-wide
    bf16x3 tile -- tiles_m x Cout / 64 workgroups, K tiles of 32 dealt over <= 16 ranges of >= 8 tiles until ~512 workgroups; without
    the operand the exact-f32 register-staged kernel answers as before; many-row layers do not split."""
    import ctypes
    from thinktwice_amd import _lib, ops
    L = _lib.lib()

    def slices(N, H, W, Cin, Cout, k, x3):
        d = ops._ConvDesc()
        d.in_, d.weight, d.out = 0x10000, 0x20000, 0x30000
        d.weight_x3 = 0x40000 if x3 else None
        d.N, d.H, d.W, d.Cin, d.in_cstride = N, H, W, Cin, Cin
        d.Cout, d.KH, d.KW, d.stride, d.pad, d.dil = Cout, k, k, 1, k // 2, 1
        d.OH, d.OW, d.out_cstride = H, W, Cout
        d.dtype = d.out_dtype = _lib.TT_F32
        return int(L.tt_conv2d_splitk_slices(ctypes.byref(d)))

    # ResNet layer 4 of a batch-1 tick: 8 images of 14 x 28 -> M = 3136 = 13 row tiles
    assert slices(8, 14, 28, 512, 512, 3, True) == 4          # 104 tiles x 4 ranges of 36 K tiles
    assert slices(8, 14, 28, 2048, 512, 1, True) == 4         # K = 2048: 64 K tiles, 4 ranges of 16
    assert slices(2, 4, 8, 512, 512, 3, True) == slices(2, 4, 8, 512, 512, 3, False)      # M = 64: not the bf16x3 tile's
    assert slices(16, 4, 8, 512, 512, 3, True) == 16          # the F7-sized trunk: M = 512, 16 tiles x 16 ranges of 9
    assert slices(8, 56, 112, 512, 512, 3, True) == 0         # M = 50,176: the chip is full without a split
    f32 = slices(8, 14, 28, 512, 512, 3, False)
    assert f32 > 0 and f32 != 4                               # the exact-f32 kernel plans its own split


def test_round5_host_side_contracts():
    """Host-only pieces of round 5: (1) the by-epoch cosine refuses to degrade silently; (2) tt_plan_add_sync refuses
    while relocations of a not-yet-added call are pending (the liveness analysis attributes a blob's pointers to the op index
    current when they were declared); (3) tt_voxel_pool_workspace_bytes covers the per-launch counting sort's layout (packed key /
    rank words, per-(chunk, cell) table, order, per-sample cell / segment tables, one partial row per segment) and answers 0 only
    for shapes no workspace path takes; (4) the agent tick's world -> ego target point equals the oracle's restatement of
    thinktwice_agent.py:354-360; (5) the fault entries answer without a device."""
    import ctypes
    import numpy as np
    from oracle import agent_ref as R
    from thinktwice_amd import _lib, agent_tick
    from thinktwice_amd.optim import warmup_cosine_lr
    with pytest.raises(ValueError, match="iters_per_epoch"):
        warmup_cosine_lr(1e-4, 10, 1000)
    assert warmup_cosine_lr(1e-4, 10, 1000, by_epoch=False) > 0
    L = _lib.lib()
    L.tt_plan_create.restype = ctypes.c_void_p
    plan = ctypes.c_void_p(L.tt_plan_create())
    try:
        L.tt_plan_add_blob.restype = ctypes.c_longlong
        blob = (ctypes.c_char * 64)()
        off = L.tt_plan_add_blob(plan, blob, ctypes.c_longlong(64))
        assert off >= 0
        assert L.tt_plan_add_reloc(plan, ctypes.c_longlong(off), ctypes.c_int(1), ctypes.c_longlong(0)) == 0
        assert L.tt_plan_add_sync(plan, ctypes.c_int(0), ctypes.c_int(1)) != 0
        assert b"relocation" in L.tt_last_error()
    finally:
        L.tt_plan_destroy(plan)
    ws = lambda B, Np, C, X, Y: int(L.tt_voxel_pool_workspace_bytes(ctypes.c_int(B), ctypes.c_int(Np), ctypes.c_int(C),
                                                                    ctypes.c_int(X), ctypes.c_int(Y)))
    B, Np, C, cells = 8, 501760, 256, 441
    al = lambda v: (v + 255) // 256 * 256
    cps, segcap = (Np + 8191) // 8192, Np // 64 + cells + 1
    want = al(4 * B * Np) + al(4 * B * cps * cells) + al(4 * B * Np) + 2 * al(4 * B * (cells + 1)) + al(4 * B * segcap * C)
    two_phase = B * ((Np + 2047) // 2048) * 64 * C * 4 + B * cells * ((Np + 2047) // 2048) * 4 + 256
    assert ws(B, Np, C, 21, 21) == max(want, two_phase)
    assert ws(1, 100, 256, 21, 21) == 0                      # tiny inputs: the single-pass kernel
    assert ws(1, 600000, 256, 21, 21) > 0                    # > 524,288 points per sample: the two-phase kernel still answers
    rng = np.random.default_rng(1)
    for _ in range(20):
        pos, yaw, tgt = rng.uniform(-50, 50, 2), float(rng.uniform(-4, 4)), rng.uniform(-50, 50, 2)
        want_t = R.offset_then_rotate(np.array([[tgt[1], -tgt[0]]]), np.stack([pos[1], -pos[0]], axis=-1), yaw).squeeze(0)
        got_t = agent_tick.offset_then_rotate((tgt[1], -tgt[0]), (pos[1], -pos[0]), yaw)
        np.testing.assert_allclose(got_t, want_t, rtol=0, atol=1e-12)
    assert L.tt_clear_device_faults() in (0, -2) and L.tt_device_faults() in (0, -2)


def test_pair_formats_hold_hi_and_lo_halves_in_the_documented_places():
    """weights.split_pairs_x3 / split_pairs_h2: per 16 K elements [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15] with hi = rne(v), lo = rne(v - hi)
    (bf16 / IEEE half) -- the layout tt_conv_desc.weight_x3 / in_pair / out_pair / weight_h2 document and the kernels' fragment reads
    assume (hi chunk 4 kc + h, lo chunk that + 2)."""
    import torch
    from thinktwice_amd import weights
    g = torch.Generator().manual_seed(0)
    w = torch.randn(6, 3, 3, 32, generator=g)
    px = weights.split_pairs_x3(w)
    assert px.shape == w.shape and px.dtype == torch.float32
    pairs = px.reshape(6, 9 * 32 // 16, 16).view(torch.bfloat16).reshape(6, 18, 2, 16)      # [row][group][hi | lo][16]
    flat = w.reshape(6, 18, 16)
    hi = flat.to(torch.bfloat16)
    assert torch.equal(pairs[:, :, 0], hi) and torch.equal(pairs[:, :, 1], (flat - hi.float()).to(torch.bfloat16))
    assert float((pairs[:, :, 0].float() + pairs[:, :, 1].float() - flat).abs().max()) < 2 ** -15 * float(flat.abs().max())
    ph = weights.split_pairs_h2(w)
    assert ph.dtype == torch.float16 and tuple(ph.shape) == (6, 3, 3, 64)
    ph = ph.reshape(6, 18, 2, 16)
    h16 = flat.to(torch.float16)
    assert torch.equal(ph[:, :, 0], h16) and torch.equal(ph[:, :, 1], (flat - h16.float()).to(torch.float16))
    assert float((ph[:, :, 0].float() + ph[:, :, 1].float() - flat).abs().max()) < 2 ** -20 * float(flat.abs().max())
    assert weights.is_x3("f32x3") and weights.is_x3("f32x3h") and not weights.is_x3(torch.float32)
    assert weights.storage_dtype("f32x3h") == torch.float32 and weights.storage_dtype("h2") == torch.float16


def test_decoder_scratch_queries_and_null_scratch_refusals():
    """Round 6: tt_dec_flatten / tt_dec_gru / tt_dec_bev_update run on more than one workgroup per sample and take a scratch buffer
    sized by a host-side query (no GPU needed); a null scratch pointer is refused before anything is launched."""
    import ctypes
    from thinktwice_amd import _lib
    L = _lib.lib()
    for f in (L.tt_dec_flatten_scratch_floats, L.tt_dec_gru_scratch_floats, L.tt_dec_bev_update_scratch_floats):
        f.restype = ctypes.c_longlong
    # grid2feat tail: x4, MLP4's two hidden maps, pooled fc1 output, gate, block output; the 2x2 level likewise; fc0's output
    per_map = 2048 + 4096 + 2048 + 128 + 128 + 2048 + 1024 + 2048 + 1024 + 256 + 256 + 1024 + 512
    assert L.tt_dec_flatten_scratch_floats(4) == 4 * per_map and L.tt_dec_flatten_scratch_floats(35) == 35 * per_map
    assert L.tt_dec_gru_scratch_floats(8) == 8 * (3 * 448 * 32 + 2)             # gate, previous state, new state + two flag words
    assert L.tt_dec_bev_update_scratch_floats(8) == 8 * (4 * 448 * 32 + 1)      # four partial maps + a ticket
    one = ctypes.c_void_p(16)                                                     # any non-null, 16 B aligned address: never dereferenced
    assert L.tt_dec_flatten(ctypes.c_int(1), one, one, None, None, one, one, one, one, None) == -1
    assert b"null" in L.tt_last_error()
    assert L.tt_dec_gru(ctypes.c_int(1), one, one, one, None, one, one, one, one, one, one, one, one, one, None) == -1
    assert L.tt_dec_bev_update(ctypes.c_int(1), one, one, one, ctypes.c_longlong(0), None, ctypes.c_longlong(0), None, one, one,
                               one, one, None) == -1
