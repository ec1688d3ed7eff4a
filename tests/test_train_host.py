"""Host logic of the training iteration that needs no GPU: the tape's storage-mirrored gradient buffers, the mapping of
prepared-weight gradients back to the reference's parameter layouts, the trainable / buffer split of a state_dict and
`_parse_losses`."""
import os

import numpy as np
import pytest
import torch

from thinktwice_amd import autodiff


def test_tape_gradient_buffers_mirror_the_activation_storage():
    buf = torch.zeros(4, 6)
    window = buf[:, 2:5]                                   # a channel-offset "concat" window of a wider buffer
    first_rows = buf[:2]                                   # a t[:n] view
    with autodiff.Tape() as tape:
        assert autodiff.TAPE is tape
        tape.seed(window, torch.ones(4, 3))
        tape.grad(first_rows).add_(2.0)
        seen = []
        tape.nodes.append(lambda: seen.append(("first", autodiff.TAPE)))
        tape.nodes.append(lambda: seen.append(("second", autodiff.TAPE)))
        with autodiff.paused():
            assert autodiff.TAPE is None
        assert autodiff.TAPE is tape
        tape.backward()
        assert autodiff.TAPE is tape
    assert autodiff.TAPE is None
    assert [s[0] for s in seen] == ["second", "first"] and all(s[1] is None for s in seen)   # reverse order, not recorded
    want = torch.zeros(4, 6)
    want[:, 2:5] += 1.0
    want[:2] += 2.0
    assert torch.equal(tape.grad(buf), want)               # both views addressed the same gradient memory
    assert tape.grad(window).data_ptr() == tape.grad(buf)[:, 2:5].data_ptr()
    tape.add_param_grad("w", torch.ones(3))
    tape.add_param_grad("w", torch.ones(3))
    assert torch.equal(tape.param_grads["w"], torch.full((3,), 2.0))


def test_tape_releases_activations_and_gradients_as_the_sweep_passes_their_first_node():
    """Tape(release=True) (the training step): a = 3 x (node 0), b = a * p (node 1), d loss / d b = 1.  After the backward of the
    node that FIRST referenced a storage, neither the forward tensor nor its gradient buffer is held any more; parameter
    gradients survive; the arithmetic equals the retaining tape's."""
    import gc
    import weakref

    def run(release):
        seen = {}
        p = torch.full((3,), 2.0)
        autodiff.register_param(p, "p.weight")

        def node0(tape, x, a, keys):
            def bwd():
                seen["grads inside node 0"] = set(tape.grads)
                seen["b alive inside node 0"] = refs["b"]() is not None
                tape.grad(x).add_(3.0 * tape.grad(a))
            return bwd

        def node1(tape, a, b):
            def bwd():
                tape.grad(a).add_(tape.grad(b) * p)
                tape.grad(p).add_((tape.grad(b) * a).sum(0))
            return bwd

        try:
            with autodiff.Tape(release=release) as tape:
                x = torch.arange(6, dtype=torch.float32).view(2, 3)
                a = x * 3.0
                keys = {"x": x.untyped_storage().data_ptr(), "a": a.untyped_storage().data_ptr()}
                tape._keep += [x, a]
                tape.nodes.append(node0(tape, x, a, keys))
                b = a * p
                keys["b"] = b.untyped_storage().data_ptr()
                tape._keep += [a, p, b]
                tape.nodes.append(node1(tape, a, b))
                tape.seed(b, torch.ones(2, 3))
                refs = {"x": weakref.ref(x), "a": weakref.ref(a), "b": weakref.ref(b)}
                gx = weakref.ref(tape.grad(x).untyped_storage())
                del x, a, b
                tape.backward()
                seen["d p"] = tape.grad(p).clone()      # (Tape.backward collects device parameters into param_grads itself)
            gc.collect()
            return tape, refs, keys, seen, gx
        finally:
            autodiff.clear_metas()

    tape, refs, keys, seen, gx = run(True)
    assert torch.equal(seen["d p"], torch.tensor([9.0, 15.0, 21.0]))      # sum over the rows of a = 3 x: parameters are not released
    assert keys["b"] not in seen["grads inside node 0"] and keys["a"] in seen["grads inside node 0"]
    assert not seen["b alive inside node 0"]
    assert all(r() is None for r in refs.values()), "forward tensors outlive the sweep"
    assert gx() is None, "the gradient buffer of x outlives its first node"
    tape2, refs2, keys2, seen2, gx2 = run(False)
    assert torch.equal(seen2["d p"], seen["d p"])
    assert keys2["b"] in seen2["grads inside node 0"] and seen2["b alive inside node 0"] and gx2() is not None


def test_tape_release_covers_gradients_created_during_the_sweep():
    """x0 -(n0)-> x -(n1)-> a -(n2)-> b.  Only b's gradient exists before the sweep (the loss seeds it); grad(a) and
    grad(x) are created INSIDE the closures of n2 / n1.  In release mode `a` (first referenced by n1) must be dead -- tensor and
    gradient buffer -- by the time n0's backward runs; the retaining tape keeps it."""
    import gc
    import weakref

    def run(release):
        seen = {}
        try:
            with autodiff.Tape(release=release) as tape:
                x0 = torch.arange(4, dtype=torch.float32)
                x = x0 * 2.0
                tape._keep += [x0, x]

                def n0(x0_, x_):
                    def bwd():
                        gc.collect()
                        seen["a alive in n0"] = refs["a"]() is not None
                        seen["grad(a) alive in n0"] = keys["a"] in tape.grads
                        tape.grad(x0_).add_(2.0 * tape.grad(x_))
                    return bwd
                tape.nodes.append(n0(x0, x))
                a = x + 1.0
                tape._keep += [x, a]
                tape.nodes.append((lambda x_, a_: lambda: tape.grad(x_).add_(tape.grad(a_)))(x, a))
                b = a * 3.0
                tape._keep += [a, b]
                tape.nodes.append((lambda a_, b_: lambda: tape.grad(a_).add_(3.0 * tape.grad(b_)))(a, b))
                tape.seed(b, torch.ones(4))
                refs = {"a": weakref.ref(a)}
                keys = {"a": a.untyped_storage().data_ptr()}
                gx0 = tape.grad(x0)                      # (a graph input: its gradient is what the test reads)
                del x, a, b
                tape.backward()
                seen["d x0"] = gx0.clone()
            return seen
        finally:
            autodiff.clear_metas()

    rel, keep = run(True), run(False)
    assert torch.equal(rel["d x0"], torch.full((4,), 6.0)) and torch.equal(keep["d x0"], rel["d x0"])
    assert not rel["a alive in n0"] and not rel["grad(a) alive in n0"]
    assert keep["a alive in n0"] and keep["grad(a) alive in n0"]


def test_prepared_weight_gradients_return_in_the_reference_layouts():
    """ConvMeta.place_weight_grad: prepared operands are [Cout][KH][KW][cin_pad]; the reference stores Conv2d as
    [Cout, Cin, KH, KW], Linear as [Cout, Cin], spconv as (Cout, kD, kH, kW, Cin), and some prepared convs are channel
    slices / row groups of a wider reference weight."""
    g = torch.Generator().manual_seed(0)
    tape = autodiff.Tape()
    ref = torch.randn(5, 3, 3, 3, generator=g)                                     # a Conv2d weight gradient
    prepared = torch.zeros(5, 3, 3, 4)
    prepared[..., :3] = ref.permute(0, 2, 3, 1)
    prepared[..., 3] = 99.0                                                        # channel padding: must be dropped
    autodiff.ConvMeta("a.conv", 3).place_weight_grad(tape, prepared)
    assert torch.equal(tape.param_grads["a.conv.weight"], ref)
    lin = torch.randn(7, 10, generator=g)
    autodiff.ConvMeta("a.fc", 10, kind="linear").place_weight_grad(tape, lin.view(7, 1, 1, 10))
    assert torch.equal(tape.param_grads["a.fc.weight"], lin)
    # Linear over a flattened (C=4, HW=6) map whose prepared columns run (HW, C)
    ref_hwc = torch.randn(7, 24, generator=g)                                      # reference columns run (C, HW)
    prep = ref_hwc.view(7, 4, 6).permute(0, 2, 1).reshape(7, 1, 1, 24)
    autodiff.ConvMeta("a.flat", 24, kind="linear_hwc", lo=6).place_weight_grad(tape, prep)      # lo = HW
    assert torch.equal(tape.param_grads["a.flat.weight"], ref_hwc)
    # two input-channel slices of one wider conv weight accumulate into the full gradient
    full = torch.randn(5, 8, 1, 1, generator=g)
    for lo, n in ((0, 5), (5, 3)):
        autodiff.ConvMeta("a.wide", n, kind="cin_slice", full_shape=(5, 8, 1, 1), lo=lo).place_weight_grad(
            tape, full[:, lo:lo + n].permute(0, 2, 3, 1).contiguous())
    assert torch.equal(tape.param_grads["a.wide.weight"], full)
    sp = torch.randn(6, 3, 3, 3, 4, generator=g)                                   # spconv (Cout, kD, kH, kW, Cin)
    autodiff.ConvMeta("a.sp", 4, kind="spconv", full_shape=tuple(sp.shape)).place_weight_grad(tape, sp.reshape(6, 1, 27, 4))
    assert torch.equal(tape.param_grads["a.sp.weight"], sp)
    dcn = torch.randn(8, 2, 3, 3, generator=g)                                     # grouped deformable conv: 4 rows per group
    for r0 in (0, 4):
        grp = dcn[r0:r0 + 4].permute(0, 2, 3, 1).reshape(4, 1, 9, 2)
        autodiff.ConvMeta("a.dcn", 2, kind="dcn_group", full_shape=(8, 2, 3, 3), lo=r0).place_weight_grad(tape, grp)
    assert torch.equal(tape.param_grads["a.dcn.weight"], dcn)


def test_trainable_split_matches_the_reference_gradient_golden():
    """trainer._trainable keeps exactly the reference's nn.Parameters: the 878 live + 90 dead names of golden F13 (whose
    backward ran on the instantiated reference modules), none of the registered buffers."""
    from thinktwice_amd import config, params
    from thinktwice_amd.trainer import _trainable
    here = os.path.dirname(__file__)
    pack = np.load(os.path.join(here, "golden", "f13_train_gradients_b2.npz"))
    spec = params.param_spec(config.model_config())        # name -> (shape, kind) of the full state_dict
    fake = {k: torch.zeros(shp if kind != "bn_nbt" else (), dtype=torch.int64 if kind in ("bn_nbt", "buf_voxel_num")
                           else torch.float32) for k, (shp, kind) in spec.items()}
    got = sorted(k for k, v in fake.items() if _trainable(k, v))
    want = sorted([str(n) for n in pack["names"]] + [str(n) for n in pack["dead"]])
    assert len(want) == 968 and got == want


@pytest.mark.parametrize("fname,B,hw,npts", [("f13_train_gradients_b2.npz", 2, (128, 256), 20000),
                                              ("f13b_train_gradients_fullsize_b1.npz", 1, (448, 896), 65536),
                                              ("f16_train_gradients_trainmode_b4.npz", 4, (128, 256), 20000)])
def test_gradient_goldens_are_complete_and_consistent(fname, B, hw, npts):
    """The three gradient fixtures the `-m gpu` training tests read (they do not exist on the GPU box unless committed): same
    968 parameters split into the same 878 live / 90 dead names, finite positive norms, 8 sampled entries each inside the
    tensor, the documented batch / image size / point count."""
    from thinktwice_amd import config, params
    pack = np.load(os.path.join(os.path.dirname(__file__), "golden", fname))
    names, dead = [str(n) for n in pack["names"]], [str(n) for n in pack["dead"]]
    assert len(names) == 878 and len(dead) == 90 and not set(names) & set(dead)
    assert [int(v) for v in pack["meta"][:4]] == [B, hw[0], hw[1], npts]
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "f13_train_gradients_b2.npz"))
    assert sorted(names) == sorted(str(n) for n in ref["names"]) and sorted(dead) == sorted(str(n) for n in ref["dead"])
    norms = pack["norms"]
    assert norms.shape == (878,) and np.isfinite(norms).all() and (norms >= 0).all() and float(np.isfinite(pack["total_loss"][0]))
    spec = params.param_spec(config.model_config(final_dim=hw))
    assert pack["idx"].shape == (878, 8) and pack["samples"].shape == (878, 8)
    for n, idx in zip(names, pack["idx"]):
        assert int(idx.max()) < int(np.prod(spec[n][0])), n


def test_parse_losses_single_process():
    from thinktwice_amd.losses import parse_losses
    losses = {"wp_loss": torch.tensor(1.0), "value_loss": torch.tensor([[0.5], [1.5]]),
              "lateral_offset": torch.tensor(10.0), "aux": [torch.tensor(2.0), torch.tensor([4.0, 6.0])]}
    loss, log_vars = parse_losses(losses)
    assert abs(float(loss) - 2.0) < 1e-6                                           # wp_loss + mean(value_loss): 'loss' names only
    assert list(log_vars) == ["wp_loss", "value_loss", "lateral_offset", "aux", "loss"]
    assert log_vars["aux"] == 7.0 and log_vars["lateral_offset"] == 10.0 and abs(log_vars["loss"] - 2.0) < 1e-6
    assert all(isinstance(v, float) for v in log_vars.values())


def test_tape_metadata_tables_are_scoped_per_model_and_never_alias():
    """the tensor -> parameter-name tables of the training tape were process-global id(tensor) dicts: one
    Trainer re-preparing its operands wiped every other model's entries, and a recycled id() could hand a dead tensor's entry to
    a new one.  They are weak, identity-checked and owner-scoped now."""
    import gc
    import torch
    from thinktwice_amd import autodiff as A
    t = A.MetaTable()

    class Model:
        pass
    m1, m2 = Model(), Model()
    a, b = torch.zeros(3), torch.zeros(3)
    with A.owned_by(m1):
        t[a] = "A"
    with A.owned_by(m2):
        t[b] = "B"
    assert t.get(a) == "A" and t.get(b) == "B" and len(t) == 2 and a in t
    t.clear(m1)                                    # model 1 re-prepares its operands ...
    assert t.get(a) is None and t.get(b) == "B"    # ... model 2 keeps its entries
    assert t.values(m2) == ["B"] and t.values(m1) == []
    del b
    gc.collect()
    assert len(t) == 0                             # entries die with their tensors
    c, d = torch.zeros(2), torch.zeros(2)
    t[c] = "C"
    t._d[id(d)] = t._d[id(c)]                      # what a recycled id() amounts to: c's entry found under d's id
    assert t.get(d) is None and t.get(c) == "C"
    # the module-level tables are MetaTables and clear_metas(owner) only touches that owner
    x, y = torch.zeros(1), torch.zeros(1)
    with A.owned_by(m1):
        A.CONV_META[x] = "x"
    with A.owned_by(m2):
        A.CONV_META[y] = "y"
    A.clear_metas(m1)
    assert A.CONV_META.get(x) is None and A.CONV_META.get(y) == "y"
    A.clear_metas(m2)


def test_lr_schedule_is_mmcv_cosine_by_epoch_with_per_iteration_warmup():
    """the reference's lr_config (configs/thinktwice.py:286-291) under EpochBasedRunner is mmcv's
    CosineAnnealingLrUpdaterHook with its default by_epoch=True: the regular rate steps per EPOCH
    (annealing_cos(base, base * min_lr_ratio, epoch / max_epochs)), the linear warmup multiplies it per ITERATION."""
    import math
    from thinktwice_amd.optim import warmup_cosine_lr
    base, ipe, epochs = 1e-4, 500, 60
    total = ipe * epochs

    def mmcv(it):       # mmcv/runner/hooks/lr_updater.py: CosineAnnealingLrUpdaterHook.get_lr + LrUpdaterHook.get_warmup_lr
        epoch = it // ipe
        target = base * 1e-3
        regular = target + 0.5 * (base - target) * (math.cos(math.pi * epoch / epochs) + 1)
        if it < 1000:
            k = (1 - it / 1000) * (1 - 1.0 / 3)
            return regular * (1 - k)
        return regular
    for it in (0, 1, 499, 500, 999, 1000, 1001, 12_345, total - 1):
        assert abs(warmup_cosine_lr(base, it, total, iters_per_epoch=ipe) - mmcv(it)) < 1e-18 + 1e-12 * base, it
    # constant inside an epoch after the warmup, a step at the epoch boundary
    a, b, c = (warmup_cosine_lr(base, it, total, iters_per_epoch=ipe) for it in (1500, 1999, 2000))
    assert a == b and c < b
    # the per-iteration form stays available
    assert warmup_cosine_lr(base, 1500, total, iters_per_epoch=ipe, by_epoch=False) > warmup_cosine_lr(base, 1999, total, by_epoch=False)
