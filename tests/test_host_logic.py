"""Host-side logic that needs no GPU: the closed-loop previous-sweep cache driver and the img_metas constants that
a captured graph takes as inputs."""
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
