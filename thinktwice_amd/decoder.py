"""`ThinkTwiceDecoder` -- host-side mirror of the reference head
(open_loop_training/code/model_code/dense_heads/thinktwice_decoder.py:262-533, inference path),
with its LookModule / SpatialCrossAttention / MSDeformableAttention3D / SpatialGRU sub-modules
(thinktwice_decoder.py:26-260, multi_scale_deformable_attn_function.py:197-526, utils.py:53-106).

Same state_dict keys and the same output dict (keys/shapes of SURVEY 8a A22).  All query packing is
done on the device (no host sync inside the 5-layer loop); dead branches of the reference
(LiDAR look features zeroed at DEC:186, PredictionModule.ffn discarded at DEC:44-46) are not
computed -- their parameters are accepted and ignored.  With `teacher_forcing_data` the decoder also runs the
teacher-forcing pass of the training step (DEC:491-533).  Inside `autodiff.Tape` the layer-wise path runs on one stream
and every op records its backward (trainer.py); the losses are thinktwice_amd/losses.py.
"""
import os

import torch

from . import _lib, autodiff, ops, weights
from .layers import conv_from_sd, conv_from_weight, linear_from_sd, rows, unrows
from .registry import HEADS

F32 = torch.float32
# A/B switch, default off (written after the round's GPU budget was spent, not yet run on hardware): assemble each
# concatenated MLP input with ONE tt_concat_rows launch instead of one `ew` copy per piece (~25 launches per layer)
_FUSED_CONCAT = os.environ.get("TT_DEC_FUSED_CONCAT", "0") == "1"


def _mlp(sd, name, idx, dev, last_act=False, in_pad=None):
    out = []
    for n, j in enumerate(idx):
        act = "relu" if (n < len(idx) - 1 or last_act) else "none"
        out.append(linear_from_sd(sd, f"{name}.{j}", dev, act=act, in_pad=in_pad if n == 0 else None))
    return out


def _run(seq, x):
    for l in seq:
        x = l(x)
    return x


class _GRU:
    """SpatialGRU (utils.py:53-106): conv-GRU over 4 steps, input constant over the map."""

    def __init__(self, sd, p, dev):
        def two(name):
            return (conv_from_sd(sd, f"{p}.{name}.0", F32, dev, pad=1, act="relu", cin_pad=40),
                    conv_from_sd(sd, f"{p}.{name}.2", F32, dev, pad=1))
        self.update, self.reset, self.tilde = two("conv_update"), two("conv_reset"), two("conv_state_tilde")
        self.dec0 = conv_from_sd(sd, p + ".conv_decoder.0", F32, dev, pad=1, act="relu")
        self.dec2 = conv_from_sd(sd, p + ".conv_decoder.2", F32, dev, pad=1)
        for pair in (self.update, self.reset):
            pair[1].act = _lib.ACT_SIGMOID

    def __call__(self, inp6, state, fut):
        """inp6 (B,4,6) f32; state (B,H,W,32); fut (B,4,H,W,32) output buffer."""
        B, H, W, _ = state.shape
        dev = state.device
        xs = torch.zeros(B, H, W, 40, dtype=F32, device=dev)      # [x 6 | state 32 | pad 2]
        xr = torch.zeros(B, H, W, 40, dtype=F32, device=dev)
        for t in range(4):
            if autodiff.TAPE is not None and t > 0:
                # the training tape keeps every step's conv inputs: no reuse of the two staging buffers across steps
                xs, xr = torch.zeros_like(xs), torch.zeros_like(xr)
            x_t = inp6[:, t]                                       # (B,6) row-strided view
            ops.broadcast_rows(x_t, xs, out_coff=0)
            ops.broadcast_rows(x_t, xr, out_coff=0)
            ops.ew(3, state, out=xs, C=32, out_coff=6)
            u = self.update[1](self.update[0](xs))
            r = self.reset[1](self.reset[0](xs))
            ops.ew(1, state, b=r, out=xr, C=32, out_coff=6)        # (1 - r) * state
            cand = self.tilde[1](self.tilde[0](xr))
            new_state = torch.empty_like(state)
            ops.ew(2, state, b=cand, g=u, out=new_state, C=32)     # (1-u)*state + u*cand
            state = new_state
            self.dec2(self.dec0(state), out=fut[:, t])
        return fut


def _layer_norm_params(sd, name, dev):
    gamma = sd[name + ".weight"].to(dev, F32).contiguous()
    beta = sd[name + ".bias"].to(dev, F32).contiguous()
    autodiff.LN_META[gamma] = (name + ".weight", name + ".bias")
    return gamma, beta


class _Layer:
    def __init__(self, sd, p, dev, dtype):
        self.gru = _GRU(sd, p + ".prediction_module.spatial_gru", dev)
        c = p + ".look_module.cam_look_module"
        self.q_ln = _layer_norm_params(sd, c + ".query_linear.0", dev)
        self.q1 = linear_from_sd(sd, c + ".query_linear.1", dev, act="gelu", in_pad=1544)
        self.q3 = linear_from_sd(sd, c + ".query_linear.3", dev, act="gelu")
        d = c + ".deformable_attention"
        self.off = linear_from_sd(sd, d + ".sampling_offsets", dev)
        self.aw = linear_from_sd(sd, d + ".attention_weights", dev)
        self.vproj = linear_from_sd(sd, d + ".value_proj", dev, dtype=dtype)
        self.vproj_w = sd[d + ".value_proj.weight"].to(dev, F32)
        self.vproj_b = sd[d + ".value_proj.bias"].to(dev, F32)
        self.vproj_name = d + ".value_proj.weight"
        self.ffn_ln = _layer_norm_params(sd, c + ".ffn.norm", dev)
        self.ffn1 = linear_from_sd(sd, c + ".ffn.w_1", dev, act="gelu")
        self.ffn2 = linear_from_sd(sd, c + ".ffn.w_2", dev)
        self.o_ln = _layer_norm_params(sd, c + ".output_proj.0", dev)
        self.o1 = linear_from_sd(sd, c + ".output_proj.1", dev, act="gelu")
        self.o3 = linear_from_sd(sd, c + ".output_proj.3", dev)
        self.mlp_ln = _layer_norm_params(sd, p + ".mlp.0", dev)
        self.mlp1 = linear_from_sd(sd, p + ".mlp.1", dev, act="relu")
        self.mlp4 = linear_from_sd(sd, p + ".mlp.4", dev, act="relu")
        self.traj = _mlp(sd, p + ".traj_offset_module", (0, 2, 4), dev, in_pad=516)
        self.ctrl = _mlp(sd, p + ".ctrl_offset_module", (0, 2, 4), dev)
        self.bev0 = conv_from_sd(sd, p + ".BEV_feat_update_module.0", F32, dev, pad=1, act="relu")
        self.bev2 = conv_from_sd(sd, p + ".BEV_feat_update_module.2", F32, dev, pad=1)
        self.flat0 = linear_from_sd(sd, p + ".flattened_BEV_feat_update_module.0", dev, act="relu")
        self.flat2 = linear_from_sd(sd, p + ".flattened_BEV_feat_update_module.2", dev)


# Composite decoder: "sample first, project after" (csrc/look_module.hip msda_sample_proj_ln_kernel) instead of projecting every
# position of every level for all five layers up front.  TT_DEC_SAMPLE_FIRST=0 (test hook: tests/test_decoder_fused.py compares
# the two forms): the value GEMM + tt_msda_sample_ln
SAMPLE_FIRST = os.environ.get("TT_DEC_SAMPLE_FIRST", "1") != "0"


@HEADS.register_module()
class ThinkTwiceDecoder:
    def __init__(self, config=None, bev_h=None, bev_w=None, BEV_feat_dim=256, flattened_BEV_feat_dim=256,
                 dtype=torch.float32, device="cuda", **kwargs):
        self.config = config
        self.bev_h, self.bev_w = bev_h, bev_w
        self.wdtype = dtype                               # precision mode of the FPN-side GEMMs (may be weights.X3)
        self.dtype = weights.storage_dtype(dtype)
        self.device = torch.device(device)
        self.refine_num = config["refine_num"]
        self.loaded = False
        self._branch = None          # second stream for the prediction branch of each refinement layer
        self._vstream = None         # third stream for the layer-independent value projections

    def load_state_dict(self, sd, prefix="decoder"):
        p, dev = prefix, self.device
        self.speed = _mlp(sd, p + ".speed_branch", (0, 2, 4), dev)
        self.join_traj = _mlp(sd, p + ".join_traj", (0, 2, 4), dev, last_act=True)
        self.value_traj = _mlp(sd, p + ".value_branch_traj", (0, 2, 4), dev)
        self.output_traj = _mlp(sd, p + ".output_traj", (0, 2), dev)
        self.join_ctrl = _mlp(sd, p + ".join_ctrl", (0, 2, 4), dev, last_act=True)
        self.value_ctrl = _mlp(sd, p + ".value_branch_ctrl", (0, 2, 4), dev)
        self.policy = _mlp(sd, p + ".policy_head", (0, 2), dev, last_act=True)
        self.dist_mu = _mlp(sd, p + ".dist_mu", (0, 2), dev)
        self.dist_sigma = _mlp(sd, p + ".dist_sigma", (0, 2), dev)
        self.fpn_linear = [conv_from_sd(sd, f"{p}.fpn_linear{i}", self.wdtype, dev) for i in range(4)]
        self.prefix = p
        # (own copies: the training tape sizes their gradient buffers by storage, and a checkpoint tensor may be a view of
        # something much larger -- the trainer's flat master buffer)
        self.temporal = autodiff.register_param(sd[p + ".temporal_embedding"].to(dev, F32).clone().contiguous(),
                                                p + ".temporal_embedding")
        self.static = autodiff.register_param(sd[p + ".static_embedding"].to(dev, F32).clone().contiguous(),
                                              p + ".static_embedding")
        cams = sd[p + ".cams_embeds"].to(dev, F32)
        lvls = sd[p + ".level_embeds"].to(dev, F32)
        self.cams_embeds, self.level_embeds = cams, lvls
        self.layers = [_Layer(sd, f"{p}.decoder_layers.{L}", dev, self.wdtype) for L in range(self.refine_num)]
        # value_proj(feat + cam_embed + level_embed) = value_proj(feat) + per-(level, cam) shift  (DEC:392-393)
        for lay in self.layers:
            emb = cams.view(1, 4, 256) + lvls.view(4, 1, 256)                       # (lvl, cam, 256)
            # W e (bias added by shift): 4 rows through the library's exact-f32 linear (tt_conv2d_fwd), not a torch GEMM
            w_e = conv_from_weight(lay.vproj_w.reshape(256, 1, 1, 256).clone(), F32)    # (a 16 B aligned copy: under the trainer
            # vproj_w is a view into the flat master buffer at an arbitrary offset)
            lay.vshift = [unrows(w_e(rows(emb[l].contiguous()))).contiguous() for l in range(4)]
            # operands of the sample-first form (tt_msda_sample_proj_ln): W^T [in][out], W e per (level, camera)
            lay.vproj_wT = lay.vproj_w.t().contiguous()
            lay.vproj_b = lay.vproj_b.contiguous()
            lay.vshift_lc = torch.stack(lay.vshift, 0).contiguous()                  # (level, cam, 256)
        self.vproj_all_shift = torch.cat([lay.vproj.shift for lay in self.layers], 0).contiguous()  # (L*256,)
        self.vproj_all = conv_from_weight(torch.cat([lay.vproj.w for lay in self.layers], 0).contiguous(),
                                          self.wdtype, shift=self.vproj_all_shift)                  # (L*256,1,1,256)
        self.vshift_all = [torch.cat([lay.vshift[l] for lay in self.layers], 1).contiguous() for l in range(4)]
        # composite execution (decoder_fused.py: ~16 launches per refinement layer, bf16x3 arithmetic).  Default: on for
        # every precision mode except the exact-f32 parity mode; TT_DEC_FUSED=0/1 overrides.
        want = os.environ.get("TT_DEC_FUSED")
        use_fused = (self.wdtype != torch.float32) if want is None else want == "1"
        self.fused = None
        if use_fused:
            from .decoder_fused import FusedDecoder
            self.fused = FusedDecoder(self, sd, p)
        self.loaded = True
        return self

    # ------------------------------------------------------------------ look module
    def _project_values(self, mlvl, B, S):
        """value_proj of ALL refinement layers as one GEMM per FPN level (Cout = layers x 256): the projections depend
        only on the FPN maps, so the activation is read once instead of once per layer (these GEMMs are HBM-bound,
        K = N = 256) and the whole thing leaves the serial layer loop.  Layer L's attention samples the channel
        window [256 L, 256 L + 256) (tt_msda_sample_strided)."""
        C = 256 * len(self.layers)
        value = torch.empty(B * 4, S, C, dtype=self.dtype, device=mlvl[0].device)
        start = 0
        for l, m in enumerate(mlvl):
            hw = m.shape[1] * m.shape[2]
            self.vproj_all(m, shift_n=self.vshift_all[l], shift_n_mod=4,
                           out=value[:, start:start + hw].unflatten(1, (m.shape[1], m.shape[2])), out_nstride=S * C)
            start += hw
        return value

    def _project_values_train(self, lay, mlvl, B, S):
        """One refinement layer's value projection under the training tape: the layer's own value_proj (so its weight and
        bias gradients land under the reference names) with the per-(level, camera) embedding shift as a recorded input."""
        autodiff.TAPE.value_shift(lay.vshift, lay.vproj_w, lay.vproj_name, self.cams_embeds, self.prefix + ".cams_embeds",
                                  self.level_embeds, self.prefix + ".level_embeds")
        value = torch.empty(B * 4, S, 256, dtype=F32, device=mlvl[0].device)
        start = 0
        for l, m in enumerate(mlvl):
            hw = m.shape[1] * m.shape[2]
            lay.vproj(m, shift_n=lay.vshift[l], shift_n_mod=4,
                      out=value[:, start:start + hw].unflatten(1, (m.shape[1], m.shape[2])), out_nstride=S * 256)
            start += hw
        return value

    def _look(self, lay, B, wp, ctrl_sp, meas, flat, lidar2img, ida_mat, mlvl, level_hw, S, values):
        ref, qos, count, max_len = ops.look_project_pack(wp, lidar2img, ida_mat, self.config["img_size"])
        qrows = ops.look_gather_query(qos, ref, wp, ctrl_sp, self.temporal, self.static, meas, flat, mlvl)
        R = qrows.shape[0]
        qn = torch.zeros_like(qrows)
        ops.layernorm_rows(qrows, lay.q_ln[0], lay.q_ln[1], out=qn, D=1543)
        q = lay.q3(lay.q1(rows(qn)))                                                # (R,1,1,256)
        off = unrows(lay.off(q))
        aw = unrows(lay.aw(q))
        value_all, L, ready = values
        if ready is not None:            # the projections were issued on their own stream (forward())
            torch.cuda.current_stream(wp.device).wait_stream(ready)
        att = ops.msda_sample(value_all, off, aw, ref, level_hw, B, coff=L * 256)    # (R,256)
        an = ops.layernorm_rows(att, lay.ffn_ln[0], lay.ffn_ln[1])
        y = unrows(lay.ffn2(lay.ffn1(rows(an)), res1=att.view(R, 1, 1, 256)))
        red = ops.sca_reduce(y, max_len, B)                                          # (B,1024)
        rn = ops.layernorm_rows(red, lay.o_ln[0], lay.o_ln[1])
        return unrows(lay.o3(lay.o1(rows(rn)))), (count, max_len)

    # ------------------------------------------------------------------ forward
    def forward(self, flattend_BEV_feat, BEV_feat, measurement_feat, target_point, parent_module,
                teacher_forcing_data=None, look_feature_metadata=None, channel_last_out=False):
        """flattend_BEV_feat (B,256), BEV_feat channel-last (B,21,21,32), measurement_feat (B,128) f32 on
        device; look_feature_metadata = [lidar2img (B,4,4,4), ida_mat (B,4,4,4), fpn (4 x (tensor, coff, C))
        channel-last, lidar feature (unused: the LiDAR look branch is zeroed, DEC:186)]."""
        flat, bev, meas = flattend_BEV_feat, BEV_feat, measurement_feat
        taped = autodiff.TAPE is not None      # training: the layer-wise path (every op records its backward), one stream
        from . import layers as _layers
        if self.fused is not None and not taped and not _layers.BN_TRAIN:    # (the composite kernels fold eval-mode BN)
            return self._forward_fused(flat, bev, meas, parent_module, teacher_forcing_data, look_feature_metadata,
                                       channel_last_out)
        B = flat.shape[0]
        dev = flat.device
        outs = {}
        outs["pred_speed"] = unrows(_run(self.speed, rows(flat)))
        fm = torch.empty(B, 384, dtype=F32, device=dev)
        ops.ew(3, flat, out=fm, C=256, out_coff=0)
        ops.ew(3, meas, out=fm, C=128, out_coff=256)
        jt = _run(self.join_traj, rows(fm))
        outs["pred_value_traj"] = unrows(_run(self.value_traj, jt))
        outs["pred_features_traj"] = unrows(jt)
        wp0 = unrows(_run(self.output_traj, jt)).view(B, 4, 2)
        jc = _run(self.join_ctrl, rows(fm))
        outs["pred_value_ctrl"] = unrows(_run(self.value_ctrl, jc))
        outs["pred_features_ctrl"] = unrows(jc)
        pol = _run(self.policy, jc)
        R1 = self.refine_num + 1
        wp_all = torch.empty(B, R1, 4, 2, dtype=F32, device=dev)
        ctrl_all = torch.empty(B, R1, 4, 4, dtype=F32, device=dev)
        # coarse outputs written straight into the stacked result buffers (torch.stack of DEC:483-484)
        ops.ew(3, wp0.reshape(B, 8), out=wp_all.view(B, R1 * 8), C=8, out_coff=0)
        mu = unrows(_run(self.dist_mu, pol))
        sg = unrows(_run(self.dist_sigma, pol))
        c0 = ctrl_all.view(B, R1 * 16)
        for t in range(4):   # cat([mu, sigma], -1) per time step
            ops.ew(3, mu, out=c0, C=2, a_coff=2 * t, out_coff=4 * t)
            ops.ew(3, sg, out=c0, C=2, a_coff=2 * t, out_coff=4 * t + 2)

        lidar2img = look_feature_metadata[0].to(dev, F32).contiguous()
        ida_mat = look_feature_metadata[1].to(dev, F32).contiguous()
        fpn = look_feature_metadata[2]
        mlvl = [self.fpn_linear[i](t, in_coff=off, cin=c) for i, (t, off, c) in enumerate(fpn)]
        level_hw = [(m.shape[1], m.shape[2]) for m in mlvl]
        S = sum(h * w for h, w in level_hw)

        # all layers' value projections, on their own stream under the coarse heads / first GRU
        fork = getattr(parent_module, "use_side_stream", True) and not taped
        fused_concat = _FUSED_CONCAT or taped   # (the broadcast pieces of the concats need the row-mapped backward)
        vready = None
        if taped:
            value_all = None
            train_values = [self._project_values_train(lay, mlvl, B, S) for lay in self.layers]
        elif fork:
            main = torch.cuda.current_stream(dev)
            if self._vstream is None:
                self._vstream = torch.cuda.Stream(dev)
            self._vstream.wait_stream(main)
            with torch.cuda.stream(self._vstream):
                value_all = self._project_values(mlvl, B, S)
            value_all.record_stream(main)
            vready = self._vstream
        else:
            value_all = self._project_values(mlvl, B, S)

        H, W = bev.shape[1:3]
        s_bev = torch.empty(B, self.refine_num, H, W, 32, dtype=F32, device=dev)
        s_flat = torch.empty(B, self.refine_num, 256, dtype=F32, device=dev)
        s_fut = torch.empty(B, self.refine_num, 4, H, W, 32, dtype=F32, device=dev)
        def run_pass(inputs_of, emit, s_bev, s_flat, s_fut, wait_values):
            """The five refinement layers (DEC:428-447).  `inputs_of(L)` -> (wp, ctrl) fed to layer L, `emit(L, d_wp,
            d_ctrl, wp, ctrl)` consumes its offsets; the main pass chains them, the teacher-forcing pass feeds the
            expert's values to every layer."""
            cur_bev, cur_flat = bev, flat
            look_info = []
            for L, lay in enumerate(self.layers):
                wp, ctrl = inputs_of(L)
                sp = ops.ew(3, ctrl.view(B * 4, 4), act=_lib.ACT_SOFTPLUS).view(B, 4, 4)
                inp6 = torch.empty(B, 4, 6, dtype=F32, device=dev)
                if fused_concat:
                    ops.concat_rows(inp6.view(B * 4, 6), [(wp.view(B * 4, 2), 2, 1, 0), (sp.view(B * 4, 4), 4, 1, 0)])
                else:
                    ops.ew(3, wp.view(B * 4, 2), out=inp6.view(B * 4, 6), C=2, out_coff=0)
                    ops.ew(3, sp.view(B * 4, 4), out=inp6.view(B * 4, 6), C=4, out_coff=2)
                fut = torch.empty(B, 4, H, W, 32, dtype=F32, device=dev)
                # The prediction branch (conv-GRU, 4 steps x 8 small convs, + the shared flatten network) and the look
                # branch (value projections, MSDA sampling, attention MLPs) only meet at the concat below: run the
                # former on a second HIP stream.  Both are chains of microsecond-scale launches on a few CUs each, so
                # they overlap almost perfectly (DEC:428-447 runs them back to back).
                if fork:
                    main = torch.cuda.current_stream(dev)
                    if self._branch is None:
                        self._branch = torch.cuda.Stream(dev)
                    self._branch.wait_stream(main)
                    fut.record_stream(self._branch)           # written by the GRU and re-read by bev_update() there
                    inp6.record_stream(self._branch)
                    with torch.cuda.stream(self._branch):
                        lay.gru(inp6, cur_bev, fut)
                        fflat = parent_module.flatten_tail(fut.view(B * 4, H, W, 32))    # (B*4,256)
                else:
                    lay.gru(inp6, cur_bev, fut)
                    fflat = parent_module.flatten_tail(fut.view(B * 4, H, W, 32))        # (B*4,256)
                values = (train_values[L], 0, None) if taped else \
                    (value_all, L, vready if (L == 0 and wait_values) else None)
                look, info = self._look(lay, B, wp, sp, meas, cur_flat, lidar2img, ida_mat, mlvl, level_hw, S, values)
                if fork:
                    main.wait_stream(self._branch)
                    fflat.record_stream(main)
                look_info.append(info)
                # [future flat 256 | look 256 | zeros 256 (LiDAR look) | temporal 128 | meas 128]
                if fused_concat:
                    hin = torch.empty(B * 4, 1024, dtype=F32, device=dev)
                    ops.concat_rows(hin, [(fflat, 256, 1, 0), (look, 256, 4, 0), (None, 256, 1, 0),
                                          (self.temporal, 128, 1, 4), (meas, 128, 4, 0)])
                else:
                    hin = torch.zeros(B * 4, 1024, dtype=F32, device=dev)
                    ops.ew(3, fflat, out=hin, C=256, out_coff=0)
                    hv = hin.view(B, 4, 1024)
                    for t in range(4):
                        ops.ew(3, look, out=hv[:, t], C=256, out_coff=256)
                        ops.ew(3, meas, out=hv[:, t], C=128, out_coff=896)
                        ops.ew(3, self.temporal[t:t + 1].expand(B, 128), out=hv[:, t], C=128, out_coff=768)
                hn = ops.layernorm_rows(hin, lay.mlp_ln[0], lay.mlp_ln[1])
                h = unrows(lay.mlp4(lay.mlp1(rows(hn))))                                 # (B*4,512)
                if fused_concat:
                    tin = torch.empty(B * 4, 516, dtype=F32, device=dev)
                    ops.concat_rows(tin, [(wp.view(B * 4, 2), 2, 1, 0), (h, 512, 1, 0), (None, 2, 1, 0)])
                else:
                    tin = torch.zeros(B * 4, 516, dtype=F32, device=dev)
                    ops.ew(3, wp.view(B * 4, 2), out=tin, C=2, out_coff=0)
                    ops.ew(3, h, out=tin, C=512, out_coff=2)
                d_wp = unrows(_run(lay.traj, rows(tin)))                                  # (B*4,2)
                cin = torch.empty(B * 4, 516, dtype=F32, device=dev)
                if fused_concat:
                    ops.concat_rows(cin, [(ctrl.view(B * 4, 4), 4, 1, 0), (h, 512, 1, 0)])
                else:
                    ops.ew(3, ctrl.view(B * 4, 4), out=cin, C=4, out_coff=0)
                    ops.ew(3, h, out=cin, C=512, out_coff=4)
                d_ctrl = unrows(_run(lay.ctrl, rows(cin)))                                # (B*4,4)
                emit(L, d_wp, d_ctrl, wp, ctrl)
                hb = h.view(B, 2048)

                def bev_update():
                    xb = torch.empty(B, H, W, 2080, dtype=F32, device=dev)
                    ops.copy_nhwc(cur_bev, xb, out_coff=0)
                    ops.broadcast_rows(hb, xb, out_coff=32)
                    nb = lay.bev2(lay.bev0(xb), res1=cur_bev)
                    ops.ew(3, nb.view(B, -1), out=s_bev[:, L].view(B, -1))
                    ops.ew(3, fut.view(B, -1), out=s_fut[:, L].view(B, -1))
                    return nb

                # the BEV-map update (K = 18,720 conv) only feeds the NEXT layer's GRU: it stays on the prediction
                # stream, beside this layer's offset heads and flattened-feature update on the main stream
                if fork:
                    self._branch.wait_stream(main)                # h is ready
                    h.record_stream(self._branch)
                    with torch.cuda.stream(self._branch):
                        new_bev = bev_update()
                else:
                    new_bev = bev_update()
                fin = torch.empty(B, 2304, dtype=F32, device=dev)
                if fused_concat:
                    ops.concat_rows(fin, [(cur_flat, 256, 1, 0), (hb, 2048, 1, 0)])
                else:
                    ops.ew(3, cur_flat, out=fin, C=256, out_coff=0)
                    ops.ew(3, hb, out=fin, C=2048, out_coff=256)
                new_flat = unrows(lay.flat2(lay.flat0(rows(fin)), res1=rows(cur_flat)))
                ops.ew(3, new_flat, out=s_flat[:, L])
                cur_bev, cur_flat = new_bev, new_flat
            if fork:
                main.wait_stream(self._branch)                    # last BEV update, s_bev / s_fut
                cur_bev.record_stream(main)
            return look_info

        def chained_inputs(L):
            # copies = the reference's .detach() of the previous layer's outputs (DEC:429-430): under the training tape no
            # gradient flows back through them (.clone(): for B = 1 the slice is already contiguous)
            return wp_all[:, L].clone(), ctrl_all[:, L].clone()

        def chained_emit(L, d_wp, d_ctrl, wp, ctrl):
            ops.ew(0, d_wp.view(B, 8), b=wp.view(B, 8), out=wp_all[:, L + 1].view(B, 8))
            ops.ew(0, d_ctrl.view(B, 16), b=ctrl.view(B, 16), out=ctrl_all[:, L + 1].view(B, 16))

        look_info = run_pass(chained_inputs, chained_emit, s_bev, s_flat, s_fut, True)

        if teacher_forcing_data is not None:
            # Teacher-forcing pass (DEC:491-533, the forward half of the training step): the same five layers from the
            # encoder's BEV state again, every layer fed the EXPERT waypoints and inv_softplus(expert Beta parameters);
            # its offsets are regressed to zero by the training losses.
            tf = teacher_forcing_data
            t_wp = tf["waypoints"].to(dev, F32).contiguous()
            spx = torch.cat([torch.cat([tf["action_mu"], tf["action_sigma"]], -1).unsqueeze(1),
                             torch.cat([torch.stack(list(tf["future_action_mu"][:-1]), 1),
                                        torch.stack(list(tf["future_action_sigma"][:-1]), 1)], -1)], 1).to(dev, F32)
            t_ctrl = (spx + torch.log(-torch.expm1(-spx))).contiguous()      # inv_softplus of B x 16 inputs (DEC:22-23)
            R = self.refine_num
            t_dwp = torch.empty(B, R, 4, 2, dtype=F32, device=dev)
            t_dctrl = torch.empty(B, R, 4, 4, dtype=F32, device=dev)
            t_bev = torch.empty(B, R, H, W, 32, dtype=F32, device=dev)
            t_flat = torch.empty(B, R, 256, dtype=F32, device=dev)
            t_fut = torch.empty(B, R, 4, H, W, 32, dtype=F32, device=dev)

            def teacher_emit(L, d_wp, d_ctrl, wp, ctrl):
                ops.ew(3, d_wp.view(B, 8), out=t_dwp[:, L].view(B, 8))
                ops.ew(3, d_ctrl.view(B, 16), out=t_dctrl[:, L].view(B, 16))

            run_pass(lambda L: (t_wp, t_ctrl), teacher_emit, t_bev, t_flat, t_fut, False)
            outs["teacher_pred_wp_offset"], outs["teacher_pred_ctrl_offset_lis"] = t_dwp, t_dctrl
            outs["teacher_refine_flattned_BEV_feature"] = t_flat
            if channel_last_out:
                outs["_teacher_refine_bev_cl"], outs["_teacher_fut_cl"] = t_bev, t_fut
            else:
                outs["teacher_refine_BEV_feature"] = ops.nhwc_to_nchw(t_bev.view(B * R, H, W, 32)).view(B, R, 32, H, W)
                outs["teacher_future_BEV_feature"] = ops.nhwc_to_nchw(t_fut.view(B * R * 4, H, W, 32)).view(
                    B, R, 4, 32, H, W)                              # plain stack (DEC:523), not the DEC:481 re-view
        ct = ops.ew(3, ctrl_all.view(B * R1 * 4, 4), act=_lib.ACT_SOFTPLUS_CLAMP).view(B, R1, 4, 4)
        outs["pred_wp"] = wp_all
        outs["mu_branches"], outs["sigma_branches"] = ct[:, :, 0, :2], ct[:, :, 0, 2:]
        outs["future_mu"], outs["future_sigma"] = ct[:, :, 1:, :2], ct[:, :, 1:, 2:]
        outs["refine_flattned_BEV_feature"] = s_flat
        outs["_look_info"] = look_info
        if channel_last_out:
            outs["_bev_cl"], outs["_refine_bev_cl"], outs["_refine_fut_cl"] = bev, s_bev, s_fut
            return outs
        outs["bev_feature"] = ops.nhwc_to_nchw(bev)
        outs["refine_BEV_feature"] = ops.nhwc_to_nchw(s_bev.view(B * self.refine_num, H, W, 32)).view(
            B, self.refine_num, 32, H, W)
        fut_nchw = ops.nhwc_to_nchw(s_fut.view(B * self.refine_num * 4, H, W, 32)).view(B, self.refine_num, 4, 32, H, W)
        # DEC:481 re-views the (B,R,4,...) stack as (B,4,R,...) and transposes (memory reinterpretation)
        outs["refine_future_BEV_feature"] = fut_nchw.view(B, 4, self.refine_num, 32, H, W).transpose(1, 2)
        return outs

    # ------------------------------------------------------------------ composite path
    def _forward_fused(self, flat, bev, meas, parent_module, teacher, meta, channel_last_out):
        """Same outputs as `forward`, sequenced through decoder_fused.FusedDecoder.  Per-layer results are kept
        layer-major (R, B, ...) so that every kernel writes and reads contiguous slices; the (B, R, ...) tensors of the
        reference's torch.stack(dim=1) are produced once at the end."""
        fz = self.fused
        B, dev = flat.shape[0], flat.device
        R, R1 = self.refine_num, self.refine_num + 1
        H, W = bev.shape[1:3]
        flat, meas, bev = flat.contiguous(), meas.contiguous(), bev.contiguous()
        outs = {}
        wp_lm = torch.empty(R1, B, 4, 2, dtype=F32, device=dev)
        ctrl_lm = torch.empty(R1, B, 4, 4, dtype=F32, device=dev)
        fz.coarse_heads(flat, meas, outs, wp_lm[0], ctrl_lm[0])
        lidar2img = meta[0].to(dev, F32).contiguous()
        ida_mat = meta[1].to(dev, F32).contiguous()
        mlvl = [self.fpn_linear[i](t, in_coff=off, cin=c) for i, (t, off, c) in enumerate(meta[2])]
        level_hw = [(m.shape[1], m.shape[2]) for m in mlvl]
        S = sum(h * w for h, w in level_hw)
        main = torch.cuda.current_stream(dev)
        fork = getattr(parent_module, "use_side_stream", True)
        vready = None
        if fork and self._branch is None:
            self._branch = torch.cuda.Stream(dev)
        if SAMPLE_FIRST and all(m.dtype == F32 for m in mlvl):
            # (f32 maps: the 16-bit storage modes keep the projected-value form, whose sampler reads 16-bit rows)
            # the composite decoder samples the raw fpn_linear maps and applies value_proj to the weighted sums
            # (tt_msda_sample_proj_ln): the (B*4, 33320, 5*256) value tensor and its GEMM do not exist
            value_all = None
        elif fork:
            if self._vstream is None:
                self._vstream = torch.cuda.Stream(dev)
            self._vstream.wait_stream(main)
            with torch.cuda.stream(self._vstream):
                value_all = self._project_values(mlvl, B, S)
            value_all.record_stream(main)
            vready = self._vstream
        else:
            value_all = self._project_values(mlvl, B, S)
        ctx = (lidar2img, ida_mat, mlvl, level_hw, value_all, vready)
        streams = (main, self._branch if fork else None)

        def run(inputs_of, wp_dst, ctrl_dst, s_bev, s_flat, s_fut, residual, wait_values):
            cur_bev, cur_flat = bev.view(B, H * W, 32), flat
            info = []
            for L in range(R):
                wp_in, ctrl_in = inputs_of(L)
                info.append(fz.layer(L, wp_in, ctrl_in, cur_bev, cur_flat, meas, ctx, s_fut[L], s_bev[L], s_flat[L],
                                     wp_dst(L), ctrl_dst(L), residual, streams, wait_values and L == 0))
                cur_bev, cur_flat = s_bev[L], s_flat[L]
            if fork:
                main.wait_stream(self._branch)          # last BEV update
            return info

        def bufs():
            return (torch.empty(R, B, H * W, 32, dtype=F32, device=dev), torch.empty(R, B, 256, dtype=F32, device=dev),
                    torch.empty(R, B, 4, H * W, 32, dtype=F32, device=dev))

        s_bev, s_flat, s_fut = bufs()
        look_info = run(lambda L: (wp_lm[L], ctrl_lm[L]), lambda L: wp_lm[L + 1], lambda L: ctrl_lm[L + 1],
                        s_bev, s_flat, s_fut, True, True)

        def to_br(t):        # (R, B, ...) -> contiguous (B, R, ...): the reference's torch.stack(dim=1)
            return t.transpose(0, 1).contiguous()

        def nchw(t, lead):   # (*lead, H*W, 32) channel-last -> (*lead, 32, H, W)
            n = 1
            for v in lead:
                n *= v
            return ops.nhwc_to_nchw(t.reshape(n, H, W, 32)).view(*lead, 32, H, W)

        if teacher is not None:
            tf = teacher
            t_wp = tf["waypoints"].to(dev, F32).contiguous()
            spx = torch.cat([torch.cat([tf["action_mu"], tf["action_sigma"]], -1).unsqueeze(1),
                             torch.cat([torch.stack(list(tf["future_action_mu"][:-1]), 1),
                                        torch.stack(list(tf["future_action_sigma"][:-1]), 1)], -1)], 1).to(dev, F32)
            t_ctrl = (spx + torch.log(-torch.expm1(-spx))).contiguous()      # inv_softplus (DEC:22-23)
            t_dwp = torch.empty(R, B, 4, 2, dtype=F32, device=dev)
            t_dctrl = torch.empty(R, B, 4, 4, dtype=F32, device=dev)
            t_bev, t_flat, t_fut = bufs()
            run(lambda L: (t_wp, t_ctrl), lambda L: t_dwp[L], lambda L: t_dctrl[L], t_bev, t_flat, t_fut, False, False)
            outs["teacher_pred_wp_offset"], outs["teacher_pred_ctrl_offset_lis"] = to_br(t_dwp), to_br(t_dctrl)
            outs["teacher_refine_flattned_BEV_feature"] = to_br(t_flat)
            if channel_last_out:
                outs["_teacher_refine_bev_cl"] = to_br(t_bev).view(B, R, H, W, 32)
                outs["_teacher_fut_cl"] = to_br(t_fut).view(B, R, 4, H, W, 32)
            else:
                outs["teacher_refine_BEV_feature"] = to_br(nchw(t_bev, (R, B)))
                outs["teacher_future_BEV_feature"] = to_br(nchw(t_fut, (R, B, 4)))
        ct = ops.ew(3, ctrl_lm.view(R1 * B * 4, 4), act=_lib.ACT_SOFTPLUS_CLAMP).view(R1, B, 4, 4).transpose(0, 1)
        outs["pred_wp"] = to_br(wp_lm)
        outs["mu_branches"], outs["sigma_branches"] = ct[:, :, 0, :2], ct[:, :, 0, 2:]
        outs["future_mu"], outs["future_sigma"] = ct[:, :, 1:, :2], ct[:, :, 1:, 2:]
        outs["refine_flattned_BEV_feature"] = to_br(s_flat)
        outs["_look_info"] = look_info
        if channel_last_out:
            outs["_bev_cl"] = bev
            outs["_refine_bev_cl"] = to_br(s_bev).view(B, R, H, W, 32)
            outs["_refine_fut_cl"] = to_br(s_fut).view(B, R, 4, H, W, 32)
            return outs
        outs["bev_feature"] = ops.nhwc_to_nchw(bev)
        outs["refine_BEV_feature"] = to_br(nchw(s_bev, (R, B)))
        fut_nchw = to_br(nchw(s_fut, (R, B, 4)))                            # (B, R, 4, 32, H, W) contiguous
        # DEC:481 re-views the (B,R,4,...) stack as (B,4,R,...) and transposes (memory reinterpretation)
        outs["refine_future_BEV_feature"] = fut_nchw.view(B, 4, R, 32, H, W).transpose(1, 2)
        return outs

    __call__ = forward

