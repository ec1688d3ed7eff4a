"""Composite ("fused") execution of the look-and-predict decoder: the same arithmetic as `decoder.ThinkTwiceDecoder`'s
layer-by-layer path (thinktwice_decoder.py:236-260, 419-489) in ~16 launches per refinement layer instead of ~180:

    prediction branch   tt_dec_gru (4-step conv-GRU, LDS resident)  ->  tt_dec_flatten (grid2feat of the 4 futures)
    look branch         tt_look_project_pack -> tt_look_query_ln -> tt_mlp_chain [query_linear.1/.3, offsets, weights]
                        -> tt_msda_sample_ln -> tt_mlp_chain [ffn] -> tt_sca_reduce_ln -> tt_mlp_chain [output_proj]
    merge               tt_dec_merge_in -> tt_mlp_chain [mlp.1, mlp.4, traj / ctrl offset heads (+ residual)]
    state update        tt_mlp_chain [broadcast-channel term of the BEV update] -> tt_dec_bev_update;
                        tt_concat_rows -> tt_mlp_chain [flattened_BEV_feat_update_module + residual]

Everything multiplies in bf16x3 (f32 operands split into bf16 hi+lo pairs, three MFMAs per product, ~1e-5 relative).
This module only prepares the weights (pair format) and sequences the launches.
"""
import torch

from . import _lib, ops, weights

F32 = torch.float32


def _fold(sd, conv, bn, eps=1e-5):
    """(weight * bn_scale, shift) of conv (+ optional bias) followed by an eval BatchNorm."""
    w = sd[conv + ".weight"].float()
    bias = sd.get(conv + ".bias")
    if bn is None:
        return w, (bias.float() if bias is not None else torch.zeros(w.shape[0]))
    scale, shift = weights.fold_bn(sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"], sd[bn + ".running_var"],
                                   eps, bias)
    return w * scale.to(w.device).view(-1, *([1] * (w.dim() - 1))), shift


def pair_conv(w, dev):
    """[N, C, KH, KW] f32 -> fragment-major pair format (weights.split_pairs_frag) of the [N rounded up to 32]
    [KH*KW*Cp] matrix (K order tap-major, channel-minor, Cp = C rounded up to 16)."""
    N, C, KH, KW = w.shape
    cp = (C + 15) // 16 * 16
    out = torch.zeros((N + 31) // 32 * 32, KH * KW, cp, dtype=F32, device=dev)
    out[:N, :, :C] = w.to(dev).permute(0, 2, 3, 1).reshape(N, KH * KW, C)
    return weights.split_pairs_frag(out.reshape(out.shape[0], KH * KW * cp).contiguous())


def _dev(t, dev):
    return t.detach().to(device=dev, dtype=F32).contiguous()


def prep_gru(sd, g, dev):
    w0, wx, b0, w2, b2 = [], [], [], [], []
    for name in ("conv_update", "conv_reset", "conv_state_tilde"):
        W0 = sd[f"{g}.{name}.0.weight"].float()                       # (32, 6 + 32, 3, 3): cat([x, state])
        w0.append(pair_conv(W0[:, 6:], dev))
        wx.append(_dev(W0[:, :6].permute(2, 3, 1, 0).reshape(9, 6, 32), dev))      # [tap][j][n]
        b0.append(_dev(sd[f"{g}.{name}.0.bias"], dev))
        w2.append(pair_conv(sd[f"{g}.{name}.2.weight"].float(), dev))
        b2.append(_dev(sd[f"{g}.{name}.2.bias"], dev))
    d = {"wd0": pair_conv(sd[g + ".conv_decoder.0.weight"].float(), dev), "bd0": _dev(sd[g + ".conv_decoder.0.bias"], dev),
         "wd2": pair_conv(sd[g + ".conv_decoder.2.weight"].float(), dev), "bd2": _dev(sd[g + ".conv_decoder.2.bias"], dev)}
    d.update({"w0": ops._ptr_array(w0), "wx": ops._ptr_array(wx), "b0": ops._ptr_array(b0), "w2": ops._ptr_array(w2),
              "b2": ops._ptr_array(b2), "_keep": [w0, wx, b0, w2, b2]})
    return d


def prep_flatten(sd, dev):
    """The 17 weight sets of tt_dec_flatten (order documented in csrc/dec_spatial.hip)."""
    ws, bs = [], []

    def add(w, b):
        ws.append(pair_conv(w, dev))
        bs.append(_dev(b, dev))

    def se(name):
        add(*_fold(sd, name + ".conv1", name + ".bn1"))
        add(*_fold(sd, name + ".conv2", name + ".bn2"))
        add(*_fold(sd, name + ".se.fc1", None))
        add(*_fold(sd, name + ".se.fc2", None))
    add(*_fold(sd, "conv21_10", None))
    se("MLP10")
    add(*_fold(sd, "conv10_4", None))
    se("MLP4")
    add(*_fold(sd, "conv4_2", None))
    se("MLP2")
    w = sd["output_fc.0.weight"].float()                              # (512, 1024) over the channel-major flatten c*4 + pix
    add(w.view(w.shape[0], 256, 2, 2), sd["output_fc.0.bias"].float())
    w3 = sd["output_fc.3.weight"].float()
    add(w3.view(w3.shape[0], w3.shape[1], 1, 1), sd["output_fc.3.bias"].float())
    scale, shift = weights.fold_bn(sd["output_fc.2.weight"], sd["output_fc.2.bias"], sd["output_fc.2.running_mean"],
                                   sd["output_fc.2.running_var"], 1e-5)
    return {"w": ops._ptr_array(ws), "b": ops._ptr_array(bs), "bn_scale": _dev(scale, dev), "bn_shift": _dev(shift, dev),
            "_keep": [ws, bs]}


def prep_bev_update(sd, q, dev):
    W0 = sd[q + ".BEV_feat_update_module.0.weight"].float()            # (128, 32 + 2048, 3, 3)
    W2 = sd[q + ".BEV_feat_update_module.2.weight"].float()            # (32, 128, 3, 3)
    w2 = [pair_conv(W2[:, 32 * c:32 * c + 32], dev) for c in range(4)]
    # broadcast-channel term: G[b, tap*128 + n] = sum_c W0[n, 32 + c, tap] * hb[b, c]
    wg = W0[:, 32:].permute(2, 3, 0, 1).reshape(9 * 128, 2048)
    return {"w0": pair_conv(W0[:, :32], dev), "b0": _dev(sd[q + ".BEV_feat_update_module.0.bias"], dev),
            "w2": ops._ptr_array(w2), "b2": _dev(sd[q + ".BEV_feat_update_module.2.bias"], dev), "_keep": w2,
            "G": ops.ChainLinear(wg, None, device=dev)}


def _lin(sd, name, act=0, side_k=0, dev="cuda", pad_k=None):
    w = sd[name + ".weight"].float()
    if pad_k is not None and w.shape[1] < pad_k:
        w = torch.nn.functional.pad(w, (0, pad_k - w.shape[1]))
    return ops.ChainLinear(w, sd.get(name + ".bias"), act=act, side_k=side_k, device=dev)


class _FusedLayer:
    def __init__(self, sd, p, dev):
        A = _lib
        self.gru = prep_gru(sd, p + ".prediction_module.spatial_gru", dev)
        c = p + ".look_module.cam_look_module"
        d = c + ".deformable_attention"
        self.q_ln = (_dev(sd[c + ".query_linear.0.weight"], dev), _dev(sd[c + ".query_linear.0.bias"], dev))
        self.chain_a = [_lin(sd, c + ".query_linear.1", A.ACT_GELU, dev=dev, pad_k=1544),
                        _lin(sd, c + ".query_linear.3", A.ACT_GELU, dev=dev),
                        _lin(sd, d + ".sampling_offsets", dev=dev), _lin(sd, d + ".attention_weights", dev=dev)]
        self.ffn_ln = (_dev(sd[c + ".ffn.norm.weight"], dev), _dev(sd[c + ".ffn.norm.bias"], dev))
        self.chain_b = [_lin(sd, c + ".ffn.w_1", A.ACT_GELU, dev=dev), _lin(sd, c + ".ffn.w_2", dev=dev)]
        self.o_ln = (_dev(sd[c + ".output_proj.0.weight"], dev), _dev(sd[c + ".output_proj.0.bias"], dev))
        self.chain_c = [_lin(sd, c + ".output_proj.1", A.ACT_GELU, dev=dev), _lin(sd, c + ".output_proj.3", dev=dev)]
        self.mlp_ln = (_dev(sd[p + ".mlp.0.weight"], dev), _dev(sd[p + ".mlp.0.bias"], dev))
        t, k = p + ".traj_offset_module", p + ".ctrl_offset_module"
        self.merge = [_lin(sd, p + ".mlp.1", A.ACT_RELU, dev=dev), _lin(sd, p + ".mlp.4", A.ACT_RELU, dev=dev),
                      _lin(sd, t + ".0", A.ACT_RELU, side_k=2, dev=dev), _lin(sd, t + ".2", A.ACT_RELU, dev=dev),
                      _lin(sd, t + ".4", dev=dev),
                      _lin(sd, k + ".0", A.ACT_RELU, side_k=4, dev=dev), _lin(sd, k + ".2", A.ACT_RELU, dev=dev),
                      _lin(sd, k + ".4", dev=dev)]
        self.bev = prep_bev_update(sd, p, dev)
        f = p + ".flattened_BEV_feat_update_module"
        self.flat = [_lin(sd, f + ".0", A.ACT_RELU, dev=dev), _lin(sd, f + ".2", dev=dev)]


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class FusedDecoder:
    """Weights of the composite path for one `ThinkTwiceDecoder` (built by its load_state_dict)."""

    def __init__(self, dec, sd, prefix):
        dev, p, A = dec.device, prefix, _lib
        self.dec = dec
        self.layers = [_FusedLayer(sd, f"{p}.decoder_layers.{L}", dev) for L in range(dec.refine_num)]
        self.flatten = prep_flatten(sd, dev)

        def mlp(name, idx, last_act=False):
            return [_lin(sd, f"{p}.{name}.{j}", A.ACT_RELU if (n < len(idx) - 1 or last_act) else 0, dev=dev)
                    for n, j in enumerate(idx)]
        self.speed = mlp("speed_branch", (0, 2, 4))
        self.traj = mlp("join_traj", (0, 2, 4), True) + mlp("value_branch_traj", (0, 2, 4)) + mlp("output_traj", (0, 2))
        self.ctrl = (mlp("join_ctrl", (0, 2, 4), True) + mlp("value_branch_ctrl", (0, 2, 4)) +
                     mlp("policy_head", (0, 2), True) + mlp("dist_mu", (0, 2)) + mlp("dist_sigma", (0, 2)))

    # ------------------------------------------------------------------ pieces
    def coarse_heads(self, flat, meas, outs, wp0, ctrl0):
        """DEC:419-445: everything before the refinement loop: 3 chains + the [mu | sigma] interleave.
        wp0 (B,4,2) / ctrl0 (B,4,4): slot 0 of the per-layer waypoint / control buffers."""
        B, dev = flat.shape[0], flat.device
        sp = torch.empty(B, self.speed[-1].N, dtype=F32, device=dev)
        ops.mlp_chain(flat, [{"lin": self.speed[0], "src": -1}, {"lin": self.speed[1], "src": 0},
                             {"lin": self.speed[2], "src": 1, "out": (sp, 0)}])
        outs["pred_speed"] = sp
        fm = torch.empty(B, 384, dtype=F32, device=dev)
        ops.concat_rows(fm, [(flat, 256, 1, 0), (meas, 128, 1, 0)])
        jt = torch.empty(B, 256, dtype=F32, device=dev)
        vt = torch.empty(B, self.traj[5].N, dtype=F32, device=dev)
        t = self.traj
        ops.mlp_chain(fm, [{"lin": t[0], "src": -1}, {"lin": t[1], "src": 0}, {"lin": t[2], "src": 1, "out": (jt, 0)},
                           {"lin": t[3], "src": 2}, {"lin": t[4], "src": 3}, {"lin": t[5], "src": 4, "out": (vt, 0)},
                           {"lin": t[6], "src": 2}, {"lin": t[7], "src": 6, "out": (wp0.view(B, 8), 0)}])
        outs["pred_value_traj"], outs["pred_features_traj"] = vt, jt
        jc = torch.empty(B, 256, dtype=F32, device=dev)
        vc = torch.empty(B, self.ctrl[5].N, dtype=F32, device=dev)
        mu = torch.empty(B, 8, dtype=F32, device=dev)
        sg = torch.empty(B, 8, dtype=F32, device=dev)
        c = self.ctrl
        ops.mlp_chain(fm, [{"lin": c[0], "src": -1}, {"lin": c[1], "src": 0}, {"lin": c[2], "src": 1, "out": (jc, 0)},
                           {"lin": c[3], "src": 2}, {"lin": c[4], "src": 3}, {"lin": c[5], "src": 4, "out": (vc, 0)},
                           {"lin": c[6], "src": 2}, {"lin": c[7], "src": 6},
                           {"lin": c[8], "src": 7}, {"lin": c[9], "src": 8, "out": (mu, 0)},
                           {"lin": c[10], "src": 7}, {"lin": c[11], "src": 10, "out": (sg, 0)}])
        outs["pred_value_ctrl"], outs["pred_features_ctrl"] = vc, jc
        # cat([mu, sigma], -1) per time step: rows (b, t) of width 2 into columns [0:2] / [2:4] of the (b, t, 4) rows
        ops.ew(3, mu.view(B * 4, 2), out=ctrl0.view(B * 4, 4), C=2, out_coff=0)
        ops.ew(3, sg.view(B * 4, 2), out=ctrl0.view(B * 4, 4), C=2, out_coff=2)

    def layer(self, L, wp, ctrl, cur_bev, cur_flat, meas, look_ctx, fut_out, bev_out, flat_out, wp_out, ctrl_out,
              residual_emit, streams, wait_values):
        """One refinement layer (DEC:236-260).  wp (B,4,2), ctrl (B,4,4) raw, cur_bev (B,441,32), cur_flat (B,256), all
        contiguous; writes fut_out (B,4,441,32), bev_out (B,441,32), flat_out (B,256), wp_out (B,4,2), ctrl_out (B,4,4).
        With `residual_emit` the outputs are wp + d_wp / ctrl + d_ctrl (main pass, DEC:435-436), else the raw offsets
        (teacher-forcing pass, DEC:513-516)."""
        dec, lay = self.dec, self.layers[L]
        B, dev = wp.shape[0], wp.device
        lidar2img, ida_mat, mlvl, level_hw, value_all, vready = look_ctx
        main, branch = streams
        # ---- prediction branch on its own stream: conv-GRU + grid2feat of the 4 future maps
        inp6 = torch.empty(B * 4, 6, dtype=F32, device=dev)
        ops.ew(3, wp.view(B * 4, 2), out=inp6, C=2, out_coff=0)
        ops.ew(3, ctrl.view(B * 4, 4), out=inp6, C=4, out_coff=2, act=_lib.ACT_SOFTPLUS)
        if branch is not None:
            branch.wait_stream(main)
            for t in (inp6, fut_out, cur_bev):
                t.record_stream(branch)
        with (torch.cuda.stream(branch) if branch is not None else _Null()):
            ops.dec_gru(lay.gru, inp6.view(B, 4, 6), cur_bev, fut_out)
            fflat = ops.dec_flatten(self.flatten, fut_out.view(B * 4, 441, 32))
        # ---- look branch
        ref, qos, count, max_len = ops.look_project_pack(wp, lidar2img, ida_mat, dec.config["img_size"])
        qn = ops.look_query_ln(qos, ref, wp, ctrl, True, dec.temporal, dec.static, meas, cur_flat, mlvl, *lay.q_ln)
        R = qn.shape[0]
        off = torch.empty(R, 512, dtype=F32, device=dev)
        aw = torch.empty(R, 256, dtype=F32, device=dev)
        a = lay.chain_a
        ops.mlp_chain(qn, [{"lin": a[0], "src": -1}, {"lin": a[1], "src": 0}, {"lin": a[2], "src": 1, "out": (off, 0)},
                           {"lin": a[3], "src": 1, "out": (aw, 0)}])
        if value_all is None:
            # sample first, project after: no projected value tensor (decoder.py SAMPLE_FIRST)
            dl = dec.layers[L]               # the layer-wise decoder's prepared value_proj operands (decoder.py load_state_dict)
            att, an = ops.msda_sample_proj_ln(mlvl, off, aw, ref, B, dl.vproj_wT, dl.vproj_b, dl.vshift_lc, *lay.ffn_ln,
                                              max_len=max_len)
        else:
            if vready is not None and wait_values:
                torch.cuda.current_stream(dev).wait_stream(vready)
            att, an = ops.msda_sample_ln(value_all, off, aw, ref, level_hw, B, L * 256, *lay.ffn_ln, max_len=max_len)
        y = torch.empty(R, 256, dtype=F32, device=dev)
        b_ = lay.chain_b
        ops.mlp_chain(an, [{"lin": b_[0], "src": -1}, {"lin": b_[1], "src": 0, "res": (att, 0), "out": (y, 0)}])
        rn = ops.sca_reduce_ln(y, max_len, B, *lay.o_ln)
        look = torch.empty(B, 256, dtype=F32, device=dev)
        c_ = lay.chain_c
        ops.mlp_chain(rn, [{"lin": c_[0], "src": -1}, {"lin": c_[1], "src": 0, "out": (look, 0)}])
        # ---- merge
        if branch is not None:
            main.wait_stream(branch)
            fflat.record_stream(main)
        hn = ops.dec_merge_in(fflat, look, dec.temporal, meas, *lay.mlp_ln)
        h = torch.empty(B * 4, 512, dtype=F32, device=dev)
        m = lay.merge
        wp2, ctrl2 = wp.view(B * 4, 2), ctrl.view(B * 4, 4)
        res_wp = (wp2, 0) if residual_emit else None
        res_ct = (ctrl2, 0) if residual_emit else None
        # the trajectory and control heads interleaved level by level: in the wide form of the chain (few rows) dependent
        # stages are separated by a barrier, and this order needs four of them instead of six
        if ops.chain_is_wide(B * 4):
            ops.mlp_chain(hn, [{"lin": m[0], "src": -1}, {"lin": m[1], "src": 0, "out": (h, 0)},
                               {"lin": m[2], "src": 1, "side": wp2}, {"lin": m[5], "src": 1, "side": ctrl2},
                               {"lin": m[3], "src": 2}, {"lin": m[6], "src": 3},
                               {"lin": m[4], "src": 4, "res": res_wp, "out": (wp_out.view(B * 4, 2), 0)},
                               {"lin": m[7], "src": 5, "res": res_ct, "out": (ctrl_out.view(B * 4, 4), 0)}])
        else:
            ops.mlp_chain(hn, [{"lin": m[0], "src": -1}, {"lin": m[1], "src": 0, "out": (h, 0)},
                               {"lin": m[2], "src": 1, "side": wp2}, {"lin": m[3], "src": 2},
                               {"lin": m[4], "src": 3, "res": res_wp, "out": (wp_out.view(B * 4, 2), 0)},
                               {"lin": m[5], "src": 1, "side": ctrl2}, {"lin": m[6], "src": 5},
                               {"lin": m[7], "src": 6, "res": res_ct, "out": (ctrl_out.view(B * 4, 4), 0)}])
        hb = h.view(B, 2048)
        # ---- state updates: the BEV map (feeds the next layer's GRU) on the prediction stream, the flat feature here
        if branch is not None:
            branch.wait_stream(main)
            for t in (h, bev_out):
                t.record_stream(branch)
        with (torch.cuda.stream(branch) if branch is not None else _Null()):
            G = torch.empty(B, 1152, dtype=F32, device=dev)
            ops.mlp_chain(hb, [{"lin": lay.bev["G"], "src": -1, "out": (G, 0)}], n_split=5, groups=36)
            ops.dec_bev_update(lay.bev, cur_bev, G, bev_out)
        fin = torch.empty(B, 2304, dtype=F32, device=dev)
        ops.concat_rows(fin, [(cur_flat, 256, 1, 0), (hb, 2048, 1, 0)])
        f_ = lay.flat
        # 2304 -> 512 over <= 32 rows is a 4.7 MB weight stream: deal its 16 column blocks over 4 workgroups (a single one
        # took 122 us), then the 512 -> 256 tail + residual as its own small launch
        if ops.chain_is_wide(B):
            ops.mlp_chain(fin, [{"lin": f_[0], "src": -1}, {"lin": f_[1], "src": 0, "res": (cur_flat, 0), "out": (flat_out, 0)}])
        else:
            f1 = torch.empty(B, 512, dtype=F32, device=dev)
            ops.mlp_chain(fin, [{"lin": f_[0], "src": -1, "out": (f1, 0)}], n_split=4)
            ops.mlp_chain(f1, [{"lin": f_[1], "src": -1, "res": (cur_flat, 0), "out": (flat_out, 0)}])
        return count, max_len
