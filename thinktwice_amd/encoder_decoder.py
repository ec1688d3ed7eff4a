"""`EncoderDecoder` -- host-side mirror of the reference model root
(open_loop_training/code/encoder_decoder_framework.py:23-250): same registry name, constructor
arguments, state_dict keys and `forward_inference(batch) -> dict` contract, so the reference's agent
(leaderboard/team_code/thinktwice_agent.py:168-172,457) can build and call it unchanged through this
package's registry.  Every tensor op runs on the gfx950 kernels behind include/thinktwice_hip.h.
"""
import torch

from . import _lib, autodiff, control, ops
from .fusion import BEVFusion
from .layers import linear_from_sd, rows, unrows
from .registry import DETECTORS, build_backbone, build_head

F32 = torch.float32


@DETECTORS.register_module()
class EncoderDecoder(torch.nn.Module):
    """A torch.nn.Module SHELL: it registers no parameters or sub-modules (the weights live in kernel layouts inside the
    plain-Python sub-objects), but it has the Module surface the reference's callers rely on -- `eval()`, `to()`,
    `state_dict()`, and `_load_from_state_dict`, the hook through which mmcv's `load_checkpoint(model, path)` /
    `load_state_dict(module, state_dict)` (mmcv/runner/checkpoint.py, used at thinktwice_agent.py:170-171 and
    train.py:238) hand a checkpoint to a module tree."""

    def __init__(self, img_encoder, decoder, lidar_encoder=None, num_cams=4, use_depth=False, use_seg=False,
                 downsample_factor=16, seg_downsample_factor=2, train_cfg=None, test_cfg=None,
                 dtype=torch.float32, device="cuda", cfg=None, lidar_dtype=None):
        super().__init__()
        self._ref_sd = None
        self.config = train_cfg if train_cfg is not None else cfg
        self.num_cams = num_cams
        self.dtype = dtype
        self._lidar_dtype = lidar_dtype
        # the constructor's module configs, kept as given: `.to(device)` rebuilds the sub-objects on another GPU from them,
        # `init_weights()` derives the state_dict layout from them
        self._ctor = dict(img_encoder=dict(img_encoder), decoder=dict(decoder),
                          lidar_encoder=None if lidar_encoder is None else dict(lidar_encoder))
        self._build(torch.device(device))
        self.training = False
        tc = self.config or {}                                                   # EDF:42-43: train_cfg decides
        self.use_depth, self.use_seg = bool(tc.get("use_depth", False)), bool(tc.get("use_seg", False))
        self.downsample_factor, self.seg_downsample_factor = downsample_factor, seg_downsample_factor
        self.d_bound = dict(img_encoder).get("d_bound")
        self.use_side_stream = True
        c = self.config or {}
        if "turn_KP" in c:   # EDF:47-48
            self.turn_controller = control.PIDController(c["turn_KP"], c["turn_KI"], c["turn_KD"], c["turn_n"])
            self.speed_controller = control.PIDController(c["speed_KP"], c["speed_KI"], c["speed_KD"], c["speed_n"])

    def _build(self, device):
        """(Re)create the sub-objects on `device` (their kernels' operand buffers live where they are built)."""
        if device.type != "cuda":
            raise _lib.TTError(f"EncoderDecoder lives on an MI355X (device '{device}' requested): there is no CPU product path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        if torch.cuda.is_available() and device.index >= torch.cuda.device_count():
            raise _lib.TTError(f"EncoderDecoder: {device} requested, {torch.cuda.device_count()} GPU(s) visible")
        self.device = device
        c = self._ctor
        # "f32x3h" (weights.X3H): the camera encoder's PAFPN on half storage / two-MFMA products, everything else -- the rest of the
        # camera encoder, the LiDAR branch, the decoder -- exactly the bf16x3 mode
        from . import weights as _w
        rest = _w.X3 if (isinstance(self.dtype, str) and self.dtype == _w.X3H) else self.dtype
        self.img_encoder = build_backbone(c["img_encoder"], dtype=self.dtype, device=device)
        # precision mode of the LiDAR branch (default: the model's; its own knob because the sparse encoder is the one
        # part of the forward whose 16-bit rounding barely reaches the outputs)
        self.lidar_encoder = (build_backbone(c["lidar_encoder"], device=device,
                                             dtype=rest if self._lidar_dtype is None else self._lidar_dtype)
                              if c["lidar_encoder"] is not None else None)
        dec = dict(c["decoder"])
        dec.setdefault("config", self.config)
        self.decoder = build_head(dec, dtype=rest, device=device)
        self.loaded = False
        self._side = None
        self._loss_red = None

    # closed-loop post-processing (thinktwice_agent.py:458-509).  `action_post()` is the one-call device path
    # (tt_action_post: control branch + waypoint PID + arbitration on the output tensors where they are, one D2H copy);
    # process_action / control_pid keep the reference's call structure on the host entries of the same C source.
    def action_post(self, stuck_threshold=800):
        """-> control.ActionPost bound to this model's cfg and device (create once per route, `.tick(pred, speed, target)`)."""
        return control.ActionPost(self.config, stuck_threshold, device=self.device)

    def process_action(self, pred, command, speed, target_point):
        return control.process_action(pred, command, speed, target_point)

    def control_pid(self, waypoints, velocity, target, stuck_desired_speed=-1):
        return control.control_pid(self.config, self.turn_controller, self.speed_controller, waypoints, velocity,
                                   target, stuck_desired_speed)

    # mmcv / torch.nn.Module surface used by the callers (AGENT:170-172, train.py:225,238)
    def train(self, mode=True):
        """model.train(): `forward_train` / `train_step` then run the reference's TRAINING semantics -- batch-statistics
        BatchNorm (SyncBN across ranks, configs/thinktwice.py:39; running statistics updated) and the live ASPP
        Dropout(0.5) (lss.py:91).  model.eval() (the default) keeps running-statistics BatchNorm, which is also the
        frozen-BN fine-tuning mode (`trainer.Trainer(frozen_bn=True)`).  `forward_inference` is eval-mode either way.
        The mode reaches the sub-objects that key on it (nn.Module.train() walks children; this shell has none): the LiDAR
        voxeliser picks its train / eval voxel cap by it (max_voxels=(120000, 160000), configs/thinktwice.py:164)."""
        self.training = bool(mode)
        for sub in (self.img_encoder, self.lidar_encoder, self.decoder):
            if sub is not None and hasattr(sub, "training"):
                sub.training = self.training
        return self

    def eval(self):
        return self.train(False)

    def to(self, *args, **kwargs):
        """nn.Module.to for the one thing callers use it for -- placing the model on a device (thinktwice_agent.py:171
        `self.model.to(self.device)`, train.py's MMDistributedDataParallel(model.cuda(), device_ids=[local_rank])).  The same
        device: no-op.  Another GPU: the sub-objects are rebuilt there and the loaded checkpoint is prepared again (the
        kernels' operand layouts are device buffers).  A dtype: refused -- the precision mode is a constructor argument
        (`dtype=`), not a cast.  The CPU: refused, there is no CPU product path."""
        device = kwargs.get("device")
        for a in args:
            if isinstance(a, (str, torch.device, int)):
                device = a
            elif isinstance(a, torch.dtype):
                raise _lib.TTError(f"EncoderDecoder.to({a}): the precision mode is chosen at construction (dtype=...), not cast")
            elif torch.is_tensor(a):
                device = a.device
        if kwargs.get("dtype") is not None:
            raise _lib.TTError("EncoderDecoder.to(dtype=...): the precision mode is chosen at construction (dtype=...)")
        if device is None:
            return self
        device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else self.device.index or 0)
        if device == self.device:
            return self
        sd, mode = self._ref_sd, self.training
        self._build(device)                    # raises for a non-GPU device before anything is torn down
        if sd is not None:
            self.load_state_dict(sd)
        self.train(mode)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else
                                    (device if isinstance(device, int) else torch.device(device).index or 0)))

    def cpu(self):
        return self.to("cpu")

    def set_epoch(self, epoch):
        self.epoch = epoch

    def model_config(self):
        """The module configs in the layout thinktwice_amd.params walks (img_encoder / lidar_encoder / decoder / cfg)."""
        return dict(img_encoder=self._ctor["img_encoder"], lidar_encoder=self._ctor["lidar_encoder"],
                    decoder=self._ctor["decoder"], cfg=self.config, num_cams=self.num_cams)

    def init_weights(self, seed=0, pretrained=None):
        """train.py:225 `model.init_weights()`: give a freshly built model its initial weights.  The reference's rules
        (dense_heads/utils.py:26-47, code/utils.py:59-80, lss.py:40-46,112-118, the MSDA sampling-offset grid
        multi_scale_deformable_attn_function.py:403-421, thinktwice_decoder.py:369-376) are what `params.init_params`
        restates per tensor, seeded per NAME (every rank builds bit-identical weights without a broadcast).  `pretrained`
        (default: img_backbone_conf.init_cfg.checkpoint when it names a local file -- 'torchvision://resnet50',
        configs/thinktwice.py:147, needs a download and is skipped offline): a torchvision-layout ResNet-50 state_dict
        loaded over `img_encoder.img_backbone.*`.  A model that already holds a checkpoint keeps it (mmcv's init_weights does
        not overwrite weights loaded through init_cfg either): call `load_state_dict` to replace weights."""
        import os
        from . import params
        if self.loaded:
            return self
        sd = params.init_params(self.model_config(), seed=seed)
        if pretrained is None:
            ic = (self._ctor["img_encoder"].get("img_backbone_conf") or {}).get("init_cfg") or {}
            pretrained = ic.get("checkpoint") if ic.get("type") == "Pretrained" else None
        if pretrained and os.path.isfile(str(pretrained)):
            ck = torch.load(pretrained, map_location="cpu")
            ck = ck.get("state_dict", ck)
            pre = "img_encoder.img_backbone."
            hit = 0
            for k, v in ck.items():
                if pre + k in sd and tuple(sd[pre + k].shape) == tuple(v.shape):
                    sd[pre + k] = v.to(sd[pre + k].dtype)
                    hit += 1
            if hit == 0:
                raise _lib.TTError(f"init_weights: no tensor of {pretrained} matches the ResNet-50 layout")
        return self.load_state_dict(sd)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        """torch / mmcv checkpoint loaders walk the module tree calling this hook; the root (this shell has no
        children) takes the whole dict."""
        sub = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)} if prefix else dict(state_dict)
        sub.pop("_metadata", None)
        self.load_state_dict(sub)

    def state_dict(self, destination=None, prefix="", keep_vars=False):
        """The reference-format state_dict this model was loaded from (what torch.save(model.state_dict()) stores)."""
        out = destination if destination is not None else {}
        for k, v in (self._ref_sd or {}).items():
            out[prefix + k] = v
        return out

    def load_state_dict(self, sd, strict=False):
        """Accepts the reference's checkpoint `state_dict` (optionally with a `module.` prefix)."""
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items() if k != "_metadata"}
        self._ref_sd = sd
        dev = self.device
        autodiff.clear_metas(self)            # the previous checkpoint's operand registrations of THIS model
        with autodiff.owned_by(self):         # (the training tape's tensor -> parameter-name tables are scoped per model)
            self.img_encoder.load_state_dict(sd, "img_encoder")
            if self.lidar_encoder is not None:
                self.lidar_encoder.load_state_dict(sd, "lidar_encoder")
            self.fusion = BEVFusion(sd, dev)
            self.meas0 = linear_from_sd(sd, "measurements_encoder.0", dev, act="relu", in_pad=12)
            self.meas2 = linear_from_sd(sd, "measurements_encoder.2", dev, act="relu")
            self.decoder.load_state_dict(sd, "decoder")
        self.loaded = True
        return self

    def flatten_tail(self, grid_feat):
        """Shared flatten network (thinktwice_decoder.py:405-415 uses parent_module's layers)."""
        return self.fusion.tail(grid_feat)

    # ------------------------------------------------------------------ forward
    def measurement_feat(self, batch):
        """speed/12, cat(speed, target_point, command) -> measurements_encoder (EDF:198,242-243)."""
        dev = self.device
        speed = batch["speed"].to(dev, F32).view(-1, 1)
        B = speed.shape[0]
        state = torch.zeros(B, 12, dtype=F32, device=dev)
        ops.affine_rows(speed, torch.full((1,), 1.0 / 12.0, device=dev), None, out=state[:, 0:1])
        ops.ew(3, batch["target_point"].to(dev, F32), out=state, C=2, out_coff=1)
        ops.ew(3, batch["target_command"].to(dev, F32), out=state, C=6, out_coff=3)
        return unrows(self.meas2(self.meas0(rows(state))))

    def extract_sensor_feat(self, img, img_metas, points, consts=None, prev_bev=None):
        # The LiDAR branch is independent of the camera trunk until the BEV fusion: run it on its own HIP
        # stream so its many small launches fill the tail of the big camera convolutions.
        main = torch.cuda.current_stream(self.device)
        pts = points[:, -1].to(self.device)
        if not self.use_side_stream or autodiff.TAPE is not None:      # (the training tape records on one stream)
            lidar = self.lidar_encoder(pts, channel_last=True, rot_flip=True)          # EDF:244-246
            cam = self.img_encoder(img.to(self.device), img_metas, channel_last=True, consts=consts,
                               prev_bev=prev_bev)
            B, H, W, C = cam["_bev_cl"].shape
            cam_bev = torch.empty(B, H, W, C, dtype=F32, device=self.device)
            ops.copy_nhwc(cam["_bev_cl"], cam_bev, rot_flip=True)      # EDF:241
            return cam, cam_bev, lidar
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            lidar = self.lidar_encoder(pts, channel_last=True, rot_flip=True)
        cam = self.img_encoder(img.to(self.device), img_metas, channel_last=True, consts=consts,
                               prev_bev=prev_bev)
        B, H, W, C = cam["_bev_cl"].shape
        cam_bev = torch.empty(B, H, W, C, dtype=F32, device=self.device)
        ops.copy_nhwc(cam["_bev_cl"], cam_bev, rot_flip=True)          # EDF:241
        main.wait_stream(self._side)
        lidar.record_stream(main)
        return cam, cam_bev, lidar

    def forward_inference(self, batch, channel_last_out=False, consts=None, prev_bev=None, teacher=None):
        """`consts`: optional device copies of `LSS.host_constants(batch["img_metas"])` (see InferenceGraph).
        `teacher`: expert waypoints / Beta parameters (`waypoints`, `action_mu/sigma`, `future_action_mu/sigma`): also
        run the decoder's teacher-forcing pass and return its `teacher_*` outputs (forward half of `forward_train`).
        `prev_bev`: cached previous-sweep BEV (see LSS.forward / PrevSweepCache); the camera trunk then runs on the
        key sweep only.  `pred["_key_bev_cl"]` is this call's key-sweep BEV for the cache."""
        if not self.loaded:
            raise _lib.TTError("EncoderDecoder: load_state_dict() first")
        # sticky device fault (a tt_mlp_chain_wide barrier that gave up in an EARLIER forward: its outputs were NaN)
        ops.raise_on_device_fault("EncoderDecoder.forward_inference")
        self.epoch = 10000
        meas = self.measurement_feat(batch)
        cam, cam_bev, lidar = self.extract_sensor_feat(batch["img"], batch["img_metas"], batch.get("points"),
                                                       consts=consts, prev_bev=prev_bev)
        flat, bev32, mids = self.fusion(cam_bev, lidar)
        pb = getattr(self, "_plan_builder", None)
        if pb is not None:
            pb.mark("decoder")          # plan compiler (thinktwice_amd/plan.py): tt_encoder_fwd ends / tt_decoder_fwd starts here
        pred = self.decoder(flat, bev32, meas, batch["target_point"], self, teacher,
                            [cam["lidar2img"], cam["ida_mat"], cam["_fpn_cl"], lidar],
                            channel_last_out=channel_last_out)
        pred["_cam_bev_cl"], pred["_lidar_bev_cl"], pred["_flat"], pred["_meas"] = cam_bev, lidar, flat, meas
        pred["_key_bev_cl"], pred["_seg_cl"] = cam["_key_bev_cl"], cam["_seg_cl"]
        pred["_depth_cl"], pred["_mid_bev_cl"] = cam["_depth_cl"], mids
        return pred

    def forward_train(self, batch):
        """EncoderDecoder.forward_train (encoder_decoder_framework.py:147-191): the forward with the decoder's
        teacher-forcing pass, then every loss term as a device reduction (thinktwice_amd/losses.py, csrc/losses.hip).
        Under model.eval() BatchNorm layers use their running statistics (golden F10 / F13: what the reference computes under
        model.eval(), and the frozen-BN fine-tuning mode); under model.train() they normalise with batch statistics and the
        ASPP dropout is live (golden F11 / F16: the reference under model.train()).  The returned device scalars
        carry no torch autograd graph: inside `with autodiff.Tape()` the forward ops and the loss terms record their
        backward on the tape (trainer.Trainer.step runs it)."""
        from . import layers
        saved, layers.BN_TRAIN = layers.BN_TRAIN, bool(self.training)
        try:
            return self._forward_train(batch)
        finally:
            layers.BN_TRAIN = saved

    def _forward_train(self, batch):
        from . import losses as LS
        if self._loss_red is None:
            self._loss_red = LS.LossReducer(self.device)
        red = self._loss_red
        tape = autodiff.TAPE
        teacher = {k: batch[k] for k in LS.TEACHER_KEYS}
        teacher = {k: ([t.to(self.device) for t in v] if isinstance(v, (list, tuple)) else v.to(self.device))
                   for k, v in teacher.items()}
        # Under the training tape (thinktwice_amd/trainer.py) every term reads the forward's own channel-last tensors and
        # records the gradient of its mean w.r.t. them; the terms are elementwise means, so the values do not change.
        pred = self.forward_inference(batch, teacher=teacher, channel_last_out=tape is not None)
        if tape is None:
            mids = [None if m is None else ops.nhwc_to_nchw(m) for m in pred["_mid_bev_cl"]]
            out = LS.decoder_loss(red, self.config, batch, pred, mids)
        else:
            out = LS.decoder_loss(red, self.config, batch, pred, pred["_mid_bev_cl"], channel_last=True)
        if self.use_seg:                                                      # EDF:172-176
            seg = pred["_seg_cl"].float().contiguous()
            out["seg_loss"] = red.seg_focal(seg, batch["seg"], num_classes=12, factor=self.seg_downsample_factor)
            if tape is not None:
                assert seg.data_ptr() == pred["_seg_cl"].data_ptr(), "training runs on f32 activation storage"
                tape.nodes.append(lambda: tape.grad(seg).add_(
                    red.seg_focal_bwd(seg, batch["seg"], num_classes=12, factor=self.seg_downsample_factor)))
        if self.use_depth:                                                    # EDF:179-190
            dep = pred["_depth_cl"].float().contiguous()
            out["depth_loss"] = red.depth_bce(dep, batch["depth"], self.d_bound, self.downsample_factor)
            if tape is not None:
                assert dep.data_ptr() == pred["_depth_cl"].data_ptr(), "training runs on f32 activation storage"
                tape.nodes.append(lambda: tape.grad(dep).add_(
                    red.depth_bce_bwd(dep, batch["depth"], self.d_bound, self.downsample_factor)))
        return out

    def _parse_losses(self, losses):
        from . import losses as LS
        return LS.parse_losses(losses)

    def train_step(self, data, optimizer=None):
        """encoder_decoder_framework.py:140-145: dict(loss, log_vars, num_samples).  `optimizer` is unused, as in the
        reference (the mmcv OptimizerHook calls loss.backward() and steps it); the backward + all-reduce + AdamW half of
        the iteration is trainer.Trainer.step, which calls this inside its tape."""
        loss, log_vars = self._parse_losses(self.forward_train(data))
        return dict(loss=loss, log_vars=log_vars, num_samples=data["img"].shape[0])

    def forward(self, is_eval=True, **kwargs):
        """EncoderDecoder.forward (encoder_decoder_framework.py:393-407): the training entry of the mmcv runner."""
        return self.train_step(kwargs, None)


class PrevSweepCache:
    """Closed-loop tick driver (thinktwice_agent.py:439-444 feeds the frame of `lag` ticks ago as
    sweep 1).  Sweep 1 is encoded and splatted with the KEY frame's matrices (reference quirk A8), so with the fixed
    evaluation rig its BEV is exactly the key-sweep BEV computed `lag` ticks earlier: keep the last `lag` key-sweep
    BEVs (B x 21 x 21 x 256 f32 = 451 KB per sample each) and skip the second camera pass -- half of the encoder,
    ~93 % of the forward's FLOPs.  Ticks before the cache is warm run the full two-sweep forward."""

    def __init__(self, model, lag=10):
        from collections import deque
        self.model, self.lag = model, lag
        self.ring = deque(maxlen=lag)

    def reset(self):
        self.ring.clear()

    def tick(self, batch, channel_last_out=False):
        prev = self.ring[0] if len(self.ring) == self.lag else None
        pred = self.model.forward_inference(batch, channel_last_out=channel_last_out, prev_bev=prev)
        self.ring.append(pred["_key_bev_cl"].contiguous().clone())
        return pred


class InferenceGraph:
    """One HIP graph for the whole `forward_inference` of a fixed batch shape.

    The forward is ~1150 kernel launches per batch, ~900 of them in the 5-stage decoder where every launch is
    microseconds of GPU work: issued one by one from the host that phase is launch-bound (rocprofv3 shows 8 us
    kernels 25-30 us apart).  Capturing the launches once and replaying the graph removes the host from the loop;
    the LiDAR side stream is captured as a fork/join inside the same graph.

    Inputs live in STATIC device buffers (the tensors of the batch given at construction): `update(batch)` copies a
    new batch into them (and re-derives the img_metas constants on the host), `replay()` runs the graph and returns
    the output dict, whose tensors are overwritten by the next replay.  Everything inside the graph is the same
    eager code path; there is no host sync, allocation or img_metas-dependent host arithmetic in it.
    """

    _KEYS = ("img", "points", "speed", "target_point", "target_command")

    def __init__(self, model, batch, channel_last_out=True, warmup=2, prev_bev=None):
        """`prev_bev`: capture the closed-loop variant (key sweep only + cached previous-sweep BEV, see
        PrevSweepCache); the tensor becomes a static input that `update(prev_bev=...)` overwrites."""
        self.model = model
        dev = model.device
        self.batch = dict(batch)
        for k in self._KEYS:
            if k in batch and torch.is_tensor(batch[k]):
                self.batch[k] = batch[k].to(dev)
        ncam = self.batch["img"].shape[-4]
        self.consts = {k: v.to(dev) for k, v in _lss_host_constants(batch["img_metas"], ncam).items()}
        self.prev_bev = None if prev_bev is None else prev_bev.to(dev).contiguous().clone()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):            # warm-up off the default stream: lazy allocations, attributes
            for _ in range(warmup):
                model.forward_inference(self.batch, channel_last_out=channel_last_out, consts=self.consts,
                                        prev_bev=self.prev_bev)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = model.forward_inference(self.batch, channel_last_out=channel_last_out, consts=self.consts,
                                               prev_bev=self.prev_bev)

    def update(self, batch=None, prev_bev=None):
        if prev_bev is not None:
            self.prev_bev.copy_(prev_bev, non_blocking=True)
        if batch is None:
            return
        for k in self._KEYS:
            if k in batch and torch.is_tensor(batch[k]) and batch[k] is not self.batch[k]:
                self.batch[k].copy_(batch[k], non_blocking=True)
        if batch.get("img_metas") is not None and batch["img_metas"] is not self.batch.get("img_metas"):
            ncam = self.batch["img"].shape[-4]
            for k, v in _lss_host_constants(batch["img_metas"], ncam).items():
                self.consts[k].copy_(v, non_blocking=True)
            self.batch["img_metas"] = batch["img_metas"]

    def replay(self):
        ops.raise_on_device_fault("InferenceGraph.replay")       # sticky: an earlier forward's barrier time-out
        self.graph.replay()
        return self.out

    def __call__(self, batch=None):
        if batch is not None:
            self.update(batch)
        return self.replay()


def _lss_host_constants(img_metas, num_cams):
    from .lss import LSS
    return LSS.host_constants(img_metas, num_cams)
