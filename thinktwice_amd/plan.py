"""Plan compiler: record ONE forward of the Python mirror as a launch plan the C runtime executes (csrc/plan.cpp, SURVEY 8b B3:
tt_encoder_fwd / tt_decoder_fwd).

`compile_forward(model, batch)` runs `forward_inference` once with
  * every C-ABI call intercepted (`_lib.lib()` hands out a recording proxy): entry name, arguments, stream;
  * every device allocation served from ONE activation arena (bump pointer; torch.empty / zeros / ... are routed through a
    TorchFunctionMode, zeros become recorded `tt_fill_u32` calls);
  * every other torch operation on a device tensor REFUSED unless it is a pure view: a plan must not depend on a torch
    kernel (contiguous same-dtype `copy_` is translated to `tt_copy_bytes`);
  * cross-stream dependencies (`Stream.wait_stream`) recorded as plan syncs.
Pointers are classified by address: arena -> buffer 1, a persistent tensor of the model (prepared weights, embeddings, folded
BatchNorm vectors: found by walking the model object) -> buffer 0 "weights" (all of them packed into one blob), an input
tensor -> buffers 2...  Host-side arguments (geometry arrays, descriptor structs, pointer tables) are copied into the plan's
blob with their embedded device pointers declared as relocations.

The result (`ForwardPlan`) binds any set of base addresses and runs with no Python in the loop: `plan.run()` is ONE ctypes
call per half (`tt_encoder_fwd`, `tt_decoder_fwd`); `plan.save(dir)` writes plan + weights for a non-Python host
(tools/plan_host.cpp)."""
import ctypes
import os
import sys

import torch
from torch.overrides import TorchFunctionMode

from . import _lib

ALIGN = 256


class PlanBuildError(RuntimeError):
    pass


_VIEW_OK = {
    "view", "reshape", "permute", "transpose", "t", "__getitem__", "unsqueeze", "squeeze", "flatten", "expand", "narrow",
    "select", "unbind", "chunk", "split", "detach", "data_ptr", "size", "stride", "dim", "numel", "is_contiguous",
    "storage_offset", "untyped_storage", "element_size", "__len__", "is_floating_point", "as_strided", "view_as",
    "record_stream", "requires_grad_", "get_device", "__get__", "unflatten", "movedim", "is_pinned", "nelement", "ndimension",
    "__repr__", "__format__", "type", "is_complex", "is_set_to", "_is_view", "is_shared", "__hash__", "__bool__",
}


class _Fn:
    """Recording wrapper of one C entry."""

    def __init__(self, builder, real, name):
        object.__setattr__(self, "_b", builder)
        object.__setattr__(self, "_real", real)
        object.__setattr__(self, "_name", name)

    def __setattr__(self, k, v):                     # .restype / .argtypes go to the real function
        setattr(self._real, k, v)

    def __getattr__(self, k):
        return getattr(self._real, k)

    def __call__(self, *args):
        self._b.record_call(self._name, args)
        return self._real(*args)


class _LibProxy:
    def __init__(self, builder, real):
        self._b, self._real = builder, real

    def __getattr__(self, name):
        real = getattr(self._real, name)
        if name.startswith("tt_") and name in self._b.recordable:
            return _Fn(self._b, real, name)
        return real


class _Mode(TorchFunctionMode):
    def __init__(self, builder):
        super().__init__()
        self.b = builder

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        b = self.b
        name = getattr(func, "__name__", None) or str(func)
        if func in (torch.empty, torch.zeros, torch.ones, torch.full):
            dev = kwargs.get("device")
            if dev is not None and torch.device(dev).type == "cuda":
                return b.alloc_like_call(func, args, kwargs)
            return func(*args, **kwargs)
        if func in (torch.empty_like, torch.zeros_like) and args[0].is_cuda and "device" not in kwargs:
            t = args[0]
            return b.alloc(tuple(t.shape), kwargs.get("dtype", t.dtype), zero=func is torch.zeros_like)
        flat = []

        def walk(o):
            if isinstance(o, torch.Tensor):
                flat.append(o)
            elif isinstance(o, (list, tuple)):
                for x in o:
                    walk(x)
            elif isinstance(o, dict):
                for x in o.values():
                    walk(x)
        walk(args)
        walk(kwargs)
        dev_t = [t for t in flat if t.is_cuda]
        if not dev_t:
            return func(*args, **kwargs)                # host-side math (img_metas constants, ...)
        if name in _VIEW_OK:
            return func(*args, **kwargs)
        if name == "contiguous":
            if args[0].is_contiguous():
                return args[0]
            raise PlanBuildError(b.where(f"Tensor.contiguous() of a non-contiguous device tensor {tuple(args[0].shape)} "
                                         f"strides {args[0].stride()} (a torch copy kernel)"))
        if name == "to":
            t = args[0]
            tgt_dev = next((a for a in list(args[1:]) + list(kwargs.values()) if isinstance(a, (torch.device, str))), None)
            tgt_dt = next((a for a in list(args[1:]) + list(kwargs.values()) if isinstance(a, torch.dtype)), None)
            if (tgt_dev is None or torch.device(tgt_dev).type == "cuda") and (tgt_dt is None or tgt_dt == t.dtype):
                return t
            raise PlanBuildError(b.where(f"Tensor.to({tgt_dev}, {tgt_dt}) of a device tensor"))
        if name in ("float",) and args[0].dtype == torch.float32:
            return args[0]
        if name == "copy_":
            dst, src = args[0], args[1]
            if (src.is_cuda and dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype
                    and dst.numel() == src.numel()):
                b.emit_copy(dst, src)
                return dst
            raise PlanBuildError(b.where(f"Tensor.copy_ {tuple(src.shape)}/{src.stride()} -> {tuple(dst.shape)}/{dst.stride()} "
                                         f"(not a contiguous same-dtype device copy)"))
        if name == "clone":
            t = args[0]
            if t.is_contiguous():
                out = b.alloc(tuple(t.shape), t.dtype)
                b.emit_copy(out, t)
                return out
            raise PlanBuildError(b.where("Tensor.clone() of a non-contiguous device tensor"))
        if name == "zero_":
            b.emit_fill(args[0], 0)
            return args[0]
        raise PlanBuildError(b.where(f"torch operation `{name}` on a device tensor inside the planned forward"))


class PlanBuilder:
    WEIGHTS, ARENA = 0, 1

    def __init__(self, device, arena_bytes):
        self.device = torch.device(device)
        self.L = _lib.lib()
        self._declare()
        self.recordable = set(self._thunk_names())
        self.plan = ctypes.c_void_p(self.L.tt_plan_create())
        self.arena = torch.empty(arena_bytes, dtype=torch.uint8, device=self.device)
        self.arena_top = 0
        self.regions = []            # (start, end, buffer id, offset of `start` inside the buffer)
        self.weights = []            # (storage tensor view uint8, blob offset)
        self.weights_bytes = 0
        self.inputs = []             # (name, tensor)
        self.streams = {}            # cuda_stream handle -> slot
        self.calls = 0
        self.marks = {}
        self.compacted_bytes = None

    # ---------------------------------------------------------------- ctypes signatures of the plan API
    def _declare(self):
        L = self.L
        L.tt_plan_create.restype = ctypes.c_void_p
        L.tt_plan_load.restype = ctypes.c_void_p
        L.tt_plan_add_blob.restype = ctypes.c_longlong
        L.tt_plan_compact_arena.restype = ctypes.c_longlong
        L.tt_plan_buffer_bytes.restype = ctypes.c_longlong
        L.tt_plan_buffer_name.restype = ctypes.c_char_p

    def _thunk_names(self):
        inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "plan_thunks.inc")
        import re
        return re.findall(r'\{"(tt_\w+)", thunk_', open(inc).read())

    # ---------------------------------------------------------------- regions
    def add_persistent(self, tensors):
        """Pack the storages of the model's persistent device tensors into the weights buffer (dedup by storage)."""
        seen = set()
        for t in tensors:
            st = t.untyped_storage()
            base, n = st.data_ptr(), st.nbytes()
            if n == 0 or base in seen or (self.arena.data_ptr() <= base < self.arena.data_ptr() + self.arena.numel()):
                continue
            seen.add(base)
            off = self.weights_bytes
            self.regions.append((base, base + n, self.WEIGHTS, off))
            view = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, 0, (n,), (1,))
            self.weights.append((view, off))
            self.weights_bytes = (off + n + ALIGN - 1) // ALIGN * ALIGN

    def add_input(self, name, t):
        assert t.is_cuda and t.is_contiguous()
        st = t.untyped_storage()
        bid = 2 + len(self.inputs)
        self.regions.append((st.data_ptr(), st.data_ptr() + st.nbytes(), bid, 0))
        self.inputs.append((name, t))
        return bid

    def classify(self, addr, what):
        a0 = self.arena.data_ptr()
        if a0 <= addr < a0 + self.arena.numel():
            return self.ARENA, addr - a0
        for s, e, bid, off in self.regions:
            if s <= addr < e:
                return bid, off + (addr - s)
        raise PlanBuildError(self.where(f"{what}: device pointer 0x{addr:x} belongs to no known buffer (a tensor the plan "
                                        f"compiler did not find: allocate it inside the forward or hang it off the model)"))

    # ---------------------------------------------------------------- allocation
    def alloc(self, shape, dtype, zero=False, fill=None):
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        off = (self.arena_top + ALIGN - 1) // ALIGN * ALIGN
        if off + nbytes > self.arena.numel():
            raise PlanBuildError(f"activation arena exhausted ({self.arena.numel()} bytes): pass a larger arena_bytes")
        self.arena_top = off + nbytes
        if nbytes:      # declared to the plan: tt_plan_compact_arena re-places the allocations by liveness after the recording
            _lib.check(self.L.tt_plan_add_arena_alloc(self.plan, ctypes.c_longlong(off), ctypes.c_longlong(nbytes)),
                       "tt_plan_add_arena_alloc")
        t = self.arena[off:off + nbytes].view(dtype).view(*shape) if nbytes else torch.empty(shape, dtype=dtype, device=self.device)
        if zero:
            self.emit_fill(t, 0)
        elif fill is not None:
            self.emit_fill(t, fill)
        return t

    def alloc_like_call(self, func, args, kwargs):
        if func is torch.full:
            shape, value = args[0], args[1]
        else:
            shape = args[0] if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)) else args
            value = None
        dtype = kwargs.get("dtype") or torch.float32
        if func is torch.empty:
            return self.alloc(tuple(shape), dtype)
        if func is torch.zeros:
            return self.alloc(tuple(shape), dtype, zero=True)
        return self.alloc(tuple(shape), dtype, fill=1.0 if func is torch.ones else value)

    # ---------------------------------------------------------------- recorded plumbing ops
    def _stream_slot(self, handle):
        if handle not in self.streams:
            self.streams[handle] = len(self.streams)
        return self.streams[handle]

    def cur_slot(self):
        return self._stream_slot(torch.cuda.current_stream(self.device).cuda_stream)

    def emit_fill(self, t, value):
        assert t.is_contiguous() and t.element_size() in (4, 8) or value == 0
        import struct
        nbytes = t.numel() * t.element_size()
        if value == 0:
            pat = 0
        elif t.dtype == torch.float32:
            pat = struct.unpack("<I", struct.pack("<f", float(value)))[0]
        elif t.dtype == torch.int32:
            pat = int(value) & 0xFFFFFFFF
        else:
            raise PlanBuildError(self.where(f"fill of a {t.dtype} tensor with {value}"))
        if nbytes % 4:
            raise PlanBuildError(self.where("fill of a buffer whose size is not a multiple of 4 bytes"))
        if nbytes:
            st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(_lib.lib().tt_fill_u32(ctypes.c_void_p(t.data_ptr()), ctypes.c_longlong(nbytes // 4), ctypes.c_uint(pat), st),
                       "tt_fill_u32")

    def emit_copy(self, dst, src):
        nbytes = dst.numel() * dst.element_size()
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.lib().tt_copy_bytes(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), ctypes.c_longlong(nbytes),
                                            st), "tt_copy_bytes")

    def record_sync(self, waiter_handle, signal_handle):
        _lib.check(self.L.tt_plan_add_sync(self.plan, self._stream_slot(waiter_handle), self._stream_slot(signal_handle)),
                   "tt_plan_add_sync")

    def mark(self, name):
        self.marks[name] = int(self.L.tt_plan_num_ops(self.plan))

    # ---------------------------------------------------------------- a C-ABI call
    def where(self, msg):
        import traceback
        frames = [f for f in traceback.extract_stack() if "thinktwice_amd" in f.filename and "plan.py" not in f.filename]
        loc = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-4:][::-1])
        return f"plan compiler: {msg} [{loc}]"

    def _blob(self, raw, relocs):
        off = int(self.L.tt_plan_add_blob(self.plan, raw, ctypes.c_longlong(len(raw))))
        if off < 0:
            raise PlanBuildError("tt_plan_add_blob failed")
        for rel_off, addr in relocs:
            bid, boff = self.classify(addr, "pointer inside a host argument")
            _lib.check(self.L.tt_plan_add_reloc(self.plan, ctypes.c_longlong(off + rel_off), ctypes.c_int(bid),
                                                ctypes.c_longlong(boff)), "tt_plan_add_reloc")
        return off

    def _struct_relocs(self, obj, base=0):
        """(offset, address) of every non-NULL pointer field of a ctypes Structure / array, recursively."""
        out = []
        if isinstance(obj, ctypes.Structure):
            for fname, ftype in obj._fields_:
                foff = getattr(type(obj), fname).offset
                val = getattr(obj, fname)
                if ftype is ctypes.c_void_p or (isinstance(ftype, type) and issubclass(ftype, ctypes._Pointer)):
                    addr = val if isinstance(val, int) else (ctypes.cast(val, ctypes.c_void_p).value if val else None)
                    if addr:
                        out.append((base + foff, addr))
                elif isinstance(val, (ctypes.Structure, ctypes.Array)):
                    out += self._struct_relocs(val, base + foff)
        elif isinstance(obj, ctypes.Array):
            et = obj._type_
            esz = ctypes.sizeof(et)
            if et is ctypes.c_void_p:
                for i, v in enumerate(obj):
                    if v:
                        out.append((base + i * esz, v))
            elif isinstance(et, type) and issubclass(et, (ctypes.Structure, ctypes.Array)):
                for i, v in enumerate(obj):
                    out += self._struct_relocs(v, base + i * esz)
        return out

    def record_call(self, name, args):
        if not args:
            return
        stream = args[-1]
        handle = stream.value if isinstance(stream, ctypes.c_void_p) else stream
        slot = self._stream_slot(handle or 0)
        n = len(args) - 1
        kinds, bufs = (ctypes.c_int * max(n, 1))(), (ctypes.c_int * max(n, 1))()
        ivals, fvals = (ctypes.c_longlong * max(n, 1))(), (ctypes.c_double * max(n, 1))()
        for i, a in enumerate(args[:-1]):
            if isinstance(a, ctypes.c_void_p) or a is None:
                v = None if a is None else a.value
                if not v:
                    kinds[i] = 4
                else:
                    kinds[i] = 2
                    bufs[i], ivals[i] = self.classify(v, f"{name} argument {i}")
            elif isinstance(a, (ctypes.c_float, ctypes.c_double)):
                kinds[i], fvals[i] = 1, float(a.value)
            elif isinstance(a, float):
                kinds[i], fvals[i] = 1, a
            elif isinstance(a, (ctypes.c_int, ctypes.c_longlong, ctypes.c_uint, ctypes.c_ulonglong, ctypes.c_long, ctypes.c_ulong)):
                kinds[i], ivals[i] = 0, int(a.value) if a.value < (1 << 63) else int(a.value) - (1 << 64)
            elif isinstance(a, bool) or isinstance(a, int):
                kinds[i], ivals[i] = 0, int(a)
            elif isinstance(a, (ctypes.Array, ctypes.Structure)):
                kinds[i], ivals[i] = 3, self._blob(bytes(a), self._struct_relocs(a))
            elif hasattr(a, "_obj"):                                # ctypes.byref(x)
                kinds[i], ivals[i] = 3, self._blob(bytes(a._obj), self._struct_relocs(a._obj))
            elif isinstance(a, ctypes._Pointer):
                raise PlanBuildError(self.where(f"{name} argument {i}: typed ctypes pointer (pass the array / byref)"))
            else:
                raise PlanBuildError(self.where(f"{name} argument {i}: unsupported argument {type(a)}"))
        _lib.check(self.L.tt_plan_add_call(self.plan, name.encode(), ctypes.c_int(n), kinds, bufs, ivals, fvals, ctypes.c_int(slot)),
                   f"tt_plan_add_call({name})")
        self.calls += 1


def _persistent_tensors(root):
    """Every CUDA tensor reachable from `root` through attributes, lists, tuples and dicts (prepared weights, folded BatchNorm
    vectors, embeddings, cached constants)."""
    seen, out, stack = set(), [], [root]
    while stack:
        o = stack.pop()
        if id(o) in seen:
            continue
        seen.add(id(o))
        if isinstance(o, torch.Tensor):
            if o.is_cuda:
                out.append(o)
            continue
        if isinstance(o, (list, tuple, set, frozenset)):
            stack.extend(o)
        elif isinstance(o, dict):
            stack.extend(o.values())
        elif hasattr(o, "__dict__") and not isinstance(o, (type, torch.cuda.Stream, torch.device)):
            mod = getattr(type(o), "__module__", "") or ""
            if mod.startswith("thinktwice_amd") or mod.startswith("torch.nn"):
                stack.extend(vars(o).values())
    return out


class ForwardPlan:
    """A compiled forward: the C plan object + the buffers it is currently bound to."""

    INPUT_KEYS = ("img", "points", "speed", "target_point", "target_command")

    def __init__(self, builder, outputs, inputs, consts):
        self.L, self.plan, self.device = builder.L, builder.plan, builder.device
        self.recorded_arena_bytes = (builder.arena_top + ALIGN - 1) // ALIGN * ALIGN
        if builder.compacted_bytes is not None:
            # the plan's arena pointers were re-placed by liveness: a fresh, smaller arena; the outputs move with it
            self.arena = torch.empty(builder.compacted_bytes, dtype=torch.uint8, device=self.device)
            outputs = self._outputs_from_plan(outputs)
            builder.arena = None
        else:
            self.arena = builder.arena[:self.recorded_arena_bytes]
        self.weights_blob = torch.zeros(max(builder.weights_bytes, ALIGN), dtype=torch.uint8, device=self.device)
        for view, off in builder.weights:
            self.weights_blob[off:off + view.numel()].copy_(view)
        self.inputs, self.consts = inputs, consts          # name -> device tensor (the buffers the host refills per frame)
        self.outputs = outputs                              # name -> arena view
        self.nstreams = int(self.L.tt_plan_num_streams(self.plan))
        self.calls = int(self.L.tt_plan_num_calls(self.plan))
        self._streams = [torch.cuda.Stream(self.device) for _ in range(self.nstreams)]
        self.bind()

    def _outputs_from_plan(self, recorded):
        """name -> view of self.arena, from the plan's own output table (buffer, offset, shape, element strides); outputs that
        do not live in the arena keep the recorded tensor."""
        L = self.L
        name, bid, off, nd = ctypes.c_char_p(), ctypes.c_int(), ctypes.c_longlong(), ctypes.c_int()
        shape, stride = (ctypes.c_longlong * 8)(), (ctypes.c_longlong * 8)()
        out = {}
        for i in range(L.tt_plan_num_outputs(self.plan)):
            _lib.check(L.tt_plan_output(self.plan, i, ctypes.byref(name), ctypes.byref(bid), ctypes.byref(off), ctypes.byref(nd),
                                        shape, stride), "tt_plan_output")
            k = name.value.decode()
            if k not in recorded:
                continue
            if bid.value != PlanBuilder.ARENA:
                out[k] = recorded[k]
                continue
            out[k] = torch.as_strided(self.arena[off.value:].view(recorded[k].dtype), list(shape)[:nd.value], list(stride)[:nd.value])
        return out

    def bases(self):
        nb = int(self.L.tt_plan_num_buffers(self.plan))
        arr = (ctypes.c_void_p * nb)()
        arr[0], arr[1] = self.weights_blob.data_ptr(), self.arena.data_ptr()
        for i, t in enumerate(self._input_list()):
            arr[2 + i] = t.data_ptr()
        return arr, nb

    def _input_list(self):
        return list(self.inputs.values()) + list(self.consts.values())

    def bind(self):
        arr, nb = self.bases()
        _lib.check(self.L.tt_plan_bind(self.plan, arr, ctypes.c_int(nb)), "tt_plan_bind")

    def _stream_array(self, main=None):
        main = main or torch.cuda.current_stream(self.device)
        hs = [main.cuda_stream] + [s.cuda_stream for s in self._streams[1:]]
        return (ctypes.c_void_p * len(hs))(*hs), len(hs)

    def run(self, halves=True):
        """Issue the forward from C on the current stream (+ the plan's side streams): two ctypes calls, or one."""
        arr, n = self._stream_array()
        if halves:
            _lib.check(self.L.tt_encoder_fwd(self.plan, arr, ctypes.c_int(n)), "tt_encoder_fwd")
            _lib.check(self.L.tt_decoder_fwd(self.plan, arr, ctypes.c_int(n)), "tt_decoder_fwd")
        else:
            _lib.check(self.L.tt_plan_run(self.plan, arr, ctypes.c_int(n)), "tt_plan_run")
        return self.outputs

    def update(self, batch):
        """Refill the input buffers in place (device copies; the img_metas constants are re-derived on the host)."""
        from .encoder_decoder import _lss_host_constants
        for k, dst in self.inputs.items():
            if k in batch and torch.is_tensor(batch[k]) and batch[k] is not dst:
                dst.copy_(batch[k].to(dst.dtype), non_blocking=True)
        if batch.get("img_metas") is not None:
            ncam = self.inputs["img"].shape[-4]
            for k, v in _lss_host_constants(batch["img_metas"], ncam).items():
                self.consts[k].copy_(v, non_blocking=True)

    def save(self, directory):
        """plan.bin + weights.bin (+ the input tensors as they are now, for tools/plan_host.cpp) -> directory."""
        os.makedirs(directory, exist_ok=True)
        _lib.check(self.L.tt_plan_save(self.plan, os.path.join(directory, "plan.bin").encode()), "tt_plan_save")
        self.weights_blob.cpu().numpy().tofile(os.path.join(directory, "weights.bin"))
        for i, t in enumerate(self._input_list()):
            t.contiguous().view(torch.uint8).cpu().numpy().tofile(os.path.join(directory, f"input{i}.bin"))
        return directory


def compile_forward(model, batch, arena_bytes=None, channel_last_out=False, prev_bev=None):
    """Record `model.forward_inference(batch)` (EncoderDecoder, loaded) into a ForwardPlan."""
    from .encoder_decoder import _lss_host_constants
    dev = model.device
    B = batch["img"].shape[0]
    if arena_bytes is None:
        # bump allocation without reuse: the sum of every temporary of one forward (~6 GB per full-size frame: f32
        # activations, the value projections of five layers, LiDAR index volumes) -- 288 GB of HBM make that the simple choice
        scale = (batch["img"].shape[-1] * batch["img"].shape[-2]) / (448 * 896)
        arena_bytes = int((4 + 7 * B * scale) * (1 << 30))
    b = PlanBuilder(dev, arena_bytes)
    inputs = {k: batch[k].to(dev).contiguous() for k in ForwardPlan.INPUT_KEYS if k in batch and torch.is_tensor(batch[k])}
    ncam = inputs["img"].shape[-4]
    consts = {k: v.to(dev).contiguous() for k, v in _lss_host_constants(batch["img_metas"], ncam).items()}
    # one plain forward first: lazy one-time state (zero pages, cached frustum / plans, stream objects) exists before recording
    warm = dict(batch)
    warm.update(inputs)
    model.forward_inference(warm, channel_last_out=channel_last_out, consts=consts, prev_bev=prev_bev)
    torch.cuda.synchronize(dev)
    b.add_persistent(_persistent_tensors(model))
    for k, t in list(inputs.items()) + list(consts.items()):
        b.add_input(k, t)
    if prev_bev is not None:
        raise PlanBuildError("prev_bev plans: pass the cached BEV as an input (not wired yet)")
    real_lib = _lib._lib
    saved_wait = torch.cuda.Stream.wait_stream

    def wait_stream(self_stream, other):
        b.record_sync(self_stream.cuda_stream, other.cuda_stream)
        return saved_wait(self_stream, other)
    model._plan_builder = b
    try:
        _lib._lib = _LibProxy(b, real_lib)
        torch.cuda.Stream.wait_stream = wait_stream
        b._stream_slot(torch.cuda.current_stream(dev).cuda_stream)        # slot 0 = the caller's stream
        with _Mode(b):
            out = model.forward_inference(warm, channel_last_out=channel_last_out, consts=consts, prev_bev=prev_bev)
    finally:
        _lib._lib = real_lib
        torch.cuda.Stream.wait_stream = saved_wait
        model._plan_builder = None
    torch.cuda.synchronize(dev)
    L = b.L
    _lib.check(L.tt_plan_set_buffer(b.plan, 0, ctypes.c_longlong(max(b.weights_bytes, ALIGN)), b"weights"), "tt_plan_set_buffer")
    _lib.check(L.tt_plan_set_buffer(b.plan, 1, ctypes.c_longlong((b.arena_top + ALIGN - 1) // ALIGN * ALIGN), b"arena"),
               "tt_plan_set_buffer")
    for i, (name, t) in enumerate(b.inputs):
        _lib.check(L.tt_plan_set_buffer(b.plan, 2 + i, ctypes.c_longlong(t.untyped_storage().nbytes()), name.encode()),
                   "tt_plan_set_buffer")
    outputs = {}
    for k, v in out.items():
        if torch.is_tensor(v) and v.is_cuda and not k.startswith("_") and v.dim() <= 8:
            bid, off = b.classify(v.data_ptr(), f"output {k}")
            shape = (ctypes.c_longlong * 8)(*list(v.shape))
            stride = (ctypes.c_longlong * 8)(*list(v.stride()))
            _lib.check(L.tt_plan_add_output(b.plan, k.encode(), ctypes.c_int(bid), ctypes.c_longlong(off), ctypes.c_int(v.dim()),
                                            shape, stride), "tt_plan_add_output")
            outputs[k] = v
    first = b.marks.get("decoder", int(L.tt_plan_num_ops(b.plan)))
    _lib.check(L.tt_plan_add_output(b.plan, b"__decoder_first_op", ctypes.c_int(-1), ctypes.c_longlong(first), ctypes.c_int(0),
                                    None, None), "tt_plan_add_output")
    # liveness-based re-placement of the arena (TT_PLAN_COMPACT=0: keep the bump layout, A/B knob)
    b.compacted_bytes = None
    if os.environ.get("TT_PLAN_COMPACT", "1") != "0":
        nb = int(L.tt_plan_compact_arena(b.plan, ctypes.c_int(PlanBuilder.ARENA), ctypes.c_longlong(ALIGN)))
        if nb > 0:
            b.compacted_bytes = nb
        else:
            # not fatal (the bump layout is valid, only ~3x larger and cold), but never silent: the usual causes are an arena
            # pointer outside every declared allocation or a relocation recorded against the wrong op
            b.compact_error = L.tt_last_error().decode()
            print(f"[plan] tt_plan_compact_arena failed (rc={nb}): {b.compact_error}; keeping the bump arena of "
                  f"{b.arena_top if hasattr(b, 'arena_top') else '?'} bytes", file=sys.stderr, flush=True)
    return ForwardPlan(b, outputs, inputs, consts)
