"""ctypes binding of libthinktwice_hip.so -- the ONLY compute backend.

There is deliberately no CPU or eager-PyTorch fallback: if the library is not
built, or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libthinktwice_hip.so")
# experiments only (tools/): load an alternative build of the same ABI, e.g. one compiled with -DTT_GLDS_DEBUG=1
LIB_PATH = os.environ.get("TT_LIB_PATH", LIB_PATH)

TT_F32, TT_BF16, TT_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_GELU, ACT_SOFTPLUS, ACT_SOFTPLUS_CLAMP = 0, 1, 2, 3, 4, 5

_lib = None


class TTError(RuntimeError):
    pass


def lib():
    """Load (once) and return the C-ABI library; raise loudly if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TTError(
                f"{LIB_PATH} is missing: run `python -m thinktwice_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no fallback path.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.tt_last_error.restype = ctypes.c_char_p
        _lib.tt_version.restype = ctypes.c_int
        if hasattr(_lib, "tt_voxel_pool_workspace_bytes"):
            _lib.tt_voxel_pool_workspace_bytes.restype = ctypes.c_longlong
        if hasattr(_lib, "tt_sp_strided_outputs_workspace_bytes"):
            _lib.tt_sp_strided_outputs_workspace_bytes.restype = ctypes.c_longlong
        if hasattr(_lib, "tt_lidar_voxelize_workspace_bytes"):
            _lib.tt_lidar_voxelize_workspace_bytes.restype = ctypes.c_longlong
        if hasattr(_lib, "tt_lift_splat_workspace_bytes"):
            _lib.tt_lift_splat_workspace_bytes.restype = ctypes.c_longlong
    return _lib


def check(rc, what):
    if rc != 0:
        raise TTError(f"{what} failed (rc={rc}): {lib().tt_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (or NULL)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def cur_stream(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise TTError("thinktwice_amd ops need device tensors (MI355X); got a CPU tensor. "
                          "There is no CPU path in the product -- see oracle/ for the checker.")


def device_faults():
    """Non-blocking read of the current device's host-mapped fault word (tt_device_faults): meaningful for work the caller
    has synchronised with."""
    return int(lib().tt_device_faults())


def clear_device_faults():
    check(lib().tt_clear_device_faults(), "tt_clear_device_faults")


def raise_on_device_fault(where):
    """The product's check: a barrier time-out of tt_mlp_chain_wide poisons its outputs with NaN and sets the fault word;
    whoever has just synchronised with the forward (or is about to start the next one) turns it into a TTError."""
    if device_faults() > 0:
        raise TTError(f"{where}: a tt_mlp_chain_wide barrier timed out on this device -- the decoder outputs since then are "
                      f"NaN / invalid (thinktwice_amd.ops.clear_device_faults() re-arms the device)")
