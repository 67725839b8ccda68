"""ctypes binding of the C ABI in include/vsrlab_hip.h (lib/libvsrlab_hip.so).

The product path has no fallback: if the HIP library is missing or a call returns a non-zero
status, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSRLAB_AMD_LIB: another build of the same library (the diagnostic `make ABL=...` / `make STAMPS=1` builds), for whole-step A/B runs
LIB_PATH = os.environ.get("VSRLAB_AMD_LIB") or os.path.join(_HERE, "lib", "libvsrlab_hip.so")

DT_F32 = 0
DT_BF16 = 1


class BasicVSRDesc(ctypes.Structure):
    """struct VsrBasicVSRDesc (include/vsrlab_hip.h)."""
    _fields_ = [("n", c_int), ("t", c_int), ("h", c_int), ("w", c_int), ("mid_channels", c_int),
                ("res_blocks", c_int), ("upscale", c_int), ("dtype", c_int), ("arena_mode", c_int)]


class DiscDesc(ctypes.Structure):
    """struct VsrDiscDesc (include/vsrlab_hip.h)."""
    _fields_ = [("n", c_int), ("h", c_int), ("w", c_int), ("mid_ch", c_int), ("dtype", c_int)]


class AttnDesc(ctypes.Structure):
    """struct VsrAttnDesc (include/vsrlab_hip.h)."""
    _fields_ = [(k, c_int) for k in ("B", "N", "heads", "head_dim", "q0", "k0", "o0", "Nq", "Nk", "Cout", "c_off", "nW", "Nm")] + \
               [("scale", c_float), ("dtype", c_int), ("mask_packed", c_int), ("mask_value", c_float)]


_P = c_void_p
_SIGNATURES = {
    "vsr_abi_version": (c_int, []),
    "vsr_status_string": (c_char_p, [c_int]),
    "vsr_basicvsr_num_params": (c_int, [ctypes.POINTER(BasicVSRDesc)]),
    "vsr_basicvsr_workspace_bytes": (c_size_t, [ctypes.POINTER(BasicVSRDesc), c_int]),
    "vsr_basicvsr_forward": (c_int, [ctypes.POINTER(BasicVSRDesc), _P, c_int, _P, _P, _P, c_size_t, c_int, _P]),
    "vsr_basicvsr_backward": (c_int, [ctypes.POINTER(BasicVSRDesc), _P, _P, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "vsr_basicvsr_get_flows": (c_int, [ctypes.POINTER(BasicVSRDesc), _P, _P, _P, _P]),
    "vsr_spynet_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "vsr_spynet_forward": (c_int, [c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, c_size_t, c_int, _P]),
    "vsr_spynet_backward": (c_int, [c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_size_t, _P]),
    "vsr_spynet_forward_ex": (c_int, [c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, _P, c_size_t, c_int, _P]),
    "vsr_spynet_backward_ex": (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "vsr_cleaner_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "vsr_cleaner_forward": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, c_size_t, c_int, _P]),
    "vsr_cleaner_backward": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "vsr_flow_warp_fwd": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vsr_flow_warp_bwd": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vsr_flow_warp_bwd_flow": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vsr_flow_warp_fwd_ex": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "vsr_flow_warp_bwd_ex": (c_int, [c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "vsr_flow_warp_bwd_flow_ex": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "vsr_planar_to_pm": (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "vsr_pm_to_planar": (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "vsr_conv3x3_c64_fwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vsr_conv3x3_c64_chain_sync_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vsr_conv3x3_c64_chain_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "vsr_conv3x3_c64_dgrad": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vsr_conv_layer_fwd": (c_int, [c_int, c_int, _P, c_int, _P, _P, _P, c_int, c_int, _P, _P, c_int, _P, c_int, c_float, c_int, c_int, c_int, c_int, _P]),
    "vsr_conv_layer_bwd_scratch_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "vsr_conv_layer_bwd": (c_int, [c_int, c_int, _P, c_int, _P, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_float, c_int, _P, _P, _P, _P,
                                   _P, c_size_t, c_int, c_int, c_int, _P]),
    "vsr_conv3x3_c64_wgrad_slab_floats": (c_size_t, []),
    "vsr_conv3x3_c64_wgrad": (c_int, [c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vsr_charbonnier_scratch_floats": (c_size_t, []),
    "vsr_charbonnier_fwd_bwd": (c_int, [_P, _P, _P, _P, _P, c_longlong, c_float, _P]),
    "vsr_optim_scratch_floats": (c_size_t, []),
    "vsr_adam_clip_step": (c_int, [_P, _P, _P, _P, c_longlong, c_float, c_float, c_float, c_float, c_float, c_int, c_float,
                                   c_float, _P, _P, _P]),
    "vsr_grad_norm": (c_int, [_P, c_longlong, c_float, _P, _P, _P]),
    "vsr_resize_bilinear": (c_int, [_P, _P, c_longlong, c_int, c_int, c_int, c_int, _P]),
    "vsr_disc_workspace_bytes": (c_size_t, [ctypes.POINTER(DiscDesc), c_int]),
    "vsr_disc_forward": (c_int, [ctypes.POINTER(DiscDesc), _P, c_int, _P, _P, _P, c_size_t, c_int, _P]),
    "vsr_disc_backward": (c_int, [ctypes.POINTER(DiscDesc), _P, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "vsr_spectral_norm": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vsr_spectral_norm_backward": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "vsr_bce_with_logits": (c_int, [_P, c_float, _P, _P, c_longlong, _P]),
    "vsr_window_attention_fwd": (c_int, [ctypes.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P]),
    "vsr_window_attention_bwd": (c_int, [ctypes.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vsr_mask_pack": (c_int, [_P, _P, c_int, c_int, _P]),
    "vsr_rpb_gather": (c_int, [_P, _P, c_int, _P, c_int, c_int, _P]),
    "vsr_rpb_scatter": (c_int, [_P, _P, c_int, _P, c_int, c_int, _P]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load libvsrlab_hip.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"vsrlab_amd: HIP library not built ({LIB_PATH} missing). Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C vsrlab_amd/csrc`.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status, what):
    if status != 0:
        msg = load().vsr_status_string(status).decode()
        raise RuntimeError(f"vsrlab_amd: {what} failed: {msg} (status {status})")
