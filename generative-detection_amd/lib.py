"""ctypes binding of libodvae_hip.so (include/odvae_hip.h).  The product path has no CPU fallback:
every op raises if the library is missing or the tensors are not on a HIP device."""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libodvae_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "odvae_hip.h")

_c = ctypes
_P, _I, _L, _F, _Z = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_size_t

ABI_VERSION = 4   # == ODVAE_ABI_VERSION in include/odvae_hip.h; bumped whenever the exported surface changes

# name -> (restype, argtypes); mirrors include/odvae_hip.h one to one (tests/test_abi.py checks that)
PROTOTYPES = {
    "odvae_last_error": (_c.c_char_p, []),
    "odvae_abi_version": (_I, []),
    "odvae_target_arch": (_c.c_char_p, []),
    "odvae_gemm_f32_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "odvae_gemm_select_staging": (_I, [_I]),
    "odvae_gemm_f32": (_I, [_I, _I, _I, _I, _I, _F, _P, _I, _L, _P, _I, _L, _P, _I, _L, _P, _P, _I, _P, _Z, _P]),
    "odvae_gemm_softmax_bwd_f32": (_I, [_I, _I, _I, _F, _P, _I, _L, _P, _I, _L, _P, _P, _L, _P, _I, _L, _I, _P]),
    "odvae_rowdot_f32": (_I, [_P, _P, _L, _I, _P, _P]),
    "odvae_attn_row_bound_f32": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "odvae_gemm_exp_bound_f32": (_I, [_I, _I, _I, _F, _P, _I, _L, _P, _I, _L, _P, _L, _P, _I, _L, _I, _P]),
    "odvae_gemm_rownorm_f32": (_I, [_I, _I, _I, _P, _I, _L, _P, _I, _L, _P, _I, _L, _P, _L, _P, _I, _P]),
    "odvae_gemm_pred_f32": (_I, [_I, _I, _I, _I, _I, _F, _P, _I, _L, _P, _I, _L, _P, _I, _L, _I, _P, _P]),
    "odvae_softmax_rows_pred_f32": (_I, [_P, _P, _L, _I, _F, _P, _P, _P]),
    "odvae_rowdot_scale_f32": (_I, [_P, _P, _P, _L, _I, _P, _P, _P]),
    "odvae_gemm_softmax_bwd_scaled_f32": (_I, [_I, _I, _I, _F, _P, _I, _L, _P, _I, _L, _P, _P, _P, _L, _P, _I, _L, _I, _P]),
    "odvae_conv3x3_pack_reduce_pad": (_I, [_I]),
    "odvae_conv3x3_pack_out_pad": (_I, [_I]),
    "odvae_conv3x3_pack_floats": (_Z, [_I, _I]),
    "odvae_conv3x3_pack_f32": (_I, [_P, _I, _I, _P, _P, _P]),
    "odvae_conv3x3_up_pack_floats": (_Z, [_I, _I]),
    "odvae_conv3x3_pack_up_f32": (_I, [_P, _I, _I, _P, _P, _P]),
    "odvae_conv3x3_wino_reduce_pad": (_I, [_I]),
    "odvae_conv3x3_wino_out_pad": (_I, [_I]),
    "odvae_conv3x3_wino_pack_floats": (_Z, [_I, _I]),
    "odvae_conv3x3_pack_wino_f32": (_I, [_P, _I, _I, _P, _P, _P]),
    "odvae_conv3x3_wino_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P]),
    "odvae_conv3x3_wino4_reduce_pad": (_I, [_I]),
    "odvae_conv3x3_wino4_out_pad": (_I, [_I]),
    "odvae_conv3x3_wino4_pack_floats": (_Z, [_I, _I]),
    "odvae_conv3x3_wino4_supported": (_I, [_I, _I, _I, _I]),
    "odvae_conv3x3_pack_wino4_f32": (_I, [_P, _I, _I, _P, _P, _P]),
    "odvae_conv3x3_pack_wino_batch": (_I, [_P, _I, _P]),
    "odvae_conv3x3_pack_wino4_batch": (_I, [_P, _I, _P]),
    "odvae_conv3x3_wino4_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P]),
    "odvae_conv3x3_wino4_stats_chunks": (_I, [_I, _I]),
    "odvae_conv3x3_wino4_stats_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P]),
    "odvae_conv3x3_wino4_up_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P]),
    "odvae_conv3x3_wino4_pool_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "odvae_conv3x3_wino4_gnbwd_f32": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "odvae_conv3x3_f32": (_I, [_I, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "odvae_conv3x3_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "odvae_conv3x3_wgrad_f32": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "odvae_conv3x3_wgrad_wino_supported": (_I, [_I, _I, _I, _I, _I]),
    "odvae_conv3x3_wgrad_wino_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "odvae_conv3x3_wgrad_wino_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "odvae_groupnorm_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "odvae_groupnorm_fwd_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _I, _P, _P, _P, _P, _Z, _P]),
    "odvae_groupnorm_fwd_partials_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _I, _P, _P, _P, _P, _I, _P]),
    "odvae_groupnorm_apply_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "odvae_groupnorm_apply_bf16": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "odvae_groupnorm_bwd_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "odvae_groupnorm_bwd_partials_f32": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "odvae_groupnorm_select_backward": (_I, [_I]),
    "odvae_groupnorm_fused_timeouts": (_I, []),
    "odvae_device_health": (_I, [_P, _P, _I, _I]),
    "odvae_attn_softmax_fallbacks": (_I, [_I]),
    "odvae_softmax_rows_f32": (_I, [_P, _P, _L, _I, _F, _P]),
    "odvae_softmax_rows_bwd_f32": (_I, [_P, _P, _P, _L, _I, _F, _P]),
    "odvae_upsample2x_bwd_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "odvae_rescale_minmax_f32": (_I, [_P, _P, _I, _I, _I, _P, _P, _Z, _P]),
    "odvae_gaussian_sample_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "odvae_gaussian_kl_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "odvae_gaussian_bwd_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "odvae_l1_masked_sum_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "odvae_l1_masked_bwd_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "odvae_colsum_workspace_bytes": (_Z, [_L, _I]),
    "odvae_colsum_f32": (_I, [_P, _L, _I, _P, _P, _Z, _P]),
    "odvae_grad_norm_f32": (_I, [_P, _L, _F, _P, _P, _Z, _P]),
    "odvae_adam_step_f32": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _P, _P]),
    "odvae_nhwc_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "odvae_mul_mask_f32": (_I, [_P, _P, _P, _L, _I, _P]),
    "odvae_latent_combine_f32": (_I, [_P, _P, _P, _P, _L, _P]),
    "odvae_im2col4x4_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "odvae_col2im4x4_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "odvae_weight4x4_reorder_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "odvae_batchnorm_workspace_bytes": (_Z, [_L, _I]),
    "odvae_batchnorm_lrelu_fwd_f32": (_I, [_P, _L, _I, _P, _P, _F, _F, _F, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "odvae_batchnorm_lrelu_bwd_f32": (_I, [_P, _P, _L, _I, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _Z, _P]),
    "odvae_leaky_relu_f32": (_I, [_P, _P, _F, _L, _P]),
    "odvae_leaky_relu_bwd_f32": (_I, [_P, _P, _P, _F, _L, _P]),
    "odvae_scaling_layer_f32": (_I, [_P, _P, _P, _P, _L, _I, _I, _P]),
    "odvae_maxpool2x2_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "odvae_maxpool2x2_bwd_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "odvae_lpips_distance_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "odvae_lpips_distance_bwd_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "odvae_pose_losses_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P]),
    "odvae_pose_losses_bwd_f32": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "odvae_linear_workspace_bytes": (_Z, [_I, _I, _I]),
    "odvae_linear_fwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "odvae_linear_bwd_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _Z, _P]),
    # bf16 mixed-precision path
    "odvae_conv_bf16_reduce_pad": (_I, [_I]),
    "odvae_conv_bf16_out_pad": (_I, [_I]),
    "odvae_conv_bf16_pack_elems": (_Z, [_I, _I, _I]),
    "odvae_conv_pack_bf16": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "odvae_conv_bf16": (_I, [_I, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "odvae_conv_bf16_stats_chunks": (_I, [_I, _I]),
    "odvae_conv_bf16_stats_supported": (_I, [_I, _I]),
    "odvae_conv_bf16_stats": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P]),
    "odvae_conv_wgrad_bf16_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "odvae_conv_wgrad_bf16": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "odvae_flash_attn_supported": (_I, [_I, _I, _I]),
    "odvae_flash_attn_fwd_bf16": (_I, [_P, _I, _I, _I, _F, _P, _P, _P]),
    "odvae_flash_attn_bwd_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "odvae_groupnorm_bf16_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "odvae_groupnorm_fwd_bf16": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _I, _P, _P, _P, _P, _Z, _P]),
    "odvae_groupnorm_fwd_partials_bf16": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _I, _P, _P, _P, _P, _I, _P]),
    "odvae_groupnorm_bwd_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "odvae_cast_pad_bf16": (_I, [_P, _L, _I, _I, _P, _P]),
    "odvae_cast_f32_from_bf16": (_I, [_P, _L, _P, _P]),
    "odvae_upsample2x_bwd_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "odvae_colsum_bf16_workspace_bytes": (_Z, [_L, _I]),
    "odvae_colsum_bf16": (_I, [_P, _L, _I, _P, _P, _Z, _P]),
    "odvae_patch_table_ints": (_I, [_I]),
    "odvae_patch_crop_resize_u8": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
}


class HipLibraryError(RuntimeError):
    """libodvae_hip.so is missing, stale or a kernel call failed."""


_lib = None


def header_symbols(path=HEADER_PATH):
    """Function names declared in include/odvae_hip.h."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(odvae_[a-z0-9_]+)\s*\(", text)))


def load():
    """dlopen the in-tree library and declare every prototype.  Raises HipLibraryError when it is absent
    (build it with `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "HIP kernel library %s not found; run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the OD-VAE hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    # the version first: a stale prebuilt library then fails here, with a message that says so, not at some symbol lookup
    try:
        lib.odvae_abi_version.restype = _I
        have = lib.odvae_abi_version()
    except AttributeError as e:
        raise HipLibraryError("%s does not export odvae_abi_version (not an OD-VAE kernel library?)" % LIB_PATH) from e
    if have != ABI_VERSION:
        raise HipLibraryError("ABI version mismatch: library %d, binding %d (stale libodvae_hip.so? rebuild with __graft_entry__.build())"
                              % (have, ABI_VERSION))
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.odvae_abi_version() != ABI_VERSION:
        raise HipLibraryError("ABI version mismatch: library %d, binding %d (stale libodvae_hip.so? rebuild with __graft_entry__.build())"
                              % (lib.odvae_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().odvae_last_error().decode("utf-8", "replace")
        raise HipLibraryError("%s failed (code %d): %s" % (what, rc, msg))


def stream_ptr():
    """hipStream_t of torch's current stream, as an integer for ctypes."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def require_device(*tensors, dtypes=(torch.float32,)):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipLibraryError(
                "OD-VAE HIP op called with a %s tensor: the hot path runs only on a HIP device (no CPU fallback)" % t.device)
        if t is not None and t.dtype not in dtypes:
            raise HipLibraryError("OD-VAE HIP op needs %s tensors, got %s" % (" / ".join(str(d) for d in dtypes), t.dtype))


class _Workspace:
    """One grow-only scratch buffer per device; kernels on one stream use it back to back."""

    def __init__(self):
        self.buf = {}

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 16384)
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b.data_ptr(), b.numel()


workspace = _Workspace()


class _Uploader:
    """Host -> device copies that do not stall the host.  `tensor.to(device)` from pageable memory synchronises the
    stream first (the host waits for every kernel already queued), which stops the host from running ahead of the GPU
    four times per forward pass (the reference draws its noise on the CPU, autoencoder.py:239-240).  Here the data is
    copied into a small ring of pinned staging buffers and sent with an asynchronous copy on the current stream; an event
    per slot keeps a buffer from being reused before its copy has run."""

    SLOTS = 8

    def __init__(self):
        self.rings = {}

    def __call__(self, t, device):
        device = torch.device(device)
        if t.device == device:
            return t
        if t.is_cuda or device.type != "cuda":
            return t.to(device)
        key = (t.dtype, t.numel())
        ring = self.rings.get(key)
        if ring is None:
            ring = self.rings[key] = {"next": 0, "slots": [None] * self.SLOTS}
        i = ring["next"]
        ring["next"] = (i + 1) % self.SLOTS
        slot = ring["slots"][i]
        if slot is None:
            slot = ring["slots"][i] = (torch.empty(t.numel(), dtype=t.dtype, pin_memory=True), torch.cuda.Event())
        else:
            slot[1].synchronize()   # eight copies ago: long done in practice
        slot[0].copy_(t.reshape(-1))
        out = slot[0].to(device, non_blocking=True).view(t.shape)
        slot[1].record(torch.cuda.current_stream(device))
        return out


upload = _Uploader()
