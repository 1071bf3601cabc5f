"""Loader + struct mirrors for include/vrpgym_hip.h."""
import ctypes as C
import os

KIND_TSP, KIND_VRP, KIND_IRP = 0, 1, 2
MAX_LAYERS = 16   # VRP_MAX_LAYERS
ABI_VERSION = 8   # include/vrpgym_hip.h: VRP_ABI_VERSION (struct layouts below mirror that header)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f32p = C.c_void_p
c_vp = C.c_void_p


class Env(C.Structure):
    """struct vrp_env"""
    _fields_ = [("kind", C.c_int32), ("B", C.c_int32), ("N", C.c_int32), ("flags", C.c_int32),
                ("pos", c_vp), ("demand", c_vp), ("depot", c_vp), ("visited", c_vp),
                ("mask", c_vp), ("cur", c_vp), ("load", c_vp)]


class EncoderLayer(C.Structure):
    """struct vrp_encoder_layer"""
    _fields_ = [(n, c_vp) for n in (
        "in_proj_weight", "in_proj_bias", "out_proj_weight", "out_proj_bias",
        "bn1_weight", "bn1_bias", "bn1_running_mean", "bn1_running_var",
        "bn1_num_batches_tracked",
        "ff0_weight", "ff0_bias", "ff2_weight", "ff2_bias",
        "bn2_weight", "bn2_bias", "bn2_running_mean", "bn2_running_var",
        "bn2_num_batches_tracked")]


class EncoderWeights(C.Structure):
    """struct vrp_encoder_weights"""
    _fields_ = [("node_dim", C.c_int32), ("depot_dim", C.c_int32), ("hidden", C.c_int32),
                ("num_layers", C.c_int32), ("heads", C.c_int32), ("reserved_", C.c_int32),
                ("node_embed_weight", c_vp), ("node_embed_bias", c_vp),
                ("depot_embed_weight", c_vp), ("depot_embed_bias", c_vp),
                ("layer", EncoderLayer * MAX_LAYERS), ("split", c_vp)]


class EncoderLayerGrads(C.Structure):
    """struct vrp_encoder_layer_grads"""
    _fields_ = [(n, c_vp) for n in (
        "in_proj_weight", "in_proj_bias", "out_proj_weight", "out_proj_bias",
        "bn1_weight", "bn1_bias", "ff0_weight", "ff0_bias", "ff2_weight", "ff2_bias",
        "bn2_weight", "bn2_bias")]


class EncoderGrads(C.Structure):
    """struct vrp_encoder_grads"""
    _fields_ = [("node_embed_weight", c_vp), ("node_embed_bias", c_vp),
                ("depot_embed_weight", c_vp), ("depot_embed_bias", c_vp),
                ("layer", EncoderLayerGrads * MAX_LAYERS)]


class DecoderWeights(C.Structure):
    """struct vrp_decoder_weights"""
    _fields_ = [(n, c_vp) for n in (
        "first_node", "last_node", "q_proj_weight", "k_proj_weight", "v_proj_weight",
        "in_proj_bias", "out_proj_weight", "out_proj_bias", "kp_weight",
        "att_output_weight", "context_proj_weight")]


class RolloutIO(C.Structure):
    """struct vrp_rollout_io"""
    _fields_ = [(n, c_vp) for n in ("acc_loss", "acc_logp", "notdone", "actions", "forced",
                                    "noise", "logits", "step_logp", "mask_trace",
                                    "load_trace")] + [("noise_seed", C.c_uint64), ("logit_clip", C.c_float)]


class DecoderGrads(C.Structure):
    """struct vrp_decoder_grads"""
    _fields_ = [(n, c_vp) for n in (
        "first_node", "last_node", "q_proj_weight", "k_proj_weight", "v_proj_weight",
        "in_proj_bias", "out_proj_weight", "out_proj_bias", "kp_weight",
        "att_output_weight", "context_proj_weight")]


def library_path():
    return os.environ.get("VRPGYM_HIP_LIB", os.path.join(_HERE, "libvrpgym_hip.so"))


def _declare(lib):
    i32, i64, vp = C.c_int, C.c_int64, C.c_void_p
    P = C.POINTER
    sig = {
        "vrp_env_reset": (i32, [P(Env), vp]),
        "vrp_env_mask": (i32, [P(Env), i32, vp]),
        "vrp_env_step": (i32, [P(Env), vp, i32, vp, vp, vp]),
        "vrp_env_features": (i32, [P(Env), vp, vp, vp]),
        "vrp_encoder_workspace_bytes": (i64, [i32, i32, i32]),
        "vrp_decoder_workspace_bytes": (i64, [i32, i32, i32]),
        "vrp_decoder_derived_bytes": (i64, []),
        "vrp_encoder_forward": (i32, [P(EncoderWeights), i32, i32, i32, vp, vp, vp, vp, vp]),
        "vrp_encoder_split_bytes": (i64, [i32, i32]),
        "vrp_encoder_prepare": (i32, [P(EncoderWeights), vp, vp]),
        "vrp_decoder_prepare": (i32, [i32, P(DecoderWeights), vp, vp]),
        "vrp_decode_prologue": (i32, [i32, vp, i32, i32, vp, vp, vp]),
        "vrp_decode_step": (i32, [i32, vp, P(DecoderWeights), P(Env), vp, vp, P(RolloutIO),
                                  i32, i32, i32, vp]),
        "vrp_decode_first_row": (i32, [i32, vp, i32, i32, vp, vp, vp]),
        "vrp_rollout": (i32, [i32, P(EncoderWeights), P(DecoderWeights), vp, P(Env), i32, i32,
                              vp, vp, vp, P(RolloutIO), i32, vp]),
        "vrp_rollout_encode": (i32, [i32, P(EncoderWeights), vp, P(Env), i32, vp, vp, vp,
                                     P(RolloutIO), i32, vp]),
        "vrp_rollout_steps": (i32, [i32, vp, P(DecoderWeights), P(Env), vp, vp, P(RolloutIO),
                                    i32, i32, vp]),
        "vrp_rollout_steps_range": (i32, [i32, vp, P(DecoderWeights), P(Env), vp, vp,
                                          P(RolloutIO), i32, i32, i32, i32, vp]),
        "vrp_draw_instances_host": (i32, [vp, vp, i32, i32, vp, vp, vp]),
        "vrp_draw_instances_host_range": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp]),
        "vrp_debug_exp1_from_bits": (i32, [vp, vp, i32, vp]),
        "vrp_draw_instances_device": (i32, [C.c_uint64, C.c_uint64, i32, i32, i32, vp, vp, vp, vp]),
        "vrp_random_rollout": (i32, [P(Env), C.c_uint64, C.c_uint64, i32, i32, vp, vp, vp, vp]),
        "vrp_encoder_tape_bytes": (i64, [i32, i32, i32, i32]),
        "vrp_encoder_forward_tape": (i32, [P(EncoderWeights), i32, i32, vp, vp, vp, vp, i32, vp]),
        "vrp_encoder_backward_workspace_bytes": (i64, [i32, i32, i32]),
        "vrp_encoder_backward": (i32, [P(EncoderWeights), P(EncoderGrads), i32, i32, vp, vp, vp,
                                       vp, vp, vp]),
        "vrp_decoder_backward_workspace_bytes": (i64, [i32, i32, i32, i32]),
        "vrp_decoder_backward": (i32, [i32, P(DecoderWeights), P(DecoderGrads), i32, i32, i32, vp,
                                       vp, vp, vp, vp, vp, vp, vp, vp]),
        "vrp_gemm_tn_workspace_bytes": (i64, [i32, i32, i32]),
        "vrp_gemm_tn": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, vp]),
        "vrp_colsum_workspace_bytes": (i64, [i32, i32]),
        "vrp_colsum": (i32, [vp, i32, i32, i32, vp, i32, vp, vp]),
        "vrp_bn_bwd_workspace_bytes": (i64, []),
        "vrp_bn_bwd": (i32, [vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp]),
        "vrp_attention_bwd": (i32, [vp, vp, vp, i32, i32, vp]),
        "vrp_gemm_nt_gated": (i32, [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
        "vrp_gemm_nt": (i32, [vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
        "vrp_step_kernel_name": (C.c_char_p, [i32, i32, i32, i32]),
        "vrp_persistent_capacity": (i32, []),
        "vrp_persistent_failures": (i32, []),
        "vrp_encoder_kernel_name": (C.c_char_p, [P(EncoderWeights), i32, i32, i32]),
        "vrp_source_hash": (C.c_char_p, []),
        "vrp_last_error": (C.c_char_p, []),
        "vrp_abi_version": (i32, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError = symbol missing from the build
        fn.restype, fn.argtypes = res, args
    return sig


EXPORTS = None


def lib():
    """The loaded library.  torch is imported first so that exactly one HIP runtime
    (torch's bundled libamdhip64, soname libamdhip64.so.7) is mapped."""
    global _LIB, EXPORTS
    if _LIB is None:
        import torch  # noqa: F401  (loads torch/lib/libamdhip64.so before our .so)
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"libvrpgym_hip.so not found at {path}: build it with "
                "`python __graft_entry__.py` or `make -C vrp-gym_amd/csrc` "
                "(there is no CPU fallback)")
        cand = C.CDLL(path, mode=C.RTLD_GLOBAL)
        cand.vrp_abi_version.restype = C.c_int
        got = cand.vrp_abi_version()
        if got != ABI_VERSION:
            # a stale build: its vrp_rollout_io / vrp_encoder_weights layouts differ from the
            # structs mirrored here -- calling into it would read garbage pointers
            raise RuntimeError(
                f"{path} implements ABI {got}, this binding needs {ABI_VERSION}: rebuild it "
                "(`python __graft_entry__.py` or `make -C vrp-gym_amd/csrc`)")
        EXPORTS = _declare(cand)
        _LIB = cand
    return _LIB


_CENSUS_DONE = set()


def require_gpu():
    """The library, on a machine with a GPU.  The first call per device also takes the
    persistent step kernel's residency census (vrp_persistent_capacity) while nothing of this
    process is in flight, so the measurement is not disturbed by a rollout's own kernels."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("vrp-gym_amd needs an AMD GPU (MI355X/gfx950) visible to "
                           "PyTorch-ROCm; there is no CPU fallback")
    L = lib()
    dev = torch.cuda.current_device()
    if dev not in _CENSUS_DONE:
        _CENSUS_DONE.add(dev)
        if not torch.cuda.is_current_stream_capturing():
            L.vrp_persistent_capacity()
    return L


def check(rc):
    if rc != 0:
        msg = lib().vrp_last_error()
        raise RuntimeError(f"libvrpgym_hip error {rc}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
