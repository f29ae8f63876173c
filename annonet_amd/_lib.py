"""ctypes binding of libannonet_hip.so (include/annonet_hip.h).  There is no CPU fallback: if the library is
missing or an entry point fails, this raises."""
import ctypes as C
import os
import subprocess

# The training step overlaps its filter-gradient kernels with the bn / backward-data chain on a second HIP stream.  HIP maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin; with torch.distributed loaded (its own
# streams) the two streams of a trainer can land on ONE queue and serialise (measured 2.29 vs 2.02 ms/step).  More
# queues avoid the collision; this only takes effect if set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ANH_LIBRARY") or os.path.join(_HERE, "lib", "libannonet_hip.so")   # ANH_LIBRARY: an instrumented build (tools/ws_phase_profile.py)
CSRC = os.path.join(_HERE, "csrc")

ANH_FP32, ANH_BF16 = 0, 1
LABEL_IGNORE = 65535


class AnnonetHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[anh_status {code}] {message}")
        self.code = code


class NetConfig(C.Structure):
    _fields_ = [("levels", C.c_int), ("in_channels", C.c_int), ("classes", C.c_int), ("width_scaler", C.c_double),
                ("min_filters", C.c_int), ("precision", C.c_int)]


class LayerDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("type", "k", "stride", "pad", "cin", "cout", "in_a", "in_b", "has_bn", "has_bias")] + \
               [(n, C.c_int64) for n in ("w_off", "b_off", "g_off", "beta_off", "rs_off")]


class WLabel(C.Structure):
    _fields_ = [("label", C.c_uint16), ("weight", C.c_float)]


class ExchangeStats(C.Structure):   # anh_exchange_stats
    _fields_ = [("replicas", C.c_int), ("early_reduce", C.c_int), ("uses_rccl", C.c_int), ("rccl_version", C.c_int),
                ("steps", C.c_int64), ("samples", C.c_int64), ("worker_calls", C.c_int64), ("bucket_bytes", C.c_int64),
                ("host_us_mean", C.c_double), ("host_us_last", C.c_double), ("host_wait_us_mean", C.c_double),
                ("allreduce_tail_us_mean", C.c_double), ("allreduce_head_us_mean", C.c_double),
                ("allreduce_tail_us_last", C.c_double), ("allreduce_head_us_last", C.c_double)]


class Rect(C.Structure):
    _fields_ = [("left", C.c_long), ("top", C.c_long), ("right", C.c_long), ("bottom", C.c_long)]

    def tuple(self):
        return (self.left, self.top, self.right, self.bottom)


class Tile(C.Structure):
    _fields_ = [("full_rect", Rect), ("unique_rect", Rect)]


class CropSpec(C.Structure):   # anh_crop_spec
    _fields_ = [("image", C.c_int), ("left", C.c_long), ("top", C.c_long), ("flip_left_right", C.c_int), ("flip_upside_down", C.c_int),
                ("brightness_change", C.c_double), ("further_downscaling_factor", C.c_double), ("noise_level", C.c_int), ("noise_seed", C.c_uint64),
                ("color_offset", C.c_int * 3)]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("type", "k", "stride", "pad", "cin", "cout")]


class OpInput(C.Structure):
    _fields_ = [("x", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p)]


class OpBnDy(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("da", "y", "scale", "shift", "mean", "invstd", "coef")]


class TilingParams(C.Structure):
    _fields_ = [("max_tile_width", C.c_int), ("max_tile_height", C.c_int), ("overlap_x", C.c_int), ("overlap_y", C.c_int)]


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_P = C.c_void_p
_SIGNATURES = {  # ConvDesc / OpInput are defined above
    # name: (restype, argtypes)
    "anh_last_error": (C.c_char_p, []),
    "anh_free": (None, [_P]),
    "anh_set_device": (C.c_int, [C.c_int]),
    "anh_device_count": (C.c_int, []),
    "anh_required_input_dim": (C.c_int, [C.POINTER(NetConfig)]),
    "anh_recommended_input_dim": (C.c_int, [C.c_int, C.c_int]),
    "anh_net_layer_count": (C.c_int, [C.POINTER(NetConfig)]),
    "anh_net_layer": (C.c_int, [C.POINTER(NetConfig), C.c_int, C.POINTER(LayerDesc)]),
    "anh_net_param_count": (C.c_int64, [C.POINTER(NetConfig)]),
    "anh_net_running_count": (C.c_int64, [C.POINTER(NetConfig)]),
    "anh_runtime_create": (C.c_int, [C.POINTER(NetConfig), C.POINTER(_P)]),
    "anh_runtime_destroy": (None, [_P]),
    "anh_runtime_config": (C.c_int, [_P, C.POINTER(NetConfig)]),
    "anh_runtime_set_params": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64]),
    "anh_runtime_get_params": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64]),
    "anh_runtime_serialize": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "anh_runtime_deserialize": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(_P)]),
    "anh_runtime_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "anh_runtime_forward_device": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "anh_infer": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.POINTER(TilingParams), _P, _P]),
    "anh_infer_device": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.POINTER(TilingParams), C.POINTER(Tile), C.c_size_t, _P, _P]),
    "anh_argmax_device": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "anh_runtime_set_stream": (C.c_int, [_P, _P]),
    "anh_runtime_get_stream": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "anh_runtime_synchronize": (C.c_int, [_P]),
    "anh_trainer_create": (C.c_int, [C.POINTER(_P)]),
    "anh_trainer_destroy": (None, [_P]),
    "anh_trainer_set_net_width": (C.c_int, [_P, C.c_double, C.c_int]),
    "anh_trainer_set_class_count": (C.c_int, [_P, C.c_size_t]),
    "anh_trainer_set_levels": (C.c_int, [_P, C.c_int]),
    "anh_trainer_set_input_channels": (C.c_int, [_P, C.c_int]),
    "anh_trainer_set_precision": (C.c_int, [_P, C.c_int]),
    "anh_trainer_set_seed": (C.c_int, [_P, C.c_uint64]),
    "anh_trainer_initialize": (C.c_int, [_P]),
    "anh_trainer_set_learning_rate": (C.c_int, [_P, C.c_double]),
    "anh_trainer_set_learning_rate_shrink_factor": (C.c_int, [_P, C.c_double]),
    "anh_trainer_set_iterations_without_progress_threshold": (C.c_int, [_P, C.c_ulong]),
    "anh_trainer_set_previous_loss_values_dump_amount": (C.c_int, [_P, C.c_ulong]),
    "anh_trainer_set_all_bn_running_stats_window_sizes": (C.c_int, [_P, C.c_ulong]),
    "anh_trainer_set_synchronization_file": (C.c_int, [_P, C.c_char_p, C.c_double]),
    "anh_trainer_be_verbose": (C.c_int, [_P]),
    "anh_trainer_set_sgd": (C.c_int, [_P, C.c_double, C.c_double]),
    "anh_trainer_get_learning_rate": (C.c_double, [_P]),
    "anh_trainer_get_last_loss": (C.c_double, [_P]),
    "anh_trainer_get_step_count": (C.c_ulong, [_P]),
    "anh_trainer_config": (C.c_int, [_P, C.POINTER(NetConfig)]),
    "anh_trainer_step": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.c_int, C.c_int, C.c_int]),
    "anh_trainer_forward_backward_device": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_double]),
    "anh_trainer_apply_update": (C.c_int, [_P, C.c_double]),
    "anh_trainer_grad_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "anh_trainer_get_params": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64]),
    "anh_trainer_set_params": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64]),
    "anh_trainer_get_grads": (C.c_int, [_P, _P, C.c_int64]),
    "anh_trainer_get_momentum": (C.c_int, [_P, _P, C.c_int64]),
    "anh_trainer_set_momentum": (C.c_int, [_P, _P, C.c_int64]),
    "anh_trainer_snapshot_runtime": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "anh_trainer_save_state": (C.c_int, [_P, C.c_char_p]),
    "anh_trainer_load_state": (C.c_int, [_P, C.c_char_p]),
    "anh_trainer_set_stream": (C.c_int, [_P, _P]),
    "anh_trainer_get_stream": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "anh_runtime_stores_activations": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "anh_trainer_early_grads": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "anh_trainer_wait_early_grads": (C.c_int, [_P, _P]),
    "anh_trainer_step_graph_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "anh_trainer_synchronize": (C.c_int, [_P]),
    "anh_trainer_layer_tensor": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64, C.POINTER(C.c_int)]),
    "anh_profile_enable": (C.c_int, [_P, C.c_int, C.c_int]),
    "anh_profile_set_filter": (C.c_int, [_P, C.c_int, C.c_char_p]),
    "anh_profile_set_sampling": (C.c_int, [_P, C.c_int, C.c_int]),
    "anh_profile_reset": (C.c_int, [_P, C.c_int]),
    "anh_profile_count": (C.c_int, [_P, C.c_int]),
    "anh_trainer_replica_params": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "anh_set_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "anh_handle_replicas": (C.c_int, [_P, C.c_int]),
    "anh_trainer_exchange_stats": (C.c_int, [_P, C.POINTER(ExchangeStats)]),
    "anh_trainer_reset_exchange_stats": (None, [_P]),
    "anh_host_register": (C.c_int, [_P, C.c_size_t]),
    "anh_host_unregister": (C.c_int, [_P]),
    "anh_labels_rect_to_host": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "anh_shard_range": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "anh_cross_replica_overlaps": (C.c_int, [C.POINTER(Tile), C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(Rect)), C.POINTER(C.c_size_t)]),
    "anh_profile_launch_order": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "anh_profile_entry": (C.c_int, [_P, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "anh_op_conv_forward": (C.c_int, [C.c_int, C.POINTER(ConvDesc), C.c_int, C.c_int, C.c_int, C.POINTER(OpInput), C.POINTER(OpInput), _P, _P, _P, C.POINTER(C.c_int)]),
    "anh_op_conv_backward_data": (C.c_int, [C.c_int, C.POINTER(ConvDesc), C.c_int, C.c_int, C.c_int, _P, _P, _P, C.POINTER(C.c_int)]),
    "anh_op_conv_backward_filter": (C.c_int, [C.c_int, C.POINTER(ConvDesc), C.c_int, C.c_int, C.c_int, C.POINTER(OpInput), C.POINTER(OpInput), _P, _P, C.POINTER(C.c_int)]),
    "anh_op_conv_forward_stats": (C.c_int, [C.c_int, C.POINTER(ConvDesc), C.c_int, C.c_int, C.c_int, C.POINTER(OpInput), C.POINTER(OpInput), _P, _P, _P, C.POINTER(C.c_int)]),
    "anh_op_conv_backward_data_bn": (C.c_int, [C.c_int, C.POINTER(ConvDesc), C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_int)]),
    "anh_op_conv_backward_filter_bn": (C.c_int, [C.c_int, C.POINTER(ConvDesc), C.c_int, C.c_int, C.c_int, _P, C.POINTER(OpBnDy), _P, C.POINTER(C.c_int)]),
    "anh_get_tiles": (C.c_int, [C.c_int, C.c_int, C.POINTER(TilingParams), C.POINTER(C.POINTER(Tile)), C.POINTER(C.c_size_t)]),
    "anh_set_weights": (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_double, _P]),
    "anh_random_rect_containing_point": (C.c_int, [C.c_uint32, C.c_uint32, C.c_long, C.c_long, C.c_long, C.c_long, C.POINTER(Rect)]),
    "anh_outpaint": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(Rect)]),
    "anh_dataset_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "anh_dataset_destroy": (None, [C.c_void_p]),
    "anh_dataset_add": (C.c_int, [C.c_void_p, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "anh_dataset_remove": (C.c_int, [C.c_void_p, C.c_int]),
    "anh_dataset_resident_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "anh_dataset_crop_batch": (C.c_int, [C.c_void_p, C.POINTER(CropSpec), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _P, _P]),
    "anh_trainer_step_crops": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CropSpec), C.c_int, C.c_int, C.c_double, C.c_double]),
    "anh_dnn_envelope_pack": (C.c_int, [C.c_char_p, C.c_size_t, C.c_double, _P, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "anh_dnn_envelope_unpack": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_double),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "anh_ignore_large_nonzero_regions": (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int64)]),
    "anh_count_steps_without_decrease": (C.c_int64, [_P, C.c_int64, C.c_double]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
        # torch wheels bundle their own libamdhip64 (same soname as /opt/rocm's).  Whichever copy is loaded first
        # serves the whole process; a second copy cannot open the GPU.  Loading torch first keeps ONE HIP runtime, so
        # device pointers and streams are interchangeable between torch (plumbing) and this library.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            if os.environ.get("ANH_LIBRARY") and not hasattr(L, name):
                continue   # a comparison build of an older tree (tools/ab_multi.sh): entry points added since are simply absent
            fn = getattr(L, name)  # AttributeError here = the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc):
    if rc != 0:
        raise AnnonetHipError(rc, lib().anh_last_error().decode(errors="replace"))
