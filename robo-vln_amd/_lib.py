"""ctypes binding of libhcm.so (include/hcm.h).  The HIP library is the product path: if it is missing
this module raises -- there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HCM_DEV_LIB=1 loads the development build (`make DEV=1` -> libhcm_dev.so: profiling knobs compiled in); never the default
LIB_PATH = os.path.join(_HERE, "libhcm_dev.so" if os.environ.get("HCM_DEV_LIB", "0") not in ("", "0") else "libhcm.so")

HCM_F32, HCM_BF16, HCM_I32, HCM_I64, HCM_U8, HCM_F16 = 0, 1, 2, 3, 4, 5
HCM_HIGH, HCM_LOW, HCM_CMA = 0, 1, 2
HCM_ENC_RESNET, HCM_ENC_SIMPLECNN = 0, 1
HCM_LSTM, HCM_GRU = 0, 1
(HCM_NUM_RECURRENT_LAYERS, HCM_HIDDEN_SIZE, HCM_NUM_ACTIONS, HCM_RECORD_WIDTH, HCM_WORKSPACE_BYTES,
 HCM_WEIGHT_BYTES, HCM_MAX_BATCH, HCM_GRAPH_LAUNCHES, HCM_EAGER_LAUNCHES, HCM_FP16_FALLBACK, HCM_CALIB_MAX_BERT, HCM_CALIB_MAX_DEPTH,
 HCM_CALIB_NONFINITE, HCM_CALIB_MAX_RGB, HCM_CALIB_MAX_VLA, HCM_STEP_NONFINITE, HCM_RANGE_FOLD, HCM_GATHER_JOINED) = range(18)
# `precision` of HCMEngine / CMAEngine -> hcm_config.precision (include/hcm.h): "fp16" is the measured 16-bit mode
PRECISIONS = {"fp32": HCM_F32, "fp16": HCM_F16, "bf16": HCM_BF16}
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
HCM_ACT_REUSE_INSTRUCTION = 1
HCM_ACT_HOST_FRAMES = 2
HCM_ACT_CHAIN_GRAPHS = 4

STATUS_EXC = {-1: ValueError, -2: RuntimeError, -3: KeyError, -4: ValueError, -5: RuntimeError, -6: ValueError,
              -7: MemoryError}


class HcmConfigStruct(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "precision", "max_batch", "rgb_h", "rgb_w", "depth_h", "depth_w", "instr_len",
        "rgb_encoder", "depth_encoder", "rgb_out", "depth_out", "depth_baseplanes",
        "vla_layers", "d_model", "vla_heads", "d_ff", "vis_in", "ins_in",
        "hidden", "rnn_type", "num_actions", "num_sub_tasks", "lo_actions",
        "bert_layers", "bert_hidden", "bert_heads", "bert_inter", "bert_vocab", "bert_max_pos",
        "build_high", "build_low", "use_prev_action", "ablate_instruction", "progress_monitor", "ablate_depth", "ablate_rgb")] + [("reserved", C.c_int32 * 8)]


class HcmCmaConfigStruct(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "precision", "max_batch", "rgb_h", "rgb_w", "depth_h", "depth_w", "instr_len",
        "vocab_size", "embedding_size", "instr_hidden", "bidirectional", "rgb_out", "depth_out", "depth_baseplanes",
        "hidden", "rnn_type", "num_actions", "use_prev_action", "rcm_state_encoder", "progress_monitor",
        "instr_rnn", "ablate_instruction", "ablate_depth", "ablate_rgb")] + [("reserved", C.c_int32 * 4)]


EXPORTS = {
    "hcm_cma_create": (C.c_int, [C.POINTER(HcmCmaConfigStruct), C.POINTER(C.c_void_p)]),
    "hcm_cma_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hcm_create": (C.c_int, [C.POINTER(HcmConfigStruct), C.POINTER(C.c_void_p)]),
    "hcm_load_tensor": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "hcm_finalize": (C.c_int, [C.c_void_p]),
    "hcm_high_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hcm_low_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hcm_high_forward_seq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hcm_low_forward_seq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hcm_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hcm_act_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "hcm_comm_unique_id": (C.c_int, [C.c_void_p]),
    "hcm_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "hcm_comm_abort": (C.c_int, [C.c_void_p]),
    "hcm_gather_poison": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hcm_act_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hcm_refresh_instruction": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_void_p]),
    "hcm_calibrate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hcm_release_host_weights": (C.c_int, [C.c_void_p]),
    "hcm_query": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    "hcm_guard_poll": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "hcm_last_error": (C.c_char_p, [C.c_void_p]),
    "hcm_destroy": (None, [C.c_void_p]),
    "hcm_debug_enable_taps": (C.c_int, [C.c_void_p, C.c_int]),
    "hcm_debug_get_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hcm_debug_igemm_prof": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "hcm_debug_gemm256_prof": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "hcm_debug_marks": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_char_p, C.c_int]),
    "hcm_op_conv2d": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 11 + [C.c_void_p]),
    "hcm_op_bottleneck_tail": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_void_p]),
    "hcm_op_bottleneck_tail_next": (C.c_int, [C.c_void_p] * 10 + [C.c_int] * 7 + [C.c_void_p]),
    "hcm_op_bottleneck_tail_ds": (C.c_int, [C.c_void_p] * 10 + [C.c_int] * 5 + [C.c_void_p]),
    "hcm_op_conv2d_gn": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 11 + [C.c_float, C.c_int, C.c_void_p]),
    "hcm_op_conv2d_gn_large": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 11 + [C.c_float, C.c_int, C.c_void_p]),
    "hcm_op_stem_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 13 + [C.c_float, C.c_int, C.c_void_p]),
    "hcm_op_stem_conv_packed": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "hcm_op_stem_pool_fused": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_float, C.c_void_p, C.c_void_p]),
    "hcm_op_stem_pool_fused_red": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_float] + [C.c_void_p] * 5),
    "hcm_op_stem_conv_packed_pool": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hcm_op_depth_conv8x8s4": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]),
    "hcm_op_stem_scratch_bytes": (C.c_int64, [C.c_int] * 3),
    "hcm_op_linear": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]),
    "hcm_op_linear_impl": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p]),
    "hcm_op_vla_layer": (C.c_int, [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 11 + [C.c_int] * 5 + [C.c_void_p]),
    "hcm_op_vla_layer_frag": (C.c_int, [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 11 + [C.c_int] * 5 + [C.c_void_p]),
    "hcm_op_attention": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 9 + [C.c_void_p]),
    "hcm_op_simplecnn3": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 3 + [C.c_void_p]),
    "hcm_op_pack_frag": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hcm_op_bert_attn_block": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 3 + [C.c_void_p, C.c_float, C.c_void_p]),
    "hcm_op_layernorm": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_float, C.c_void_p]),
    "hcm_op_layernorm_post": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_void_p] + [C.c_int] * 3 + [C.c_float, C.c_void_p]),
    "hcm_op_groupnorm": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_void_p]),
    "hcm_op_conv2d_gn_pool": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 11 + [C.c_float, C.c_void_p]),
    "hcm_op_conv2d_gn_res2": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 9 + [C.c_float, C.c_int, C.c_void_p]),
    "hcm_op_maxpool3x3s2": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 5 + [C.c_void_p]),
}

_lib = None


def lib():
    """Load libhcm.so (built in-tree by `__graft_entry__.build()` / `make -C robo-vln_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP library is the only compute path; there is no CPU fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(l, name)      # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error(handle=None):
    msg = lib().hcm_last_error(handle)
    return msg.decode() if msg else ""


def check(rc, handle=None):
    if rc == 0:
        return
    msg = lib().hcm_last_error(handle)
    msg = msg.decode() if msg else f"libhcm error {rc}"
    raise STATUS_EXC.get(rc, RuntimeError)(msg)
