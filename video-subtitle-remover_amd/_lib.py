"""ctypes binding of libvsr_hip.so (C-ABI declared in include/vsr_hip.h).

There is deliberately no fallback: if the shared library has not been built
(``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C csrc``) importing this
module raises, and every compute entry point returns VSR_ERR_NOGPU without a HIP device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvsr_hip.so")

VSR_OK, VSR_ERR_ARG, VSR_ERR_STATE, VSR_ERR_HIP, VSR_ERR_NOGPU = 0, -1, -2, -3, -4
VARIANT = {"auto": 0, "det": 1}
BMODE_NK, BMODE_KN = 0, 1
ACT_NONE, ACT_LRELU02 = 0, 1
TILE_128x128, TILE_256x32, TILE_256x64, TILE_128x64 = 0, 1, 2, 3
TILE_256x128, TILE_256x256 = 4, 5      # 8-wave tiles of the fp16-operand kernels (variant 6, NK)
TILE_DIMS = {TILE_128x128: (128, 128), TILE_256x32: (256, 32), TILE_256x64: (256, 64), TILE_128x64: (128, 64), TILE_256x128: (256, 128), TILE_256x256: (256, 256)}


class VsrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvsr_hip error {code}: {msg}")
        self.code = code


class GGProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p),
                ("rowA", C.c_void_p), ("colA", C.c_void_p), ("rowB", C.c_void_p), ("colB", C.c_void_p),
                ("rowC", C.c_void_p), ("colC", C.c_void_p), ("rowR", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("tilesM", C.c_int32), ("tilesN", C.c_int32),
                ("splitK", C.c_int32), ("chunksPerSplit", C.c_int32), ("tileStart", C.c_int32), ("act", C.c_int32),
                ("alpha", C.c_float), ("splitStride", C.c_int64)]


class SMProblem(C.Structure):
    _fields_ = [("S", C.c_void_p), ("P", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("ldS", C.c_int32),
                ("ldP", C.c_int32), ("nsplit", C.c_int32), ("rowStart", C.c_int32), ("scale", C.c_float),
                ("flags", C.c_int32), ("splitStride", C.c_int64)]


class VsrOpInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("nitems", C.c_int32), ("tile_cfg", C.c_int32), ("bmode", C.c_int32),
                ("buf_src", C.c_int32), ("buf_dst", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("halo_src", C.c_int32), ("halo_dst", C.c_int32), ("n", C.c_int32), ("ldy", C.c_int32),
                ("pix", C.c_int32), ("t_frame_idx", C.c_int32), ("t_first", C.c_int32), ("premask", C.c_int32),
                ("M", C.c_int32), ("N", C.c_int32), ("nsplit", C.c_int32), ("t_rowC", C.c_int32), ("t_colC", C.c_int32),
                ("buf_mask", C.c_int32), ("off_src", C.c_int64), ("off_dst", C.c_int64), ("split_stride", C.c_int64),
                ("flops", C.c_double), ("tag", C.c_char * 32),
                ("ew", C.c_int32), ("ibuf", C.c_int32 * 4), ("ioff", C.c_int64 * 4), ("ipar", C.c_int32 * 16),
                ("fpar", C.c_float * 4)]


class VsrGemmInfo(C.Structure):
    _fields_ = [("bufA", C.c_int32), ("bufB", C.c_int32), ("bufC", C.c_int32), ("bufR", C.c_int32),
                ("offA", C.c_int64), ("offB", C.c_int64), ("offC", C.c_int64), ("offR", C.c_int64),
                ("offBias", C.c_int64),
                ("tRowA", C.c_int32), ("tColA", C.c_int32), ("tRowB", C.c_int32), ("tColB", C.c_int32),
                ("tRowC", C.c_int32), ("tColC", C.c_int32), ("tRowR", C.c_int32),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("tilesM", C.c_int32), ("tilesN", C.c_int32),
                ("splitK", C.c_int32), ("chunksPerSplit", C.c_int32), ("splitStride", C.c_int64),
                ("alpha", C.c_float), ("act", C.c_int32)]


class VsrSoftmaxInfo(C.Structure):
    _fields_ = [("bufS", C.c_int32), ("bufP", C.c_int32), ("offS", C.c_int64), ("offP", C.c_int64),
                ("splitStride", C.c_int64), ("M", C.c_int32), ("N", C.c_int32), ("ldS", C.c_int32),
                ("ldP", C.c_int32), ("nsplit", C.c_int32), ("scale", C.c_float)]


# every symbol include/vsr_hip.h declares: name -> (restype, argtypes)
_P, _I, _L, _D = C.c_void_p, C.c_int, C.c_int64, C.c_double
SIGNATURES = {
    "vsr_version": (_I, []),
    "vsr_last_error": (C.c_char_p, []),
    "vsr_device_count": (_I, []),
    "vsr_sttn_create": (_I, [_I, C.POINTER(_P)]),
    "vsr_sttn_set_param": (_I, [_P, C.c_char_p, _P, C.POINTER(_L), _I]),
    "vsr_sttn_finalize": (_I, [_P, _I]),
    "vsr_sttn_destroy": (None, [_P]),
    "vsr_sttn_geometry": (_I, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vsr_sttn_set_window": (_I, [_P, _I, _I]),
    "vsr_sttn_packed_weights": (_L, [_P, _P, _L]),
    "vsr_sttn_inpaint": (_I, [_P, _P, _I, _P, _P, _P]),
    "vsr_sttn_auto_chunk": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _I, _P]),
    "vsr_sttn_auto_chunk_rows": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P]),
    "vsr_sttn_auto_chunk_box": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P]),
    "vsr_sttn_det_batch_box": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P]),
    "vsr_sttn_decode_rows": (_I, [_P, _I, _I, _I, _P, _P]),
    "vsr_sttn_decode_cols": (_I, [_P, _I, _I, _I, _P, _P]),
    "vsr_sttn_flops_box": (_D, [_P, _I, _I, _I, _I, _I]),
    "vsr_sttn_flops_rows": (_D, [_P, _I, _I, _I]),
    "vsr_sttn_det_inpaint": (_I, [_P, _P, _P, _I, _P, _P, _P]),
    "vsr_sttn_det_batch": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P]),
    "vsr_sttn_det_batch_rows": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P]),
    "vsr_sttn_set_precision": (_I, [_P, _I]),
    "vsr_sttn_set_lanes": (_I, [_P, _I]),
    "vsr_sttn_fallbacks": (_L, [_P]),
    "vsr_sttn_flops": (_D, [_P, _I]),
    "vsr_sttn_flops_reference": (_D, [_P, _I]),
    "vsr_sttn_timing": (_I, [_P, _I]),
    "vsr_sttn_timing_get": (_I, [_P, C.c_char_p, C.POINTER(_D), C.POINTER(C.c_int32), C.POINTER(_D)]),
    "vsr_sttn_timing_reset": (_I, [_P]),
    "vsr_run_gather_gemm": (_I, [C.POINTER(GGProblem), _I, _I, _I, _P]),
    "vsr_run_gather_gemm_variant": (_I, [C.POINTER(GGProblem), _I, _I, _I, _I, _P]),
    "vsr_run_softmax": (_I, [C.POINTER(SMProblem), _I, _P]),
    "vsr_launch_to_split": (_I, [_P, _P, C.c_int64, _P]),
    "vsr_launch_kn_to_nk_split": (_I, [_P, _P, _P, C.c_int, C.c_int, C.c_int64, _P, _P]),
    "vsr_launch_resize_u8": (_I, [_P, _L, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vsr_launch_norm_im2col": (_I, [_P, _I, _I, _I, _P, _I, _P, _P]),
    "vsr_launch_reduce_scatter": (_I, [_P, _I, _L, _I, _I, _P, _P, _P, _P]),
    "vsr_launch_upsample2x": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "vsr_launch_decode_out": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vsr_launch_upscale_blend": (_I, [_P, _I, _I, _P, _P, _L, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "vsr_cv2_linear_tables": (_I, [_I, _I, _I, _P, _P, _P]),
    "vsr_plan_create": (_I, [_P, _I, C.POINTER(_P)]),
    "vsr_plan_create_rows": (_I, [_P, _I, _I, _I, C.POINTER(_P)]),
    "vsr_plan_create_box": (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "vsr_raft_plan_create": (_I, [_P, _I, _I, _I, _I, C.POINTER(_P)]),
    "vsr_raft_create": (_I, [C.POINTER(_P)]),
    "vsr_raft_set_param": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "vsr_raft_finalize": (_I, [_P, _I]),
    "vsr_raft_destroy": (None, [_P]),
    "vsr_raft_packed_weights": (_L, [_P, _P, _L]),
    "vsr_raft_flows": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "vsr_raft_flops": (_D, [_P, _I, _I, _I, _I]),
    "vsr_raft_read_buffer": (_I, [_P, _I, _L, _L, _P]),
    "vsr_raft_set_precision": (_I, [_P, _I]),
    "vsr_raft_fallbacks": (_L, [_P]),
    "vsr_rfc_plan_create": (_I, [_P, _I, _I, _I, C.POINTER(_P)]),
    "vsr_rfc_create": (_I, [C.POINTER(_P)]),
    "vsr_rfc_set_param": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "vsr_rfc_finalize": (_I, [_P, _I]),
    "vsr_rfc_destroy": (None, [_P]),
    "vsr_rfc_packed_weights": (_L, [_P, _P, _L]),
    "vsr_rfc_complete": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "vsr_rfc_read_buffer": (_I, [_P, _I, _L, _L, _P]),
    "vsr_rfc_set_precision": (_I, [_P, _I]),
    "vsr_rfc_fallbacks": (_L, [_P]),
    "vsr_rfc_flops": (_D, [_P, _I, _I, _I]),
    "vsr_det_launch_conv2d": (_I, [_P, _P, _P] + [_I] * 15 + [_P, _P]),
    "vsr_det_launch_deconv2x2": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "vsr_det_launch_binary": (_I, [_P, _P, _I, _L, _I, _L, _I, _P, _P]),
    "vsr_det_launch_unary": (_I, [_P, _L, _I, C.c_float, C.c_float, _P, _P]),
    "vsr_det_launch_affine": (_I, [_P, _P, _P, _L, _I, _L, _P, _P]),
    "vsr_det_launch_gap": (_I, [_P, _L, _L, _P, _P]),
    "vsr_det_launch_maxpool": (_I, [_P, _L] + [_I] * 10 + [_P, _P]),
    "vsr_det_launch_nearest": (_I, [_P, _L, _I, _I, _I, _P, _P]),
    "vsr_det_launch_normalize": (_I, [_P, _I, _I, _P, _P]),
    "vsr_det_launch_copy": (_I, [_P, _L, _P, _L, _L, _L, _P]),
    "vsr_host_trace_borders": (_I, [_P, _I, _I, _P, _L, _P, _I, _P, _P]),
    "vsr_det_launch_ccl": (_I, [_P, _I, _I, C.c_float, _P, _P, _P, _I, _P, _P]),
    "vsr_det_launch_db_boxes": (_I, [_P, _I, _I, C.c_float, _I, _I, C.c_float, C.c_float, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "vsr_det_launch_nchw_to_nhwc": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "vsr_det_launch_nhwc_to_nchw": (_I, [_P, _I, _I, _L, _I, _P, _P, _I, _P, _P]),
    "vsr_det_launch_to_view": (_I, [_P, _I, _I, _I, _I, _I, _P, _L, _L, _I, _P]),
    "vsr_det_launch_from_view": (_I, [_P, _L, _L, _I, _I, _I, _I, _I, _P, _L, _P]),
    "vsr_det_launch_dwconv_view": (_I, [_P, _L, _L, _I, _P, _P, _P] + [_I] * 11 + [_P, _L, _L, _I, _P]),
    "vsr_det_launch_nearest_view": (_I, [_P, _L, _L, _I, _I, _I, _I, _I, _I, _P, _L, _L, _I, _P]),
    "vsr_det_launch_im2col_view": (_I, [_P] + [_I] * 8 + [_P, _L, _L, _I, _P]),
    "vsr_det_launch_dots_view": (_I, [_P, _L, _L, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P]),
    "vsr_gemm_plan_create": (_I, [C.POINTER(GGProblem), _I, _I, _I, _I, C.POINTER(_P)]),
    "vsr_gemm_plan_run": (_I, [_P, _P]),
    "vsr_gemm_plan_destroy": (None, [_P]),
    "vsr_io_yuv_to_bgr": (_I, [_P, _L, _I, _I, _I, _I, _I, _P, _I, _P]),
    "vsr_io_bgr_to_yuv": (_I, [_P, _I, _I, _I, _I, _P, _L, _I, _P]),
    "vsr_launch_bgr2hsv_u8": (_I, [_P, _P, _L, _P]),
    "vsr_launch_absdiff_sums_u8x3": (_I, [_P, _I, _L, _P, _P]),
    "vsr_pp_create": (_I, [_I, C.POINTER(_P)]),
    "vsr_pp_destroy": (None, [_P]),
    "vsr_pp_img_propagation": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "vsr_pp_imgprop_plan_create": (_I, [_I, _I, _I, C.POINTER(_P)]),
    "vsr_pp_set_param": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "vsr_pp_finalize": (_I, [_P]),
    "vsr_pp_packed_weights": (_L, [_P, _P, _L]),
    "vsr_pp_window_flags": (_I, [_P, _I, _I, _I, _P, _I]),
    "vsr_pp_forward": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "vsr_pp_read_buffer": (_I, [_P, _I, _L, _L, _P]),
    "vsr_pp_prepare_frames": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "vsr_pp_compose_frames": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "vsr_pp_blend_window": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "vsr_pp_set_precision": (_I, [_P, _I]),
    "vsr_pp_fallbacks": (_L, [_P]),
    "vsr_pp_flops": (_D, [_P, _I, _I, _I, _I, _P, _I]),
    "vsr_pp_forward_box": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P]),
    "vsr_pp_encode": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "vsr_pp_forward_cached": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P]),
    "vsr_pp_token_count": (_I, [_I, _I]),
    "vsr_pp_flops_box": (_D, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P]),
    "vsr_pp_gen_plan_create": (_I, [_P, _I, _I, _I, _I, _P, _I, C.POINTER(_P)]),
    "vsr_pp_gen_plan_create_mode": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "vsr_pp_gen_plan_create_box": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "vsr_lama_create": (_I, [C.POINTER(_P)]),
    "vsr_lama_set_param": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    "vsr_lama_finalize": (_I, [_P, _I]),
    "vsr_lama_destroy": (None, [_P]),
    "vsr_lama_blocks": (_I, [_P]),
    "vsr_lama_packed_weights": (_L, [_P, _P, _L]),
    "vsr_lama_inpaint": (_I, [_P, _P, _L, _P, _L, _I, _I, _I, _P, _L, _P]),
    "vsr_lama_set_precision": (_I, [_P, _I]),
    "vsr_lama_fallbacks": (_L, [_P]),
    "vsr_lama_read_buffer": (_I, [_P, _I, _L, _L, _P]),
    "vsr_lama_flops": (_D, [_P, _I, _I, _I]),
    "vsr_lama_plan_create": (_I, [_P, _I, _I, _I, C.POINTER(_P)]),
    "vsr_plan_consts": (_L, [_P, _P, _L]),
    "vsr_plan_destroy": (None, [_P]),
    "vsr_plan_num_buffers": (_I, [_P]),
    "vsr_plan_buffer_elems": (_L, [_P, _I]),
    "vsr_plan_num_tables": (_I, [_P]),
    "vsr_plan_table_len": (_L, [_P, _I]),
    "vsr_plan_table_copy": (_I, [_P, _I, _P]),
    "vsr_plan_num_ops": (_I, [_P]),
    "vsr_plan_op": (_I, [_P, _I, C.POINTER(VsrOpInfo)]),
    "vsr_plan_op_lane": (_I, [_P, _I]),
    "vsr_plan_op_gemm": (_I, [_P, _I, _I, C.POINTER(VsrGemmInfo)]),
    "vsr_plan_op_softmax": (_I, [_P, _I, _I, C.POINTER(VsrSoftmaxInfo)]),
    "vsr_plan_counts": (_I, [_P, _P]),
    "vsr_plan_flops": (_D, [_P]),
    "vsr_switch_state": (_I, [C.c_char_p]),
    "vsr_flow_timing": (_I, [_I]),
    "vsr_flow_timing_reset": (_I, []),
    "vsr_flow_timing_get": (_I, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "vsr_flow_timing_keys": (C.c_int64, [_P, C.c_int64]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP extension first (__graft_entry__.build() or "
        f"`make -C {os.path.join(_HERE, 'csrc')}`); this package has no CPU fallback")

# torch first: the library shares the process's HIP runtime with PyTorch (its streams and device pointers cross this ABI), and
# that has to be the runtime PyTorch ships and loads.  Loaded the other way round (/opt/rocm's libamdhip64 first, through this
# library's NEEDED entry) PyTorch reports no device at all -- seen on the GPU box when a test module imported this package
# before anything had imported torch (profiles/r02_cli_e2e.log).
import torch  # noqa: E402,F401

from . import switches  # noqa: E402

switches.export_defaults()             # the library reads some switches itself, once per process (switches.py)
lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = library does not match the header
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    return (lib.vsr_last_error() or b"").decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise VsrError(rc, last_error())
    return rc
