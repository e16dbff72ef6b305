"""ctypes binding of libvitsmi.so (include/vitsmi.h).  Fails loudly if the library is missing
or cannot be built: there is no CPU fallback on the product path."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VITS_MAX_DIMS = 4


class VitsNoise(C.Structure):
    _fields_ = [("noise_dp", C.c_void_p), ("noise_z", C.c_void_p), ("noise_z_stride", C.c_int64),
                ("seed", C.c_uint64)]


class VitsOutput(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("dims", C.c_int64 * 4), ("y_lengths", C.POINTER(C.c_int64))]


class VitsStats(C.Structure):
    _fields_ = [("conv_flops", C.c_double), ("conv_bytes", C.c_double), ("dec_flops", C.c_double),
                ("dec_bytes", C.c_double), ("flow_flops", C.c_double), ("enc_flops", C.c_double),
                ("dp_flops", C.c_double), ("conv_ms", C.c_float), ("dec_ms", C.c_float), ("flow_ms", C.c_float),
                ("enc_ms", C.c_float), ("dp_ms", C.c_float), ("total_ms", C.c_float), ("conv_launches", C.c_int),
                ("total_launches", C.c_int), ("sx_flops", C.c_double), ("sx_ms", C.c_float), ("sx_launches", C.c_int),
                ("f16_peak_max", C.c_float), ("f16_peak_min", C.c_float), ("f16_tracked", C.c_int),
                ("f16_saturated", C.c_int), ("sx_bytes", C.c_double)]


class VitsLaunchRecord(C.Structure):
    _fields_ = [("kernel", C.c_char * 128), ("flops", C.c_double), ("bytes", C.c_double), ("ms", C.c_float),
                ("stage", C.c_int), ("cin", C.c_int), ("cout", C.c_int), ("k", C.c_int), ("dil", C.c_int), ("t", C.c_int)]


class VitsOpenOptions(C.Structure):
    _fields_ = [("device_id", C.c_int), ("gen_precision", C.c_char_p), ("arena_dev", C.c_void_p),
                ("arena_bytes", C.c_size_t), ("host_only", C.c_int), ("layout_only", C.c_int)]


# int fn(void *user, const float *samples, int B, int64 first_sample, int64 n_samples, int64 total_samples)
CHUNK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int64, C.c_int64, C.c_int64)

VITS_E_RANGE = -6


EXPORTS = [
    "vits_open", "vits_open_with_arena", "vits_open_host", "vits_open_layout", "vits_open_opts", "vits_close",
    "vits_run_chunked", "vits_run_vocoder_chunked", "vits_last_error", "vits_num_inputs",
    "vits_input_name", "vits_meta", "vits_hparam", "vits_arena_bytes", "vits_arena_host", "vits_arena_device",
    "vits_run", "vits_free_output", "vits_run_device", "vits_sync", "vits_last_y_lengths", "vits_last_pcm16", "vits_run_vocoder",
    "vits_tap",
    "vits_set_timing", "vits_set_tails", "vits_reserve", "vits_get_stats", "vits_stream", "vits_test_conv1d", "vits_test_conv_transpose1d",
    "vits_test_attention", "vits_test_attention16", "vits_bench_conv1d", "vits_test_conv1d_sx", "vits_test_conv1d_sx_planar", "vits_test_conv1d_sx_gate", "vits_test_set_sx_small_max",
    "vits_test_conv_transpose1d_sx",
    "vits_bench_conv1d_sx", "vits_test_conv_pair_sx", "vits_fetch_output", "vits_run_async", "vits_host_alloc", "vits_host_free",
    "vits_launch_records",
]


G2P_EXPORTS = ["g2p_open", "g2p_close", "g2p_last_error", "g2p_hparam", "g2p_num_outputs", "g2p_output_name", "g2p_bucket",
               "g2p_run", "g2p_generate", "g2p_generate_batch", "g2p_test_forced_steps"]


def lib_path():
    # VITSMI_LIB: another build of the library (kernel experiments: variants compiled with -D switches side by side)
    return os.environ.get("VITSMI_LIB") or os.path.join(_HERE, "libvitsmi.so")


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.environ.get("VITSMI_LIB"):
        # the in-tree library must be the build of the in-tree sources: compared by content hash (_build_info.json, written
        # by phoonnx_amd/build.py), rebuilt when it is not - or, without a compiler, refused: never a stale binary
        from . import build as _build
        if _build.stale():
            try:
                _build.build()
            except Exception as e:
                raise RuntimeError(f"{path} is missing or was not built from the sources in this tree (source_sha "
                                   f"{_build.source_sha()} vs recorded {_build.read_build_info().get('source_sha')}) "
                                   f"and cannot be rebuilt here: {e}") from e
    lib = C.CDLL(path)
    vp, i64p, f32p = C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_float)
    lib.vits_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.vits_open_with_arena.argtypes = [C.c_char_p, C.c_int, vp, C.c_size_t, C.POINTER(vp)]
    lib.vits_open_host.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.vits_open_layout.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.vits_open_opts.argtypes = [C.c_char_p, C.POINTER(VitsOpenOptions), C.POINTER(vp)]
    lib.vits_run_chunked.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, C.POINTER(VitsNoise), C.c_int, CHUNK_FN, vp]
    lib.vits_run_vocoder_chunked.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, CHUNK_FN, vp]
    lib.vits_close.argtypes = [vp]
    lib.vits_close.restype = None
    lib.vits_last_error.argtypes = [vp]
    lib.vits_last_error.restype = C.c_char_p
    lib.vits_num_inputs.argtypes = [vp]
    lib.vits_input_name.argtypes = [vp, C.c_int]
    lib.vits_input_name.restype = C.c_char_p
    lib.vits_meta.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
    lib.vits_hparam.argtypes = [vp, C.c_char_p, i64p]
    lib.vits_arena_bytes.argtypes = [vp]
    lib.vits_arena_bytes.restype = C.c_size_t
    lib.vits_arena_host.argtypes = [vp]
    lib.vits_arena_host.restype = vp
    lib.vits_arena_device.argtypes = [vp]
    lib.vits_arena_device.restype = vp
    run_args = [vp, vp, vp, C.c_int, C.c_int, vp, vp, C.POINTER(VitsNoise), C.POINTER(VitsOutput)]
    lib.vits_run.argtypes = run_args
    lib.vits_run_device.argtypes = run_args
    lib.vits_run_async.argtypes = run_args[:-1]
    lib.vits_free_output.argtypes = [vp, C.POINTER(VitsOutput)]
    lib.vits_free_output.restype = None
    lib.vits_sync.argtypes = [vp]
    lib.vits_last_y_lengths.argtypes = [vp, i64p, C.c_int]
    lib.vits_last_pcm16.argtypes = [vp, C.c_int, C.c_float, vp, C.c_size_t]
    lib.vits_run_vocoder.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.POINTER(VitsOutput)]
    lib.vits_tap.argtypes = [vp, C.c_char_p, vp, C.c_size_t, i64p]
    lib.vits_set_timing.argtypes = [vp, C.c_int]
    lib.vits_set_tails.argtypes = [vp, C.c_int]
    lib.vits_reserve.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.vits_get_stats.argtypes = [vp, C.POINTER(VitsStats)]
    lib.vits_stream.argtypes = [vp]
    lib.vits_stream.restype = vp
    lib.vits_fetch_output.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
    lib.vits_host_alloc.argtypes = [C.c_size_t]
    lib.vits_host_alloc.restype = vp
    lib.vits_host_free.argtypes = [vp]
    lib.vits_host_free.restype = None
    lib.vits_launch_records.argtypes = [vp, C.POINTER(VitsLaunchRecord), C.c_int]
    lib.vits_test_conv1d.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_float, vp]
    lib.vits_test_conv_transpose1d.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int,
                                               C.c_int, vp]
    lib.vits_test_attention.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp]
    lib.vits_test_attention16.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp,
                                          C.c_int, C.c_int, f32p]
    lib.vits_bench_conv1d.argtypes = [C.c_int] * 11 + [f32p]
    lib.vits_test_conv1d_sx.argtypes = lib.vits_test_conv1d.argtypes
    lib.vits_test_conv_transpose1d_sx.argtypes = lib.vits_test_conv_transpose1d.argtypes
    lib.vits_test_conv1d_sx_planar.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                               vp, vp, C.c_int, C.c_int, vp, vp]
    lib.vits_test_set_sx_small_max.argtypes = [C.c_longlong]
    lib.vits_test_set_sx_small_max.restype = C.c_longlong
    lib.vits_test_conv1d_sx_gate.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.vits_bench_conv1d_sx.argtypes = [C.c_int] * 9 + [f32p]
    lib.vits_test_conv_pair_sx.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_float, vp, f32p]
    lib.g2p_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.g2p_close.argtypes = [vp]
    lib.g2p_close.restype = None
    lib.g2p_last_error.argtypes = [vp]
    lib.g2p_last_error.restype = C.c_char_p
    lib.g2p_hparam.argtypes = [vp, C.c_char_p, i64p]
    lib.g2p_num_outputs.argtypes = [vp]
    lib.g2p_output_name.argtypes = [vp, C.c_int]
    lib.g2p_output_name.restype = C.c_char_p
    lib.g2p_bucket.argtypes = [vp, C.c_int, C.c_int]
    lib.g2p_run.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp]
    lib.g2p_generate.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int64, vp, C.POINTER(C.c_int)]
    lib.g2p_generate_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int64, vp, vp]
    lib.g2p_test_forced_steps.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp]
    _LIB = lib
    return lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def last_error(h=None):
    return load().vits_last_error(h).decode("utf-8", "replace")
