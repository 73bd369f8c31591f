"""ctypes binding of ``libbiscuit_hip.so`` (C ABI declared in ``include/biscuit_hip.h``).

There is no CPU fallback: if the shared library is missing or does not export the ABI,
importing this module raises.  Build it with ``make -C biscuit_amd/csrc`` (or
``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libbiscuit_hip.so')

BQ_DTYPE_F32, BQ_DTYPE_BF16, BQ_DTYPE_F16 = 0, 1, 2
BQ_MC_HEAD, BQ_MC_FULL = 0, 1
BQ_PROF_MAX = 64


class BqConfig(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('tile_px', C.c_int32), ('n_classes', C.c_int32),
                ('dropout', C.c_float), ('max_batch', C.c_int32), ('max_mc', C.c_int32)]


class BqProfEntry(C.Structure):
    _fields_ = [('name', C.c_char * 48), ('launches', C.c_int64), ('ms', C.c_double),
                ('flops', C.c_double), ('bytes', C.c_double)]


_vp, _i, _i64, _u64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_size_t, C.c_float

# name -> (restype, argtypes): exactly the symbols include/biscuit_hip.h declares
ABI = {
    'bq_create': (_vp, [_i, C.POINTER(BqConfig)]),
    'bq_destroy': (None, [_vp]),
    'bq_last_error': (C.c_char_p, [_vp]),
    'bq_workspace_bytes': (_sz, [_vp, _i, _i]),
    'bq_load_weights': (_i, [_vp, _vp, _sz]),
    'bq_stage': (_i, [_vp, _vp, _i, _vp, _vp]),
    'bq_stage_f32': (_i, [_vp, _vp, _i, _vp, _vp]),
    'bq_png_unfilter': (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    'bq_png_inflate_scratch_bytes': (_sz, [_i]),
    'bq_png_inflate': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp, _sz, _vp, _vp]),
    'bq_png_unfilter_strided': (_i, [_vp, _vp, _sz, _i, _i, _vp, _vp]),
    'bq_stream_create_masked': (_i, [_vp, C.POINTER(C.c_uint32), _i, C.POINTER(_vp)]),
    'bq_stream_destroy': (_i, [_vp, _vp]),
    'bq_set_num_cus': (_i, [_vp, _i]),
    'bq_set_option': (_i, [_vp, C.c_char_p, _i]),
    'bq_stain_reinhard_fast': (_i, [_vp, _vp, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _vp]),
    'bq_stain_lab_stats': (_i, [_vp, _vp, _i, _vp, _vp]),
    'bq_backbone': (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp]),
    'bq_backbone_u8': (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp]),
    'bq_mc_head': (_i, [_vp, _vp, _i, _i64, _i, _i, _u64, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'bq_set_tile_index_ptr': (_i, [_vp, _vp]),
    'bq_set_tile_index_array': (_i, [_vp, _vp]),
    'bq_mc_infer': (_i, [_vp, _vp, _i, _i64, _i, _u64, _i, _vp, _vp, _vp, _sz, _vp]),
    'bq_slide_reduce': (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    'bq_roc_workspace_bytes': (_sz, [_i64]),
    'bq_roc_youden': (_i, [_vp, _vp, _vp, _i64, _vp, _sz, _vp, _vp]),
    'bq_slide_finish': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    'bq_profile_enable': (_i, [_vp, _i]),
    'bq_profile_read': (_i, [_vp, C.POINTER(BqProfEntry), _i]),
    'bq_debug_activation': (_i64, [_vp, C.c_char_p, _vp, _i, _vp, _sz, _vp, _sz, _vp]),
    'bq_debug_activation_u8': (_i64, [_vp, C.c_char_p, _vp, _i, _vp, _sz, _vp, _sz, _vp]),
}


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f'{path} not found: the HIP extension is mandatory (no CPU fallback). '
            'Build it with `make -C biscuit_amd/csrc`.')
    lib = C.CDLL(path)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()
