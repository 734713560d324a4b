"""ctypes binding of ``libhip_dsp.so`` (the C ABI declared in ``include/hip_dsp.h``).

The product path has no CPU fallback: if the HIP library has not been built this
module raises at import time, loudly.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AUDIAN_AMD_LIB') or os.path.join(_HERE, 'libhip_dsp.so')     # (AUDIAN_AMD_LIB: another build, tools/ab_two_builds.sh)

OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_TOO_SHORT, ERR_NOMEM = range(6)
MAX_SECTIONS = 4


class HipDspError(RuntimeError):
    """A libhip_dsp call failed (status code in ``.status``)."""

    def __init__(self, status, message):
        super().__init__(f'libhip_dsp error {status}: {message}')
        self.status = status


if not os.path.exists(LIB_PATH) and os.path.exists('/opt/rocm/bin/hipcc') and \
   os.environ.get('AUDIAN_AMD_NO_AUTOBUILD') != '1':
    # a source checkout without the built extension: build it (hipcc, gfx950) rather than fail;
    # there is still no CPU path -- without hipcc or on a build error the import raises below
    import subprocess as _sp
    try:
        _sp.check_call(['make', '-C', os.path.join(_HERE, 'csrc'), '-j4', '-s'])
    except Exception:
        pass

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} is missing: build the HIP extension first '
        '(`make -C audian_amd/csrc` or `python -c "import __graft_entry__ as g; g.build()"`). '
        'audian_amd has no CPU fallback.')

# PyTorch wheels bundle their own libamdhip64/libhsa-runtime64 (same SONAME as ROCm's).
# Two HIP runtimes in one process cannot both own the GPU, so when torch is going to be
# used in this process (multi-GPU runs: torch.distributed/RCCL) it has to be loaded
# FIRST; libhip_dsp.so then binds to the runtime that is already there.
import sys as _sys
if 'torch' not in _sys.modules and (int(os.environ.get('WORLD_SIZE', '1')) > 1 or
                                    os.environ.get('AUDIAN_AMD_TORCH') == '1'):
    import torch as _torch  # noqa: F401

lib = ctypes.CDLL(LIB_PATH)


def hip_runtime_path():
    """Which libamdhip64 this process resolved (diagnostics for the note above)."""
    with open('/proc/self/maps') as f:
        return sorted({ln.split()[-1] for ln in f if 'libamdhip64' in ln})

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double
_sz = ctypes.c_size_t
_pp = ctypes.POINTER(ctypes.c_void_p)

_SIGNATURES = {
    'hipdsp_version': ([], _int),
    'hipdsp_last_error': ([], ctypes.c_char_p),
    'hipdsp_device_count': ([ctypes.POINTER(_int)], _int),
    'hipdsp_ctx_create': ([_int, _vp, _pp], _int),
    'hipdsp_ctx_destroy': ([_vp], _int),
    'hipdsp_ctx_set_stream': ([_vp, _vp], _int),
    'hipdsp_ctx_synchronize': ([_vp], _int),
    'hipdsp_ctx_set_max_segments': ([_vp, _int], _int),
    'hipdsp_ctx_set_option': ([_vp, ctypes.c_char_p, ctypes.c_longlong], _int),
    'hipdsp_ctx_reserve': ([_vp, _sz], _int),
    'hipdsp_ctx_set_mid_event': ([_vp, _vp], _int),
    'hipdsp_stream_create': ([_vp, _pp], _int),
    'hipdsp_stream_destroy': ([_vp, _vp], _int),
    'hipdsp_graph_begin': ([_vp], _int),
    'hipdsp_graph_end': ([_vp, _pp], _int),
    'hipdsp_graph_launch': ([_vp, _vp], _int),
    'hipdsp_graph_destroy': ([_vp, _vp], _int),
    'hipdsp_malloc': ([_vp, _sz, _pp], _int),
    'hipdsp_malloc_probed': ([_vp, _sz, _int, _pp], _int),
    'hipdsp_free': ([_vp, _vp], _int),
    'hipdsp_pool_stats': ([_vp, ctypes.POINTER(_sz), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)], _int),
    'hipdsp_pool_trim': ([_vp], _int),
    'hipdsp_memset': ([_vp, _vp, _int, _sz], _int),
    'hipdsp_memcpy_h2d': ([_vp, _vp, _vp, _sz], _int),
    'hipdsp_memcpy_d2h': ([_vp, _vp, _vp, _sz], _int),
    'hipdsp_memcpy_d2d': ([_vp, _vp, _vp, _sz], _int),
    'hipdsp_memcpy2d_d2d': ([_vp, _vp, _sz, _vp, _sz, _sz, _sz], _int),
    'hipdsp_event_create': ([_vp, _pp], _int),
    'hipdsp_event_destroy': ([_vp, _vp], _int),
    'hipdsp_event_record': ([_vp, _vp], _int),
    'hipdsp_event_wait': ([_vp, _vp], _int),
    'hipdsp_event_elapsed_ms': ([_vp, _vp, _vp, ctypes.POINTER(ctypes.c_float)], _int),
    'hipdsp_pack_f64': ([_vp, _vp, _vp, _i64, _i64, _i64], _int),
    'hipdsp_pack_f32': ([_vp, _vp, _vp, _i64, _i64, _i64], _int),
    'hipdsp_unpack_f64': ([_vp, _vp, _i64, _vp, _i64, _i64], _int),
    'hipdsp_unpack_spectrum_f64': ([_vp, _vp, _i64, _vp, _i64, _i64, _i64], _int),
    'hipdsp_sosplan_create': ([_vp, _pp], _int),
    'hipdsp_sosplan_destroy': ([_vp, _vp], _int),
    'hipdsp_sosplan_set': ([_vp, _vp, _vp, _int], _int),
    'hipdsp_sosplan_set_host': ([_vp, _vp, _vp, _int], _int),
    'hipdsp_sosplan_upload': ([_vp, _vp], _int),
    'hipdsp_sos_plan_host': ([_vp, _int, ctypes.POINTER(_i64), ctypes.POINTER(_int), _vp], _int),
    'hipdsp_sos_segments_host': ([_i64, _int, _int, _int, _i64, _i64, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_int)], _int),
    'hipdsp_sosplan_info': ([_vp, _vp, ctypes.POINTER(_i64), ctypes.POINTER(_int)], _int),
    'hipdsp_sosfilt': ([_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64], _int),
    'hipdsp_envelope': ([_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _int, _dbl, _int], _int),
    'hipdsp_envelope_multi': ([_vp, _vp, _int, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _int, _dbl, _int], _int),
    'hipdsp_sosfilt_envelope': ([_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _int, _dbl, _int, _int, _i64], _int),
    'hipdsp_chain_forward': ([_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _int, _dbl, _int, _int, _dbl, _vp, _vp, _i64, _i64, _i64,
                              _i64, _i64], _int),
    'hipdsp_chain_backward': ([_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _int, _dbl, _int, _int, _int, _dbl, _vp, _i64, _i64], _int),
    'hipdsp_chain_backward_plan': ([_vp, _vp, _i64, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_int)], _int),
    'hipdsp_chain_plan': ([_vp, _vp, _vp, _i64, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_int)], _int),
    'hipdsp_spectrogram': ([_vp, _vp, _i64, _i64, _i64, _int, _int, _dbl, _vp, _vp, _i64, _i64], _int),
    'hipdsp_decibel': ([_vp, _vp, _vp, _i64, _dbl, _dbl], _int),
    'hipdsp_decibel_image': ([_vp, _vp, _vp, _i64, _i64, _dbl, _dbl], _int),
    'hipdsp_decibel_image_decimate': ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl], _int),
    'hipdsp_channel_mean': ([_vp, _vp, _i64, ctypes.POINTER(_int), _int, _i64, _i64, _dbl, _vp], _int),
    'hipdsp_stride_copy': ([_vp, _vp, _i64, _i64, _vp], _int),
    'hipdsp_max_nonneg': ([_vp, _vp, _i64, _vp], _int),
    'hipdsp_band_order_stats': ([_vp, _vp, _i64, _i64, _i64, _i64, _vp], _int),
    'hipdsp_unwrap': ([_vp, _vp, _i64, _i64, _i64, _dbl, _dbl, _int, _int, _vp, _i64], _int),
    'hipdsp_pcm_unpack': ([_vp, _vp, _int, _i64, _i64, _dbl, _vp, _i64], _int),
    'hipdsp_minmax_decimate': ([_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _i64], _int),
    'hipdsp_mean_spectrum_db': ([_vp, _vp, _i64, _i64, _i64, _dbl, _dbl, _dbl, _vp], _int),
    'hipdsp_comm_unique_id': ([_vp], _int),
    'hipdsp_comm_create': ([_vp, _vp, _int, _int, _pp], _int),
    'hipdsp_comm_destroy': ([_vp, _vp], _int),
    'hipdsp_allgather_f32': ([_vp, _vp, _vp, _vp, _i64], _int),
    'hipdsp_copy_probe': ([_vp, _vp, _vp, _sz], _int),
    'hipdsp_synth': ([_vp, _vp, _i64, _i64, _i64, _dbl, ctypes.c_uint64, _i64, _i64], _int),
}

for _name, (_args, _res) in _SIGNATURES.items():
    _fn = getattr(lib, _name)      # AttributeError here = header and library disagree
    _fn.argtypes = _args
    _fn.restype = _res


def last_error():
    return lib.hipdsp_last_error().decode('utf-8', 'replace')


def check(status):
    """Raise the Python exception matching a libhip_dsp status code."""
    if status == OK:
        return
    msg = last_error()
    if status == ERR_TOO_SHORT:
        raise ValueError(msg)                 # scipy.signal.sosfiltfilt raises ValueError
    if status == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if status == ERR_NOMEM:
        raise MemoryError(msg)
    if status == ERR_INVALID:
        raise ValueError(msg)
    raise HipDspError(status, msg)
