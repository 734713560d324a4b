"""Thin Python host layer over the libhip_dsp C ABI: context, device arrays, filter
plans and the hot-path calls.  Device-native layout is planar float32
(channels, frames); see ``include/hip_dsp.h``.
"""

import ctypes

import numpy as np

from . import _lib
from ._lib import check, lib


class Context:
    """A device + HIP stream on which libhip_dsp enqueues its work."""

    def __init__(self, device=0, stream=None):
        h = ctypes.c_void_p()
        check(lib.hipdsp_ctx_create(int(device), ctypes.c_void_p(stream or 0), ctypes.byref(h)))
        self._h = h
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream):
        check(lib.hipdsp_ctx_set_stream(self._h, ctypes.c_void_p(stream or 0)))

    def synchronize(self):
        check(lib.hipdsp_ctx_synchronize(self._h))

    def set_max_segments(self, n):
        check(lib.hipdsp_ctx_set_max_segments(self._h, int(n)))

    def set_option(self, name, value):
        check(lib.hipdsp_ctx_set_option(self._h, name.encode(), int(value)))

    def set_mid_event(self, ev):
        check(lib.hipdsp_ctx_set_mid_event(self._h, ev if ev is not None else ctypes.c_void_p(0)))

    def pool_stats(self):
        """(cached bytes, hits, misses) of the context's block cache behind hipdsp_malloc/free."""
        cached, hits, misses = ctypes.c_size_t(), ctypes.c_uint64(), ctypes.c_uint64()
        check(lib.hipdsp_pool_stats(self._h, ctypes.byref(cached), ctypes.byref(hits), ctypes.byref(misses)))
        return int(cached.value), int(hits.value), int(misses.value)

    def pool_trim(self):
        check(lib.hipdsp_pool_trim(self._h))

    def reserve(self, nbytes):
        check(lib.hipdsp_ctx_reserve(self._h, int(nbytes)))

    # streams and graphs ---------------------------------------------------
    def create_stream(self):
        st = ctypes.c_void_p()
        check(lib.hipdsp_stream_create(self._h, ctypes.byref(st)))
        return st.value

    def destroy_stream(self, stream):
        check(lib.hipdsp_stream_destroy(self._h, ctypes.c_void_p(stream)))

    def graph_begin(self):
        check(lib.hipdsp_graph_begin(self._h))

    def graph_end(self):
        g = ctypes.c_void_p()
        check(lib.hipdsp_graph_end(self._h, ctypes.byref(g)))
        return g

    def graph_launch(self, graph):
        check(lib.hipdsp_graph_launch(self._h, graph))

    def graph_destroy(self, graph):
        check(lib.hipdsp_graph_destroy(self._h, graph))

    # events -------------------------------------------------------------
    def event(self):
        ev = ctypes.c_void_p()
        check(lib.hipdsp_event_create(self._h, ctypes.byref(ev)))
        return ev

    def record(self, ev):
        check(lib.hipdsp_event_record(self._h, ev))

    def wait_event(self, ev):
        """Later work on this context's stream waits for `ev` (recorded on any stream)."""
        check(lib.hipdsp_event_wait(self._h, ev))

    def elapsed_ms(self, start, stop):
        ms = ctypes.c_float()
        check(lib.hipdsp_event_elapsed_ms(self._h, start, stop, ctypes.byref(ms)))
        return float(ms.value)

    def destroy_event(self, ev):
        check(lib.hipdsp_event_destroy(self._h, ev))

    def close(self):
        if self._h is not None and self._h.value:
            lib.hipdsp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    """The process-wide context on device 0 / LOCAL_RANK (created on first use)."""
    global _default_ctx
    if _default_ctx is None:
        import os
        _default_ctx = Context(int(os.environ.get('LOCAL_RANK', '0')))
    return _default_ctx


class DeviceArray:
    """A caller-owned block of HBM with a NumPy-like shape/dtype (C-contiguous)."""

    def __init__(self, ctx, shape, dtype=np.float32, ptr=None, owner=None, write_probe=0):
        """write_probe = N > 1: the block kernels will write a trace into -- the best of N allocations by the time of
        a memset over each (hipdsp_malloc_probed: where a block lies in HBM moves a write stream by up to 12 %)."""
        self.ctx = ctx
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64))*self.dtype.itemsize
        self._own = ptr is None
        self._owner = owner
        if ptr is None:
            p = ctypes.c_void_p()
            if write_probe > 1:
                check(lib.hipdsp_malloc_probed(ctx.handle, self.nbytes, int(write_probe), ctypes.byref(p)))
            else:
                check(lib.hipdsp_malloc(ctx.handle, self.nbytes, ctypes.byref(p)))
            self.ptr = p.value or 0
        else:
            self.ptr = int(ptr)

    @classmethod
    def from_host(cls, ctx, array, dtype=None):
        a = np.ascontiguousarray(array, dtype=dtype)
        d = cls(ctx, a.shape, a.dtype)
        d.copy_from_host(a)
        return d

    def copy_from_host(self, array):
        a = np.ascontiguousarray(array, dtype=self.dtype)
        if a.nbytes != self.nbytes:
            raise ValueError('size mismatch in copy_from_host')
        check(lib.hipdsp_memcpy_h2d(self.ctx.handle, ctypes.c_void_p(self.ptr),
                                    ctypes.c_void_p(a.ctypes.data), self.nbytes))

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(lib.hipdsp_memcpy_d2h(self.ctx.handle, ctypes.c_void_p(out.ctypes.data),
                                    ctypes.c_void_p(self.ptr), self.nbytes))
        return out

    def zero_(self):
        check(lib.hipdsp_memset(self.ctx.handle, ctypes.c_void_p(self.ptr), 0, self.nbytes))
        return self

    def view(self, offset_elems, shape):
        """A non-owning sub-block starting `offset_elems` elements into this array."""
        v = DeviceArray(self.ctx, shape, self.dtype,
                        ptr=self.ptr + int(offset_elems)*self.dtype.itemsize, owner=self)
        if v.ptr + v.nbytes > self.ptr + self.nbytes:
            raise ValueError('view exceeds the parent array')
        return v

    def free(self):
        if self._own and self.ptr:
            lib.hipdsp_free(self.ctx.handle, ctypes.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _p(x):
    """Device pointer of a DeviceArray, a torch tensor, an int, or None."""
    if x is None:
        return ctypes.c_void_p(0)
    if isinstance(x, DeviceArray):
        return ctypes.c_void_p(x.ptr)
    if hasattr(x, 'data_ptr'):
        return ctypes.c_void_p(x.data_ptr())
    return ctypes.c_void_p(int(x))


class SosPlan:
    """Device-resident plan for one SOS table (see hipdsp_sosplan_* in hip_dsp.h)."""

    def __init__(self, ctx, sos=None):
        self.ctx = ctx
        h = ctypes.c_void_p()
        check(lib.hipdsp_sosplan_create(ctx.handle, ctypes.byref(h)))
        self._h = h
        self.n_sections = 0
        if sos is not None:
            self.set(sos)

    @staticmethod
    def _table(sos):
        sos = np.ascontiguousarray(sos, dtype=np.float64)
        if sos.ndim != 2 or sos.shape[1] != 6:
            raise ValueError('sos must be shape (n_sections, 6)')
        return sos

    def set(self, sos):
        sos = self._table(sos)
        check(lib.hipdsp_sosplan_set(self.ctx.handle, self._h, ctypes.c_void_p(sos.ctypes.data),
                                     len(sos)))
        self.n_sections = len(sos)

    def set_host(self, sos):
        sos = self._table(sos)
        check(lib.hipdsp_sosplan_set_host(self.ctx.handle, self._h,
                                          ctypes.c_void_p(sos.ctypes.data), len(sos)))
        self.n_sections = len(sos)

    def upload(self):
        check(lib.hipdsp_sosplan_upload(self.ctx.handle, self._h))

    def info(self):
        w = ctypes.c_int64()
        e = ctypes.c_int()
        check(lib.hipdsp_sosplan_info(self.ctx.handle, self._h, ctypes.byref(w), ctypes.byref(e)))
        return int(w.value), int(e.value)

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h is not None and self._h.value:
            lib.hipdsp_sosplan_destroy(self.ctx.handle, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator behind the C ABI (hipdsp_comm_*): rank 0 calls
    `Comm.unique_id()` and ships the 128 bytes to the other ranks."""

    def __init__(self, ctx, unique_id, rank, nranks):
        self.ctx = ctx
        self.rank, self.nranks = int(rank), int(nranks)
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        h = ctypes.c_void_p()
        check(lib.hipdsp_comm_create(ctx.handle, buf, self.rank, self.nranks, ctypes.byref(h)))
        self._h = h

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        check(lib.hipdsp_comm_unique_id(buf))
        return buf.raw

    def allgather(self, send, recv, count_per_rank):
        check(lib.hipdsp_allgather_f32(self.ctx.handle, self._h, _p(send), _p(recv),
                                       int(count_per_rank)))

    def close(self):
        if self._h is not None and self._h.value:
            lib.hipdsp_comm_destroy(self.ctx.handle, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _plan(plan):
    return plan.handle if plan is not None else ctypes.c_void_p(0)


# How often each hot-path entry point has been called in this process: tests (and integrators) read these to see
# WHICH launches a BufferedData.recompute_all() turned into (e.g. that the fused forward sweep ran).
launches = {}


def _count(name):
    launches[name] = launches.get(name, 0) + 1


def sosfilt(ctx, plan, x, x_pitch, y, y_pitch, channels, frames, skip=0):
    _count('sosfilt')
    check(lib.hipdsp_sosfilt(ctx.handle, _plan(plan), _p(x), int(x_pitch), _p(y), int(y_pitch),
                             int(channels), int(frames), int(skip)))


def envelope(ctx, plan, x, x_pitch, y, y_pitch, channels, frames, skip=0, rectify=True,
             gain=np.pi/2, clamp=True):
    _count('envelope')
    check(lib.hipdsp_envelope(ctx.handle, _plan(plan), _p(x), int(x_pitch), _p(y), int(y_pitch),
                              int(channels), int(frames), int(skip), int(bool(rectify)),
                              float(gain), int(bool(clamp))))


def envelope_multi(ctx, plans, x, x_pitch, y, y_pitch, channels, frames, skip=0, rectify=True,
                   gain=np.pi/2, clamp=True):
    """sosfiltfilt envelope over a cascade split into several plans (hipdsp_envelope_multi)."""
    arr = (ctypes.c_void_p*len(plans))(*[p.handle for p in plans])
    _count('envelope_multi')
    check(lib.hipdsp_envelope_multi(ctx.handle, arr, len(plans), _p(x), int(x_pitch), _p(y), int(y_pitch),
                                    int(channels), int(frames), int(skip), int(bool(rectify)), float(gain),
                                    int(bool(clamp))))


def sosfilt_envelope(ctx, fplan, eplan, x, x_pitch, yf, yf_pitch, env, env_pitch, channels, frames,
                     rectify=True, gain=np.pi/2, clamp=True, phase=0, env_first=0):
    """yf = sosfilt(fplan, x); env = sosfiltfilt(eplan, gain*|yf[env_first:]|) (hipdsp_sosfilt_envelope): env rows hold
    frames - env_first samples.  phase 0 = both sweeps, 1 = forward only, 2 = backward only (after phase 1 or
    chain_forward with the same env_first)."""
    _count('sosfilt_envelope:%d' % phase)
    check(lib.hipdsp_sosfilt_envelope(ctx.handle, fplan.handle, eplan.handle, _p(x), int(x_pitch),
                                      _p(yf), int(yf_pitch), _p(env), int(env_pitch), int(channels),
                                      int(frames), int(bool(rectify)), float(gain), int(bool(clamp)),
                                      int(phase), int(env_first)))


def chain_forward(ctx, fplan, eplan, x, x_pitch, yf, yf_pitch, channels, frames, nfft, hop, fs, psd,
                  frames_out, psd_pitch=0, rectify=True, gain=np.pi/2, db_out=None, spec_frames=0, spec_first=0,
                  env_first=0):
    """Band-pass + envelope state sweep + spectrogram of the filtered trace in one pass over x
    (nfft/hop 2048/1024, 2048/512, 1024/512, 1024/256, 512/256, 256/128; NotImplementedError otherwise).  The envelope follows with
    sosfilt_envelope(..., phase=2).  eplan None: no envelope (filter + spectrogram only); spec_frames: the
    spectrogram is handed only that many samples of the filtered trace (0 = all); spec_first: frame 0 of the
    spectrogram starts at that sample of the filtered trace (spec_frames counts from there); env_first: the envelope
    is that of yf[env_first:] (pass the same env_first to sosfilt_envelope(phase=2))."""
    _count('chain_forward')
    check(lib.hipdsp_chain_forward(ctx.handle, fplan.handle, _plan(eplan), _p(x), int(x_pitch), _p(yf),
                                   int(yf_pitch), int(channels), int(frames), int(bool(rectify)),
                                   float(gain), int(nfft), int(hop), float(fs), _p(psd), _p(db_out),
                                   int(frames_out), int(psd_pitch), int(spec_frames), int(spec_first),
                                   int(env_first)))


def chain_backward(ctx, eplan, yf, yf_pitch, env, env_pitch, channels, frames, nfft, hop, fs, psd, frames_out,
                   psd_pitch=0, rectify=True, gain=np.pi/2, clamp=True):
    """Envelope backward sweep + the odd spectrogram frames (after chain_forward with the context option
    "chain_split_frames"; hipdsp_chain_backward)."""
    check(lib.hipdsp_chain_backward(ctx.handle, eplan.handle, _p(yf), int(yf_pitch), _p(env), int(env_pitch),
                                    int(channels), int(frames), int(bool(rectify)), float(gain), int(bool(clamp)),
                                    int(nfft), int(hop), float(fs), _p(psd), int(frames_out), int(psd_pitch)))


def chain_backward_plan(ctx, eplan, channels, frames):
    """(first_border, segment_frames, n_segments) of chain_backward: its internal borders are
    first_border - s*segment_frames, s = 0 ... n_segments - 2 (hipdsp_chain_backward_plan)."""
    fb, seg, n = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
    check(lib.hipdsp_chain_backward_plan(ctx.handle, eplan.handle, int(channels), int(frames), ctypes.byref(fb),
                                         ctypes.byref(seg), ctypes.byref(n)))
    return int(fb.value), int(seg.value), int(n.value)


def chain_plan(ctx, fplan, eplan, channels, frames):
    """(segment_frames, n_segments) of chain_forward for this shape (hipdsp_chain_plan)."""
    seg, n = ctypes.c_int64(), ctypes.c_int()
    check(lib.hipdsp_chain_plan(ctx.handle, fplan.handle, _plan(eplan), int(channels), int(frames),
                                ctypes.byref(seg), ctypes.byref(n)))
    return int(seg.value), int(n.value)


def spectrogram(ctx, x, x_pitch, channels, frames, nfft, hop, fs, out, frames_out, db_out=None,
                out_pitch=0):
    _count('spectrogram')
    check(lib.hipdsp_spectrogram(ctx.handle, _p(x), int(x_pitch), int(channels), int(frames),
                                 int(nfft), int(hop), float(fs), _p(out), _p(db_out),
                                 int(frames_out), int(out_pitch)))


def decibel(ctx, p, out, n, ref_power=1.0, min_power=1e-20):
    check(lib.hipdsp_decibel(ctx.handle, _p(p), _p(out), int(n), float(ref_power),
                             float(min_power)))


def decibel_image(ctx, spec_tf, image_ft, frames, nfreq, ref_power=1.0, min_power=1e-20):
    check(lib.hipdsp_decibel_image(ctx.handle, _p(spec_tf), _p(image_ft), int(frames), int(nfreq),
                                   float(ref_power), float(min_power)))


def decibel_image_decimate(ctx, spec_tf, image_fc, frames, nfreq, start, stop, step, ref_power=1.0,
                           min_power=1e-20):
    check(lib.hipdsp_decibel_image_decimate(ctx.handle, _p(spec_tf), _p(image_fc), int(frames), int(nfreq),
                                            int(start), int(stop), int(step), float(ref_power),
                                            float(min_power)))


def pack(ctx, src_tc, dst, dst_pitch, frames, channels, src_dtype=np.float64):
    fn = lib.hipdsp_pack_f64 if np.dtype(src_dtype) == np.float64 else lib.hipdsp_pack_f32
    check(fn(ctx.handle, _p(src_tc), _p(dst), int(dst_pitch), int(frames), int(channels)))


def unpack(ctx, src, src_pitch, dst_tc, frames, channels):
    check(lib.hipdsp_unpack_f64(ctx.handle, _p(src), int(src_pitch), _p(dst_tc), int(frames),
                                int(channels)))


def unpack_spectrum(ctx, src, dst_tcf, frames, channels, nfreq, src_pitch=0):
    check(lib.hipdsp_unpack_spectrum_f64(ctx.handle, _p(src), int(src_pitch), _p(dst_tcf),
                                         int(frames), int(channels), int(nfreq)))


def channel_mean(ctx, x, x_pitch, channels, start, n, out, heterodyne_cycles_per_sample=0.0):
    arr = (ctypes.c_int*len(channels))(*[int(c) for c in channels])
    check(lib.hipdsp_channel_mean(ctx.handle, _p(x), int(x_pitch), arr, len(channels), int(start),
                                  int(n), float(heterodyne_cycles_per_sample), _p(out)))


def stride_copy(ctx, x, n, step, out):
    check(lib.hipdsp_stride_copy(ctx.handle, _p(x), int(n), int(step), _p(out)))


def max_nonneg(ctx, x, n, out):
    check(lib.hipdsp_max_nonneg(ctx.handle, _p(x), int(n), _p(out)))


def band_order_stats(ctx, x, rows, cols, row_stride, rank, out2):
    check(lib.hipdsp_band_order_stats(ctx.handle, _p(x), int(rows), int(cols), int(row_stride), int(rank),
                                      _p(out2)))


def unwrap(ctx, x, x_pitch, channels, frames, thresh, y, y_pitch, ampl_max=1.0, clips=False, down_scale=True):
    check(lib.hipdsp_unwrap(ctx.handle, _p(x), int(x_pitch), int(channels), int(frames), float(thresh),
                            float(ampl_max), int(bool(clips)), int(bool(down_scale)), _p(y), int(y_pitch)))


def pcm_unpack(ctx, pcm_tc, sample_bytes, frames, channels, scale, dst, dst_pitch):
    check(lib.hipdsp_pcm_unpack(ctx.handle, _p(pcm_tc), int(sample_bytes), int(frames), int(channels),
                                float(scale), _p(dst), int(dst_pitch)))


def minmax_decimate(ctx, x, x_pitch, channels, start, stop, step, out, out_pitch):
    check(lib.hipdsp_minmax_decimate(ctx.handle, _p(x), int(x_pitch), int(channels), int(start),
                                     int(stop), int(step), _p(out), int(out_pitch)))


def mean_spectrum_db(ctx, spec_tf, nfreq, i0, i1, out, ref_power=1.0, min_power=1e-20,
                     floor_db=-200.0):
    check(lib.hipdsp_mean_spectrum_db(ctx.handle, _p(spec_tf), int(nfreq), int(i0), int(i1),
                                      float(ref_power), float(min_power), float(floor_db), _p(out)))


def memcpy2d(ctx, dst, dst_pitch_bytes, src, src_pitch_bytes, width_bytes, height):
    check(lib.hipdsp_memcpy2d_d2d(ctx.handle, _p(dst), int(dst_pitch_bytes), _p(src),
                                  int(src_pitch_bytes), int(width_bytes), int(height)))


def synth(ctx, x, x_pitch, channels, frames, rate, seed, c0=0, c_total=None):
    check(lib.hipdsp_synth(ctx.handle, _p(x), int(x_pitch), int(channels), int(frames),
                           float(rate), int(seed), int(c0),
                           int(c_total if c_total is not None else channels)))
