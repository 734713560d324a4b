"""Helpers shared by the -m gpu parity tests: run the HIP path through the C ABI on
host arrays laid out like the reference's buffers ((T, C) float64/float32)."""

import numpy as np


def ctx():
    from audian_amd import hipdsp
    return hipdsp.default_context()


def to_planar(c, x_tc):
    """(T, C) host array -> planar (C, T) float32 DeviceArray via the pack kernel."""
    from audian_amd import hipdsp
    x_tc = np.ascontiguousarray(x_tc)
    T, C = x_tc.shape
    src = hipdsp.DeviceArray.from_host(c, x_tc)
    dst = hipdsp.DeviceArray(c, (C, T), np.float32)
    hipdsp.pack(c, src, dst, T, T, C, src_dtype=x_tc.dtype)
    return dst


def from_planar(c, d, T, C, pitch=None):
    """planar float32 DeviceArray -> (T, C) float64 host array via the unpack kernel."""
    from audian_amd import hipdsp
    out = hipdsp.DeviceArray(c, (T, C), np.float64)
    hipdsp.unpack(c, d, pitch if pitch is not None else T, out, T, C)
    return out.to_host()


def gpu_sosfilt(sos, x_tc, skip=0, max_segments=0):
    from audian_amd import hipdsp
    c = ctx()
    c.set_max_segments(max_segments)
    T, C = x_tc.shape
    dx = to_planar(c, x_tc)
    dy = hipdsp.DeviceArray(c, (C, max(T - skip, 1)), np.float32)
    plan = hipdsp.SosPlan(c, sos) if sos is not None else None
    hipdsp.sosfilt(c, plan, dx, T, dy, max(T - skip, 1), C, T, skip)
    y = from_planar(c, dy, T - skip, C, pitch=max(T - skip, 1)) if T - skip > 0 \
        else np.zeros((0, C))
    c.set_max_segments(0)
    return y


def gpu_envelope(sos, x_tc, skip=0, rectify=True, clamp=True, max_segments=0):
    from audian_amd import hipdsp
    c = ctx()
    c.set_max_segments(max_segments)
    T, C = x_tc.shape
    dx = to_planar(c, x_tc)
    dy = hipdsp.DeviceArray(c, (C, max(T - skip, 1)), np.float32)
    plan = hipdsp.SosPlan(c, sos) if sos is not None else None
    try:
        hipdsp.envelope(c, plan, dx, T, dy, max(T - skip, 1), C, T, skip, rectify=rectify,
                        clamp=clamp)
    finally:
        c.set_max_segments(0)
    return from_planar(c, dy, T - skip, C, pitch=max(T - skip, 1))


def gpu_spectrogram(x_tc, rate, nfft, hop, frames_out, want_db=False):
    """Returns dest (frames_out, C, F) float64 like BufferedSpectrogram's buffer."""
    from audian_amd import hipdsp
    c = ctx()
    T, C = x_tc.shape
    F = nfft//2 + 1
    dx = to_planar(c, x_tc) if T > 0 else hipdsp.DeviceArray(c, (C, 1), np.float32)
    out = hipdsp.DeviceArray(c, (C, frames_out, F), np.float32)
    db = hipdsp.DeviceArray(c, (C, frames_out, F), np.float32) if want_db else None
    hipdsp.spectrogram(c, dx, max(T, 1), C, T, nfft, hop, rate, out, frames_out, db_out=db)
    dst = hipdsp.DeviceArray(c, (frames_out, C, F), np.float64)
    hipdsp.unpack_spectrum(c, out, dst, frames_out, C, F)
    res = dst.to_host()
    if want_db:
        return res, db.to_host().transpose(1, 0, 2)
    return res
