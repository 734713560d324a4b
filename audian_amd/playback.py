"""Audio playback data of a region: the arithmetic of ``DataBrowser.play_region``
(``src/audian/databrowser.py:1702-1729`` in /root/reference) on the device mirror of a
trace -- channel-group means, optional heterodyne (multiply by a sine carrier, zero-phase
20 kHz low-pass, down-sample).  Fading and the sound card stay with the caller."""

import numpy as np

from .buffereddata import BufferedData, _covers
from .design import butter_sos


def play_data(trace, t0, t1, show_channels, heterodyne_freq=None):
    """Returns ``(playdata, rate)``: playdata is (frames, min(2, len(show_channels))) float64,
    the left/right channel being the mean over the first/second half of `show_channels`
    (databrowser.py:1711-1715); with `heterodyne_freq` the heterodyne chain of
    databrowser.py:1716-1727 is applied and the rate reduced accordingly."""
    from . import hipdsp
    rate = trace.rate
    i0 = int(np.round(t0*rate))
    i1 = int(np.round(t1*rate))
    if i0 < 0:
        i0 = 0
    if i1 > len(trace):
        i1 = len(trace)
    n = i1 - i0
    n2 = (len(show_channels) + 1)//2
    groups = [list(show_channels[:n2])]
    if len(show_channels) > 1:
        groups.append(list(show_channels[n2:]))
    trace.update_buffer(i0, i1)
    a = i0 - trace.offset
    on_device = isinstance(trace, BufferedData) and trace._dev is not None and \
        _covers(trace._dev_valid, a, a + n) and n > 0
    if not on_device:
        data = trace[i0:i1, :]
        play = np.zeros((n, len(groups)))
        for k, grp in enumerate(groups):
            play[:, k] = np.mean(data[:, grp], 1)
        if heterodyne_freq:
            raise NotImplementedError('heterodyne playback needs the trace on the device')
        return play, rate
    ctx = trace.ctx
    cap = trace._pitch() if isinstance(trace, BufferedData) else len(trace._hostbuf)
    cps = float(heterodyne_freq)/rate if heterodyne_freq else 0.0
    mixed = hipdsp.DeviceArray(ctx, (len(groups), n), np.float32)
    for k, grp in enumerate(groups):
        hipdsp.channel_mean(ctx, trace._dev, cap, grp, a, n, mixed.view(k*n, (1,)),
                            heterodyne_cycles_per_sample=cps)
    if not heterodyne_freq:
        return mixed.to_host().T.astype(np.float64), rate
    fcutoff = 20000.0
    plan = hipdsp.SosPlan(ctx, butter_sos(2, fcutoff, 'lowpass', rate))
    nstep = int(np.round(rate/(2*fcutoff)))
    if nstep < 1:
        nstep = 1
    low = hipdsp.DeviceArray(ctx, (len(groups), n), np.float32)
    hipdsp.envelope(ctx, plan, mixed, n, low, n, len(groups), n, 0, rectify=False, clamp=False)
    m = (n + nstep - 1)//nstep
    dec = hipdsp.DeviceArray(ctx, (len(groups), m), np.float32)
    for k in range(len(groups)):
        hipdsp.stride_copy(ctx, low.view(k*n, (1,)), n, nstep, dec.view(k*m, (1,)))
    return dec.to_host().T.astype(np.float64), rate/nstep
