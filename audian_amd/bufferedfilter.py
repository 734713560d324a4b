"""Filter data on the fly: ``BufferedFilter`` of audian
(``src/audian/bufferedfilter.py`` in /root/reference) with the per-channel
``scipy.signal.sosfilt`` loop replaced by the block-parallel biquad cascade kernel
(``hipdsp_sosfilt``)."""

import numpy as np

from .buffereddata import BufferedData
from .design import butter_sos


def make_plans(ctx, sos, max_sections):
    """SOS table -> list of device plans of at most `max_sections` sections each.
    Splitting a zero-state cascade is exact; the hand-over between plans is float32."""
    from . import hipdsp
    return [hipdsp.SosPlan(ctx, sos[i:i + max_sections])
            for i in range(0, len(sos), max_sections)]


def choose_design(highpass, lowpass, rate):
    """Which Butterworth response BufferedFilter.update asks for (bufferedfilter.py:39-52):
    a high-pass below 0.1 % of Nyquist counts as "off", a low-pass within 1e-8 Hz of Nyquist too.
    Returns (btype, Wn) or None for the pass-through."""
    nyquist = rate/2
    hp_off = highpass < 0.001*nyquist
    lp_off = lowpass >= nyquist - 1e-8
    if hp_off and lp_off:
        return None
    if hp_off:
        return 'lowpass', lowpass
    if lp_off:
        return 'highpass', highpass
    return 'bandpass', (highpass, lowpass)


class BufferedFilter(BufferedData):
    """Same constructor and attributes as audian's class (bufferedfilter.py:11-21):
    highpass_cutoff, lowpass_cutoff, filter_order, sos."""

    def __init__(self, name='filtered', source='data', panel='trace', color='#00ee00', lw_thin=1.1,
                 lw_thick=2):
        BufferedData.__init__(self, name, source, tbefore=10, panel=panel, panel_type='trace',
                              color=color, lw_thin=lw_thin, lw_thick=lw_thick)
        self._reset(lowpass=1)
        self._plans = []

    def _reset(self, lowpass):
        self.highpass_cutoff, self.lowpass_cutoff, self.filter_order, self.sos = 0, lowpass, 2, None

    def open(self, source):
        """Link to `source`; the filter starts wide open (0 Hz ... Nyquist, order 2)."""
        BufferedData.open(self, source)
        self._reset(lowpass=self.rate/2)
        self.update()

    def process(self, source, dest, nbefore):
        """dest = sosfilt(sos, source, axis=0)[nbefore:] per channel, zero initial state;
        pass-through copy when no filter is set (bufferedfilter.py:31-36)."""
        from . import hipdsp
        if len(dest) != len(source) - nbefore:
            raise ValueError(f'could not broadcast input array from shape '
                             f'({len(source) - nbefore},) into shape ({len(dest)},)')
        call = self._take_call(source, dest)
        ns = len(source)
        if len(dest) == 0:
            return
        dsrc, spitch, keep = self._device_source(source, call)
        ddst, dpitch, is_mirror = self._device_dest(dest, call)
        if self.sos is None:
            hipdsp.sosfilt(self.ctx, None, dsrc, spitch, ddst, dpitch, self.channels, ns, nbefore)
        else:
            plans = self._plans
            cur, cpitch = dsrc, spitch
            for i, plan in enumerate(plans):
                last = i == len(plans) - 1
                if last:
                    hipdsp.sosfilt(self.ctx, plan, cur, cpitch, ddst, dpitch, self.channels, ns,
                                   nbefore)
                else:
                    tmp = hipdsp.DeviceArray(self.ctx, (self.channels, ns), np.float32)
                    hipdsp.sosfilt(self.ctx, plan, cur, cpitch, tmp, ns, self.channels, ns, 0)
                    cur, cpitch = tmp, ns
        self._finish_dest(dest, ddst, dpitch, is_mirror, call)
        if keep is not None or not is_mirror:
            self.ctx.synchronize()

    def update(self):
        """Design the filter for the current cut-offs and order, refresh the device plans and
        recompute this trace and everything derived from it."""
        from . import _lib
        design = choose_design(self.highpass_cutoff, self.lowpass_cutoff, self.rate)
        self.sos = None if design is None else butter_sos(self.filter_order, design[1], design[0], self.rate)
        chunk = _lib.MAX_SECTIONS
        if self.sos is None:
            self._plans = []
        elif len(self._plans) == -(-len(self.sos)//chunk):
            for i, plan in enumerate(self._plans):        # re-use the device blocks
                plan.set(self.sos[i*chunk:(i + 1)*chunk])
        else:
            self._plans = make_plans(self.ctx, self.sos, chunk)
        self.recompute_all()
