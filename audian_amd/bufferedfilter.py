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


class BufferedFilter(BufferedData):

    def __init__(self, name='filtered', source='data', panel='trace',
                 color='#00ee00', lw_thin=1.1, lw_thick=2):
        super().__init__(name, source, tbefore=10, panel=panel,
                         panel_type='trace', color=color,
                         lw_thin=lw_thin, lw_thick=lw_thick)
        self.highpass_cutoff = 0
        self.lowpass_cutoff = 1
        self.filter_order = 2
        self.sos = None
        self._plans = []

    def open(self, source):
        super().open(source)
        self.highpass_cutoff = 0
        self.lowpass_cutoff = self.rate/2
        self.filter_order = 2
        self.sos = None
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
        from . import _lib
        if self.highpass_cutoff < 0.001*self.rate/2 and \
           self.lowpass_cutoff >= self.rate/2 - 1e-8:
            self.sos = None
        elif self.highpass_cutoff < 0.001*self.rate/2:
            self.sos = butter_sos(self.filter_order, self.lowpass_cutoff,
                                  'lowpass', self.rate)
        elif self.lowpass_cutoff >= self.rate/2 - 1e-8:
            self.sos = butter_sos(self.filter_order, self.highpass_cutoff,
                                  'highpass', self.rate)
        else:
            self.sos = butter_sos(self.filter_order,
                                  (self.highpass_cutoff, self.lowpass_cutoff),
                                  'bandpass', self.rate)
        if self.sos is None:
            self._plans = []
        elif len(self._plans) == (len(self.sos) + _lib.MAX_SECTIONS - 1)//_lib.MAX_SECTIONS:
            for i, plan in enumerate(self._plans):        # re-use the device blocks
                plan.set(self.sos[i*_lib.MAX_SECTIONS:(i + 1)*_lib.MAX_SECTIONS])
        else:
            self._plans = make_plans(self.ctx, self.sos, _lib.MAX_SECTIONS)
        self.recompute_all()
