"""Compute envelope on the fly: ``BufferedEnvelope`` of audian
(``src/audian/bufferedenvelope.py`` in /root/reference).  The
``sosfiltfilt(sos, (pi/2)*|x|, axis=0)`` call becomes one fused device call
(``hipdsp_envelope``: rectify + odd padding + zi-scaled forward and backward biquad
passes + clamp)."""

import numpy as np

from .buffereddata import BufferedData
from .design import butter_sos


class BufferedEnvelope(BufferedData):
    """Same constructor and attributes as audian's class (bufferedenvelope.py:13-26):
    envelope_cutoff, highpass_cutoff, filter_order, sos."""

    def __init__(self, name='envelope', source='filtered', panel='trace', color='#ff8800', lw_thin=2.5,
                 lw_thick=4, envelope_cutoff=500, filter_order=2, highpass_cutoff=0):
        BufferedData.__init__(self, name, source, tbefore=1, panel=panel, panel_type='trace',
                              color=color, lw_thin=lw_thin, lw_thick=lw_thick)
        self.envelope_cutoff, self.highpass_cutoff = envelope_cutoff, highpass_cutoff
        self.filter_order, self.sos = filter_order, None
        self._plan = None
        self._plans = []           # cascades longer than one plan: chained plans (hipdsp_envelope_multi)

    def open(self, source):
        BufferedData.open(self, source)
        self.sos = None
        self.update()

    def _fusable_with(self, filt):
        """The sample of the filtered buffer this envelope starts at if its whole-buffer recompute is what the
        filter's fused launch can compute (BufferedFilter._plan_fusion), else None: one plan of at most two
        decaying sections over the filtered frames from that sample to the buffer's end.  At the start of a
        recording that is sample 0; after a scroll align_buffer trims the envelope's second of pre-roll
        (buffereddata.py:75-88) and the reference's sosfiltfilt pads and starts there: the launch's env_first."""
        if not self._builtin(BufferedEnvelope) or self.sos is None or self._plans or self._plan is None or \
           len(self.sos) > 2:
            return None
        warm, edge = self._plan.info()
        if warm >= 1 << 40:
            return None
        if len(filt._hostbuf) > 0:
            self.allocate_buffer()               # what recompute() does first
        n = len(self._hostbuf)
        first, count, lead = self._load_geometry(self.offset, n) if n > 0 else (-1, 0, 0)
        if first < 0 or lead != 0 or first + count != len(filt._hostbuf) or n != count or count <= edge:
            return None
        return first

    def process(self, source, dest, nbefore):
        """dest = sosfiltfilt(sos, (pi/2)|source|, axis=0)[nbefore:], negatives clamped to
        0 unless a high-pass is set; zeros when the design failed
        (bufferedenvelope.py:34-41).  Raises ValueError like scipy when the slab is not
        longer than the pad length."""
        from . import hipdsp
        if self.sos is not None and len(dest) != len(source) - nbefore:
            raise ValueError(f'could not broadcast input array from shape '
                             f'({len(source) - nbefore},) into shape ({len(dest)},)')
        call = self._take_call(source, dest)
        if len(dest) == 0:
            return
        if self._take_fused(call):
            return
        ddst, dpitch, is_mirror = self._device_dest(dest, call)
        keep = None
        if self.sos is None:
            hipdsp.envelope(self.ctx, None, None, 0, ddst, dpitch, self.channels,
                            len(dest), 0)
        else:
            dsrc, spitch, keep = self._device_source(source, call)
            if self._plans:
                # more sections than one plan holds: sosfiltfilt step by step over the chained plans
                hipdsp.envelope_multi(self.ctx, self._plans, dsrc, spitch, ddst, dpitch, self.channels,
                                      len(source), nbefore, rectify=True, gain=np.pi/2,
                                      clamp=self.highpass_cutoff == 0)
            else:
                hipdsp.envelope(self.ctx, self._plan, dsrc, spitch, ddst, dpitch, self.channels,
                                len(source), nbefore, rectify=True, gain=np.pi/2,
                                clamp=self.highpass_cutoff == 0)
        self._finish_dest(dest, ddst, dpitch, is_mirror, call)
        if keep is not None or not is_mirror:
            self.ctx.synchronize()

    def update(self):
        """Low-pass at envelope_cutoff, or band-pass (highpass_cutoff, envelope_cutoff) when a
        high-pass is set; a design scipy would reject leaves sos = None, i.e. a zero envelope
        (bufferedenvelope.py:44-55)."""
        from . import hipdsp, _lib
        band = self.highpass_cutoff > 0
        wn = (self.highpass_cutoff, self.envelope_cutoff) if band else self.envelope_cutoff
        try:
            self.sos = butter_sos(self.filter_order, wn, 'bandpass' if band else 'lowpass', self.rate)
        except ValueError:
            self.sos = None
        self._plans = []
        if self.sos is not None and len(self.sos) > _lib.MAX_SECTIONS:
            # any filter_order is legal in the reference (bufferedenvelope.py:13-16): cascades that do not
            # fit one plan run as chained plans (hipdsp_envelope_multi)
            chunk = _lib.MAX_SECTIONS
            self._plans = [hipdsp.SosPlan(self.ctx, self.sos[i:i + chunk]) for i in range(0, len(self.sos), chunk)]
        elif self.sos is not None:
            if self._plan is None:
                self._plan = hipdsp.SosPlan(self.ctx, self.sos)
            else:
                self._plan.set(self.sos)
        self.recompute_all()
