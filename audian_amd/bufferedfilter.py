"""Filter data on the fly: ``BufferedFilter`` of audian
(``src/audian/bufferedfilter.py`` in /root/reference) with the per-channel
``scipy.signal.sosfilt`` loop replaced by the block-parallel biquad cascade kernel
(``hipdsp_sosfilt``)."""

import numpy as np

from .buffereddata import BufferedData
from .design import butter_sos


MIN_FUSED_FRAMES = 8192       # hipdsp_chain_forward / the prefetching sweeps want at least four 2048-sample tiles

# The fused launch must not lose to the launches it replaces (round 4 shipped one that did: 14.7 against 12.4 ms at the
# reference's default window, bufferedspectrogram.py:14-16).  fusion_costs.json holds both, measured per window and per
# length of the two cascades at BASELINE configs[2]'s shape (tools/fusion_cost_bench.py, picoseconds per channel-sample);
# tests/test_gpu_facade.py::test_fused_launch_never_loses re-measures on the box it runs on.
_FUSION_COSTS = None
FUSION_MARGIN = 1.0           # the fused launch is taken when it costs at most this times the separate launches


def fusion_costs():
    global _FUSION_COSTS
    if _FUSION_COSTS is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fusion_costs.json')) as f:
            _FUSION_COSTS = json.load(f)
    return _FUSION_COSTS


def fused_spectrogram_pays(nfft, hop, filter_sections, envelope_sections):
    """Does hipdsp_chain_forward for this window and these cascades cost no more than the filter's own sweep
    (hipdsp_sosfilt, or the forward sweep of hipdsp_sosfilt_envelope when an envelope rides along) plus
    hipdsp_spectrogram?  Shapes the table does not hold: no."""
    t = fusion_costs()
    fused = t['fused'].get(f'{nfft}/{hop} {filter_sections}+{envelope_sections}')
    alone = t['filter'].get(f'{filter_sections}+{envelope_sections}')
    spec = t['spectrogram'].get(f'{nfft}/{hop}')
    if fused is None or alone is None or spec is None:
        return False
    return fused <= FUSION_MARGIN*(alone + spec)


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
        self._fuse = None          # set by recompute_all() for the one process() call it triggers

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
        if self.sos is None and call is not None and call.doffset == 0 and call.dnframes == len(self._hostbuf) and \
           not isinstance(self.source, BufferedData) and self._builtin(BufferedFilter) and keep is None and \
           dsrc.shape == (max(1, self.channels), spitch):
            # No filter set -- the state a session opens in (bufferedfilter.py:40-42, 32-33: dest = source[nbefore:])
            # -- and the whole buffer is being (re)computed: nothing is copied.  The filtered trace's device mirror
            # becomes a VIEW of the raw slab's device copy (kept between recomputes, _device_source: replaced, never
            # rewritten, when the loader's slab changes), so the spectrogram behind it is the only launch of the
            # update.  The host semantics are unchanged: the host copy is stale and is read back lazily, like any result.
            self._fuse = None
            self._alias_mirror(dsrc.view(nbefore, (dsrc.shape[0]*spitch - nbefore,)), spitch, call)
            return
        ddst, dpitch, is_mirror = self._device_dest(dest, call)
        fuse, self._fuse = self._fuse, None
        if fuse is not None and is_mirror and nbefore == 0 and call.doffset == 0 and \
           call.dnframes == len(self._hostbuf) and self._process_fused(fuse, dsrc, spitch, ddst, dpitch, ns):
            fuse['done'] = True
        elif self.sos is None:
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

    # ---- one launch for the filter and the traces derived from it -----------------------------------
    def _plan_fusion(self):
        """Which of the traces derived from this one can be computed by the filter's own launch:
        a spectrogram whose frames the fused forward sweep covers (hipdsp_chain_forward) and/or an
        envelope of at most two sections over the filtered frames up to the buffer's end (its state
        sweep rides along, the backward sweep follows) -- wherever the user has scrolled to: the
        spectrogram's first frame and the envelope (pre-roll trimmed after a scroll) may start anywhere
        inside the filtered buffer (spec_first / env_first).  None when there is nothing to fuse;
        everything else -- no filter, a cascade longer than one plan, short buffers, other windows,
        subclasses with their own process() -- keeps the separate process() calls of the dependency walk."""
        from .bufferedspectrogram import BufferedSpectrogram
        from .bufferedenvelope import BufferedEnvelope
        if not self._builtin(BufferedFilter) or self.sos is None or len(self._plans) != 1:
            return None
        n = len(self._hostbuf)
        if n < MIN_FUSED_FRAMES or self._plans[0].info()[0] >= 1 << 40:
            return None
        first, count, lead = self._load_geometry(self.offset, n)
        if lead != 0 or count != n:
            return None
        spec = env = None
        for dest in self.dests:
            if not dest.need_update:
                continue
            if spec is None and isinstance(dest, BufferedSpectrogram):
                geom = dest._fusable_with(self)
                if geom is not None:
                    spec = (dest, geom)
            elif env is None and isinstance(dest, BufferedEnvelope):
                env_first = dest._fusable_with(self)
                if env_first is not None:
                    env = (dest, env_first)
        if spec is not None:
            # the cost gate: a fused launch that would be slower than the launches it replaces is not taken (the
            # envelope's state sweep alone still rides on the filter: hipdsp_sosfilt_envelope, one pass over the slab)
            n_env = len(env[0].sos) if env is not None else 0
            if not fused_spectrogram_pays(spec[0].nfft, spec[0].hop, len(self.sos), n_env):
                spec = None
        if spec is None and env is None:
            return None
        return {'spec': spec, 'env': env, 'done': False}

    def _process_fused(self, fuse, dsrc, spitch, ddst, dpitch, ns):
        """The fused launch(es); False when the library does not cover the case after all."""
        from . import hipdsp
        plan = self._plans[0]
        spec, env = fuse['spec'], fuse['env']
        env, env_first = env if env is not None else (None, 0)
        eplan = env._plan if env is not None else None
        edev = env._mirror() if env is not None else None
        clamp = env is not None and env.highpass_cutoff == 0
        try:
            if spec is not None:
                trace, (nd, spec_first, spec_frames) = spec
                F = trace.nfft//2 + 1
                hipdsp.chain_forward(self.ctx, plan, eplan, dsrc, spitch, ddst, dpitch, self.channels, ns,
                                     trace.nfft, trace.hop, self.rate, trace._mirror(), nd, psd_pitch=nd*F,
                                     rectify=True, gain=np.pi/2, spec_frames=spec_frames, spec_first=spec_first,
                                     env_first=env_first)
                if env is not None:
                    hipdsp.sosfilt_envelope(self.ctx, plan, eplan, dsrc, spitch, ddst, dpitch, edev, ns - env_first,
                                            self.channels, ns, rectify=True, gain=np.pi/2, clamp=clamp, phase=2,
                                            env_first=env_first)
            else:
                hipdsp.sosfilt_envelope(self.ctx, plan, eplan, dsrc, spitch, ddst, dpitch, edev, ns - env_first,
                                        self.channels, ns, rectify=True, gain=np.pi/2, clamp=clamp, phase=0,
                                        env_first=env_first)
        except NotImplementedError:
            return False
        if spec is not None:
            spec[0]._fused_token = True
        if env is not None:
            env._fused_token = True
        return True

    def recompute_all(self):
        """Recompute this trace and the traces derived from it (buffereddata.py:149-153) -- depth first as
        in the reference, but the filter's own launch already fills the mirrors of the derived traces it
        can serve (_plan_fusion); their recompute() in the walk below then only does its bookkeeping
        (buffer_changed, spec_rect, frequencies)."""
        if not self.need_update:
            return
        try:
            if not self._builtin(BufferedFilter):
                self.recompute()                 # a subclass with its own process() / recompute(): the plain walk
            else:
                if self._source_len() > 0:       # (what recompute() does, with the fusion planned in between)
                    self.allocate_buffer()
                self._fuse = self._plan_fusion() if len(self._hostbuf) > 0 else None
                self.reload_buffer()
            for dest in self.dests:
                dest.recompute_all()
        finally:
            # whatever happened on the way: no token outlives the walk it was issued for
            self._fuse = None
            for dest in self.dests:
                dest._fused_token = False

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
