"""Spectrogram of source data on the fly: ``BufferedSpectrogram`` of audian
(``src/audian/bufferedspectrogram.py`` in /root/reference).  thunderlab's
``spectrogram`` / ``decibel`` (scipy.signal.spectrogram with a Hann window, constant
detrend, density scaling) become ``hipdsp_spectrogram`` / ``hipdsp_decibel_image``."""

import numpy as np

from .buffereddata import BufferedData


def decibel(power, ref_power=1.0, min_power=1e-20):
    """thunderlab.powerspectrum.decibel for host arrays and scalars (used by the
    once-per-trace colour-range estimate and by cursors; the image path is on the
    device): 10*log10(power/ref_power), -inf at or below min_power."""
    p = np.asarray(power, dtype=np.float64)
    out = np.full(p.shape, -np.inf)
    m = p > min_power
    out[m] = 10.0*np.log10(p[m]/ref_power)
    return out if out.ndim else float(out)


# nfft / hop the filter's fused forward sweep writes (hipdsp_chain_forward)
FUSED_WINDOWS = {(2048, 1024), (2048, 512), (1024, 512), (1024, 256), (512, 256), (256, 128)}

# parameter limits of the reference (bufferedspectrogram.py:83-100)
MIN_NFFT = 8
MAX_NFFT = 2**30
MAX_OVERLAP = 0.99999


def hop_for(nfft, overlap_frac):
    """Frame advance for `nfft`-sample windows overlapping by `overlap_frac`: rounded to the
    nearest sample and kept inside 1 ... nfft (what BufferedSpectrogram.set_hop settles on)."""
    return int(min(max(np.round(nfft*(1 - overlap_frac)), 1), nfft))


class BufferedSpectrogram(BufferedData):
    """Same constructor and attributes as audian's class (bufferedspectrogram.py:14-29):
    nfft / hop / overlap_frac, frequencies, fresolution, tresolution, spec_rect, use_spec, init."""

    def __init__(self, name='spectrogram', source='filtered', panel='spectrogram', nfft=256,
                 overlap_frac=0.5):
        BufferedData.__init__(self, name, source, tafter=10, panel=panel, panel_type='spectrogram')
        self.nfft, self.overlap_frac, self.hop = nfft, overlap_frac, 0
        self.set_hop()
        self.frequencies = np.zeros(0)
        self.fresolution = self.tresolution = 1
        self.spec_rect, self.use_spec, self.init = [], True, True

    def _bins(self):
        return self.nfft//2 + 1

    def _set_resolutions(self, source_rate):
        self.fresolution = source_rate/self.nfft
        self.tresolution = self.hop/source_rate

    def open(self, source):
        """Link to `source` at a frame rate of one spectrum per hop.  Note that the reference
        truncates the hop here (bufferedspectrogram.py:32) while set_hop() rounds it."""
        self.hop = int((1 - self.overlap_frac)*self.nfft)
        self._set_resolutions(source.rate)
        self.frequencies = np.arange(0, 0.5*(source.rate + self.fresolution), self.fresolution)
        self.spec_rect, self.use_spec = [], True
        BufferedData.open(self, source, self.hop, more_shape=(self._bins(),))
        self.unit = self.unit + '^2/Hz'
        # the y axis of a spectrogram is frequency
        self.ampl_min, self.ampl_max = 0, 0.5*self.source.rate

    def _fusable_with(self, filt):
        """(frames, first source sample, source samples) of this spectrogram's whole-buffer recompute if the
        filter's fused forward sweep can write it (BufferedFilter._plan_fusion), else None: the window must be
        one of the sweep's.  After a scroll the filtered buffer starts at an arbitrary sample of the recording and
        frame 0 `first` = ceil(offset / hop) hop - offset samples into it (align_buffer, buffereddata.py:75-88):
        the sweep shifts its tile grid by that (hipdsp_chain_forward's spec_first)."""
        if not self._builtin(BufferedSpectrogram) or (self.nfft, self.hop) not in FUSED_WINDOWS:
            return None
        if len(filt._hostbuf) > 0:
            self.allocate_buffer()               # what recompute() does first
        nd = len(self._hostbuf)
        if nd == 0:
            return None
        first, count, lead = self._load_geometry(self.offset, nd)
        if first < 0 or lead != 0 or count <= 0 or first + count > len(filt._hostbuf):
            return None
        return nd, first, count

    def process(self, source, dest, nbefore):
        """dest[k, c, :] = one-sided PSD of source[k*hop : k*hop + nfft, c]; frames that do
        not fit into the source are zero (bufferedspectrogram.py:45-66)."""
        from . import hipdsp
        call = self._take_call(source, dest)
        nd = len(dest)
        F = self.nfft//2 + 1
        if nd > 0 and self._take_fused(call):
            nsource = min((nd - 1)*self.hop + self.nfft, len(source))
            if nsource >= self.nfft:
                self.frequencies = np.arange(F)*self.source.rate/self.nfft
        elif nd > 0:
            dsrc, spitch, keep = self._device_source(source, call)
            ddst, dpitch, is_mirror = self._device_dest(dest, call)
            hipdsp.spectrogram(self.ctx, dsrc, spitch, self.channels, len(source), self.nfft,
                               self.hop, self.source.rate, ddst, nd, out_pitch=dpitch)
            self._finish_dest(dest, ddst, dpitch, is_mirror, call)
            if keep is not None or not is_mirror:
                self.ctx.synchronize()
            nsource = min((nd - 1)*self.hop + self.nfft, len(source))
            if nsource >= self.nfft:
                self.frequencies = np.arange(F)*self.source.rate/self.nfft
        # extent of the full buffer:
        self.spec_rect = [self.offset/self.rate, 0,
                          len(self._hostbuf)/self.rate,
                          self.source.rate/2 + self.fresolution]

    def set_hop(self):
        """Recompute hop from nfft and overlap_frac; True if it changed (then overlap_frac is
        snapped to the value the integer hop really gives)."""
        hop = hop_for(self.nfft, self.overlap_frac)
        changed = hop != self.hop
        if changed:
            self.hop = hop
            self.overlap_frac = 1 - hop/self.nfft
        return changed

    def update(self, nfft=None, overlap_frac=None):
        """New window length and/or overlap (DataBrowser.set_resolution, databrowser.py:1195):
        nfft is limited to 8 ... min(len(source)//2, 2**30) with the upper limit winning,
        overlap_frac to 0 ... 0.99999; everything is recomputed if nfft or hop changed."""
        dirty = False
        if nfft is not None:
            nfft = min(max(nfft, MIN_NFFT), min(len(self.source)//2, MAX_NFFT))
            dirty = nfft != self.nfft
            self.nfft = nfft
        if overlap_frac is not None:
            self.overlap_frac = min(max(overlap_frac, 0.0), MAX_OVERLAP)
        dirty = self.set_hop() or dirty
        if not dirty:
            return
        self._set_resolutions(self.source.rate)
        self.update_step(self.hop, more_shape=(self._bins(),))
        self.recompute_all()

    def decibel_image(self, channel, ref_power=1.0, min_power=1e-20):
        """decibel(buffer[:, channel, :].T) as SpecItem.update_plot needs it
        (src/audian/specitem.py:36), computed on the device from the mirror when it is
        valid; returns a (F, frames) float32 host array."""
        from . import hipdsp
        from .buffereddata import _covers
        n = len(self._hostbuf)
        F = self.nfft//2 + 1
        if n == 0:
            return np.zeros((F, 0), dtype=np.float32)
        if self._dev is not None and _covers(self._dev_valid, 0, n):
            img = hipdsp.DeviceArray(self.ctx, (F, n), np.float32)
            hipdsp.decibel_image(self.ctx, self._dev.view(channel*n*F, (1,)), img, n, F,
                                 ref_power, min_power)
            return img.to_host()
        return decibel(self.buffer[:, channel, :].T, ref_power, min_power).astype(np.float32)

    def decimated_image(self, start, stop, step, channel, ref_power=1.0, min_power=1e-20):
        """The dB image of frames [start, stop) (absolute frame indices inside the current buffer)
        of one channel at screen resolution: every column is the maximum over `step` frames --
        TraceItem.update_plot's np.maximum.reduceat (src/audian/traceitem.py:55-61) applied to the
        spectrogram, the reference's open TODO (README.md:96) -- then decibel(...).T as
        SpecItem.update_plot shows it (src/audian/specitem.py:36).  Device reduction when the
        mirror is valid, so only (F, ceil((stop-start)/step)) float32 values cross PCIe."""
        from . import hipdsp
        from .buffereddata import _covers
        a, b = int(start) - self.offset, int(stop) - self.offset
        n = len(self._hostbuf)
        F = self.nfft//2 + 1
        if a < 0 or b > n or b < a or step < 1:
            raise IndexError('range outside the loaded buffer')
        ncols = (b - a + step - 1)//step
        if ncols == 0:
            return np.zeros((F, 0), dtype=np.float32)
        if self._dev is not None and _covers(self._dev_valid, a, b):
            img = hipdsp.DeviceArray(self.ctx, (F, ncols), np.float32)
            hipdsp.decibel_image_decimate(self.ctx, self._dev.view(channel*n*F, (1,)), img, n, F, a, b, step,
                                          ref_power, min_power)
            return img.to_host()
        seg = np.arange(0, b - a, step)
        block = np.maximum.reduceat(self.buffer[a:b, channel, :], seg, axis=0)
        return decibel(block, ref_power, min_power).T.astype(np.float32)

    def mean_power_db(self, i0, i1, channel, floor_db=-200.0):
        """Power spectrum of frames [i0, i1) (absolute frame indices) of one channel as
        SpectrogramPlot.update_plot shows it (src/audian/spectrogramplot.py:158-160):
        decibel(mean over frames), floored at -200 dB.  Device reduction when the mirror is
        valid; returns float64 (F,)."""
        from . import hipdsp
        from .buffereddata import _covers
        F = self.nfft//2 + 1
        a, b = int(i0) - self.offset, int(i1) - self.offset
        n = len(self._hostbuf)
        if a < 0 or b > n or b <= a:
            raise IndexError('range outside the loaded buffer')
        if self._dev is not None and _covers(self._dev_valid, a, b):
            out = hipdsp.DeviceArray(self.ctx, (F,), np.float32)
            hipdsp.mean_spectrum_db(self.ctx, self._dev.view(channel*n*F, (1,)), F, a, b, out,
                                    floor_db=floor_db)
            return out.to_host().astype(np.float64)
        power = decibel(np.mean(self.buffer[a:b, channel, :], axis=0))
        power[power < floor_db] = floor_db
        return power

    def estimate_noiselevels(self, channel):
        """Colour range for the spectrogram image (bufferedspectrogram.py:109-126): 95th
        percentile of the dB values in the top 1/16 of the band, and the maximum dB.  With a
        valid device mirror both are device reductions (radix select of the two order statistics the
        percentile interpolates between, max reduction; dB is monotonic): three floats cross PCIe
        instead of the whole slab."""
        if not self.init or len(self._hostbuf) == 0 or len(self._hostbuf.shape) < 3:
            return None, None
        from . import hipdsp
        from .buffereddata import _covers
        F = self._hostbuf.shape[2]
        n = len(self._hostbuf)
        nf = max(F//16, 1)                       # top 1/16 of the band
        if self._dev is not None and _covers(self._dev_valid, 0, n) and self._stale:
            # everything on the device: the two order statistics np.percentile(.., 95) interpolates between
            # (radix select over the top band) and the maximum; three floats cross PCIe.  decibel() is
            # monotonic, so the percentile of the dB values is the interpolation of their dB.
            base = self._dev.view(channel*n*F, (1,))
            pos = 0.95*(n*nf - 1)                      # np.percentile, linear interpolation
            k = int(np.floor(pos))
            stats = hipdsp.DeviceArray(self.ctx, (3,), np.float32)
            hipdsp.band_order_stats(self.ctx, self._dev.view(channel*n*F + F - nf, (1,)), n, nf, F, k, stats)
            hipdsp.max_nonneg(self.ctx, base, n*F, stats.view(2, (1,)))
            lo, hi, top = (float(v) for v in stats.to_host().astype(np.float64))
            with np.errstate(all='ignore'):
                dlo, dhi = decibel(lo), decibel(hi)
                zmin = dlo + (pos - k)*(dhi - dlo) if pos > k else dlo
            zmax = decibel(top)
        else:
            with np.errstate(all='ignore'):
                zmin = np.percentile(decibel(self.buffer[:, channel, -nf:]), 95)
            zmax = np.max(decibel(self.buffer[:, channel, :]))
        if not (np.isfinite(zmin) and np.isfinite(zmax)):
            return None, None
        self.init = False                       # once per trace
        # upper end 5 % below the maximum; at least 20 dB of range, at most 80 dB
        lift = max(0.95*(zmax - zmin), 20)
        zmax = zmin + lift
        return (zmax - 80 if lift > 80 else zmin), zmax
