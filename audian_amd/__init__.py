"""audian_amd -- MI355X-native (gfx950) implementation of audian's BufferedData DSP
hot path: Butterworth SOS band-pass (BufferedFilter), rectified zero-phase envelope
(BufferedEnvelope) and Hann STFT power spectrogram + dB (BufferedSpectrogram),
as hand-written HIP kernels behind the C ABI of ``include/hip_dsp.h``.

Importing this package does not touch the GPU; ``audian_amd.hipdsp`` (and anything
that computes) loads ``libhip_dsp.so`` and fails loudly if it is missing.
"""

__version__ = '0.1.0'
