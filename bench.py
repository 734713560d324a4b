#!/usr/bin/env python3
"""Benchmark of the BufferedData DSP hot path on MI355X.

One "step" = one pass of the chain  data -> BufferedFilter (Butterworth band-pass)
-> {BufferedSpectrogram (Hann STFT PSD), BufferedEnvelope (rectify + sosfiltfilt)}
over one batch of synthetic multichannel float32 audio that is already resident in
HBM.  --config picks the BASELINE.json workload:

    2 (default)  configs[2]: 64 ch x 600 s x 96 kHz per GPU, nfft 2048 / hop 1024, band-pass 300-3000 Hz
                 order 2, envelope low-pass 20 Hz -- the configuration the metric is quoted on
    3            configs[3]: 256 ch x 600 s x 96 kHz sharded over 8 GPUs = 32 ch per GPU, same chain, plus the
                 RCCL all-gather of the merged spectrogram tile (three tile sizes, see --tile)
    1            configs[1]: 4 ch x 60 s x 48 kHz, nfft 1024 / hop 256, band-pass order 4 (four sections)

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N ...        (no launcher: bench.py starts its own N ranks, see self_launch())

Rank 0 prints ONE JSON line: whole-job Msamples/s, the HBM roofline of the dominant
kernel (HIP-event timed inside the timed region) and, at N=1, the reference's scipy
CPU path timed on this box's host cores on a bounded sample.  At N > 1 the timed step includes the
all-gather of the --tile spectrogram tile; after the timed region the same K steps are repeated without
any gather and with each of the three tile sizes (`legs`: compute_ms / gather_ms separately).
"""

import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

CONFIGS = {
    1: dict(channels=4, seconds=60.0, rate=48000.0, nfft=1024, hop=256, order=4, env=20.0,
            name='BASELINE configs[1]'),
    2: dict(channels=64, seconds=600.0, rate=96000.0, nfft=2048, hop=1024, order=2, env=20.0,
            name='BASELINE configs[2]'),
    3: dict(channels=32, seconds=600.0, rate=96000.0, nfft=2048, hop=1024, order=2, env=20.0,
            name='BASELINE configs[3] (256 ch over 8 GPUs = 32 ch/GPU)'),
}
# spectrogram tiles that are all-gathered at N > 1 (seconds of the recording)
TILES = {'visible': 10.0,      # what the display shows (plotranges.py:141-144): "gather only the visible tile"
         'window': 80.0,       # SURVEY 8e's interactive tile: the resident window (data.py:17,168), 0.98 GB/rank at 32 ch
         'full': None}         # the whole spectrogram of the rank (7.4 GB at 32 ch): xGMI-bound by construction


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument('--channels', type=int, default=None, help='channels per GPU (default: the config\'s)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help="'weak' (default): the config's channels PER GPU (64 at configs[2]: 512 at N = 8); 'strong': "
                         "the config's channels in TOTAL, sharded over the N GPUs (64 / N per GPU) -- BASELINE.json's "
                         "metric as it is worded (\"64ch x 96kHz, 1/2/4/8 GPU\")")
    ap.add_argument('--strong-world', type=int, default=None,
                    help="rehearsal on fewer GPUs: run one rank's share of a strong-scaling run over this many GPUs")
    ap.add_argument('--seconds', type=float, default=None)
    ap.add_argument('--rate', type=float, default=None)
    ap.add_argument('--nfft', type=int, default=None)
    ap.add_argument('--hop', type=int, default=None)
    ap.add_argument('--hp', type=float, default=300.0)
    ap.add_argument('--lp', type=float, default=3000.0)
    ap.add_argument('--order', type=int, default=None)
    ap.add_argument('--env', type=float, default=None)
    ap.add_argument('--tile', default='visible', choices=sorted(TILES),
                    help='N>1: the spectrogram tile all-gathered in every TIMED step (the legs after the timed '
                         'region cover all three)')
    ap.add_argument('--tile-seconds', type=float, default=None, help='N>1: override the length of --tile')
    ap.add_argument('--gather', default='torch', choices=['torch', 'c-abi'],
                    help="N>1: 'torch' = torch.distributed all_gather_into_tensor, 'c-abi' = hipdsp_allgather_f32 "
                         '(the same RCCL call through libhip_dsp, on a second context/stream)')
    ap.add_argument('--reserve-cus', type=int, default=None,
                    help='CUs the fused forward sweep leaves free (default 8 at N>1, 0 at N=1): its workgroups take '
                         'a whole CU each, RCCL\'s resident all-gather kernel needs some of its own')
    ap.add_argument('--no-legs', action='store_true', help='N>1: skip the compute-only / per-tile legs')
    ap.add_argument('--full-leg', action='store_true',
                    help="N>1: also run the leg that gathers every rank's WHOLE spectrogram (7.4-14.8 GB per rank, "
                         'merged on every rank: 59-118 GB at N=8, seconds per gather); by default only the visible '
                         'and the window tile are rehearsed after the timed region')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-seconds', type=float, default=60.0)
    ap.add_argument('--max-segments', type=int, default=0)
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="N>1 collective backend; 'gloo' (host staging) only to rehearse the "
                         "multi-rank path on a one-GPU box")
    ap.add_argument('--same-device', action='store_true',
                    help='rehearsal: every rank uses GPU 0 (needs --backend gloo)')
    ap.add_argument('--write-probe', type=int, default=4,
                    help='the envelope buffer is the best of this many allocations by the time of a memset over each '
                         '(hipdsp_malloc_probed, what BufferedData does for its device mirrors; 1: a plain allocation)')
    ap.add_argument('--no-fuse', action='store_true',
                    help='four launches (band-pass, spectrogram, envelope forward, backward) instead '
                         'of fusing the envelope forward pass into the band-pass kernel')
    ap.add_argument('--no-overlap', action='store_true',
                    help='spectrogram on the main stream instead of a second stream next to the '
                         'envelope backward sweep')
    ap.add_argument('--no-fuse-spectrogram', action='store_true',
                    help='by default band-pass + envelope state sweep + spectrogram run as ONE launch '
                         '(hipdsp_chain_forward: FFT waves take the filtered tiles from LDS, 20 instead of '
                         '24 B/sample for the step) whenever the library covers the shape; this flag keeps the '
                         'separate launches')
    ap.add_argument('--force-dist', action='store_true',
                    help='rehearsal: take the multi-rank code path even with one rank')
    ap.add_argument('--rendezvous-only', action='store_true',
                    help='launcher check without a GPU: every rank joins a gloo process group, one all-reduce, rank 0 '
                         'prints a JSON line with the world size it saw -- tests/test_bench_contract.py uses it to '
                         'cover the self-launch of --gpus N on the CPU')
    ap.add_argument('--rehearse-legs', action='store_true',
                    help='the multi-rank bookkeeping of the N > 1 line without a GPU (gloo, host tensors): shards, tile '
                         'geometry, the all-gathers into the merged tiles, max over ranks, the legs record; no kernel runs')
    ap.add_argument('--no-facade', action='store_true',
                    help='N=1: skip the leg (outside the timed region) that runs the same slab through the drop-in '
                         'surface -- ArrayLoader -> BufferedFilter.update() -> recompute_all() -- and reports '
                         'facade_ms_per_step next to ms_per_step')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    for key in ('channels', 'seconds', 'rate', 'nfft', 'hop', 'order', 'env'):
        if getattr(args, key) is None:
            setattr(args, key, cfg[key])
    args.config_name = cfg['name']
    return args


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota)/int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(args, sos, esos):
    """The reference's own CPU path (scipy call pattern, float64, one thread) on a
    bounded sample of the same workload; falls back to the C/NumPy oracle port."""
    C, T = args.channels, int(min(args.cpu_sample_seconds, args.seconds)*args.rate)
    rng = np.random.default_rng(1234 + 2)
    t = np.arange(T)/args.rate
    x = rng.uniform(-1.0, 1.0, size=(T, C))
    for c in range(C):
        x[:, c] = 0.5*x[:, c] + 0.5*np.sin(2*np.pi*1000.0*(1 + c/C)*t)
    try:
        from oracle import scipy_path as path
        impl = 'scipy %s call pattern of the reference (oracle/scipy_path.py)' % \
            __import__('scipy').__version__
        stamps = {}
        run = lambda: path.chain(x, args.rate, sos, esos, args.nfft, args.hop, stamps=stamps)
    except ImportError:
        stamps = {}
        from oracle import oracle as path
        impl = 'C/NumPy oracle port (oracle/oracle.py); scipy not installed'

        def run():
            filt = np.zeros_like(x)
            path.filter_process(sos, x, filt, 0)
            spec = np.zeros(((T + args.hop - 1)//args.hop, C, args.nfft//2 + 1))
            path.spectrogram_process(filt, spec, args.rate, args.nfft, args.hop)
            env = np.zeros_like(x)
            path.envelope_process(esos, filt, env, 0)
    t0 = time.perf_counter()
    run()
    dt = time.perf_counter() - t0
    # "reference": scipy.signal itself, called exactly as the reference's process() bodies call it (the
    # reference is pure Python over scipy; its package cannot be imported here -- PyQt -- so the three
    # call sites are restated in oracle/scipy_path.py); "port": this repo's C/NumPy oracle instead
    out = {'value': C*T/dt/1e6, 'unit': 'Msamples/s', 'cores': 1,
           'kind': 'reference' if impl.startswith('scipy') else 'port',
           'host_cores': os.cpu_count(), 'usable_cores': usable_cores(),
           'sample': f'{C} ch x {T/args.rate:g} s x {args.rate/1000:g} kHz float64, '
                     f'same chain, {dt:.1f} s wall; {impl}'}
    if stamps:
        # per process() body, so that the figure explains itself: SURVEY 6's indicative 2.2 Msamples/s for this very call
        # pattern came from the survey container (8 virtual cores of a 2.1 GHz Xeon: band-pass 39, spectrogram 12,
        # envelope 10 Msamples/s when timed alone, 2.1 for the chain -- its transposes and the first touch of 0.5 GB of
        # float64 temporaries are memory-bound there); the GPU boxes' host cores run the same scipy calls ~9x faster
        out['stages_Msamples_per_s'] = {k: round(C*T/v/1e6, 1) for k, v in stamps.items() if v > 0}
    # for fairness also an all-cores figure: channels split over worker processes, one per channel up to the
    # cores this process may use (affinity mask and cgroup quota), in a separate process tree that never
    # touches the GPU
    try:
        import subprocess
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'scipy_path.py'), str(C),
                            str(T/args.rate), str(args.rate), str(args.nfft), str(args.hop),
                            str(args.hp), str(args.lp), str(args.order), str(args.env), str(usable_cores())],
                           capture_output=True, text=True, timeout=180)
        if r.returncode == 0:
            allc = json.loads(r.stdout.strip().split('\n')[-1])
            out['all_cores'] = {'value': allc['value'], 'unit': 'Msamples/s', 'cores': allc['cores'],
                                'sample': 'same sample, channels split over worker processes'}
    except Exception:
        pass
    return out


def facade_leg(args, hipdsp, ctx, dx, df, ds, de, C, T, nd, F):
    """The same slab through the plug-in surface the browser sees (DataBrowser.update_filter ->
    BufferedFilter.update() -> recompute_all(), databrowser.py:1264-1288, buffereddata.py:149-153): a host
    recording behind an ArrayLoader, BufferedFilter -> {BufferedSpectrogram, BufferedEnvelope} opened on it like
    audian's Data model opens them, then K cut-off updates.  The raw slab is uploaded once (first update) and its
    device copy is reused, as in an interactive cut-off sweep; everything downstream stays in HBM.  Reports the
    wall-clock per update() including all Python bookkeeping, which launches an update() turned into, and whether
    the results equal the direct C-ABI path's bit for bit on sampled windows."""
    from audian_amd.bufferedfilter import BufferedFilter
    from audian_amd.bufferedenvelope import BufferedEnvelope
    from audian_amd.bufferedspectrogram import BufferedSpectrogram
    from audian_amd.tracegraph import TraceGraph
    t_setup = time.perf_counter()
    # the recording as the loader holds it: (frames, channels) on the host, float32 like the file's samples
    host = np.empty((T, C), dtype=np.float32)
    chunk = 1 << 20
    tmp = hipdsp.DeviceArray(ctx, (chunk, C), np.float64)
    for a in range(0, T, chunk):
        n = min(chunk, T - a)
        hipdsp.unpack(ctx, dx.view(a, (1,)), T, tmp, n, C)
        host[a:a + n] = tmp.to_host().reshape(-1)[:n*C].reshape(n, C)
    tmp.free()
    hipdsp._default_ctx = ctx                         # the traces compute on the bench's context

    class Shown:
        def isVisible(self):
            return True

        def setVisible(self, show):
            pass
    g = TraceGraph(args.seconds, 0.0)                 # the whole recording is resident, as in the timed region
    filt = BufferedFilter()
    spec = BufferedSpectrogram(nfft=args.nfft, overlap_frac=1.0 - args.hop/args.nfft)
    env = BufferedEnvelope(envelope_cutoff=args.env)
    for t in (filt, spec, env):
        g.add_trace(t)
    g.setup_traces()
    g.open(host, args.rate, view=True)
    for t in g.traces:
        t.plot_items = [Shown() for _ in range(t.channels)]
    g.set_need_update()
    g.update_times(0.0, args.seconds)
    filt.filter_order = args.order
    filt.highpass_cutoff, filt.lowpass_cutoff = args.hp, args.lp
    filt.update()                                     # uploads the raw slab, first recompute
    ctx.synchronize()
    setup_s = time.perf_counter() - t_setup
    before = dict(hipdsp.launches)
    filt.update()
    per_update = {k: v - before.get(k, 0) for k, v in hipdsp.launches.items() if v != before.get(k, 0)}
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        filt.update()
    ctx.synchronize()
    ms = (time.perf_counter() - t0)/args.steps*1e3
    # same numbers as the direct path?  (sampled windows; the spectrogram's last frame is zero here because
    # load_buffer hands it one sample "after" only, buffereddata.py:99 -- the reference's behaviour)
    rng = np.random.default_rng(5)
    same = True
    nsp = len(spec._hostbuf)               # (floor(T / hop) frames: align_buffer rounds down, buffereddata.py:88)
    if len(filt._hostbuf) == T and len(env._hostbuf) == T and 8 < nsp <= nd:
        for _ in range(6):
            c = int(rng.integers(0, C))
            off = int(rng.integers(0, max(1, T - 100000)))
            n = min(100000, T - off)
            same &= np.array_equal(filt._dev.view(c*T + off, (n,)).to_host(), df.view(c*T + off, (n,)).to_host())
            same &= np.array_equal(env._dev.view(c*T + off, (n,)).to_host(), de.view(c*T + off, (n,)).to_host())
            k = int(rng.integers(0, max(1, nsp - 40)))
            m = min(32, nsp - 4 - k)
            if m > 0:
                same &= np.array_equal(spec._dev.view((c*nsp + k)*F, (m*F,)).to_host(),
                                       ds.view((c*nd + k)*F, (m*F,)).to_host())
    else:
        same = None
    out = {'facade_ms_per_step': round(ms, 4), 'launches_per_update': per_update,
           'equals_direct_path_on_sampled_windows': bool(same) if same is not None else None,
           'setup_s': round(setup_s, 2),
           'what': 'ArrayLoader(host float32 recording) -> BufferedFilter.update() -> recompute_all() over '
                   '{BufferedSpectrogram, BufferedEnvelope}; wall clock per update() incl. Python bookkeeping, '
                   'raw slab resident on the device after the first update'}
    for t in (filt, spec, env):
        if t._dev is not None:
            t._dev.free()
    filt._raw_cache = None
    return out


def parity_subset(args, hipdsp, ctx, dx, df, ds, de, T, nd, sos, esos, extra=()):
    """HIP outputs vs the CPU oracle (max|a-b|/max|b| per channel, per frame for the PSD): the first 2 s
    of channels {0, C/2, C-1}, plus the windows in `extra` = [(channel, first frame)] deep
    inside the run -- around an internal segment border of the fused plan and at the end of the last
    channel -- for which the oracle starts early enough to have forgotten its zero initial state."""
    from oracle import oracle
    C, F = args.channels, args.nfft//2 + 1
    n_cmp = min(T, int(2*args.rate))
    lead_f, lead_e = 20000, 80000            # samples after which the band-pass / the 20 Hz envelope have forgotten
    worst = 0.0
    windows = [(c, 0) for c in sorted({0, C//2, C - 1})] + list(extra)
    for c, off in windows:
        off = max(0, min(int(off), T - n_cmp))
        a0 = max(0, off - lead_f - lead_e)                     # first sample handed to the oracle
        a1 = min(T, off + n_cmp + lead_e)                      # the backward pass needs a look-ahead
        x = dx.view(c*T + a0, (a1 - a0,)).to_host().astype(np.float64)[:, None]
        filt = np.zeros_like(x)
        oracle.filter_process(sos, x, filt, 0)
        o = off - a0
        g = df.view(c*T + off, (n_cmp,)).to_host()
        worst = max(worst, np.max(np.abs(g - filt[o:o + n_cmp, 0]))/np.max(np.abs(filt[o:o + n_cmp, 0])))
        # the envelope of the oracle's filtered trace, from where that one has converged
        e0 = 0 if a0 == 0 else lead_f
        env = np.zeros_like(filt[e0:])
        oracle.envelope_process(esos, filt[e0:], env, 0)
        g = de.view(c*T + off, (n_cmp,)).to_host()
        want = env[o - e0:o - e0 + n_cmp, 0]
        worst = max(worst, np.max(np.abs(g - want))/np.max(np.abs(want)))
        k0 = (off + args.hop - 1)//args.hop
        nfr = min(nd - k0, (off + n_cmp - k0*args.hop - args.nfft)//args.hop)
        if nfr > 0:
            spec = np.zeros((nfr, 1, F))
            s0 = k0*args.hop - a0
            oracle.spectrogram_process(filt[s0:s0 + (nfr - 1)*args.hop + args.nfft], spec, args.rate,
                                       args.nfft, args.hop)
            g = ds.view((c*nd + k0)*F, (nfr, F)).to_host()
            for k in range(nfr):
                worst = max(worst, np.max(np.abs(g[k] - spec[k, 0]))/np.max(np.abs(spec[k, 0])))
    return float(worst)


class TorchGather:
    """All-gather of the spectrogram tile with torch.distributed (RCCL at backend 'nccl').  The tile is copied out
    of the rank's spectrogram ON THE COMPUTE STREAM right behind the forward sweep (0.03 ms for the visible tile) into
    one of two buffers; the gather itself is enqueued on a side stream and waits for the step's BACKWARD sweep: it
    then runs under the forward sweep of the next step, which plans no workgroup for `reserve` CUs.  The backward
    sweep runs alone on the chip.  (Round 2 issued copy and gather on a side stream right behind the forward sweep and
    lost 0.7-1.4 ms per step to it with one rank: the backward sweep was a kernel of persistent SINGLE-WAVE workgroups,
    and any small kernel in front of its launch left the dispatcher's round-robin such that some SIMDs got three of
    its waves and others one -- 2.96 -> 3.8-4.5 ms at 32 channels, profiles/r03_forcedist_trace_before.txt.  Its
    workgroups are four waves now, one per SIMD of a CU.)"""

    def __init__(self, torch, dist, hipdsp, ctx, tspec, tile_frames, world, backend, cstream):
        self.torch, self.dist, self.hipdsp, self.ctx, self.tspec, self.tf = torch, dist, hipdsp, ctx, tspec, tile_frames
        self.world, self.backend, self.cstream = world, backend, cstream
        C, nd, F = tspec.shape
        self.C, self.nd, self.F = C, nd, F
        self.whole = tile_frames == nd
        self.side = torch.cuda.Stream()
        self.tile = [None, None] if self.whole else \
            [torch.empty((C, tile_frames, F), dtype=torch.float32, device='cuda') for _ in range(2)]
        self.merged = torch.empty((world*C, tile_frames, F), dtype=torch.float32,
                                  device='cuda' if backend == 'nccl' else 'cpu')
        self.n = 0
        self.src = None
        self.gathered = [torch.cuda.Event(), torch.cuda.Event()]     # buffer b may be overwritten
        for ev in self.gathered:
            ev.record(cstream)
        self.computed = torch.cuda.Event()

    def before_forward(self):
        """The forward sweep is about to overwrite the spectrogram: a gather that reads it in place (whole
        spectrogram) must have finished; a tile has been copied out on this very stream."""
        if self.whole:
            self.cstream.wait_event(self.gathered[0])

    def after_forward(self):
        """Right behind the forward sweep, on the compute stream: copy the tile out."""
        if self.whole:
            self.src = self.tspec
            return
        b = self.n % 2
        self.cstream.wait_event(self.gathered[b])          # the gather of two steps ago has read this buffer
        self.src = self.tile[b]
        self.hipdsp.memcpy2d(self.ctx, self.src.data_ptr(), 4*self.tf*self.F, self.tspec.data_ptr(), 4*self.nd*self.F,
                             4*self.tf*self.F, self.C)

    def issue(self):
        """After the backward sweep has been enqueued: the gather follows it on the side stream."""
        torch = self.torch
        b = 0 if self.whole else self.n % 2
        self.n += 1
        self.computed.record(self.cstream)
        self.side.wait_event(self.computed)
        with torch.cuda.stream(self.side):
            if self.backend == 'nccl':
                self.dist.all_gather_into_tensor(self.merged, self.src)      # (the side stream waits for it)
            else:
                self.side.synchronize()
                self.dist.all_gather_into_tensor(self.merged, self.src.cpu())
            self.gathered[b].record(self.side)

    def drain(self):
        self.side.synchronize()

    def alone(self, k):
        """k gathers of the (already copied) tile with nothing else on the device: seconds per gather."""
        torch = self.torch
        src = self.tspec if self.whole else self.tile[0]
        torch.cuda.synchronize()
        self.dist.barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            if self.backend == 'nccl':
                self.dist.all_gather_into_tensor(self.merged, src)
            else:
                self.dist.all_gather_into_tensor(self.merged, src.cpu())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0)/k


class AbiGather:
    """The same exchange step through the C ABI (hipdsp_comm_* / hipdsp_allgather_f32): the tile is copied on the
    compute context (hipdsp_memcpy2d_d2d) behind the forward sweep, a second context on its own stream enqueues
    ncclAllGather behind the backward sweep; ordered with hipdsp events.  The 128-byte unique id travels over
    torch.distributed."""

    def __init__(self, torch, dist, hipdsp, ctx, ds, shape, tile_frames, world, rank, local_rank):
        self.torch, self.dist, self.hipdsp, self.ctx, self.ds, self.tf = torch, dist, hipdsp, ctx, ds, tile_frames
        C, nd, F = shape
        self.C, self.nd, self.F, self.world = C, nd, F, world
        self.whole = tile_frames == nd
        self.gctx = hipdsp.Context(local_rank, ctx.create_stream())
        ids = [hipdsp.Comm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        self.comm = hipdsp.Comm(self.gctx, ids[0], rank, world)
        self.tile = [None, None] if self.whole else \
            [hipdsp.DeviceArray(self.gctx, (C, tile_frames, F), np.float32) for _ in range(2)]
        self.merged = hipdsp.DeviceArray(self.gctx, (world*C, tile_frames, F), np.float32)
        self.ev_computed = ctx.event()
        self.ev_gathered = [ctx.event(), ctx.event()]
        self.n = 0
        self.src = None
        for ev in self.ev_gathered:
            self.gctx.record(ev)

    def before_forward(self):
        if self.whole:
            self.ctx.wait_event(self.ev_gathered[0])

    def after_forward(self):
        h = self.hipdsp
        if self.whole:
            self.src = self.ds
            return
        b = self.n % 2
        self.ctx.wait_event(self.ev_gathered[b])
        self.src = self.tile[b]
        h.memcpy2d(self.ctx, self.src, 4*self.tf*self.F, self.ds, 4*self.nd*self.F, 4*self.tf*self.F, self.C)

    def issue(self):
        b = 0 if self.whole else self.n % 2
        self.n += 1
        self.ctx.record(self.ev_computed)
        self.gctx.wait_event(self.ev_computed)
        self.comm.allgather(self.src, self.merged, self.C*self.tf*self.F)
        self.gctx.record(self.ev_gathered[b])

    def drain(self):
        self.gctx.synchronize()

    def alone(self, k):
        src = self.ds if self.whole else self.tile[0]
        self.ctx.synchronize()
        self.gctx.synchronize()
        self.dist.barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            self.comm.allgather(src, self.merged, self.C*self.tf*self.F)
        self.gctx.synchronize()
        return (time.perf_counter() - t0)/k

    def close(self):
        self.gctx.synchronize()
        self.comm.close()
        for a in list(self.tile) + [self.merged]:
            if a is not None:
                a.free()
        self.gctx.pool_trim()


def tile_leg_record(name, seconds_of_recording, gb, world, steps, gdt, compute_dt, alone_s):
    """One entry of legs['tiles']: the K steps with this tile gathered (gdt seconds, max over ranks), without any gather
    (compute_dt) and the gather alone on the device (alone_s per gather); `gb` = GB of the tile per rank."""
    return {'seconds_of_recording': seconds_of_recording,
            'GB_per_rank': round(gb, 3), 'GB_received_per_rank': round(gb*(world - 1), 3),
            'step_ms': round(gdt/steps*1e3, 4),                          # compute + gather, overlapped
            'gather_ms': round(alone_s*1e3, 4),                          # the gather alone on the device
            'gather_exposed_ms': round((gdt - compute_dt)/steps*1e3, 4),
            'gather_GBps_per_rank_in': round(gb*(world - 1)/alone_s, 1) if world > 1 and alone_s > 0 else None}


def rehearse_legs(args):
    """--rehearse-legs: the N-rank BOOKKEEPING of the multi-GPU line without a GPU (gloo, host tensors) -- channel shards
    (audian_amd.dist.shard_channels), tile geometry (tile_frames), the all-gather of each tile into the merged
    (world x channels, frames', F) tensor with every rank's block checked where it must land, max-over-ranks times,
    the legs' record (tile_leg_record, shared with the timed path) and rank 0's ONE JSON line.  No kernel runs: `value`
    is null and the line says so.  tests/test_bench_contract.py runs it with eight ranks -- the driver's N = 8 -- which
    no GPU box of a round may (at most six processes on the card)."""
    import torch
    import torch.distributed as dist
    from audian_amd.dist import shard_channels, tile_frames as n_tile_frames
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29535')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    C, T = args.channels, int(round(args.seconds*args.rate))
    F, nd = args.nfft//2 + 1, (T + args.hop - 1)//args.hop
    c0, c1 = shard_channels(world*C, rank, world)
    assert (c0, c1) == (rank*C, rank*C + C), (c0, c1)            # equal shards: what synth()'s c0 / c_total assume
    tiles = {}
    for name in ('visible', 'window', 'full'):
        tf = nd if TILES[name] is None else n_tile_frames(nd, args.rate, args.hop, TILES[name])
        # every value names its rank, channel and frame: a block that lands in the wrong place is seen
        ch = torch.arange(c0, c1, dtype=torch.float32).view(C, 1, 1)
        fr = torch.arange(tf, dtype=torch.float32).view(1, tf, 1)
        local = (ch*4096.0 + fr).expand(C, tf, F).contiguous()
        merged = torch.empty((world*C, tf, F), dtype=torch.float32)
        dist.barrier()
        t0 = time.perf_counter()
        k = 2
        for _ in range(k):
            dist.all_gather_into_tensor(merged, local)
        alone = torch.tensor([(time.perf_counter() - t0)/k], dtype=torch.float64)
        dist.all_reduce(alone, op=dist.ReduceOp.MAX)
        want = (torch.arange(world*C, dtype=torch.float32).view(-1, 1)*4096.0 + torch.arange(tf, dtype=torch.float32).view(1, -1))
        ok = torch.tensor([1.0 if torch.equal(merged[:, :, 0], want) and torch.equal(merged[:, :, F - 1], want) else 0.0])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        gb = 4.0*C*tf*F/1e9
        rec = tile_leg_record(name, TILES[name] if TILES[name] is not None else args.seconds, gb, world, args.steps,
                              float(alone.item())*args.steps, 0.0, float(alone.item()))
        rec['frames'] = tf
        rec['merged_ok_on_every_rank'] = bool(ok.item() == 1.0)
        tiles[name] = rec
    if rank == 0:
        print(json.dumps({
            'metric': 'Msamples/s spectrogram+bandpass, 64ch x 96kHz', 'value': None, 'unit': 'Msamples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True,
            'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32 I/O, f64 IIR state', 'data': 'synthetic',
            'config': {'workload': f'{args.config_name}: {C} ch/GPU x {args.seconds:g} s x {args.rate/1000:g} kHz float32',
                       'channels_per_gpu': C, 'frames': T, 'spectrogram_frames': nd,
                       'parallelism': f'channel shard x{world}, all-gather (gloo, host tensors) of the spectrogram tiles'},
            'legs': {'tile_in_timed_region': args.tile, 'tiles': tiles},
            'invalid': 'rehearsal of the multi-rank bookkeeping on the CPU: no kernel ran'}), flush=True)
    dist.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves.  This process has not
    imported torch nor touched HIP (a process that has initialised the GPU must not be replaced or forked into
    ranks); it starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD --
    fresh processes, one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by the launcher, rendezvous on
    127.0.0.1 at a port that is free now -- forwards rank 0's single JSON line to its own stdout and exits with the
    children's worst return code (no line counts as a failure)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')         # dmabuf IPC: the only kind this pool's driver has
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench.py: --gpus %d without a launcher: starting %d ranks (%s)' % (args.gpus, args.gpus, ' '.join(cmd[1:8])),
          file=sys.stderr)
    sys.stderr.flush()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)        # (stderr goes where ours goes)
    out = proc.communicate()[0].decode(errors='replace')
    rc = proc.returncode
    lines = [ln for ln in out.splitlines() if ln.startswith('{') and ln.rstrip().endswith('}')]
    for ln in out.splitlines():
        if ln not in lines[-1:]:
            print(ln, file=sys.stderr)                       # whatever else a rank wrote to stdout
    if lines:
        sys.stdout.write(lines[-1] + '\n')
        sys.stdout.flush()
    elif rc == 0:
        rc = 1
    # a signal-killed launcher has a negative return code; the driver wants non-zero, not a wrapped value
    sys.exit(rc if 0 <= rc < 256 else 1)


def rendezvous_only(args):
    """--rendezvous-only: the launcher plumbing without a GPU (gloo)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29534')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    if os.environ.get('BENCH_RENDEZVOUS_FAIL_RANK') == os.environ['RANK']:
        sys.exit(3)                                          # test hook: a rank that dies before the rendezvous
    dist.init_process_group('gloo')
    t = torch.ones(1, dtype=torch.float64)*(dist.get_rank() + 1)
    dist.all_reduce(t)
    dist.barrier()
    if dist.get_rank() == 0:
        print(json.dumps({'rendezvous_only': True, 'n_gpus': world, 'gpus_flag': args.gpus,
                          'rank_sum': float(t.item())}), flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args)                                    # does not return
    if args.rendezvous_only:
        return rendezvous_only(args)
    if args.rehearse_legs:
        return rehearse_legs(args)
    # stdout carries exactly ONE line, the JSON result: whatever libraries print on the way (RCCL
    # writes a version banner to stdout when its communicator comes up) goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    # enough hardware queues that the compute, spectrogram and RCCL streams do not share one
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    args.gpus = world        # under a launcher its WORLD_SIZE is authoritative (without one, self_launch() above)

    dist = None
    torch = None
    multi = world > 1 or args.force_dist
    if multi:
        # The fused forward sweep plans no workgroup for `reserve` CUs (its 1024-thread workgroups take a whole CU
        # each): RCCL's resident all-gather kernel must fit into them, or workgroups of the sweep queue for a second
        # round and the sweep takes twice as long.  RCCL launches one workgroup per channel, so its channel count is
        # capped at the number of reserved CUs (NCCL_MAX_NCHANNELS; an explicit setting in the environment wins).
        reserve_for_rccl = args.reserve_cus if args.reserve_cus is not None else 8
        if reserve_for_rccl > 0:
            os.environ.setdefault('NCCL_MAX_NCHANNELS', str(reserve_for_rccl))
        import torch            # before libhip_dsp: one HIP runtime per process (_lib.py)
        import torch.distributed as dist
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos

    if multi:
        from audian_amd.dist import tile_frames as n_tile_frames
        if args.same_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if world == 1:                        # --force-dist without a launcher
            for key, val in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', '29533'), ('RANK', '0'),
                             ('WORLD_SIZE', '1')):
                os.environ.setdefault(key, val)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo')
        # compute on a non-default stream so that the RCCL gather (its own stream) can
        # overlap the kernels; the legacy default stream would serialise them
        cstream = torch.cuda.Stream()
        torch.cuda.set_stream(cstream)
        stream = cstream.cuda_stream
    else:
        cstream = None
        stream = None
    ctx = hipdsp.Context(local_rank, stream)
    if args.max_segments:
        ctx.set_max_segments(args.max_segments)
    reserve = args.reserve_cus if args.reserve_cus is not None else (8 if multi else 0)
    if reserve:
        ctx.set_option('chain_reserve_cus', reserve)

    total_channels = None
    if args.scaling == 'strong':
        share = args.strong_world or world
        total_channels = args.channels
        if total_channels % share:
            sys.exit(f'bench.py --scaling strong: {total_channels} channels do not divide over {share} GPUs')
        args.channels = total_channels//share
    C, T = args.channels, int(round(args.seconds*args.rate))
    F = args.nfft//2 + 1
    nd = (T + args.hop - 1)//args.hop                 # BufferedData.update_step frames
    sos = butter_sos(args.order, (args.hp, args.lp), 'bandpass', args.rate)
    esos = butter_sos(2, args.env, 'lowpass', args.rate)
    plan, eplan = hipdsp.SosPlan(ctx, sos), hipdsp.SosPlan(ctx, esos)
    warm_f, _ = plan.info()
    warm_e, edge = eplan.info()

    dx = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    df = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    # (the envelope is the one output whose sweep is bound by its write stream: where its buffer lies in HBM moves that
    # sweep by up to 12 %; the facade allocates its mirrors the same way, audian_amd/buffereddata.py: WRITE_PROBE)
    de = hipdsp.DeviceArray(ctx, (C, T), np.float32, write_probe=args.write_probe)
    if multi:
        tspec = torch.empty((C, nd, F), dtype=torch.float32, device='cuda')
        ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32, ptr=tspec.data_ptr(), owner=tspec)
    else:
        tspec = None
        ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
    ctx.reserve(8*C*((T + edge + 2047)//2048 + 1)*2*len(esos))  # envelope state checkpoints (+ the state a channel ends with)
    hipdsp.synth(ctx, dx, T, C, T, args.rate, 1234 + args.config, c0=rank*C, c_total=world*C)
    ctx.synchronize()

    # measured device-copy ceiling (read + write of one trace), reported next to the 8 TB/s spec peak as SURVEY 8d
    # asks; untimed, before the steps: the float4-per-thread copy that reaches the part's streaming ceiling
    # (hipdsp_copy_probe) and, for comparison with earlier rounds' lines, hipMemcpy D2D
    ca, cb, cc = ctx.event(), ctx.event(), ctx.event()
    nbytes = 4*C*T//16*16
    hipdsp.check(hipdsp.lib.hipdsp_copy_probe(ctx.handle, hipdsp._p(de), hipdsp._p(dx), nbytes))
    ctx.record(ca)
    for _ in range(3):
        hipdsp.check(hipdsp.lib.hipdsp_copy_probe(ctx.handle, hipdsp._p(de), hipdsp._p(dx), nbytes))
    ctx.record(cb)
    for _ in range(3):
        hipdsp.lib.hipdsp_memcpy_d2d(ctx.handle, hipdsp._p(de), hipdsp._p(dx), 4*C*T)
    ctx.record(cc)
    ctx.synchronize()
    copy_gbps = 3*2.0*nbytes/(ctx.elapsed_ms(ca, cb)*1e-3)/1e9
    memcpy_gbps = 3*8.0*C*T/(ctx.elapsed_ms(cb, cc)*1e-3)/1e9

    # The fused forward sweep (band-pass + envelope states + spectrogram in one launch) whenever the library
    # covers the shape: one untimed trial decides.
    fuse3 = not args.no_fuse_spectrogram and not args.no_fuse
    if fuse3:
        try:
            hipdsp.chain_forward(ctx, plan, eplan, dx, T, df, T, C, T, args.nfft, args.hop, args.rate, ds, nd,
                                 rectify=True, gain=np.pi/2)
            ctx.synchronize()
        except NotImplementedError as err:
            # HIPDSP_ERR_UNSUPPORTED only (a shape or plan the fused sweep does not cover).  Anything else --
            # a HIP error, a fault reported by the kernel -- ends the run: a number measured on a device
            # whose headline kernel has just failed would mask the failure.
            print(f'bench.py: fused forward sweep not used ({err}); separate launches',
                  file=sys.stderr)
            fuse3 = False
    # The spectrogram and the envelope backward sweep both only read the filtered trace: with separate
    # launches they run next to each other on two streams, ordered by events.
    overlap = not args.no_overlap and not fuse3 and not multi
    sctx = ctx
    if overlap:
        ctx.set_stream(ctx.create_stream())
        sctx = hipdsp.Context(local_rank, ctx.create_stream())
    ev_filtered, ev_spec = ctx.event(), ctx.event()
    if multi:
        # The IIR sweeps launch one wave per (channel, segment) and want all of them resident at once: at most
        # 12 waves per CU leave a wave slot per SIMD free for whatever else is resident (round 3: the planner
        # picks 8 anyway, and the gather no longer runs next to the backward sweep)
        ctx.set_option('sos_waves_per_cu', 12)

    def tile_frames_of(name):
        secs = TILES[name] if not (name == args.tile and args.tile_seconds) else args.tile_seconds
        return nd if secs is None else n_tile_frames(nd, args.rate, args.hop, secs)

    def make_gatherer(name):
        tf = tile_frames_of(name)
        if args.gather == 'c-abi' and args.backend == 'nccl':
            return AbiGather(torch, dist, hipdsp, ctx, ds, (C, nd, F), tf, world, rank, local_rank)
        return TorchGather(torch, dist, hipdsp, ctx, tspec, tf, world, args.backend, cstream)

    n_ev = 7
    events = [[ctx.event() for _ in range(n_ev)] for _ in range(args.steps)]
    mids = [ctx.event() for _ in range(args.steps)]
    fused = not args.no_fuse

    def step(i, gatherer):
        ev = events[i] if i >= 0 else None
        if gatherer is not None:
            gatherer.before_forward()
        if ev:
            ctx.record(ev[0])
        if fuse3:
            hipdsp.chain_forward(ctx, plan, eplan, dx, T, df, T, C, T, args.nfft, args.hop, args.rate, ds, nd,
                                 rectify=True, gain=np.pi/2)
        elif fused:
            # band-pass + envelope forward sweep in one pass over x: writes the filtered trace and
            # the envelope state entering every 2048-sample tile; the backward sweep (which
            # recomputes the forward output tile by tile) follows after the spectrogram
            hipdsp.sosfilt_envelope(ctx, plan, eplan, dx, T, df, T, de, T, C, T, rectify=True,
                                    gain=np.pi/2, clamp=True, phase=1)
        else:
            hipdsp.sosfilt(ctx, plan, dx, T, df, T, C, T, 0)
        if ev:
            ctx.record(ev[1])
        if overlap:
            ctx.record(ev_filtered)
            sctx.wait_event(ev_filtered)
        if ev:
            sctx.record(ev[5])
        if not fuse3:
            hipdsp.spectrogram(sctx, df, T, C, T, args.nfft, args.hop, args.rate, ds, nd)
        if ev:
            sctx.record(ev[6])
        if gatherer is not None:
            # the tile of the merged spectrogram leaves the rank's spectrogram right behind the forward sweep
            gatherer.after_forward()
        if ev and fused:
            sctx.record(ev[2])
        if overlap:
            sctx.record(ev_spec)
        if ev:
            ctx.set_mid_event(mids[i])
            if not fused:
                ctx.record(ev[2])
        if fused:
            if ev:
                ctx.record(mids[i])
            ctx.set_mid_event(None)
            hipdsp.sosfilt_envelope(ctx, plan, eplan, dx, T, df, T, de, T, C, T, rectify=True,
                                    gain=np.pi/2, clamp=True, phase=2)
        else:
            hipdsp.envelope(ctx, eplan, df, T, de, T, C, T, 0, rectify=True, gain=np.pi/2, clamp=True)
        if ev:
            ctx.set_mid_event(None)
            ctx.record(ev[3])
        if gatherer is not None:
            # merged spectrogram tile on every rank: one RCCL all-gather over xGMI per step, enqueued behind the
            # backward sweep so that it runs under the forward sweep of the next step (which leaves `reserve` CUs
            # to RCCL's kernel) and the backward sweep's persistent waves are dispatched onto an idle chip
            gatherer.issue()
        if overlap:
            ctx.wait_event(ev_spec)       # the next step overwrites the filtered trace
        if ev:
            ctx.record(ev[4])

    def fence(gatherer):
        if gatherer is not None:
            gatherer.drain()             # every gather issued so far is part of the job
        sctx.synchronize()
        ctx.synchronize()
        if multi:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def timed_run(gatherer, record=True):
        """W warm-up steps, then exactly K steps between barriers; seconds (max over ranks)."""
        for _ in range(args.warmup):
            step(-1, gatherer)
        fence(gatherer)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i if record else -1, gatherer)
        fence(gatherer)
        dt = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([dt], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    main_gather = make_gatherer(args.tile) if multi else None
    dt = timed_run(main_gather)

    # per-kernel averages from the HIP events recorded inside the timed region
    if fuse3:
        names = ['chain_fwd<S=%d+%d,filt+env_state+psd>' % (len(sos), len(esos)), 'spectrogram(fused)',
                 'gather_issue', 'env_bwd<S=%d>' % len(esos), 'unused']
    elif fused:
        names = ['sos_ckpt<S=%d+%d,filt+env_state>' % (len(sos), len(esos)), 'spectrogram', 'gather_issue',
                 'env_bwd<S=%d>' % len(esos), 'unused']
    else:
        names = ['sos_scan<S=%d,filt>' % len(sos), 'spectrogram', 'sos_ckpt<S=0+%d,env_state>' % len(esos),
                 'env_bwd<S=%d>' % len(esos), 'unused']
    ms = dict.fromkeys(names, 0.0)
    pair_ms = 0.0
    for i in range(args.steps):
        e = events[i]
        pair_ms += max(ctx.elapsed_ms(e[5], e[6]), ctx.elapsed_ms(e[5], e[3]))/args.steps
        ms[names[0]] += ctx.elapsed_ms(e[0], e[1])
        ms[names[1]] += ctx.elapsed_ms(e[5], e[6])
        ms[names[2]] += ctx.elapsed_ms(e[6], e[2]) if fused else ctx.elapsed_ms(e[2], mids[i])
        ms[names[3]] += ctx.elapsed_ms(mids[i], e[3])
        ms[names[4]] += ctx.elapsed_ms(e[3], e[4])
    for k in ms:
        ms[k] /= args.steps
    ckpt_bytes = 8.0*C*((T + edge + 2047)//2048)*2*len(esos)
    if fuse3:
        alg_bytes = {
            names[0]: 8.0*C*T + ckpt_bytes + 4.0*C*nd*F,     # x read; filtered trace, tile states and PSD written
            names[3]: 8.0*C*T + ckpt_bytes,
        }
    elif fused:
        alg_bytes = {                   # algorithmic HBM bytes per launch
            names[0]: 8.0*C*T + ckpt_bytes,            # x read, filtered trace + tile states written
            names[1]: 4.0*C*T + 4.0*C*nd*F,
            names[3]: 8.0*C*T + ckpt_bytes,            # filtered trace + tile states read, envelope written
        }
    else:
        alg_bytes = {
            names[0]: 8.0*C*T,
            names[1]: 4.0*C*T + 4.0*C*nd*F,
            names[2]: 4.0*C*T + ckpt_bytes,
            names[3]: 8.0*C*T + ckpt_bytes,
        }
    # with the spectrogram on its own stream its event-bracketed time and the envelope sweeps'
    # overlap; the roofline entry is taken from the kernels that run alone
    alone = [names[0]] if (overlap or fuse3) else list(alg_bytes)
    dom = max(alone, key=lambda k: ms[k])
    achieved = alg_bytes[dom]/(ms[dom]*1e-3)/1e9
    # HBM bytes per launch from the rocprofv3 PMC passes of this same command (FETCH_SIZE x 2
    # on gfx950, WRITE_SIZE; tools/summarize_profiles.py) -- only valid for the profiled shape
    traffic = traffic_source = None
    pmc_file = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))
        if pmc.get('shape') == [C, T, args.nfft, args.hop] and dom in pmc.get('kernels', {}):
            traffic = pmc['kernels'][dom]['hbm_bytes']
            traffic_source = 'profiles/' + str(pmc.get('source'))
    kernels = {k: {'ms': round(ms[k], 4),
                   'GBps': round(alg_bytes[k]/(ms[k]*1e-3)/1e9, 1) if k in alg_bytes and ms[k] > 0 else None}
               for k in names if k in alg_bytes}
    if overlap:
        shared = [k for k in alg_bytes if k != names[0]]
        for k in shared:
            kernels[k]['concurrent'] = True      # shares the device with the others marked so
        kernels['||'.join(shared)] = {'ms': round(pair_ms, 4),
                                      'GBps': round(sum(alg_bytes[k] for k in shared)/(pair_ms*1e-3)/1e9, 1)}

    # the fused forward sweep does the work of two BufferedData stages; SURVEY 8d counts those per
    # stage (band-pass 4 R + 4 W, spectrogram 4 R + 4.004 W): reported next to the launch's own bytes
    stage_gbps = None
    if fuse3:
        stage_gbps = round((12.0*C*T + 4.0*C*nd*F + ckpt_bytes)/(ms[names[0]]*1e-3)/1e9, 1)

    # the engine clock the fused sweep actually runs at (shader clocks against the 100 MHz wall clock of one wave
    # that lives through the launch, "chain_debug" bit 16): the chip's power cap, not HBM, sets this kernel's time
    engine_mhz = None
    if fuse3 and rank == 0:
        try:
            if main_gather is not None:
                main_gather.drain()
            ctx.set_option('chain_debug', 16)
            hipdsp.chain_forward(ctx, plan, eplan, dx, T, df, T, C, T, args.nfft, args.hop, args.rate, ds, nd,
                                 rectify=True, gain=np.pi/2)
            ctx.synchronize()
            words = np.zeros(2, dtype=np.int64)
            hipdsp.check(hipdsp.lib.hipdsp_memcpy_d2h(ctx.handle, words.ctypes.data_as(ctypes.c_void_p),
                                                      ctypes.c_void_p(ds.ptr), 16))
            if words[1] > 0:
                engine_mhz = round(float(words[0])/float(words[1])*100.0, 0)
        except Exception:
            engine_mhz = None
        finally:
            ctx.set_option('chain_debug', 0)
            # (the 16 bytes are the first PSD values of channel 0: the launch below writes them again)
            hipdsp.chain_forward(ctx, plan, eplan, dx, T, df, T, C, T, args.nfft, args.hop, args.rate, ds, nd,
                                 rectify=True, gain=np.pi/2)
            ctx.synchronize()

    # N > 1: the same K steps again without any gather and with each tile size, gathers alone as well
    legs = None
    if multi:
        tile_gb = lambda name: 4.0*C*tile_frames_of(name)*F/1e9
        legs = {'tile_in_timed_region': args.tile,
                'step_ms': round(dt/args.steps*1e3, 4), 'tile_GB_per_rank': round(tile_gb(args.tile), 3)}
        if not args.no_legs:
            main_gather.drain()
            compute_dt = timed_run(None, record=False)
            legs['compute_ms'] = round(compute_dt/args.steps*1e3, 4)
            legs['tiles'] = {}
            for name in ('visible', 'window', 'full'):
                if name == 'full' and not (args.full_leg or args.tile == 'full'):
                    legs['tiles'][name] = {'skipped': 'opt-in (--full-leg): %.1f GB per rank, %.1f GB merged on every rank'
                                                      % (tile_gb(name), tile_gb(name)*world)}
                    continue
                g = None
                try:
                    g = main_gather if name == args.tile else make_gatherer(name)
                    gdt = timed_run(g, record=False)          # (without the HIP events of the timed region, like compute_ms)
                    alone_s = g.alone(max(2, min(args.steps, 5)))
                    legs['tiles'][name] = tile_leg_record(name, TILES[name] if TILES[name] is not None else args.seconds,
                                                          tile_gb(name), world, args.steps, gdt, compute_dt, alone_s)
                except Exception as err:      # a leg must never cost the line of the timed region
                    legs['tiles'][name] = {'failed': f'{type(err).__name__}: {err}'[:300]}
                if g is not None and g is not main_gather:
                    g.drain()
                    if hasattr(g, 'close'):
                        g.close()
                    del g
                    torch.cuda.empty_cache()

    parity = None
    cpu = None
    if rank == 0:
        extra = []
        if T > int(6*args.rate):
            # around an internal segment border of the forward sweep's plan, and the end of the last channel
            seg_frames, n_seg = (hipdsp.chain_plan(ctx, plan, eplan, C, T) if fuse3 else (T//2, 2))
            if n_seg > 1:
                extra.append((C//2, (n_seg//2)*seg_frames - int(args.rate)))
            extra.append((C - 1, T - int(2*args.rate)))
        parity = parity_subset(args, hipdsp, ctx, dx, df, ds, de, T, nd, sos, esos, extra)
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, sos, esos)
    # Latency-sized jobs (BASELINE configs[1]: four launches of 5-100 us): the same step captured once into a
    # hipGraph and replayed, as the interactive path does (configs[4], tests/test_gpu_graph.py) -- reported next to
    # the launch-by-launch figure, which stays `value`
    graph_ms = None
    if rank == 0 and world == 1 and not multi and fuse3 and dt/args.steps < 2e-3:
        try:
            gctx = hipdsp.Context(local_rank, ctx.create_stream())
            gctx.reserve(8*C*((T + edge + 2047)//2048 + 1)*2*len(esos))

            def gstep():
                hipdsp.chain_forward(gctx, plan, eplan, dx, T, df, T, C, T, args.nfft, args.hop, args.rate, ds, nd,
                                     rectify=True, gain=np.pi/2)
                hipdsp.sosfilt_envelope(gctx, plan, eplan, dx, T, df, T, de, T, C, T, rectify=True, gain=np.pi/2,
                                        clamp=True, phase=2)
            gstep()
            gctx.synchronize()
            gctx.graph_begin()
            gstep()
            graph = gctx.graph_end()
            for _ in range(args.warmup):
                gctx.graph_launch(graph)
            gctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                gctx.graph_launch(graph)
            gctx.synchronize()
            graph_ms = (time.perf_counter() - t0)/args.steps*1e3
            gctx.graph_destroy(graph)
        except Exception as err:
            graph_ms = f'failed: {type(err).__name__}: {err}'[:200]
    facade = None
    if rank == 0 and world == 1 and not multi and not args.no_facade and fuse3:
        try:
            facade = facade_leg(args, hipdsp, ctx, dx, df, ds, de, C, T, nd, F)
        except Exception as err:                  # a leg must never cost the line of the timed region
            facade = {'failed': f'{type(err).__name__}: {err}'[:300]}

    if rank == 0:
        samples = float(C)*T*world
        line = {
            'metric': 'Msamples/s spectrogram+bandpass, 64ch x 96kHz',
            'value': samples/(dt/args.steps)/1e6,
            'unit': 'Msamples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt/args.steps*1e3,
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f32 I/O, f64 IIR state',
            'data': 'synthetic',
            'config': {
                'workload': (f'{args.config_name}, STRONG scaling: {total_channels} channels in total = ' if total_channels
                             else f'{args.config_name}: synthetic ') + f'{C} ch/GPU x {args.seconds:g} s x '
                            f'{args.rate/1000:g} kHz float32; bandpass {args.hp:g}-{args.lp:g} Hz '
                            f'order {args.order} -> spectrogram nfft {args.nfft} hop {args.hop} '
                            f'+ envelope {args.env:g} Hz',
                'channels_per_gpu': C, 'frames': T, 'spectrogram_frames': nd,
                'parallelism': f'channel shard x{world}' +
                               (f', all-gather ({args.gather}) of the {args.tile} spectrogram tile '
                                f'({4*C*tile_frames_of(args.tile)*F/1e9:.2f} GB per rank) copied out behind the forward sweep, gathered behind the backward sweep (under the next forward sweep), '
                                f'{reserve} CUs left to RCCL' if multi else ''),
                'iir_warmup_samples': {'bandpass': warm_f, 'envelope': warm_e},
                'streams': ('spectrogram on a second stream next to the envelope backward sweep '
                            '(their event-bracketed times overlap)' if overlap else 'one compute stream'),
                'envelope_forward': ('state checkpoints, ' + ('fused into the band-pass kernel' if fused else 'own launch')),
                'envelope_buffer': (f'best of {args.write_probe} allocations by memset time (hipdsp_malloc_probed)'
                                    if args.write_probe > 1 else 'plain allocation'),
                'spectrogram': ('FFT waves inside the forward sweep (filtered tiles from LDS)' if fuse3 else 'own launch'),
            },
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved/HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_source,
                         # (the PMC passes cannot run inside the timed command: `traffic` is the committed figure of the
                         # rocprofv3 --pmc passes of this same command and shape, profiles/pmc_traffic.json)
                         'traffic_measured_in_this_run': False,
                         'algorithmic_bytes': alg_bytes[dom],
                         'device_copy_GBps': round(copy_gbps, 1),         # float4-per-thread copy kernel (hipdsp_copy_probe)
                         'hipMemcpy_d2d_GBps': round(memcpy_gbps, 1),
                         'per_stage_accounting_GBps': stage_gbps,
                         'engine_clock_MHz_in_kernel': engine_mhz},
            'kernels': kernels,
            'chain_algorithmic_GBps': round(sum(alg_bytes.values())/(dt/args.steps)/1e9, 1),
            'parity_max_rel_err': parity,
            'cpu_baseline': cpu,
        }
        if graph_ms is not None:
            line['graph_ms_per_step'] = graph_ms if isinstance(graph_ms, str) else round(graph_ms, 4)
        if facade is not None:
            line['facade'] = facade
            line['facade_ms_per_step'] = facade.get('facade_ms_per_step')
        if legs is not None:
            line['legs'] = legs
            line['compute_ms'] = legs.get('compute_ms')
            line['gather_ms'] = (legs.get('tiles', {}).get(args.tile) or {}).get('gather_ms')
        if parity is None or not parity < 1e-4:
            line['invalid'] = 'parity gate failed'
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + '\n').encode())
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
