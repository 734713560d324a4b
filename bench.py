#!/usr/bin/env python3
"""Benchmark of the BufferedData DSP hot path on MI355X.

One "step" = one pass of the chain  data -> BufferedFilter (Butterworth band-pass)
-> {BufferedSpectrogram (Hann STFT PSD), BufferedEnvelope (rectify + sosfiltfilt)}
over one batch of synthetic multichannel float32 audio that is already resident in
HBM (BASELINE.json configs[2]: 64 ch x 600 s x 96 kHz per GPU, nfft 2048 / hop 1024,
band-pass 300-3000 Hz order 2, envelope low-pass 20 Hz).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line: whole-job Msamples/s, the HBM roofline of the dominant
kernel (HIP-event timed inside the timed region) and, at N=1, the reference's scipy
CPU path timed on this box's host cores on a bounded sample.
"""

import argparse
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--channels', type=int, default=64, help='channels per GPU')
    ap.add_argument('--seconds', type=float, default=600.0)
    ap.add_argument('--rate', type=float, default=96000.0)
    ap.add_argument('--nfft', type=int, default=2048)
    ap.add_argument('--hop', type=int, default=1024)
    ap.add_argument('--hp', type=float, default=300.0)
    ap.add_argument('--lp', type=float, default=3000.0)
    ap.add_argument('--order', type=int, default=2)
    ap.add_argument('--env', type=float, default=20.0)
    ap.add_argument('--tile-seconds', type=float, default=10.0,
                    help='N>1: length of the spectrogram tile that is all-gathered every step: the visible '
                         'window of the browser (10 s by default, plotranges.py:141-144) -- "gather only the '
                         'visible tile when interactive" (SURVEY 7-6).  61 gathers the whole resident buffer '
                         'of the spectrogram trace instead (buffer_time 60 s + 11 s + 10 s of raw pre/post-roll '
                         'minus the 10 s + 10 s trimmed in align_buffer; data.py:17,168, buffereddata.py:75-88), '
                         'which is xGMI-bound: 1.5 GB per rank and step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-seconds', type=float, default=60.0)
    ap.add_argument('--max-segments', type=int, default=0)
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="N>1 collective backend; 'gloo' (host staging) only to rehearse the "
                         "multi-rank path on a one-GPU box")
    ap.add_argument('--same-device', action='store_true',
                    help='rehearsal: every rank uses GPU 0 (needs --backend gloo)')
    ap.add_argument('--no-fuse', action='store_true',
                    help='four launches (band-pass, spectrogram, envelope forward, backward) instead '
                         'of fusing the envelope forward pass into the band-pass kernel')
    ap.add_argument('--no-overlap', action='store_true',
                    help='spectrogram on the main stream instead of a second stream next to the '
                         'envelope backward sweep')
    ap.add_argument('--no-fuse-spectrogram', action='store_true',
                    help='by default band-pass + envelope state sweep + spectrogram run as ONE launch '
                         '(hipdsp_chain_forward: FFT waves take the filtered tiles from LDS, 20 instead of '
                         '24 B/sample for the step) whenever the shape allows it (nfft 2048 / hop 1024, plans of '
                         '<= 2 sections); this flag keeps the separate launches.  At N > 1 the fused kernel, which '
                         'wants a whole CU per workgroup, waits for the previous all-gather to leave the device')
    ap.add_argument('--force-dist', action='store_true',
                    help='rehearsal: take the multi-rank code path even with one rank')
    return ap.parse_args()


def cpu_baseline(args, sos, esos):
    """The reference's own CPU path (scipy call pattern, float64, one thread) on a
    bounded sample of the same workload; falls back to the C/NumPy oracle port."""
    C, T = args.channels, int(args.cpu_sample_seconds*args.rate)
    rng = np.random.default_rng(1234 + 2)
    t = np.arange(T)/args.rate
    x = rng.uniform(-1.0, 1.0, size=(T, C))
    for c in range(C):
        x[:, c] = 0.5*x[:, c] + 0.5*np.sin(2*np.pi*1000.0*(1 + c/C)*t)
    try:
        from oracle import scipy_path as path
        impl = 'scipy %s call pattern of the reference (oracle/scipy_path.py)' % \
            __import__('scipy').__version__
        run = lambda: path.chain(x, args.rate, sos, esos, args.nfft, args.hop)
    except ImportError:
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
           'host_cores': os.cpu_count(),
           'sample': f'{C} ch x {args.cpu_sample_seconds:g} s x {args.rate/1000:g} kHz float64, '
                     f'same chain, {dt:.1f} s wall; {impl}'}
    # for fairness also an all-cores figure: channels split over 16 worker processes (the
    # box's CPU share for one GPU), in a separate process tree that never touches the GPU
    try:
        import subprocess
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'scipy_path.py'), str(C),
                            str(args.cpu_sample_seconds), str(args.rate), str(args.nfft), str(args.hop),
                            str(args.hp), str(args.lp), str(args.order), str(args.env), '16'],
                           capture_output=True, text=True, timeout=180)
        if r.returncode == 0:
            allc = json.loads(r.stdout.strip().split('\n')[-1])
            out['all_cores'] = {'value': allc['value'], 'unit': 'Msamples/s', 'cores': allc['cores'],
                                'sample': 'same sample, channels split over worker processes'}
    except Exception:
        pass
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
        # the envelope of the GPU's own filtered trace where the oracle's has not converged yet
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


def main():
    args = parse()
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
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus N > 1 must be launched with torch.distributed.run '
                     '(one rank per GPU)')
        args.gpus = world

    dist = None
    torch = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch            # before libhip_dsp: one HIP runtime per process (_lib.py)
        import torch.distributed as dist
    from audian_amd import hipdsp
    from audian_amd.design import butter_sos

    if multi:
        from audian_amd.dist import allgather_tiles, tile_frames as n_tile_frames
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
        # overlap the envelope kernels; the legacy default stream would serialise them
        cstream = torch.cuda.Stream()
        torch.cuda.set_stream(cstream)
        stream = cstream.cuda_stream
    else:
        stream = None
    ctx = hipdsp.Context(local_rank, stream)
    if args.max_segments:
        ctx.set_max_segments(args.max_segments)
    # The spectrogram and the envelope backward sweep both only read the filtered trace: they run next to each other on two streams, ordered by events.
    fuse3 = (not args.no_fuse_spectrogram and not args.no_fuse
             and args.nfft == 2048 and args.hop == 1024 and args.order <= 2
             and args.seconds*args.rate >= 8192)
    overlap = not args.no_overlap and not fuse3      # nothing left to run next to the backward sweep
    sctx, sstream = ctx, None
    if overlap:
        if multi:
            sstream = torch.cuda.Stream()
            sctx = hipdsp.Context(local_rank, sstream.cuda_stream)
        else:
            ctx.set_stream(ctx.create_stream())
            sctx = hipdsp.Context(local_rank, ctx.create_stream())
    ev_filtered, ev_spec = ctx.event(), ctx.event()
    if multi:
        # The IIR sweeps launch one wave per (channel, segment) and want all of them resident at
        # once.  Next to RCCL's all-gather kernel a few would have to wait for a second round:
        # 12 waves per CU (instead of 16) leave a wave slot per SIMD free and cost < 2 % alone
        # (tools/coresidency_probe.hip: +9 % instead of +18 % next to a spinning kernel).
        ctx.set_option('sos_waves_per_cu', 12)

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
    de = hipdsp.DeviceArray(ctx, (C, T), np.float32)
    if multi:
        tspec = torch.empty((C, nd, F), dtype=torch.float32, device='cuda')
        ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32, ptr=tspec.data_ptr(), owner=tspec)
        tile_frames = n_tile_frames(nd, args.rate, args.hop, args.tile_seconds)
        # two tiles in flight: the gather of step i runs under the kernels of step i + 1
        tile_buf = [torch.empty((C, tile_frames, F), dtype=torch.float32, device='cuda') for _ in range(2)]
        merged = [torch.empty((world*C, tile_frames, F), dtype=torch.float32,
                              device='cuda' if args.backend == 'nccl' else 'cpu') for _ in range(2)]
        works = [None, None]
        counter = [0]
    else:
        ds = hipdsp.DeviceArray(ctx, (C, nd, F), np.float32)
    ctx.reserve(8*C*((T + edge + 2047)//2048)*2*len(esos))      # envelope state checkpoints
    hipdsp.synth(ctx, dx, T, C, T, args.rate, 1234 + 2, c0=rank*C, c_total=world*C)
    ctx.synchronize()

    # measured device-copy ceiling (read + write of one trace, hipMemcpy D2D), reported next
    # to the 8 TB/s spec peak as SURVEY 8d asks; untimed, before the steps
    ca, cb = ctx.event(), ctx.event()
    hipdsp.lib.hipdsp_memcpy_d2d(ctx.handle, hipdsp._p(de), hipdsp._p(dx), 4*C*T)
    ctx.record(ca)
    for _ in range(3):
        hipdsp.lib.hipdsp_memcpy_d2d(ctx.handle, hipdsp._p(de), hipdsp._p(dx), 4*C*T)
    ctx.record(cb)
    copy_gbps = 3*8.0*C*T/(ctx.elapsed_ms(ca, cb)*1e-3)/1e9

    if fuse3:
        # one untimed trial of the fused forward sweep: whatever it cannot do (shape, plan) or the
        # device refuses falls back to the separate launches instead of ending the run without a number
        try:
            hipdsp.chain_forward(ctx, plan, eplan, dx, T, df, T, C, T, args.nfft, args.hop, args.rate, ds, nd,
                                 rectify=True, gain=np.pi/2)
            ctx.synchronize()
        except NotImplementedError as err:
            # HIPDSP_ERR_UNSUPPORTED only (a shape or plan the fused sweep does not cover).  Anything else --
            # a HIP error, a fault reported by the kernel -- ends the run: a number measured on a device
            # whose headline kernel has just failed would mask the failure.
            print(f'bench.py: fused forward sweep not used ({err}); separate launches on one stream',
                  file=sys.stderr)
            fuse3 = False

    n_ev = 7
    events = [[ctx.event() for _ in range(n_ev)] for _ in range(args.steps)]
    mids = [ctx.event() for _ in range(args.steps)]

    fused = not args.no_fuse

    def step(i):
        ev = events[i] if i >= 0 else None
        if ev:
            ctx.record(ev[0])
        if fuse3 and multi:
            # the fused forward sweep wants every CU to itself (one 1024-thread workgroup per CU):
            # it starts only when the previous step's all-gather has left the device, so that gather
            # overlaps the backward sweep of its own step and nothing else
            with torch.cuda.stream(cstream):
                for b in range(2):
                    if works[b] is not None:
                        works[b].wait()
                        works[b] = None
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
        if multi:
            # merged spectrogram tile of the resident window on every rank: one RCCL all-gather
            # over xGMI per step, double-buffered so that it overlaps the envelope backward
            # sweep of this step and the kernels of the next one
            with torch.cuda.stream(sstream if overlap else cstream):
                b = counter[0] % 2
                counter[0] += 1
                if works[b] is not None:
                    works[b].wait()              # tile b is about to be overwritten
                    works[b] = None
                tile_buf[b].copy_(tspec[:, :tile_frames, :])
                if args.backend == 'nccl':
                    _, works[b] = allgather_tiles(tile_buf[b], world*C, out=merged[b], async_op=True)
                else:
                    allgather_tiles(tile_buf[b].cpu(), world*C, out=merged[b])
        if ev and fused:
            sctx.record(ev[2])            # end of tile copy + wait for the gather two steps back
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
        if overlap:
            ctx.wait_event(ev_spec)       # the next step overwrites the filtered trace
        if ev:
            ctx.record(ev[4])

    def fence():
        if multi:
            for b in range(2):
                if works[b] is not None:
                    works[b].wait()          # every gather issued so far is part of the job
                    works[b] = None
        sctx.synchronize()
        ctx.synchronize()
        if multi:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(-1)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device='cuda' if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-kernel averages from the HIP events recorded inside the timed region
    if fuse3:
        names = ['chain_fwd<S=%d+%d,filt+env_state+psd>' % (len(sos), len(esos)), 'spectrogram(fused)',
                 'tile_copy+gather_wait', 'env_bwd<S=%d>' % len(esos), 'unused']
    elif fused:
        names = ['sos_ckpt<S=%d+%d,filt+env_state>' % (len(sos), len(esos)), 'spectrogram', 'tile_copy+gather_wait',
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
    traffic = None
    pmc_file = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))
        if pmc.get('shape') == [C, T, args.nfft, args.hop] and dom in pmc.get('kernels', {}):
            traffic = pmc['kernels'][dom]['hbm_bytes']
    kernels = {k: {'ms': round(ms[k], 4),
                   'GBps': round(alg_bytes[k]/(ms[k]*1e-3)/1e9, 1) if k in alg_bytes and ms[k] > 0 else None}
               for k in names if k in alg_bytes or (multi and k == 'tile_copy+gather_wait')}
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

    if rank == 0:
        samples = float(C)*T*world
        line = {
            'metric': 'Msamples/s spectrogram+bandpass, 64ch x 96kHz',
            'value': samples/(dt/args.steps)/1e6,
            'unit': 'Msamples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt/args.steps*1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32 I/O, f64 IIR state',
            'data': 'synthetic',
            'config': {
                'workload': f'BASELINE configs[2]: synthetic {C} ch/GPU x {args.seconds:g} s x '
                            f'{args.rate/1000:g} kHz float32; bandpass {args.hp:g}-{args.lp:g} Hz '
                            f'order {args.order} -> spectrogram nfft {args.nfft} hop {args.hop} '
                            f'+ envelope {args.env:g} Hz',
                'channels_per_gpu': C, 'frames': T, 'spectrogram_frames': nd,
                'parallelism': f'channel shard x{world}' +
                               (f', all-gather of the {args.tile_seconds:g} s spectrogram tile '
                                f'({4*C*tile_frames*F/1e9:.2f} GB per rank) under the ' +
                                ('backward sweep of its step' if fuse3 else 'kernels of this and the next step')
                                if multi else ''),
                'iir_warmup_samples': {'bandpass': warm_f, 'envelope': warm_e},
                'streams': ('spectrogram on a second stream next to the envelope backward sweep '
                            '(their event-bracketed times overlap)' if overlap else 'one stream'),
                'envelope_forward': ('state checkpoints, ' + ('fused into the band-pass kernel' if fused else 'own launch')),
                'spectrogram': ('FFT waves inside the forward sweep (filtered tiles from LDS)' if fuse3 else 'own launch'),
            },
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved/HBM_PEAK_GBS, 4), 'traffic': traffic,
                         'algorithmic_bytes': alg_bytes[dom],
                         'device_copy_GBps': round(copy_gbps, 1),
                         'per_stage_accounting_GBps': stage_gbps},
            'kernels': kernels,
            'chain_algorithmic_GBps': round(sum(alg_bytes.values())/(dt/args.steps)/1e9, 1),
            'parity_max_rel_err': parity,
            'cpu_baseline': cpu,
        }
        if parity is None or not parity < 1e-4:
            line['invalid'] = 'parity gate failed'
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + '\n').encode())
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
