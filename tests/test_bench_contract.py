"""bench.py keeps the driver's contract: flags, defaults that finish within minutes, and (on a GPU)
one JSON line with the required keys."""

import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = ['metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
            'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline']


def test_flags_and_defaults():
    sys.path.insert(0, ROOT)
    import bench
    argv, sys.argv = sys.argv, ['bench.py']
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert (a.gpus, a.steps, a.warmup) == (1, 10, 3)
    # BASELINE configs[2]
    assert (a.channels, a.seconds, a.rate, a.nfft, a.hop) == (64, 600.0, 96000.0, 2048, 1024)
    assert (a.hp, a.lp, a.order, a.env) == (300.0, 3000.0, 2, 20.0)
    # BASELINE configs[3]: 256 channels over 8 GPUs = 32 per GPU; configs[1]: 4 ch x 60 s x 48 kHz, 1024/256, order 4
    sys.argv = ['bench.py', '--config', '3']
    try:
        a3 = bench.parse()
    finally:
        sys.argv = argv
    assert (a3.channels, a3.seconds, a3.rate, a3.nfft, a3.hop) == (32, 600.0, 96000.0, 2048, 1024)
    assert 'configs[3]' in a3.config_name and a3.tile == 'visible'
    sys.argv = ['bench.py', '--config', '1', '--channels', '8']
    try:
        a1 = bench.parse()
    finally:
        sys.argv = argv
    assert (a1.channels, a1.seconds, a1.rate, a1.nfft, a1.hop, a1.order) == (8, 60.0, 48000.0, 1024, 256, 4)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True)
    for flag in ('--gpus', '--steps', '--warmup'):
        assert flag in out.stdout


def _no_launcher_env():
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_WORLD_SIZE', 'GROUP_RANK'):
        env.pop(k, None)
    return env


def test_n_gt_1_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus N` as the driver's N = 1 command is shaped: the parent starts N fresh ranks before
    anything touches the GPU, forwards rank 0's single JSON line and nothing else (here: the gloo rendezvous check,
    no GPU needed)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--rendezvous-only'],
                         capture_output=True, text=True, env=_no_launcher_env(), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert d == {'rendezvous_only': True, 'n_gpus': 3, 'gpus_flag': 3, 'rank_sum': 6.0}
    assert 'starting 3 ranks' in out.stderr


def test_self_launch_passes_the_ranks_failure_on():
    """A rank that dies makes the parent exit non-zero without a JSON line on stdout."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--rendezvous-only'],
                         capture_output=True, text=True, env=dict(_no_launcher_env(), BENCH_RENDEZVOUS_FAIL_RANK='1'),
                         timeout=300)
    assert out.returncode != 0 and out.stdout.strip() == '' and 'starting 2 ranks' in out.stderr


def test_under_a_launcher_world_size_wins():
    """Started by torch.distributed.run (WORLD_SIZE set) bench.py does not launch again."""
    env = dict(_no_launcher_env(), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT='29579')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--rendezvous-only'],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.splitlines()[-1])['n_gpus'] == 1 and 'starting' not in out.stderr


def test_eight_rank_bookkeeping_of_the_multi_gpu_line():
    """the driver's N = 8 bookkeeping -- shard bounds, tile sizes, the all-gather of every tile
    into the merged tensor, max over ranks, ONE line from rank 0 -- had only ever run with 2 and 3 ranks.  Eight ranks
    on the CPU (gloo, host tensors; a GPU box of a round may hold six processes at most): no kernel runs, the line says
    so, everything else is the N = 8 line's shape (bench.py --rehearse-legs, tile_leg_record shared with the timed path)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--config', '3', '--backend', 'gloo',
                          '--same-device', '--channels', '8', '--seconds', '5', '--rehearse-legs'],
                         capture_output=True, text=True, env=_no_launcher_env(), timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['scaling'] == 'weak' and d['value'] is None and 'no kernel ran' in d['invalid']
    assert d['config']['channels_per_gpu'] == 8 and 'configs[3]' in d['config']['workload']
    tiles = d['legs']['tiles']
    assert set(tiles) == {'visible', 'window', 'full'}
    nd = (int(5*96000) + 1023)//1024
    for name, t in tiles.items():
        assert t['merged_ok_on_every_rank'] is True, (name, t)
        assert t['gather_GBps_per_rank_in'] is not None and t['gather_GBps_per_rank_in'] > 0, (name, t)
        assert t['frames'] == nd                     # 5 s of recording: every tile is the whole spectrogram
        assert abs(t['GB_per_rank'] - 4.0*8*nd*1025/1e9) < 1e-3 and abs(t['GB_received_per_rank'] - 7*4.0*8*nd*1025/1e9) < 1e-2
        assert t['step_ms'] > 0 and t['gather_ms'] > 0


@pytest.mark.gpu
def test_four_ranks_on_one_gpu_through_the_real_path():
    """... and the real path with as many ranks as a round's GPU box holds next to the test runner (six processes on the
    card at most, this one among them): compute on GPU 0 in every rank, the tile gathers over gloo -- N > 3 for the
    first time through the kernels, the gatherers and the legs."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--config', '3', '--backend', 'gloo',
                          '--same-device', '--channels', '4', '--seconds', '20', '--steps', '2', '--warmup', '1'],
                         capture_output=True, text=True, env=_no_launcher_env(), timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 4 and d['scaling'] == 'weak' and d['config']['channels_per_gpu'] == 4
    assert d['parity_max_rel_err'] < 1e-4 and 'invalid' not in d
    legs = d['legs']
    assert legs['compute_ms'] > 0 and set(legs['tiles']) == {'visible', 'window', 'full'}
    for name in ('visible', 'window'):
        t = legs['tiles'][name]
        assert 'failed' not in t, t
        assert t['gather_GBps_per_rank_in'] is not None and t['GB_received_per_rank'] > 0


@pytest.mark.gpu
def test_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--seconds', '20', '--steps', '2',
                          '--warmup', '1', '--cpu-sample-seconds', '1'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:2000]            # nothing but the JSON line on stdout
    d = json.loads(lines[0])
    for key in REQUIRED:
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['unit'] == 'Msamples/s' and d['vs_baseline'] is None and 'workload' in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved']/r['peak']) < 1e-3
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0
    assert d['parity_max_rel_err'] < 1e-4 and 'invalid' not in d
    assert d['roofline']['kernel'].startswith('chain_fwd')      # N = 1 default: the fused forward sweep
    assert 500 < r['engine_clock_MHz_in_kernel'] < 2600         # measured inside the kernel, live


@pytest.mark.gpu
def test_separate_launches_on_request():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--seconds', '20', '--steps', '2',
                          '--warmup', '1', '--no-cpu-baseline', '--no-fuse-spectrogram'], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.splitlines()[-1])
    assert d['roofline']['kernel'].startswith('sos_ckpt') and 'spectrogram' in d['kernels']
    assert d['parity_max_rel_err'] < 1e-4 and 'invalid' not in d


@pytest.mark.gpu
def test_multi_rank_code_path_prints_only_the_json_line():
    """The N > 1 code path (torch.distributed over RCCL, non-default streams, tile all-gather)
    with a single rank: RCCL's start-up banner must not end up on stdout."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--force-dist', '--seconds', '20',
                          '--steps', '2', '--warmup', '1', '--no-cpu-baseline'], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29577'))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and 'all-gather' in d['config']['parallelism']
    assert d['parity_max_rel_err'] < 1e-4 and 'invalid' not in d
    # compute and gather are reported separately, for the visible tile, the resident window and the whole spectrogram
    legs = d['legs']
    assert d['compute_ms'] == legs['compute_ms'] > 0 and d['gather_ms'] is not None
    assert set(legs['tiles']) == {'visible', 'window', 'full'}
    assert 'skipped' in legs['tiles']['full']                 # seconds per gather at N = 8: opt-in (--full-leg)
    for name in ('visible', 'window'):
        t = legs['tiles'][name]
        assert t['step_ms'] > 0 and t['gather_ms'] >= 0 and t['GB_per_rank'] > 0


@pytest.mark.gpu
def test_configs3_rehearsal_through_the_c_abi_gather():
    """BASELINE configs[3]'s shard (32 channels per GPU) with the exchange step taken through the C ABI
    (hipdsp_comm_* / hipdsp_allgather_f32 on a second context) instead of torch.distributed -- one rank here,
    the driver runs the real thing; the fused forward sweep leaves CUs to RCCL's kernel."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--force-dist', '--config', '3', '--seconds',
                          '60', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--gather', 'c-abi', '--tile',
                          'window'], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29578'))
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.splitlines()[-1])
    assert d['config']['channels_per_gpu'] == 32 and 'configs[3]' in d['config']['workload']
    assert 'c-abi' in d['config']['parallelism'] and '8 CUs left to RCCL' in d['config']['parallelism']
    assert d['legs']['tile_in_timed_region'] == 'window' and d['parity_max_rel_err'] < 1e-4 and 'invalid' not in d


@pytest.mark.gpu
def test_configs3_shard_at_full_size():
    """BASELINE configs[3] at the size it states: one rank's shard of 256 ch x 600 s x 96 kHz over 8 GPUs = 32 channels
    x 57.6 M frames, the multi-rank code path (non-default streams, CUs left to RCCL, the window tile of SURVEY 8e
    all-gathered through the C ABI every step) with the one rank a one-GPU box has.  bench.py's parity subset compares
    the first 2 s of channels {0, 16, 31}, a window at an internal segment border of the fused plan and the last 2 s
    of the last channel with the oracle (filtered trace, envelope, every PSD frame)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--force-dist', '--config', '3', '--steps',
                          '2', '--warmup', '1', '--no-cpu-baseline', '--gather', 'c-abi', '--tile', 'window'],
                         capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29581'))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert 'configs[3]' in d['config']['workload'] and '32 ch/GPU x 600 s x 96 kHz' in d['config']['workload']
    assert d['config']['channels_per_gpu'] == 32 and d['config']['frames'] == 57600000
    assert d['config']['spectrogram_frames'] == 56250
    assert 'c-abi' in d['config']['parallelism'] and '8 CUs left to RCCL' in d['config']['parallelism']
    assert d['legs']['tile_in_timed_region'] == 'window' and abs(d['legs']['tile_GB_per_rank'] - 0.984) < 0.01
    assert d['parity_max_rel_err'] < 1e-4 and 'invalid' not in d
    assert d['roofline']['kernel'].startswith('chain_fwd')
    assert 0 < d['compute_ms'] <= d['ms_per_step']*1.05


@pytest.mark.gpu
def test_gpus_2_without_a_launcher_on_one_gpu():
    """The driver's scaling command in the shape of its N = 1 command -- `python bench.py --gpus 2 ...` with no
    launcher around it -- starts two ranks itself; here both on GPU 0 with gloo as the collective backend (a one-GPU
    box), so the whole N > 1 path runs: rendezvous, channel shard, tile gather, max over ranks, ONE line."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo',
                          '--same-device', '--seconds', '20', '--channels', '8', '--steps', '2', '--warmup', '1',
                          '--no-legs'], capture_output=True, text=True, timeout=900, env=_no_launcher_env())
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['config']['channels_per_gpu'] == 8
    assert d['parity_max_rel_err'] < 1e-4 and 'invalid' not in d
    assert abs(d['value'] - 2*8*20*96000/(d['ms_per_step']*1e-3)/1e6) < 1e-6*d['value']


@pytest.mark.gpu
def test_configs1_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', '1', '--steps', '3', '--warmup',
                          '1', '--cpu-sample-seconds', '2'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.splitlines()[-1])
    assert 'configs[1]' in d['config']['workload'] and d['config']['channels_per_gpu'] == 4
    assert d['parity_max_rel_err'] < 1e-4 and 'invalid' not in d


def test_strong_scaling_flag_parses():
    sys.path.insert(0, ROOT)
    import bench
    argv = sys.argv
    sys.argv = ['bench.py', '--scaling', 'strong', '--strong-world', '8']
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert a.scaling == 'strong' and a.strong_world == 8 and a.channels == 64      # divided over the GPUs in main()
    assert bench.usable_cores() >= 1


@pytest.mark.gpu
def test_facade_leg_runs_the_fused_launch_and_agrees_with_the_direct_path():
    """At N = 1 the line carries a leg OUTSIDE the timed region that pushes the same slab through the plug-in surface
    (ArrayLoader -> BufferedFilter.update() -> recompute_all()): it must issue the fused launch, give the direct
    path's numbers bit for bit and cost about the same per step."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--seconds', '30', '--steps', '3', '--warmup',
                          '1', '--no-cpu-baseline'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.splitlines()[-1])
    f = d['facade']
    assert 'failed' not in f, f
    assert f['launches_per_update'] == {'chain_forward': 1, 'sosfilt_envelope:2': 1}
    assert f['equals_direct_path_on_sampled_windows'] is True
    assert d['facade_ms_per_step'] == f['facade_ms_per_step'] < 1.5*d['ms_per_step'] + 0.5
    assert d['roofline']['traffic_source'] is None or d['roofline']['traffic_source'].startswith('profiles/')
    assert d['roofline']['device_copy_GBps'] > d['roofline']['hipMemcpy_d2d_GBps'] > 1000


@pytest.mark.gpu
def test_strong_scaling_share_and_graph_leg():
    """--scaling strong: the config's channels in TOTAL over the GPUs (rehearsed here as one rank's share of eight);
    a latency-sized job also reports the step as a replayed hipGraph."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--scaling', 'strong', '--strong-world', '8',
                          '--seconds', '30', '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--no-facade'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.splitlines()[-1])
    assert d['scaling'] == 'strong' and d['config']['channels_per_gpu'] == 8 and 'STRONG' in d['config']['workload']
    assert d['parity_max_rel_err'] < 1e-4 and 'invalid' not in d
    assert isinstance(d.get('graph_ms_per_step'), float) and 0 < d['graph_ms_per_step'] < 2*d['ms_per_step']
