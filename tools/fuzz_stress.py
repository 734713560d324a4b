"""More seeds of the seeded random GPU tests (tests/test_gpu_fuzz.py) than the suite runs: every test function of
that module that takes (oracle, seed), seeds 1000 .. 1000 + N, time-boxed; needs an MI355X.
    python tools/fuzz_stress.py [N=400] [minutes=8] [first seed=1000]
    FUZZ_MODULE=test_gpu_facade python tools/fuzz_stress.py 60 9        (random walks through the facade)
    FUZZ_OPTIONS=1 python tools/fuzz_stress.py ...                      (random tuning options per case as well)
    FUZZ_ONLY=spectrogram python tools/fuzz_stress.py ...               (the families whose name contains the word)
"""
import sys, os, time, inspect
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.chdir(ROOT)
from oracle import oracle
import importlib
t = importlib.import_module(os.environ.get('FUZZ_MODULE', 'test_gpu_fuzz'))     # FUZZ_MODULE=test_gpu_facade: the random walks
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
budget = 60.0*float(sys.argv[2]) if len(sys.argv) > 2 else 480.0
first = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
# every case is announced in a file BEFORE it runs (a kernel fault kills the process: the last line names the case)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
progress = open(os.path.join(ROOT, 'gpurun_out', 'fuzz_progress.log'), 'w')
failures = open(os.path.join(ROOT, 'gpurun_out', 'fuzz_failures.log'), 'w')
tests = [(n, f) for n, f in inspect.getmembers(t, inspect.isfunction)
         if n.startswith('test_random') and list(inspect.signature(f).parameters) == ['oracle', 'seed']
         and os.environ.get('FUZZ_ONLY', '') in n]                      # FUZZ_ONLY=spectrogram: one family only
t0 = time.time()
bad, done = 0, {n: 0 for n, _ in tests}
for seed in range(first, first + N):
    for name, f in tests:
        if time.time() - t0 > budget:
            break
        opts = {}
        if os.environ.get('FUZZ_OPTIONS') == '1':
            # tuning options whose values must not change any result beyond rounding, drawn per case
            import numpy as _np, gpu_helpers as _gh
            r = _np.random.default_rng(seed*131 + len(name))
            opts = {'sos_waves_per_cu': int(r.choice([4, 8, 12, 16])), 'sos_prefetch': int(r.integers(0, 2)),
                    'spec_kernel': int(r.choice([0, 0, 2, 3])), 'spec_no_half': int(r.integers(0, 2)),
                    'spec_fpw': int(r.choice([0, 0, 1, 3, 16])), 'sos_fair': int(r.integers(0, 2)),
                    'sos_no_pin': int(r.integers(0, 2)), 'chain_reserve_cus': int(r.choice([0, 0, 8, 100])),
                    'force_generic_fft': int(r.integers(0, 8) == 0)}
            for k, v in opts.items():
                _gh.ctx().set_option(k, v)
        progress.write(f'{name} {seed} {opts}\n'); progress.flush(); os.fsync(progress.fileno())
        try:
            f(oracle, seed)
        except AssertionError as e:
            bad += 1
            print('FAIL', name, seed, opts, str(e)[:300], flush=True)
            failures.write(f'FAIL {name} {seed} {opts} {str(e)[:300]}\n'); failures.flush()
        except Exception as e:
            bad += 1
            print('ERROR', name, seed, type(e).__name__, str(e)[:300], flush=True)
            failures.write(f'ERROR {name} {seed} {type(e).__name__} {str(e)[:300]}\n'); failures.flush()
        done[name] += 1
    if time.time() - t0 > budget:
        break
    if seed % 25 == 0:
        print('seed', seed, 'elapsed %.0f s' % (time.time() - t0), flush=True)
print('cases per test:', done)
print('done, failures:', bad)
