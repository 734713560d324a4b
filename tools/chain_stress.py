"""3000 more seeds of tests/test_gpu_fuzz.py::test_random_chain_forward_cases (both hand-over variants of the
fused forward sweep are drawn at random); needs an MI355X.  Round 1: 0 failures."""
import sys, os
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.chdir(ROOT)
from oracle import oracle
import test_gpu_fuzz as t
bad = 0
for seed in range(1000, 4000):
    try:
        t.test_random_chain_forward_cases(oracle, seed)
    except AssertionError as e:
        bad += 1; print('CHAIN_FORWARD FAIL', seed, str(e)[:200], flush=True)
    if seed % 500 == 0:
        print('seed', seed, flush=True)
print('done, failures:', bad)
