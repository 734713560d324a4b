"""A wider one-off run of tests/test_gpu_fuzz.py (about 1250 seeds instead of 84); needs an MI355X."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.chdir(ROOT)
from oracle import oracle
import test_gpu_fuzz as t
bad = 0
for seed in range(24, 424):
    try:
        t.test_random_envelope_cases(oracle, seed)
    except AssertionError as e:
        bad += 1; print('ENV FAIL', seed, str(e)[:200])
for seed in range(12, 162):
    try:
        t.test_random_filter_envelope_chain_cases(oracle, seed)
    except AssertionError as e:
        bad += 1; print('CHAIN FAIL', seed, str(e)[:200])
for seed in range(20, 320):
    try:
        t.test_random_spectrogram_cases(oracle, seed)
    except AssertionError as e:
        bad += 1; print('SPEC FAIL', seed, str(e)[:200])
for seed in range(12, 212):
    try:
        t.test_random_sosfilt_cases(oracle, seed)
    except AssertionError as e:
        bad += 1; print('SOSFILT FAIL', seed, str(e)[:200])
for seed in range(16, 216):
    try:
        t.test_random_chain_forward_cases(oracle, seed)
    except AssertionError as e:
        bad += 1; print('CHAIN_FORWARD FAIL', seed, str(e)[:200])
    if seed % 50 == 0:
        print('chain_forward seed', seed, flush=True)
print('done, failures:', bad)
