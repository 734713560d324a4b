"""A wider one-off run of tests/test_gpu_fuzz.py (about 1000 seeds instead of 68); needs an MI355X."""
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
print('done, failures:', bad)
