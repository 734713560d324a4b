"""Host half of the IIR plan (no GPU): warm-up length, pad length and steady-state initial
conditions computed by libhip_dsp against brute-force NumPy and the oracle."""

import ctypes

import numpy as np
import pytest

from audian_amd import _lib
from audian_amd.design import butter_sos

TILE = 2048


def plan_host(sos):
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    warm = ctypes.c_int64()
    edge = ctypes.c_int()
    zi = np.zeros(2*len(sos))
    _lib.check(_lib.lib.hipdsp_sos_plan_host(ctypes.c_void_p(sos.ctypes.data), len(sos), ctypes.byref(warm),
                                             ctypes.byref(edge), ctypes.c_void_p(zi.ctypes.data)))
    return int(warm.value), int(edge.value), zi


def transition_matrix(sos):
    """State-transition matrix of the DF-II-transposed cascade, column by column."""
    S = len(sos)
    A = np.zeros((2*S, 2*S))
    for c in range(2*S):
        z = np.zeros(2*S)
        z[c] = 1.0
        cur = 0.0
        for s in range(S):
            b0, b1, b2, _, a1, a2 = sos[s]
            y = b0*cur + z[2*s]
            z[2*s] = b1*cur - a1*y + z[2*s + 1]
            z[2*s + 1] = b2*cur - a2*y
            cur = y
        A[:, c] = z
    return A


@pytest.mark.parametrize('btype,order,wn,rate', [
    ('bandpass', 2, (300.0, 3000.0), 96000.0), ('bandpass', 4, (300.0, 3000.0), 48000.0),
    ('lowpass', 2, 20.0, 96000.0), ('lowpass', 2, 500.0, 48000.0), ('highpass', 3, 100.0, 192000.0),
    ('bandpass', 2, (5.0, 3000.0), 96000.0), ('lowpass', 1, 4000.0, 48000.0)])
def test_warmup_is_the_smallest_tile_multiple_below_tolerance(oracle, btype, order, wn, rate):
    sos = butter_sos(order, wn, btype, rate)
    warm, edge, zi = plan_host(sos)
    assert warm % TILE == 0 and warm >= TILE
    A = transition_matrix(sos)
    tol = 2.0**-60
    norm = lambda M: np.max(np.sum(np.abs(M), axis=1))
    assert norm(np.linalg.matrix_power(A, warm)) < tol
    if warm > TILE:
        assert norm(np.linalg.matrix_power(A, warm - TILE)) >= tol*0.5     # not wastefully long
    assert edge == oracle.sosfiltfilt_edge(sos)
    assert np.allclose(zi.reshape(-1, 2), oracle.sosfilt_zi(sos), rtol=1e-12, atol=1e-15)


def test_non_decaying_filter_never_segments():
    # a pure integrator section (pole on the unit circle) keeps its history for ever
    sos = np.array([[1.0, 0.0, 0.0, 1.0, -1.0, 0.0]])
    warm, edge, _ = plan_host(sos)
    assert warm >= 2**40


def test_plan_rejects_bad_tables():
    with pytest.raises(ValueError):
        plan_host(np.array([[1.0, 0.0, 0.0, 2.0, 0.0, 0.0]]))            # a0 != 1
    with pytest.raises(NotImplementedError):
        plan_host(np.tile(np.array([[1.0, 0.0, 0.0, 1.0, 0.0, 0.0]]), (5, 1)))   # > 4 sections
    with pytest.raises(ValueError):
        plan_host(np.array([[np.nan, 0.0, 0.0, 1.0, 0.0, 0.0]]))


def segments(n_cus, w_max, per_simd, frames, channels, warm, max_segments=0):
    length = ctypes.c_int64()
    count = ctypes.c_int()
    _lib.check(_lib.lib.hipdsp_sos_segments_host(n_cus, w_max, per_simd, max_segments, frames, channels, warm,
                                                 ctypes.byref(length), ctypes.byref(count)))
    return int(length.value), int(count.value)


def step_cost(w, w_max, per_simd):
    if per_simd > 0:
        sw = -(-w//per_simd)
        return 0.65 if sw <= 1 else 0.5*sw
    if per_simd == -1:
        return 7.0 + 0.5625*w
    if per_simd == -2:
        return 1.4 + 0.9125*w
    return w_max*(1 + 0.25*(1 - w/w_max))


def plan_cost(n_cus, w_max, per_simd, frames, channels, warm, n):
    """The planner's cost model (include/hip_dsp.h): rounds x (segment + warm-up) x cost of a tile step."""
    ln = max(TILE, -(-(-(-frames//n))//TILE)*TILE)
    cnt = -(-frames//ln)
    units = channels*cnt
    span = ln + (warm if cnt > 1 else 0)
    if units <= n_cus*w_max:
        return span*step_cost(-(-units//n_cus), w_max, per_simd)
    return -(-units//(n_cus*w_max))*span*step_cost(w_max, w_max, per_simd)


def test_segment_planner_properties():
    """Segments are whole tiles, cover the slab, and their number follows the cost model."""
    rng = np.random.default_rng(0)
    for _ in range(300):
        n_cus = int(rng.choice([256, 248, 64]))
        w_max, per_simd = [(16, 4), (12, 4), (8, 0), (16, -1), (16, -2), (8, 4)][int(rng.integers(0, 6))]
        channels = int(rng.integers(1, 300))
        frames = int(rng.integers(1, 60_000_000))
        warm = int(rng.choice([0, 2048, 4096, 53248, 400*2048, 2**50*2048]))
        cap = int(rng.choice([0, 0, 1, 7]))
        length, count = segments(n_cus, w_max, per_simd, frames, channels, warm, cap)
        assert length % TILE == 0 and length >= TILE and count >= 1
        assert count*length >= frames > (count - 1)*length
        if cap:
            assert count <= cap
        if warm >= 2**40:
            assert count == 1                                  # a filter that never forgets
            continue
        # no other segment count (that is allowed) is cheaper by the model
        mine = plan_cost(n_cus, w_max, per_simd, frames, channels, warm, count)
        top = min(cap or 10**9, -(-frames//TILE), 65536)
        for n in {1, 2, 3, count + 1, max(1, count - 1), max(1, n_cus*w_max//channels), max(1, n_cus*4//channels),
                  max(1, n_cus*8//channels), max(1, n_cus//channels)}:
            if n <= top:
                assert mine <= plan_cost(n_cus, w_max, per_simd, frames, channels, warm, n)*(1 + 1e-9), (n, count)


def test_segment_planner_bench_configuration():
    # BASELINE configs[2], backward sweep on 256 CUs: 8 waves per CU (as fast as 16, half the warm-ups) = 32 segments
    length, count = segments(256, 16, 4, 57_600_000, 64, 53248)
    assert count == 32 and length == -(-57_600_000//32//TILE)*TILE
    # ... the fused forward sweep (8 pairs per CU, band-pass warm-up only): 32 segments per channel
    length, count = segments(256, 8, 0, 57_600_000, 64, 4096)
    assert count == 32
    # few channels per GPU (strong scaling, 8 of 64): 8 waves per CU again
    length, count = segments(256, 16, 4, 57_600_000, 8, 53248)
    assert count == 256
    # BASELINE configs[1] (4 ch x 60 s x 48 kHz): a few tiles per segment at one wave per SIMD, not 1024 segments per channel
    length, count = segments(256, 16, 4, 2_880_000, 4, 26624)
    assert count <= 256 and length >= 4*TILE
    # a short interactive slab is not cut below what the warm-up makes worthwhile
    length, count = segments(256, 16, 4, 200_000, 2, 53248)
    assert count*length >= 200_000 and length >= TILE
    # the band-pass alone and the band-pass + envelope-state sweep fill the CU at configs[2]'s size: 16 waves = 64 segments
    for model in (-1, -2):
        length, count = segments(256, 16, model, 57_600_000, 64, 4096)
        assert count == 64, model
    # ... and do not shred a short slab into one-tile segments that mostly warm up (4 ch x 60 s x 48 kHz, warm-up 2 tiles)
    length, count = segments(256, 16, -2, 2_880_000, 4, 4096)
    assert length >= 2*TILE
    with pytest.raises(ValueError):
        segments(0, 16, 4, 10, 1, 0)


@pytest.mark.parametrize('btype,order,wn,rate', [
    ('bandpass', 2, (300.0, 3000.0), 96000.0), ('bandpass', 4, (300.0, 3000.0), 48000.0),
    ('lowpass', 7, 500.0, 48000.0), ('highpass', 5, 100.0, 192000.0)])
def test_transition_powers_are_block_lower_triangular(btype, order, wn, rate):
    """The scan kernels skip the entries of A^(32*2^k) above the 2 x 2 block diagonal
    (csrc/sos_cascade.inc): a section's state never depends on the sections behind it, and powers
    computed by repeated squaring of such a matrix keep EXACT zeros there."""
    sos = butter_sos(order, wn, btype, rate)
    A = transition_matrix(sos)
    D = len(A)
    P = np.linalg.matrix_power(A, 32)
    for k in range(6):
        for r in range(D):
            for c in range(D):
                if c//2 > r//2:
                    assert A[r, c] == 0.0 and P[r, c] == 0.0, (k, r, c)
        P = P @ P
