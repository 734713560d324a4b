"""The ISA checker that guards the hand-counted prefetches (tools/check_prefetch_isa.py) must
accept the patterns the kernels rely on and catch the ways they can break."""

import os
import subprocess
import sys

from conftest import ROOT


def asm(*lines):
    return '\t;;#ASMSTART\n' + ''.join('\t%s\n' % l for l in lines) + '\t;;#ASMEND\n'


LOADS = asm('global_load_dwordx2 v[0:1], v[20:21], off') + asm('global_load_dwordx2 v[2:3], v[20:21], off offset:512') + \
    asm('global_load_dwordx2 v[4:5], v[20:21], off offset:1024') + asm('global_load_dwordx2 v[6:7], v[20:21], off offset:1536')

GOOD = '_Zkernel:\n' + LOADS + asm('s_waitcnt vmcnt(0)') + '''.LBB0_1:
	v_add_f32_e32 v9, v0, v1
''' + LOADS + '''	v_mul_f32_e32 v10, v11, v12
	global_store_dword v[30:31], v10, off
	global_store_dword v[30:31], v11, off
''' + asm('s_waitcnt vmcnt(2)') + '''	s_cbranch_scc0 .LBB0_1
	s_endpgm
.Lfunc_end0:
'''


def run(text, tmp_path):
    f = tmp_path/'k.s'
    f.write_text(text)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_prefetch_isa.py'), str(f)],
                          capture_output=True, text=True)


def test_checker_passes_clean_loop_and_flags_early_use(tmp_path):
    r = run(GOOD, tmp_path)
    assert r.returncode == 0 and '0 hazard' in r.stdout and '8 of them' in r.stdout, r.stdout
    # reading a destination register while the load is in flight
    bad = GOOD.replace('\tv_mul_f32_e32 v10, v11, v12\n', '\tv_mul_f32_e32 v10, v2, v12\n')
    r = run(bad, tmp_path)
    assert r.returncode == 1 and 'HAZARD' in r.stdout
    # a wait that does not cover the loads (three operations may remain, only two are younger)
    bad = GOOD.replace('s_waitcnt vmcnt(2)', 's_waitcnt vmcnt(3)')
    r = run(bad, tmp_path)
    assert r.returncode == 1 and 'HAZARD' in r.stdout
    # one store less than counted (a merged store, say)
    bad = GOOD.replace('\tglobal_store_dword v[30:31], v11, off\n', '')
    r = run(bad, tmp_path)
    assert r.returncode == 1 and 'HAZARD' in r.stdout


def test_checker_sees_register_copies_on_the_back_edge(tmp_path):
    # what hipcc does for a conditionally assigned prefetch buffer: a phi copy before the wait
    bad = GOOD.replace(asm('s_waitcnt vmcnt(2)'), '\tv_mov_b32_e32 v40, v4\n' + asm('s_waitcnt vmcnt(2)'))
    r = run(bad, tmp_path)
    assert r.returncode == 1 and 'v_mov_b32_e32 v40, v4' in r.stdout
    # the same copy after the wait is fine
    ok = GOOD.replace(asm('s_waitcnt vmcnt(2)'), asm('s_waitcnt vmcnt(2)') + '\tv_mov_b32_e32 v40, v4\n')
    assert run(ok, tmp_path).returncode == 0


def test_checker_follows_all_paths_and_uniform_flags(tmp_path):
    # a branch around the stores whose own wait is vmcnt(0): fine on both paths
    two_way = '_Zkernel:\n' + LOADS + asm('s_waitcnt vmcnt(0)') + '''.LBB0_1:
	v_add_f32_e32 v9, v0, v1
''' + LOADS + '''	s_cbranch_scc1 .LBB0_2
	global_store_dword v[30:31], v10, off
	global_store_dword v[30:31], v11, off
''' + asm('s_waitcnt vmcnt(2)') + '''	s_branch .LBB0_3
.LBB0_2:
''' + asm('s_waitcnt vmcnt(0)') + '''.LBB0_3:
	s_cbranch_scc0 .LBB0_1
	s_endpgm
.Lfunc_end0:
'''
    assert run(two_way, tmp_path).returncode == 0
    # the no-store path without its wait is a hazard even though the other path is clean
    r = run(two_way.replace('.LBB0_2:\n' + asm('s_waitcnt vmcnt(0)'), '.LBB0_2:\n'), tmp_path)
    assert r.returncode == 1
    # hipcc's lowering of "wait unless the stores were issued": a 0 / -1 flag in an SGPR pair,
    # tested through vcc -- the checker must not walk the combination the flag rules out
    flagged = '_Zkernel:\n' + LOADS + asm('s_waitcnt vmcnt(0)') + '''.LBB0_1:
	v_add_f32_e32 v9, v0, v1
''' + LOADS + '''	s_mov_b64 s[16:17], -1
	s_cbranch_scc1 .LBB0_2
	global_store_dword v[30:31], v10, off
	global_store_dword v[30:31], v11, off
''' + asm('s_waitcnt vmcnt(2)') + '''	s_mov_b64 s[16:17], 0
.LBB0_2:
	s_andn2_b64 vcc, exec, s[16:17]
	s_cbranch_vccnz .LBB0_3
''' + asm('s_waitcnt vmcnt(0)') + '''.LBB0_3:
	s_cbranch_scc0 .LBB0_1
	s_endpgm
.Lfunc_end0:
'''
    assert run(flagged, tmp_path).returncode == 0
    # with the flag clobbered by something the checker cannot evaluate, it has to assume the worst
    r = run(flagged.replace('\ts_mov_b64 s[16:17], -1\n', '\ts_mov_b64 s[16:17], -1\n\ts_and_b64 s[16:17], s[16:17], s[30:31]\n'), tmp_path)
    assert r.returncode == 1


def test_compiler_tracked_loads_only_take_a_queue_slot(tmp_path):
    # a load outside an asm block is hipcc's business (it inserts its own waits); it still counts
    # as a younger operation behind the prefetch
    text = GOOD.replace('\tglobal_store_dword v[30:31], v11, off\n', '\tglobal_load_dword v50, v[30:31], off\n')
    assert run(text, tmp_path).returncode == 0


def test_isa_tally_classifies_a_listing(tmp_path):
    """tools/isa_tally.py (where do a sweep's VALU issue slots go): classes and per-block counts of a small listing."""
    text = '''_Zmykernel_v1:
	s_load_dwordx4 s[0:3], s[4:5], 0x0
	v_cvt_f64_f32_e32 v[2:3], v1
.LBB0_1:
	v_fma_f64 v[4:5], s[0:1], v[2:3], v[4:5]
	v_pk_fma_f32 v[6:7], v[6:7], v[8:9], v[10:11]
	v_mov_b32_dpp v12, v13 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
	v_cndmask_b32_e64 v14, v15, v16, s[0:1]
	ds_read_b128 v[20:23], v24
	global_store_dwordx4 v[30:31], v[20:23], off
	s_waitcnt vmcnt(0)
	s_cbranch_scc0 .LBB0_1
	s_endpgm
.Lfunc_end0:
'''
    f = tmp_path/'k.s'
    f.write_text(text)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_tally.py'), str(f), 'mykernel', '1'],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    for want in ('valu f64 fma', 'valu f64 cvt', 'valu pk_f32', 'valu dpp', 'valu cndmask', 'lds', 'vmem', 'smem',
                 's_waitcnt', 'branch'):
        assert want in out, (want, out)
    assert 'whole function: 11 instructions' in out
    assert '.LBB0_1: 8' in out and '.LBB0_1+: 1' in out        # (a branch ends a block: what follows it is the fall-through path's)


# ---- tools/entry_points_gate.py: the regression gate over the per-entry-point timing logs (Missing 2)
BASE_LOG = '''hipdsp_sosfilt, band-pass of 2 sections (BufferedFilter alone)                    5.528 ms    5335 GB/s
hipdsp_chain_forward 2048/1024, 2 + 1 sections                                   10.379 ms    4263 GB/s
hipdsp_chain_forward 256/128, 2 + 1 sections                                     10.683 ms    4152 GB/s
hipdsp_spectrogram 8192/4096 (BufferedSpectrogram alone)                         11.322 ms    2605 GB/s
nfft   2048 hop   1024 PSD   :    1.535 ms    4800 GB/s
some chatter that is not a timing line
'''


def gate(tmp_path, new_text, *args):
    (tmp_path/'base.log').write_text(BASE_LOG)
    (tmp_path/'new.log').write_text(new_text)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'entry_points_gate.py'), str(tmp_path/'base.log'),
                           str(tmp_path/'new.log'), *args], capture_output=True, text=True)


def test_entry_points_gate_flags_a_slower_line_and_nothing_else(tmp_path):
    # round 4's own step: the fused launch at the reference's default window, 10.68 -> 14.68 ms
    new = BASE_LOG.replace('10.683 ms    4152', '14.678 ms    3022').replace('5.528 ms', '5.630 ms').replace('11.322 ms', '7.815 ms')
    r = gate(tmp_path, new)
    assert r.returncode == 1, r.stdout
    flagged = [l for l in r.stdout.splitlines() if 'SLOWER' in l]
    assert len(flagged) == 1 and '256/128' in flagged[0] and '+37.4 %' in flagged[0], r.stdout
    assert 'faster' in r.stdout and '1 slower than the tolerance' in r.stdout and 'FAILED' in r.stdout
    # within the tolerance (here 1.8 %): passes; a tighter tolerance catches it
    ok = BASE_LOG.replace('5.528 ms', '5.630 ms')
    assert gate(tmp_path, ok).returncode == 0
    assert gate(tmp_path, ok, '--tol', '0.01').returncode == 1
    # an accepted regression is written down where the gate runs
    r = gate(tmp_path, new, '--allow', '256/128=the reason')
    assert r.returncode == 0 and '[the reason]' in r.stdout and 'allowed' in r.stdout, r.stdout


def test_entry_points_gate_lines_that_come_and_go(tmp_path):
    new = BASE_LOG.replace('hipdsp_spectrogram 8192/4096 (BufferedSpectrogram alone)                         11.322 ms    2605 GB/s\n', '') + \
        'hipdsp_new_entry_point                                                            1.000 ms    1000 GB/s\n'
    r = gate(tmp_path, new)
    assert r.returncode == 0 and 'gone' in r.stdout and 'new ' in r.stdout and '1 missing' in r.stdout, r.stdout
    assert gate(tmp_path, new, '--strict').returncode == 1
    # a log without timing lines is an error of its own, not a pass
    assert gate(tmp_path, 'nothing here\n').returncode == 2
    # the spec_sizes format (name ends in a colon) is read too
    r = gate(tmp_path, BASE_LOG.replace('1.535 ms', '1.700 ms'))
    assert r.returncode == 1 and 'nfft 2048 hop 1024 PSD' in r.stdout, r.stdout
