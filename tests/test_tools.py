"""The ISA hazard checker that guards the hand-counted prefetch must catch a hazard."""

import os
import subprocess
import sys

from conftest import ROOT

GOOD = '''
_Zkernel:
.LBB0_1:
	s_waitcnt vmcnt(2)
	v_add_f32_e32 v9, v0, v1
	global_load_dwordx2 v[0:1], v[20:21], off
	global_load_dwordx2 v[2:3], v[20:21], off offset:512
	global_load_dwordx2 v[4:5], v[20:21], off offset:1024
	global_load_dwordx2 v[6:7], v[20:21], off offset:1536
	v_mul_f32_e32 v10, v11, v12
	global_store_dword v[30:31], v10, off
	global_store_dword v[30:31], v11, off
	s_cbranch_scc0 .LBB0_1
	s_endpgm
'''


def run(text, tmp_path):
    f = tmp_path/'k.s'
    f.write_text(text)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_prefetch_isa.py'), str(f)],
                          capture_output=True, text=True)


def test_checker_passes_clean_loop_and_flags_early_use(tmp_path):
    r = run(GOOD, tmp_path)
    assert r.returncode == 0 and '0 hazard' in r.stdout, r.stdout
    # reading a destination register before the counted wait
    bad = GOOD.replace('\tv_mul_f32_e32 v10, v11, v12\n', '\tv_mul_f32_e32 v10, v2, v12\n')
    r = run(bad, tmp_path)
    assert r.returncode == 1 and 'HAZARD' in r.stdout
    # a wait that does not cover the loads (only one younger store would be allowed to remain)
    bad = GOOD.replace('s_waitcnt vmcnt(2)', 's_waitcnt vmcnt(3)')
    r = run(bad, tmp_path)
    assert r.returncode == 1 and 'HAZARD' in r.stdout
