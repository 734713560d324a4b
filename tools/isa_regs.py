"""Registers, scratch and LDS per kernel of a gfx950 ISA listing (the .s files -save-temps leaves in csrc/build),
optionally next to a second listing (another build): spills that a change brought in show up as scratch bytes.
    python tools/isa_regs.py build/chain-hip-amdgcn-amd-amdhsa-gfx950.s [other.s] [--filter chain_fwd]"""
import re
import subprocess
import sys


def kernels(path):
    s = open(path).read()
    out = {}
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(r'\.amdhsa_%s (\d+)' % k, body).group(1))
        out[m.group(1)] = (g('next_free_vgpr'), g('next_free_sgpr'), g('private_segment_fixed_size'), g('group_segment_fixed_size'))
    return out


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    flt = next((a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--filter=')), '')
    a = kernels(args[0])
    b = kernels(args[1]) if len(args) > 1 else {}
    dm = demangle(list(a))
    for k, v in a.items():
        name = re.sub(r'\(.*', '', dm.get(k, k).replace('(anonymous namespace)::', '').replace('void ', ''))
        if flt and flt not in name:
            continue
        line = f'{name:64s} vgpr {v[0]:3d} sgpr {v[1]:3d} scratch {v[2]:4d} lds {v[3]:6d}'
        if k in b:
            o = b[k]
            line += f'   | other: vgpr {o[0]:3d} scratch {o[2]:4d} lds {o[3]:6d}' + ('   <-- scratch changed' if o[2] != v[2] else '')
        print(line)
