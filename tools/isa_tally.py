"""Tally the instructions of one kernel of a hipcc -S listing by basic block and class -- where do the VALU
issue slots of a sweep go?
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -S --cuda-device-only -o /tmp/sos.s audian_amd/csrc/sos.hip
    python tools/isa_tally.py /tmp/sos.s env_bwd_kernelILi1ELb1ELb1ELb0E [min_block_size]
"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
minsize = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
def cls(m, ops):
    if m.startswith('v_'):
        if 'dpp' in ops or m.endswith('_dpp'): return 'valu dpp'
        if re.search(r'_f64|f64_', m):
            return 'valu f64 ' + ('cvt' if 'cvt' in m else 'fma' if 'fma' in m else 'mul' if 'mul' in m else 'add' if 'add' in m else 'other')
        if m.startswith('v_pk_'): return 'valu pk_f32'
        if re.search(r'_f32', m): return 'valu f32'
        if m.startswith(('v_mov', 'v_accvgpr')): return 'valu mov'
        if m.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): return 'valu lane'
        if m.startswith('v_cndmask'): return 'valu cndmask'
        if m.startswith('v_cmp'): return 'valu cmp'
        if m.startswith('v_permlane'): return 'valu permlane'
        return 'valu int/other'
    if m.startswith('ds_'): return 'lds'
    if m.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if m.startswith(('s_load', 's_buffer_load')): return 'smem'
    if m.startswith('s_waitcnt'): return 's_waitcnt'
    if m.startswith(('s_cbranch', 's_branch')): return 'branch'
    if m.startswith('s_nop'): return 's_nop'
    if m.startswith('s_'): return 'salu'
    return 'other'
blocks, cur, name = [], collections.Counter(), 'entry'
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(('.', ';')) and not re.match(r'^\.LBB\S*:', t):
        continue
    mlab = re.match(r'^(\.LBB\S*):', t)
    if mlab:
        blocks.append((name, cur)); cur, name = collections.Counter(), mlab.group(1)
        continue
    parts = t.split(None, 1)
    cur[cls(parts[0], parts[1] if len(parts) > 1 else '')] += 1
    if parts[0].startswith(('s_cbranch', 's_branch')):
        # a branch ends a basic block although the fall-through path carries no label (round 5: a once-per-sweep
        # selection loop behind `s_cbranch_scc1` read as part of the per-tile block in front of it)
        blocks.append((name, cur)); cur, name = collections.Counter(), name.rstrip('+') + '+'
blocks.append((name, cur))
total = collections.Counter()
for n, c in blocks: total.update(c)
print('whole function:', sum(total.values()), 'instructions')
for k, v in sorted(total.items(), key=lambda kv: -kv[1]): print(f'   {k:18s} {v}')
print('blocks with at least', minsize, 'instructions:')
for n, c in blocks:
    if sum(c.values()) >= minsize:
        print(f'{n}: {sum(c.values())}  ' + ', '.join(f'{k} {v}' for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
