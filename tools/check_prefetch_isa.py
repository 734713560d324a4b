"""Build-time guard for the hand-counted prefetches (sos.hip, spectrogram.hip).

The inline-asm loads write their destination VGPRs asynchronously and hipcc does not know it:
until an `s_waitcnt vmcnt(N)` retires them nothing may read or write those registers, and a
register copy the compiler slips in between (a phi copy on a loop back edge, say) silently
reads stale data.  This script re-checks the generated ISA with a model of the hardware rule:

  * vector-memory operations of a wave complete in issue order; `s_waitcnt vmcnt(N)` returns
    when at most the N youngest are outstanding;
  * the state at a program point is the queue of outstanding operations (destination
    registers of loads, nothing for stores), oldest pending load first;
  * the states are propagated over the control-flow graph of every kernel to a fixed point
    (all paths, loops included); an instruction that touches a register with a load in flight
    in ANY reachable state is reported.

A second, straight-line rule covers the SGPR bases of inline-asm VMEM instructions (sgpr_hazards below).

usage: python tools/check_prefetch_isa.py file.s [kernel-substring ...]
"""
import re
import sys

VMEM = re.compile(r'^(global|buffer|scratch|flat)_(load|store|atomic)\w*\s')
LOAD = re.compile(r'^(global|buffer|scratch|flat)_(load|atomic)\w*\s+(v\[(\d+):(\d+)\]|v(\d+))\s*,')
MAX_STATES = 20000


def regs_of(text):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', text):
        out.add(int(m.group(1)))
    return out


def kernels(lines):
    name, start = None, None
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name, start = m.group(1), i
        if name and l.startswith('.Lfunc_end'):
            yield name, lines[start + 1:i]
            name = None


def parse(body):
    """Instructions (text) with label positions; comments and directives dropped.  Loads inside
    an inline-asm block are marked with a leading '@': only those are invisible to hipcc's own
    wait insertion, everything else merely takes a place in the queue."""
    insts, labels = [], {}
    in_asm = False
    for l in body:
        if l.strip().startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if l.strip().startswith(';;#ASMEND'):
            in_asm = False
            continue
        t = l.split(';')[0].strip()
        if not t:
            continue
        m = re.match(r'^(\.L\w+):$', t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if t.startswith('.'):
            continue
        insts.append(('@' + t) if in_asm and VMEM.match(t) else t)
    return insts, labels


def canon(queue):
    """Drop everything older than the oldest pending load; cap at the 6-bit counter."""
    for i, e in enumerate(queue):
        if e:
            return tuple(queue[i:][-64:])
    return ()


FLAG_SET = re.compile(r'^s_mov_b64\s+(s\[\d+:\d+\]),\s*(0|-1)$')
VCC_FROM = re.compile(r'^s_(and|andn2)_b64\s+vcc,\s*exec,\s*(s\[\d+:\d+\])$')
FIRST_OP = re.compile(r'^\S+\s+(s\[(\d+):(\d+)\]|s(\d+))(?=[\s,]|$)')


def scalar_step(inst, flags, vcc):
    """Tiny constant propagation for the uniform-condition idiom of hipcc: a condition lives in
    an SGPR pair as 0 / -1 (`s_mov_b64`), is turned into vcc by `s_and(n2)_b64 vcc, exec, s[..]`
    and consumed by `s_cbranch_vccz/vccnz`.  Without it the check would walk paths such as
    "took the no-store branch, then skipped that branch's wait", which the flag rules out."""
    m = FLAG_SET.match(inst)
    if m:
        flags = dict(flags)
        flags[m.group(1)] = int(m.group(2))
        return frozenset(flags.items()), vcc
    m = VCC_FROM.match(inst)
    if m:
        val = dict(flags).get(m.group(2))
        if val is None:
            return flags, None
        nonzero = (val == -1) if m.group(1) == 'and' else (val == 0)
        return flags, ('nz' if nonzero else 'z')
    if re.match(r'^s_(cbranch|branch|waitcnt|nop|barrier|endpgm|sleep)', inst):
        return flags, vcc
    if 'vcc' in inst:
        vcc = None
    m = FIRST_OP.match(inst)
    if m and flags:
        lo_ = int(m.group(2)) if m.group(2) else int(m.group(4))
        hi_ = int(m.group(3)) if m.group(3) else lo_
        keep = {}
        for name, val in flags:
            a_, b_ = [int(x) for x in name[2:-1].split(':')]
            if b_ < lo_ or a_ > hi_:
                keep[name] = val
        flags = frozenset(keep.items())
    return flags, vcc


def step(inst, state, report, where):
    queue, flags, vcc = state
    untracked = inst.startswith('@')
    if untracked:
        inst = inst[1:]
    flags, vcc = scalar_step(inst, flags, vcc)
    pending = set().union(*queue) if queue else set()
    m = re.search(r's_waitcnt.*vmcnt\((\d+)\)', inst)
    if m:
        n = int(m.group(1))
        q = list(queue)
        q = q[len(q) - n:] if 0 < n < len(q) else ([] if n == 0 else q)
        return canon(q), flags, vcc
    if inst.startswith('s_'):
        return queue, flags, vcc
    used = regs_of(inst)
    if pending and used & pending:
        report(where, inst, sorted(used & pending))
    if VMEM.match(inst):
        ld = LOAD.match(inst)
        dest = frozenset()
        if untracked and ld and 'store' not in inst.split()[0]:
            dest = frozenset(range(int(ld.group(4)), int(ld.group(5)) + 1)) if ld.group(4) else \
                frozenset([int(ld.group(6))])
        return canon(list(queue) + [dest]), flags, vcc
    return queue, flags, vcc


def check(name, body):
    insts, labels = parse(body)
    if not any(i.startswith('@') for i in insts):
        return 0, []
    # basic-block leaders
    leaders = {0} | set(labels.values())
    for k, t in enumerate(insts):
        if re.match(r'^s_(cbranch|branch|endpgm)', t):
            leaders.add(k + 1)
    leaders = sorted(x for x in leaders if x < len(insts))
    block_of = {}
    for b, st in enumerate(leaders):
        end = leaders[b + 1] if b + 1 < len(leaders) else len(insts)
        for k in range(st, end):
            block_of[k] = b
    problems = {}

    def report(where, inst, regs):
        problems.setdefault((where, inst), regs)

    entry = {0: {((), frozenset(), None)}}
    work = [0]
    n_loads = sum(1 for i in insts if i.startswith('@'))
    while work:
        b = work.pop()
        st = leaders[b]
        end = leaders[b + 1] if b + 1 < len(leaders) else len(insts)
        last = insts[end - 1]
        for state in list(entry[b]):
            s = state
            for k in range(st, end):
                s = step(insts[k], s, report, k)
            vcc = s[2]
            m = re.match(r'^s_cbranch_(\w+)\s+(\.L\w+)', last)
            if m:
                succ = [labels[m.group(2)], end]
                if m.group(1) == 'vccnz' and vcc is not None:
                    succ = [succ[0]] if vcc == 'nz' else [succ[1]]
                if m.group(1) == 'vccz' and vcc is not None:
                    succ = [succ[0]] if vcc == 'z' else [succ[1]]
            elif re.match(r'^s_branch\s+(\.L\w+)', last):
                succ = [labels[re.match(r'^s_branch\s+(\.L\w+)', last).group(1)]]
            elif last.startswith('s_endpgm'):
                succ = []
            else:
                succ = [end]
            for t in succ:
                if t >= len(insts):
                    continue
                tb = block_of[t]
                cur = entry.setdefault(tb, set())
                if s not in cur:
                    cur.add(s)
                    if len(cur) > MAX_STATES:
                        raise SystemExit(f'{name}: state explosion in the prefetch check')
                    if tb not in work:
                        work.append(tb)
    return n_loads, [(name, k, inst, regs) for (k, inst), regs in sorted(problems.items())]


VALU_SGPR_WRITE = re.compile(r'^v_(readlane_b32|readfirstlane_b32)\s+s(\d+)\b')


def sgpr_hazards(name, body):
    """Second rule, straight-line: a VMEM instruction that reads an SGPR (its `saddr` pair) a VALU instruction has written
    (v_readlane_b32 / v_readfirstlane_b32: an SGPR coming back from its spill slot) needs five wait states in between.
    hipcc inserts them for its own instructions and cannot see into an inline-asm block: the asm has to carry its own
    `s_nop 4` (spec_chipx.h took stale bases without it and faulted)."""
    out = []
    recent = []                                     # (wait states since, sgpr) of the latest VALU writes of SGPRs
    in_asm = False
    for l in body:
        st = l.strip()
        if st.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if st.startswith(';;#ASMEND'):
            in_asm = False
            continue
        t = l.split(';')[0].strip()
        if not t or t.startswith('.') or t.endswith(':'):
            if t.endswith(':'):
                recent = []                         # (a label: what came before is another path's business)
            continue
        m = re.match(r'^s_nop\s+(\d+)', t)
        states = int(m.group(1)) + 1 if m else 1
        if in_asm and VMEM.match(t):
            sm = re.search(r'\bs\[(\d+):(\d+)\]\s*$', t.split('offset')[0].strip())
            if sm:
                used = set(range(int(sm.group(1)), int(sm.group(2)) + 1))
                for ws, reg in recent:
                    if reg in used and ws < 5:
                        out.append((name, 'sgpr', t, [reg, ws]))
        recent = [(ws + states, reg) for ws, reg in recent if ws + states < 5]
        w = VALU_SGPR_WRITE.match(t)
        if w:
            recent.append((0, int(w.group(2))))
    return out


def main():
    path = sys.argv[1]
    want = sys.argv[2:]
    lines = open(path).read().split('\n')
    total, bad, nk = 0, [], 0
    for name, body in kernels(lines):
        if not want or any(w in name for w in want):
            bad += sgpr_hazards(name, body)
    todo = [(name, body) for name, body in kernels(lines) if not want or any(w in name for w in want)]
    # the kernels are independent: one worker process per core (155 kernels of the fused sweeps take minutes in one)
    import multiprocessing as mp
    import os
    workers = max(1, min(len(todo), int(os.environ.get('CHECK_ISA_JOBS', '0')) or (os.cpu_count() or 1)))
    if workers > 1:
        with mp.get_context('fork').Pool(workers) as pool:
            results = pool.starmap(check, sorted(todo, key=lambda kb: -len(kb[1])), chunksize=1)
    else:
        results = [check(name, body) for name, body in todo]
    for n, pr in results:
        if n:
            nk += 1
        total += n
        bad += pr
    for b in bad[:40]:
        print('HAZARD', *b)
    print(f'{path}: {nk} kernel(s) with untracked loads, {total} of them checked on all paths, {len(bad)} hazard(s)')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
