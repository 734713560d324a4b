"""Build-time guard for the hand-counted prefetches (sos.hip, spectrogram.hip).

The inline-asm loads write their destination VGPRs asynchronously; until the counted
`s_waitcnt vmcnt(N)` of the next iteration nothing may read or write those registers.
hipcc does not know that, so this script re-checks the generated ISA: for every kernel it
finds the block of asm loads inside the main loop and verifies that no instruction between
the last of them and the end of the loop body (and none from the loop head up to the
counted wait) touches their destination registers.

usage: python tools/check_prefetch_isa.py file.s [kernel-substring ...]
"""
import re
import sys


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
        if name and 's_endpgm' in l:
            yield name, lines[start:i + 1]
            name = None


VMEM = re.compile(r'^\s*(global_|buffer_|scratch_|flat_)(load|store|atomic)')


def check(name, body, width):
    """Groups of >= 4 close-together `global_load_dwordxN ... off` inside a loop are taken
    to be a hand-counted prefetch.  From the last load of a group the scan walks forward
    (following the loop's back edge once), counting the younger VMEM operations, until an
    `s_waitcnt vmcnt(k)` with k <= that count retires the loads; any instruction touching
    the destination registers before that point is reported."""
    loads = [i for i, l in enumerate(body) if re.search(r'global_load_dwordx%d\s' % width, l)]
    if not loads:
        return None
    groups, cur = [], [loads[0]]
    for i in loads[1:]:
        if i - cur[-1] <= 8:
            cur.append(i)
        else:
            groups.append(cur)
            cur = [i]
    groups.append(cur)
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r'^(\.LBB\w+):', l)
        if m:
            labels[m.group(1)] = k
    problems, checked = [], 0
    for g in groups:
        if len(g) < 4:
            continue
        dest = set()
        for i in g:
            m = re.search(r'global_load_dwordx\d+\s+v\[(\d+):(\d+)\]', body[i])
            dest.update(range(int(m.group(1)), int(m.group(2)) + 1))
        # in-order model of the vector-memory queue from the first load of the group on
        k, jumped, steps = g[0], False, 0
        queue = []                               # destination register sets, oldest first
        seen_group = False
        while k < len(body) and steps < 40000:
            steps += 1
            l = body[k].strip()
            m = re.search(r's_waitcnt.*vmcnt\((\d+)\)', l)
            if m:
                n = int(m.group(1))
                queue = queue[-n:] if n > 0 else []
            if seen_group and not any(q & dest for q in queue):
                break                            # every load of the group has been retired
            if 's_endpgm' in l:
                break
            br = re.search(r's_cbranch_\w+\s+(\.LBB\w+)', l) or re.search(r's_branch\s+(\.LBB\w+)', l)
            if br and labels.get(br.group(1), 10**9) < g[0] and not jumped and k > g[-1]:
                jumped = True                    # loop back edge: continue at the loop head
                k = labels[br.group(1)]
                continue
            if l and not l.startswith(';') and not l.startswith('.'):
                pending = set().union(*queue) if queue else set()
                ld = re.search(r'^(global|scratch|buffer)_load_\w+\s+v\[(\d+):(\d+)\]', l) or \
                    re.search(r'^(global|scratch|buffer)_load_\w+\s+v(\d+)()\b', l)
                used = regs_of(l)
                if ld:
                    lo_ = int(ld.group(2))
                    hi_ = int(ld.group(3)) if ld.group(3) else lo_
                    d = set(range(lo_, hi_ + 1))
                    src = regs_of(l.split(',', 1)[1]) if ',' in l else set()
                    if (src | d) & pending & dest:
                        problems.append((name, k, l, sorted((src | d) & pending & dest)))
                    queue.append(d)
                    if k in g:
                        seen_group = seen_group or k == g[-1]
                else:
                    if used & pending & dest:
                        problems.append((name, k, l, sorted(used & pending & dest)))
                    if VMEM.match(l):
                        queue.append(set())
            k += 1
        checked += 1
    return checked, problems


def main():
    path = sys.argv[1]
    want = sys.argv[2:]
    lines = open(path).read().split('\n')
    total, bad = 0, []
    for name, body in kernels(lines):
        if want and not any(w in name for w in want):
            continue
        for width in (2, 4):
            r = check(name, body, width)
            if r:
                total += r[0]
                bad += r[1]
    for b in bad:
        print('HAZARD', *b)
    print(f'{path}: {total} prefetch group(s) checked, {len(bad)} hazard(s)')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
