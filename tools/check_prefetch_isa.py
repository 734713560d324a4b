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


def check(name, body, width):
    """Find runs of >= 8 consecutive-ish asm loads (marked by ';APP' regions are not kept
    by hipcc -S, so identify them as `global_load_dwordxN ... off` groups of equal width
    that are followed by a counted vmcnt wait earlier in the same loop)."""
    loads = [i for i, l in enumerate(body) if re.search(r'global_load_dwordx%d\s' % width, l)]
    if not loads:
        return None
    # cluster loads that are close together
    groups, cur = [], [loads[0]]
    for i in loads[1:]:
        if i - cur[-1] <= 8:
            cur.append(i)
        else:
            groups.append(cur)
            cur = [i]
    groups.append(cur)
    problems = []
    checked = 0
    for g in groups:
        if len(g) < 8:
            continue
        # inside a loop?  next backward branch target label after the group
        dest = set()
        for i in g:
            m = re.search(r'global_load_dwordx\d+\s+v\[(\d+):(\d+)\]', body[i])
            dest.update(range(int(m.group(1)), int(m.group(2)) + 1))
        # scan forward until a backward branch (loop end) or s_endpgm
        labels = {re.match(r'^(\.LBB\w+):', l).group(1): k for k, l in enumerate(body)
                  if re.match(r'^(\.LBB\w+):', l)}
        end = None
        for k in range(g[-1] + 1, len(body)):
            m = re.search(r's_cbranch_\w+\s+(\.LBB\w+)', body[k]) or re.search(r's_branch\s+(\.LBB\w+)', body[k])
            if m and labels.get(m.group(1), 10**9) < g[0]:
                end = k
                break
        if end is None:
            continue            # not in a loop (prologue fetch): waited for by vmcnt(0) right after
        checked += 1
        for k in range(g[-1] + 1, end):
            l = body[k].strip()
            if not l or l.startswith(';') or l.startswith('.'):
                continue
            hit = regs_of(l) & dest
            if hit:
                problems.append((name, k, l, sorted(hit)))
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
