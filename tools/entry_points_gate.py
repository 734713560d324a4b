"""Regression gate over the per-entry-point timing logs (tools/entry_points_bench.py, tools/spec_sizes_bench.py,
tools/chain_shapes_bench.py, tools/next_rows_bench.py): compares the lines of a NEW log with the lines of the same name
in a BASE log and exits non-zero when any of them is slower by more than the tolerance (default 3 %).

    python tools/entry_points_gate.py BASE.log NEW.log [--tol 0.03] [--allow 'substring=reason' ...]

A line is  <name> <milliseconds> ms <rate> GB/s  (the name is everything in front of the time; for the facade line of
entry_points_bench.py the launch counts inside the name are part of it: a changed launch count is a changed line).
Both logs should come from ONE box lease (tools/ab_two_builds.sh, tools/entry_points_ab.sh run the two libraries in
turn): boxes differ by 5-10 %, more than the tolerance.  Lines only one log has are listed, not failed -- unless
--strict.  `--allow` exempts the lines whose name contains the substring and prints the reason next to them, so that
an accepted regression is written down where the gate runs (DESIGN.md section 8 quotes the same reasons).

Round 4 shipped a 37 % slower fused launch at the reference's default window (profiles/r03_entry_points.log ->
profiles/r04z5_entry_points.log, hipdsp_chain_forward 256/128: 10.68 -> 14.68 ms) because nothing read two of these
logs side by side (Missing 2)."""
import argparse
import re
import sys

LINE = re.compile(r'^(?P<name>.*?\S)\s+(?P<ms>\d+(?:\.\d+)?) ms\s+(?P<rate>\d+(?:\.\d+)?) GB/s\s*$')


def parse(text):
    """name -> milliseconds, in the order of the log; a name that occurs twice keeps its LAST time (a rerun)."""
    rows = {}
    for raw in text.splitlines():
        m = LINE.match(raw.rstrip())
        if m:
            name = re.sub(r'\s+', ' ', m.group('name').rstrip(':').strip())
            rows[name] = float(m.group('ms'))
    return rows


def compare(base, new, tol=0.03, allow=()):
    """-> (report lines, failed names).  `allow`: (substring, reason) pairs."""
    report, failed = [], []
    for name, ms in new.items():
        if name not in base:
            report.append(f'  new   {ms:9.3f} ms            {name}')
            continue
        b = base[name]
        ratio = ms/b if b > 0 else float('inf')
        mark = 'ok'
        if ratio > 1.0 + tol:
            why = next((r for s, r in allow if s in name), None)
            if why is None:
                mark = 'SLOWER'
                failed.append(name)
            else:
                mark = 'allowed'
                name = f'{name}   [{why}]'
        elif ratio < 1.0 - tol:
            mark = 'faster'
        report.append(f'  {mark:7s} {b:9.3f} -> {ms:9.3f} ms  {100.0*(ratio - 1.0):+6.1f} %  {name}')
    gone = [n for n in base if n not in new]
    for n in gone:
        report.append(f'  gone  {base[n]:9.3f} ms            {n}')
    return report, failed, gone


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('base')
    ap.add_argument('new')
    ap.add_argument('--tol', type=float, default=0.03)
    ap.add_argument('--allow', action='append', default=[], metavar='SUBSTRING=REASON')
    ap.add_argument('--strict', action='store_true', help='a line of BASE that NEW lacks fails the gate too')
    ap.add_argument('--also-base', action='append', default=[], metavar='LOG', help='a repeat of BASE: the faster of the times counts')
    ap.add_argument('--also-new', action='append', default=[], metavar='LOG', help='a repeat of NEW: the faster of the times counts')
    a = ap.parse_args(argv)
    allow = []
    for item in a.allow:
        sub, _, why = item.partition('=')
        allow.append((sub, why or 'accepted'))
    def best_of(paths):
        rows = {}
        for path in paths:
            with open(path) as f:
                for name, ms in parse(f.read()).items():
                    rows[name] = min(ms, rows.get(name, ms))
        return rows
    # (short launches scatter by several per cent from run to run: repeats of a log may be given, the faster time counts)
    base = best_of([a.base] + a.also_base)
    new = best_of([a.new] + a.also_new)
    if not base or not new:
        print(f'entry_points_gate: no timing lines in {a.base if not base else a.new}')
        return 2
    report, failed, gone = compare(base, new, a.tol, allow)
    print(f'entry_points_gate: {a.base} -> {a.new}, tolerance {100*a.tol:.1f} %')
    print('\n'.join(report))
    bad = len(failed) + (len(gone) if a.strict else 0)
    print(f'entry_points_gate: {len(new)} lines, {len(failed)} slower than the tolerance'
          + (f', {len(gone)} missing' if gone else '') + (' -- FAILED' if bad else ' -- passed'))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
