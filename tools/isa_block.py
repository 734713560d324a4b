"""Print one basic block (or a class summary of every block of at least MIN instructions) of a kernel of a hipcc listing.
    python tools/isa_block.py FILE.s KERNEL_KEY [.LBBn_m | --sizes MIN] [--filter REGEX]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labs = [i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)] + [len(body)]
flt = next((a.split('=', 1)[1] for a in sys.argv if a.startswith('--filter=')), None)
if len(sys.argv) > 3 and sys.argv[3].startswith('.LBB'):
    i0 = next(i for i in labs[:-1] if body[i].startswith(sys.argv[3] + ':'))
    i1 = labs[labs.index(i0) + 1]
    for l in body[i0:i1]:
        t = l.strip()
        if t and not t.startswith(';') and (flt is None or re.search(flt, t)):
            print(l)
else:
    mn = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    for a, b in zip(labs[:-1], labs[1:]):
        ins = [l.strip() for l in body[a + 1:b] if l.strip() and not l.strip().startswith((';', '.'))]
        if len(ins) >= mn:
            print(body[a].split(':')[0], len(ins))
