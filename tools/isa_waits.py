"""Where a kernel waits for global loads: for every kernel of a hipcc -S file (or those matching a substring) the histogram of
s_waitcnt vmcnt(N) next to the number of global loads and MFMAs -- vmcnt(0) / vmcnt(1) in a loop that is meant to keep loads
in flight across iterations means the compiler lost their order (a load under a branch: it then waits for everything).
usage: isa_waits.py file.s [name-substring]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
name, stats = None, None
out = []
for l in txt:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        name = m.group(1)
        stats = dict(w=collections.Counter(), loads=0, mfma=0, bar=0)
        continue
    if name is None:
        continue
    t = l.strip()
    if t.startswith('.Lfunc_end'):
        if want in name:
            out.append((name, stats))
        name = None
        continue
    if t.startswith('global_load') or t.startswith('buffer_load'):
        stats['loads'] += 1
    elif t.startswith('v_mfma'):
        stats['mfma'] += 1
    elif t.startswith('s_barrier'):
        stats['bar'] += 1
    elif t.startswith('s_waitcnt'):
        m = re.search(r'vmcnt\((\d+)\)', t)
        if m:
            stats['w'][int(m.group(1))] += 1
for name, st in out:
    w = st['w']
    low = sum(v for k, v in w.items() if k <= 1)
    print(f"{name[:90]:90s} loads {st['loads']:5d} mfma {st['mfma']:5d} bar {st['bar']:3d} vmcnt<=1: {low:4d} of {sum(w.values()):4d}  max {max(w) if w else -1}")
