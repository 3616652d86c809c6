"""Skeleton of a kernel's MFMA region from a hipcc -S file: runs of instruction classes between the first and the last MFMA
(GL global load, GS store, dsr / dsw, MFMA, V / S other vector / scalar ops, waits and branches verbatim).
usage: isa_loop.py file.s <kernel-name-substring> [max chars]"""
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 6000
on, rows = False, []
for l in txt:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        on = want in m.group(1)
        continue
    if not on:
        continue
    t = l.strip()
    if t.startswith('.Lfunc_end'):
        break
    if not t or t.startswith(';'):
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): k = 'MFMA'
    elif op.startswith('global_load') or op.startswith('buffer_load'): k = 'GL'
    elif op.startswith('global_store'): k = 'GS'
    elif op.startswith('ds_read'): k = 'dsr'
    elif op.startswith('ds_write'): k = 'dsw'
    elif op.startswith('s_waitcnt'): k = t.split(';')[0].replace('s_waitcnt ', 'W:').replace(' ', '')
    elif op.startswith('s_barrier'): k = 'BAR'
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): k = op.replace('s_cbranch_', 'br_').replace('s_branch', 'jmp') + '>' + t.split()[1].replace('.LBB', '')
    elif op.startswith('.LBB'): k = op.replace('.LBB', 'L')
    elif op.startswith('v_'): k = 'V'
    elif op.startswith('s_'): k = 'S'
    else: k = op
    rows.append(k)
first = next(i for i, k in enumerate(rows) if k == 'MFMA')
last = len(rows) - 1 - next(i for i, k in enumerate(reversed(rows)) if k == 'MFMA')
rows = rows[max(0, first - 40):last + 2]
out, prev, n = [], None, 0
for k in rows + [None]:
    if k == prev:
        n += 1
        continue
    if prev is not None:
        out.append(f"{n}{prev}" if n > 1 or prev in ('V', 'S', 'MFMA', 'GL', 'dsr', 'dsw') else prev)
    prev, n = k, 1
print(' '.join(out)[:lim])
