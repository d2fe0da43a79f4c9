#!/usr/bin/env python3
"""Which VGPRs does the main loop of a kernel keep live without ever writing them (hoisted
loop invariants)?  usage: tools/loop_regs.py file.s <kernel-name-substring>"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):', s, re.M)
i = m.start()
body = s[i:s.index('s_endpgm', i)]
lines = body.split('\n')
labels = {l.split(':')[0]: k for k, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
best = None
for k, l in enumerate(lines):  # backward branch spanning the most MFMAs
    m2 = re.search(r's_cbranch\w+ (\.LBB\d+_\d+)', l)
    if m2 and m2.group(1) in labels and labels[m2.group(1)] < k:
        h = labels[m2.group(1)]
        n = sum('v_mfma' in x for x in lines[h:k + 1])
        if best is None or n > best[0]:
            best = (n, h, k)
n, h, e = best
loop = lines[h:e + 1]


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


written, used = set(), set()
for l in loop:
    l = l.split(';')[0].strip()
    if not l or l.endswith(':') or l.startswith('.'):
        continue
    parts = l.split(None, 1)
    if len(parts) < 2:
        continue
    op, ops = parts[0], parts[1].split(',')
    if op.startswith(('ds_write', 'global_store', 's_', 'v_cmp', 'global_load_lds', 'buffer_store')) and not op.startswith('v_cmpx'):
        dst, src = set(), set().union(*[regs(o) for o in ops])
    else:
        dst = regs(ops[0])
        src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
    used |= src | dst
    written |= dst
inv = sorted(used - written)
print(f"loop: {len(loop)} lines, {n} MFMAs, {len(used)} VGPRs touched, {len(inv)} never written inside (hoisted invariants)")
print(inv)
