"""Dev tool: per basic block of one kernel in a hipcc -S listing: MFMA / LDS / global / scratch instruction counts
(are the register spills inside the K loop or only around it?).  usage: isa_spills.py <file.s> <kernel-name-substring>"""
import re, sys
src = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):', src, re.M):
    name = m.group(1)
    body = src[m.end():]
    body = body[:body.index('s_endpgm')]
    cur, rows, order = 'entry', {}, ['entry']
    for l in body.split('\n'):
        lm = re.match(r'^(\.LBB\d+_\d+):', l)
        if lm:
            cur = lm.group(1); order.append(cur)
        t = l.strip().split(' ')[0] if l.strip() else ''
        d = rows.setdefault(cur, dict(n=0, mfma=0, ds=0, gload=0, gstore=0, sload=0, sstore=0))
        if t and not t.startswith(('.', ';')):
            d['n'] += 1
        for key, pre in (('mfma', 'v_mfma'), ('ds', 'ds_'), ('gload', 'global_load'), ('gstore', 'global_store'),
                         ('sload', 'scratch_load'), ('sstore', 'scratch_store')):
            if t.startswith(pre):
                d[key] += 1
    print(name)
    for k in order:
        d = rows[k]
        if d['mfma'] or d['sload'] + d['sstore'] >= 4 or d['gstore'] >= 4:
            print(f"   {k:12s} {d}")
