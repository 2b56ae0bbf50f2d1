"""Static VALU/SALU/LDS/VMEM instruction counts of one kernel by source line, from `hipcc -S -gline-tables-only` output.
usage: python tools/isa_lines.py build/kernels_dbg.s k_encode_persistentILi4E [bucket-size]"""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
files = {}
cur = None
on = False
cnt = collections.Counter()
ops = collections.defaultdict(collections.Counter)
for line in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
    if re.match(r'^_Z.*:', line):
        on = kern in line
        continue
    if not on:
        continue
    if '.end_amdhsa_kernel' in line or line.startswith('.Lfunc_end'):
        on = False
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'\s+([a-z_0-9]+)\s', line)
    if not m or line.strip().startswith('.') or line.strip().startswith(';'):
        continue
    op = m.group(1)
    kind = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other'
    cnt[(cur, kind)] += 1
    ops[cur][op] += 1
lines = sorted(set(k[0] for k in cnt), key=lambda x: (x[0], x[1]))
tot = collections.Counter()
for l in lines:
    v = {k: cnt[(l, k)] for k in ('valu', 'salu', 'lds', 'vmem')}
    for k in v: tot[k] += v[k]
    top = ", ".join("%s x%d" % (o, c) for o, c in ops[l].most_common(4))
    print("%-28s %5d  valu %4d salu %4d lds %3d vmem %3d   %s" % (l[0], l[1], v['valu'], v['salu'], v['lds'], v['vmem'], top))
print("total", dict(tot))
