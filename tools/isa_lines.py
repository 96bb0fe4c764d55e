"""Static VALU/SALU/VMEM/LDS instruction counts of one kernel by source line, from the assembly hipcc writes with
-gline-tables-only -save-temps=obj.  usage: isa_lines.py <file.s> <kernel-name-substring> [top-n]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files, counts, kinds = {}, collections.Counter(), collections.Counter()
inside, loc = False, None
for line in open(path):
    t = line.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', t)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
        continue
    if re.match(r'^_Z\w+:', t):
        inside = key in t
        continue
    if not inside:
        continue
    if t.startswith('.Lfunc_end'):
        inside = False
        continue
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if m:
        loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'^(v_|s_|ds_|global_|buffer_|flat_|scratch_)(\w+)', t)
    if not m:
        continue
    kind = {'v_': 'valu', 's_': 'salu', 'ds_': 'lds'}.get(m.group(1), 'vmem')
    kinds[kind] += 1
    if kind == 'valu':
        counts[loc] += 1
print(dict(kinds))
byfile = collections.Counter()
for (f, l), c in counts.items():
    byfile[f] += c
print(dict(byfile))
for (f, l), c in counts.most_common(top):
    print(f"{c:5d}  {f}:{l}")
