"""Static instruction counts of the instrumented render kernel's sections, from the assembly hipcc writes with
-save-temps=obj: the code between two section clocks (SectionClock::lap<K>, marked '; crt_section_end K') in layout order.
usage: isa_sections.py <file.s> [kernel-name-substring]"""
import re, sys
path = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else 'render_kernelILb1ELi16ELb0E'
names = ["votes", "hit+shade", "store+fetch", "philox", "scatter", "camera", "spheres+root box", "node steps", "leaf steps"]
inside, cur = False, dict(valu=0, salu=0, mem=0)
out = []
for line in open(path):
    t = line.strip()
    if re.match(r'^_Z\w+:', t):
        inside = key in t
        continue
    if not inside:
        continue
    if t.startswith('.Lfunc_end'):
        break
    m = re.match(r'; crt_section_end (\d+)', t)
    if m:
        out.append((names[int(m.group(1))], cur))
        cur = dict(valu=0, salu=0, mem=0)
        continue
    if t.startswith('v_'):
        cur['valu'] += 1
    elif t.startswith('s_'):
        cur['salu'] += 1
    elif re.match(r'^(global_|ds_|buffer_|flat_|scratch_)', t):
        cur['mem'] += 1
out.append(("(after the last clock)", cur))
for name, c in out:
    print(f"{name:24s} valu {c['valu']:4d}  salu {c['salu']:4d}  mem {c['mem']:3d}")
