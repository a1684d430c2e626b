"""Instruction mix of the kernels in a hipcc -S listing whose mangled name matches a regex.
usage: python tools/isa_stats.py file.s 'sepconv_march_kernelILi3ELi2E' [top]"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 22
for m in re.finditer(r'^(\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not pat.search(name):
        continue
    code = body.split('.amdhsa_kernel')[0]
    ins = re.findall(r'^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+)', code, re.M)
    c = Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print(f"{name}\n  total {len(ins)}  valu {valu}")
    for k in ('next_free_vgpr', 'next_free_sgpr', 'group_segment_fixed_size', 'private_segment_fixed_size'):
        g = re.search(r'\.amdhsa_%s (\S+)' % k, body)
        if g: print(f"  {k} {g.group(1)}")
    g = re.search(r'%s\.num_vgpr, (\d+)' % re.escape(name), txt)
    if g: print("  num_vgpr", g.group(1))
    print("  " + ", ".join(f"{k}:{v}" for k, v in c.most_common(top)))
