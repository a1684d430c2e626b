"""The instruction sequence of a kernel in a hipcc -S listing as one letter per instruction, basic block by basic block
(development aid): M mfma, v VALU, L vector memory load, T store, d LDS, s SALU, w s_waitcnt, b barrier, j branch.
usage: python tools/isa_shape.py file.s 'sepconv_mfma2_rgb_kernelILi15E'"""
import re, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for m in re.finditer(r'^(\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not pat.search(name): continue
    print(name)
    code = body.split('.amdhsa_kernel')[0]
    cur, label = [], "entry"
    def out():
        if cur:
            s = "".join(cur)
            print(f"  {label:12s} {len(s):4d}  M={s.count('M'):3d} v={s.count('v'):3d}  {s}")
    for line in code.splitlines():
        lm = re.match(r'^(\.LBB\w+):', line)
        if lm:
            out(); cur, label = [], lm.group(1); continue
        im = re.match(r'^\s+([a-z]\w+)', line)
        if not im: continue
        op = im.group(1)
        if op.startswith('v_mfma') or op.startswith('v_smfma'): c = 'M'
        elif op.startswith('v_'): c = 'v'
        elif op.startswith('s_waitcnt'): c = 'w'
        elif op.startswith('s_barrier'): c = 'b'
        elif op.startswith('s_cbranch') or op.startswith('s_branch'): c = 'j'
        elif op.startswith('s_nop'): c = 'n'
        elif op.startswith('s_'): c = 's'
        elif op.startswith('ds_'): c = 'd'
        elif 'store' in op: c = 'T'
        elif op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_'): c = 'L'
        else: c = '?'
        cur.append(c)
    out()
