"""Opcode-class sequence of the basic blocks of a gfx950 .s file that hold >= N MFMAs (M mfma, T transcendental, v VALU,
c cvt_pk_bf16, L LDS, G global/scratch, w waitcnt, s scalar, n nop).  Usage: isa_seq.py file.s [min_mfma]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
minm = int(sys.argv[2]) if len(sys.argv) > 2 else 30
blocks = []; cur = None
for l in lines:
    if re.match(r'^\.LBB\d+_\d+:', l) or re.match(r'^_Z\w+:', l):
        cur = [l, []]; blocks.append(cur)
    elif cur is not None and l.startswith('\t') and not l.startswith('\t.'):
        cur[1].append(l.strip())
for name, ins in blocks:
    n = sum('v_mfma' in x for x in ins)
    if n < minm: continue
    seq = []
    for x in ins:
        op = x.split()[0]
        if 'mfma' in op: c = 'M'
        elif op.split('_e')[0] in ('v_exp_f32', 'v_rcp_f32', 'v_sqrt_f32', 'v_rsq_f32', 'v_log_f32'): c = 'T'
        elif op.startswith('v_cvt_pk'): c = 'c'
        elif op.startswith('ds_'): c = 'L'
        elif op.startswith('s_waitcnt'): c = 'w'
        elif op.startswith('s_nop'): c = 'n'
        elif op.startswith('s_'): c = 's'
        elif op.startswith('v_'): c = 'v'
        elif op.startswith(('global', 'scratch', 'buffer', 'flat')): c = 'G'
        else: c = '?'
        seq.append(c)
    print(name, len(ins), 'instrs, mfma', n); print(''.join(seq))
