"""Basic blocks of an assembly listing between two line numbers with their instruction counts by class (which blocks are on the hot path?).
usage: python tools/isa_basic_blocks.py <file.s> <first line> <last line>"""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
lo,hi=int(sys.argv[2]),int(sys.argv[3])
# split into basic blocks at labels and after branches
blocks=[]; cur={'start':lo,'label':None,'ins':[]}
for i in range(lo-1,hi):
    t=lines[i].strip()
    if not t or t.startswith(';') or t.startswith('.loc') or t.startswith('.Ltmp'): continue
    m=re.match(r'(\.LBB\d+_\d+):',t)
    if m:
        if cur['ins'] or cur['label']: blocks.append(cur)
        cur={'start':i+1,'label':m.group(1),'ins':[]}; continue
    if t.startswith('.') : continue
    cur['ins'].append(t)
    if t.startswith('s_cbranch') or t.startswith('s_branch'):
        blocks.append(cur); cur={'start':i+2,'label':None,'ins':[]}
blocks.append(cur)
def cnt(b,p): return sum(1 for x in b['ins'] if x.startswith(p))
for b in blocks:
    last=b['ins'][-1] if b['ins'] else ''
    print("%5d %-12s valu=%3d salu=%3d lds=%2d vmem=%2d  %s"%(b['start'],b['label'] or '',cnt(b,'v_'),cnt(b,'s_'),cnt(b,'ds_'),cnt(b,'global_')+cnt(b,'flat_'),last if last.startswith('s_c') or last.startswith('s_b') else ''))
