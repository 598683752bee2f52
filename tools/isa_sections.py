"""Static instruction counts of one kernel of a hipcc -g -save-temps assembly listing, attributed to source-line ranges (sections).
usage: python tools/isa_sections.py <file.s> <mangled kernel name> <source file name> <sections file: python list of (name, first line, last line)> [--mn]"""
import re,sys,collections
S=sys.argv[1]; kern=sys.argv[2]; srcfile=sys.argv[3]
# sections: list of (name, lo, hi) by source line in srcfile
sections=eval(open(sys.argv[4]).read()) if len(sys.argv)>4 else []
lines=open(S,errors='replace').read().split('\n')
# file table
files={}
for l in lines:
    m=re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',l)
    if m: files[int(m.group(1))]=(m.group(3) or m.group(2))
start=None
for i,l in enumerate(lines):
    if l.startswith(kern+':'): start=i; break
end=None
for i in range(start,len(lines)):
    if lines[i].startswith('.Lfunc_end'): end=i; break
cur=('?',0)
def cls(m):
    if m.startswith('v_'): 
        if m.startswith('v_readlane') or m.startswith('v_readfirstlane') or m.startswith('v_writelane'): return 'VALU'
        return 'VALU'
    if m.startswith('s_waitcnt'): return 'WAIT'
    if m.startswith('s_'): return 'SALU'
    if m.startswith('ds_'): return 'LDS'
    if m.startswith('global_') or m.startswith('flat_') or m.startswith('buffer_'): return 'VMEM'
    return 'OTHER'
by=collections.defaultdict(lambda: collections.Counter())
mn=collections.defaultdict(lambda: collections.Counter())
for l in lines[start:end]:
    t=l.strip()
    m=re.match(r'\.loc\s+(\d+)\s+(\d+)',t)
    if m: cur=(files.get(int(m.group(1)),'?'),int(m.group(2))); continue
    if not t or t.startswith('.') or t.startswith(';') or t.endswith(':'): continue
    op=t.split()[0]
    f,ln=cur
    sec='other:'+f.split('/')[-1]
    if f.endswith(srcfile):
        sec='unsectioned'
        for name,lo,hi in sections:
            if lo<=ln<=hi: sec=name;break
    by[sec][cls(op)]+=1
    mn[sec][op]+=1
tot=collections.Counter()
for sec in sorted(by):
    print("%-28s"%sec, dict(by[sec])); tot.update(by[sec])
print("TOTAL",dict(tot))
if '--mn' in sys.argv:
    for sec in sorted(mn):
        print(sec, mn[sec].most_common(12))
