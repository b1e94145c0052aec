#!/usr/bin/env python
"""Summary of tools/wgrad_micro.py outputs: per layer the automatic choice against the best forced one (and an older run)."""
import re, sys
def parse(fn):
    rows=[]
    for ln in open(fn):
        m=re.match(r'(\S+)\s+(\d+)x\d+\s+k\s*(\d+) n\s*(\d+)\s+cfg\s+(\d+)\s+(\S+)\s+ksplit\s+(\d+)\s+kernel\s+([\d.]+) us\s+reduce\s+([\d.]+) us\s+pair\s+([\d.]+)',ln)
        if m: rows.append((m.group(1),int(m.group(5)),m.group(6),int(m.group(7)),float(m.group(8)),float(m.group(9)),float(m.group(10))))
    return rows
newf=sys.argv[1]; oldf=sys.argv[2] if len(sys.argv)>2 else None
old={r[0]:r for r in parse(oldf) if r[1]==0} if oldf else {}
by={}
for r in parse(newf): by.setdefault(r[0],[]).append(r)
to=ta=tb=0
for l,rs in by.items():
    o=old.get(l,(0,0,'',0,0,0,0)); auto=[r for r in rs if r[1]==0][0]; best=min(rs,key=lambda r:r[4]+r[5])
    to+=o[4]+o[5]; ta+=auto[4]+auto[5]; tb+=best[4]+best[5]
    print('%-8s old %5.1f+%4.1f (%5.1f TF) | auto %-22s ks%3d %5.1f+%4.1f (%5.1f TF) | best cfg %3d %-22s ks%3d %5.1f+%4.1f (%5.1f TF)'%(l,o[4],o[5],o[6],auto[2],auto[3],auto[4],auto[5],auto[6],best[1],best[2],best[3],best[4],best[5],best[6]))
print('sum old %.1f auto %.1f best %.1f'%(to,ta,tb))
