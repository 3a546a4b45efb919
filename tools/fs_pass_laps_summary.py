import re, sys
passes=[]; cur=[]
for l in open(sys.argv[1]):
    if l.startswith("PASS"):
        passes.append((float(l.split()[2]), cur)); cur=[]
    elif "[bath timing]" in l:
        m=re.match(r"\[bath timing\] (.*?)\s+([0-9.]+) ms", l.strip())
        if m: cur.append((m.group(1).strip(), float(m.group(2))))
passes=passes[2:]
names=[]
for t,c in passes:
    for n,_ in c:
        if n not in names: names.append(n)
tot=sorted(t for t,_ in passes); med=tot[len(tot)//2]
fast=[c for t,c in passes if t<=med+2]; slow=[c for t,c in passes if t>med+5]
print("passes",len(passes),"median",med,"slow",len(slow))
def avg(cs,n):
    v=[dict(c).get(n,0.0) for c in cs]; return sum(v)/max(1,len(v))
for n in names:
    a,b=avg(fast,n),avg(slow,n)
    if abs(a-b)>0.5: print("%-90s fast %.2f slow %.2f"%(n[:90],a,b))
