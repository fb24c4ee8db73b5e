import sqlite3, re, collections, sys
c=sqlite3.connect(sys.argv[1])
rows=list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
idx=[i for i,r in enumerate(rows) if 'edge_fwd_kernel' in r[0]]
a,b=idx[-2],idx[-1]
step=rows[a:b]
def short(n):
    n=n.replace('(anonymous namespace)::','').replace('void ','').replace('at::native::','')
    n=re.sub(r'\((?!anonymous).*','',n)
    return n[:70]
if len(sys.argv)>2 and sys.argv[2]=='seq':
    for i,r in enumerate(step): print('%3d %8.1f  %-55s grid %d'%(i,(r[2]-r[1])/1e3,short(r[0])[:55],r[3]//max(r[4],1)))
    sys.exit()
agg=collections.OrderedDict()
for r in step:
    k=short(r[0]); d=(r[2]-r[1])/1e3
    a_=agg.setdefault(k,[0,0.0]); a_[0]+=1; a_[1]+=d
tot=sum(v[1] for v in agg.values())
print('total',round(tot,1),'us; dispatches',len(step))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[3]) if len(sys.argv)>3 else 26]:
    print('%4d %9.1f us %6.1f avg  %s'%(v[0],v[1],v[1]/v[0],k))
