import itertools, sys, random
def model(nrb,ncb,r,TB,order_mode="md",seed=0,verbose=False, custom_order=None):
    edges=[];hid={};vid={}
    for p in range(1,nrb):
        for q in range(ncb):
            hid[(p,q)]=len(edges); edges.append((0,p,q,(p-1)*ncb+q,p*ncb+q))
    for q in range(1,ncb):
        for p in range(nrb):
            vid[(p,q)]=len(edges); edges.append((1,p,q,p*ncb+(q-1),p*ncb+q))
    E=len(edges)
    crosses=[(p,q) for p in range(1,nrb) for q in range(1,ncb)]
    xid={c:i for i,c in enumerate(crosses)}
    bside=[]
    for p in range(nrb):
        for q in range(ncb):
            bside.append([hid.get((p,q),-1) if p>=1 else -1, hid.get((p+1,q),-1) if p+1<nrb else -1,
                          vid.get((p,q),-1) if q>=1 else -1, vid.get((p,q+1),-1) if q+1<ncb else -1])
    xc=[]
    for e,(hv,p,q,b0,b1) in enumerate(edges):
        if hv==0:
            if q>=1: xc.append((xid[(p,q)],e))
            if q+1<ncb: xc.append((xid[(p,q+1)],e))
        else:
            if p>=1: xc.append((xid[(p,q)],e))
            if p+1<nrb: xc.append((xid[(p+1,q)],e))
    adj=[set() for _ in range(E)]
    for s in bside:
        for x in s:
            for y in s:
                if x!=y and x>=0 and y>=0: adj[x].add(y)
    is_pre=[0]*E; busy=[0]*(nrb*ncb)
    for e in range(E):
        if not busy[edges[e][3]] and not busy[edges[e][4]]:
            is_pre[e]=1; busy[edges[e][3]]=busy[edges[e][4]]=1
    g=[set() for _ in range(E)]
    for e in range(E):
        if not is_pre[e]:
            for x in adj[e]:
                if not is_pre[x]: g[e].add(x)
    for e in range(E):
        if is_pre[e]:
            for x in adj[e]:
                for y in adj[e]:
                    if x!=y and not is_pre[x] and not is_pre[y]: g[x].add(y)
    g0=[set(s) for s in g]
    act=[e for e in range(E) if not is_pre[e]]
    if custom_order is not None: order=custom_order
    else:
        gg=[set(s) for s in g]; done=set(); order=[]
        for step in range(len(act)):
            best=min((e for e in act if e not in done), key=lambda e:(len(gg[e]),e))
            done.add(best); order.append(best)
            nb=list(gg[best])
            for x in nb:
                gg[x].discard(best)
                for y in nb:
                    if x!=y: gg[x].add(y)
    # scalar layout
    pos={}; n=0; ordof={e:i for i,e in enumerate(order)}
    xhost={}
    for x,e in xc:
        if not is_pre[e] and (x not in xhost or ordof[e]>ordof[xhost[x]]): xhost[x]=e
    xpos={}
    for e in order:
        pos[e]=(n,n+r); n+=r
        for x in range(len(crosses)):
            if xhost.get(x)==e: xpos[x]=n; n+=1
    for x in range(len(crosses)):
        if x not in xpos: xpos[x]=n; n+=1
    T=(n+TB-1)//TB
    mask=[[0]*T for _ in range(T)]
    def mark(r0,r1,c0,c1):
        for tr in range(r0//TB,(r1-1)//TB+1):
            for tc in range(c0//TB,(c1-1)//TB+1): mask[tr][tc]=mask[tc][tr]=1
    for t in range(T): mask[t][t]=1
    for e in order:
        mark(*pos[e],*pos[e])
        for y in g0[e]: mark(*pos[e],*pos[y])
    # cross couplings: cross x couples to edges adjacent (incl. pre edges' neighbours -> approximated: all active edges adjacent to cross or adjacent to a pre edge at that cross)
    for x,e in xc:
        es=[e] if not is_pre[e] else [y for y in adj[e] if not is_pre[y]]
        for y in es: mark(xpos[x],xpos[x]+1,*pos[y])
        for x2,e2 in xc:
            if e2==e or (is_pre[e] and e2 in adj[e]) : mark(xpos[x],xpos[x]+1,xpos[x2],xpos[x2]+1)
    for k in range(T):
        for i in range(k+1,T):
            if mask[i][k]:
                for j in range(k+1,i+1):
                    if mask[j][k]: mask[i][j]=mask[j][i]=1
    fl=0; nsl=0
    for j in range(T):
        for i in range(j,T):
            if mask[i][j]:
                nsl+=1
                for k in range(j):
                    if mask[i][k] and mask[j][k]: fl+=2*TB**3
                fl+= TB**3/3 if i==j else 2*TB**3
    return n,T,nsl,fl/1e6,order
for (nrb,ncb,r) in ((4,4,43),(3,3,39)):
    for TB in (64,32,16):
        n,T,nsl,fl,order=model(nrb,ncb,r,TB)
        print(nrb,ncb,"TB",TB,"n",n,"T",T,"slots",nsl,"MF",round(fl,1),"dense",round(n**3/3e6,1))
