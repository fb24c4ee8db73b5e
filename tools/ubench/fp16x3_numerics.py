import numpy as np
rng=np.random.default_rng(0)
def split_bf16x3(x):
    def trunc(v):
        u=v.astype(np.float32).view(np.uint32) & np.uint32(0xffff0000)
        return u.view(np.float32)
    h=trunc(x); r=x-h; m=trunc(r); l=trunc(r-m)
    return h,m,l
def split_f16(x):
    h=x.astype(np.float16).astype(np.float32)
    l=(x-h).astype(np.float16).astype(np.float32)
    return h,l
def gemm_bf16x6(A,B):
    ah,am,al=split_bf16x3(A); bh,bm,bl=split_bf16x3(B)
    acc=np.zeros((A.shape[0],B.shape[1]),np.float32)
    for (p,q) in [(al,bh),(ah,bl),(am,bm),(am,bh),(ah,bm),(ah,bh)]:
        acc=acc+(p.astype(np.float64)@q.astype(np.float64)).astype(np.float32)
    return acc
def gemm_f16x3(A,B,rowscale=True):
    # per-row scale of A, per-column scale of B (powers of two), RN split, three products
    def p2(v): 
        e=np.floor(np.log2(np.maximum(v,1e-38)))
        return np.exp2(13-e).astype(np.float32)
    sa=p2(np.abs(A).max(1,keepdims=True)) if rowscale else p2(np.abs(A).max())*np.ones((A.shape[0],1),np.float32)
    sb=p2(np.abs(B).max(0,keepdims=True))
    ah,al=split_f16(A*sa); bh,bl=split_f16(B*sb)
    assert np.isfinite(ah).all() and np.isfinite(bh).all()
    acc=np.zeros((A.shape[0],B.shape[1]),np.float32)
    for (p,q) in [(al,bh),(ah,bl),(ah,bh)]:
        acc=acc+(p.astype(np.float64)@q.astype(np.float64)).astype(np.float32)
    return acc/(sa*sb)
def fp32chain(A,B):
    acc=np.zeros((A.shape[0],B.shape[1]),np.float32)
    for k in range(A.shape[1]):
        acc=(acc+A[:,k:k+1]*B[k:k+1,:]).astype(np.float32)
    return acc
def report(name,A,B):
    ref=A.astype(np.float64)@B.astype(np.float64)
    rms=lambda y: np.sqrt(((y-ref)**2).mean())/np.sqrt((ref**2).mean())
    rowmax=lambda y: (np.sqrt(((y-ref)**2).sum(1))/np.sqrt((ref**2).sum(1))).max()
    y6=gemm_bf16x6(A,B); y3=gemm_f16x3(A,B); y3t=gemm_f16x3(A,B,rowscale=False); yc=fp32chain(A,B)
    print("%-34s bf16x6 %.2e (row max %.2e) | f16x3 row-scaled %.2e (%.2e) | f16x3 tensor-scaled %.2e (%.2e) | fp32 chain %.2e (%.2e)"%(name,rms(y6),rowmax(y6),rms(y3),rowmax(y3),rms(y3t),rowmax(y3t),rms(yc),rowmax(yc)))
M,K,F=256,1024,128
A=rng.standard_normal((M,K)).astype(np.float32); B=(0.05*rng.standard_normal((K,F))).astype(np.float32)
report("normal x normal*0.05",A,B)
A2=(A*np.exp2(rng.integers(-20,4,size=(M,1)))).astype(np.float32)
report("rows scaled 2^-20..2^3",A2,B)
A3=(A*np.exp2(rng.integers(-12,1,size=(M,K)))).astype(np.float32)
report("elements scaled 2^-12..1",A3,B)
A4=np.maximum(A,0)*np.float32(1e-6)
report("relu * 1e-6 (gradient-like)",A4,B)
A5=A.copy(); A5[:,0]*=1e4
report("one huge column (1e4)",A5,B)
B2=(B*np.exp2(rng.integers(-10,1,size=(K,1)))).astype(np.float32)
report("weights rows 2^-10..1",A,B2)
