"""numpy model of the bf16x3 contraction (gemm_bf16x3.h): three exact bf16 pieces per float32 operand, the six leading piece
products, float32 accumulation per 16-deep k-block -- against a float32 FMA chain, both measured against float64.  Truncation
and round-to-nearest splits; the error is dominated by the accumulations (6 per 16 k-values against 16), not by the dropped
terms: bf16x3 comes out slightly MORE accurate than the float32 chain."""
import numpy as np
rng=np.random.default_rng(0)
def trunc_bf16(x):
    u=x.astype(np.float32).view(np.uint32)&np.uint32(0xffff0000); return u.view(np.float32)
def rne_bf16(x):
    u=x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r=((u+0x7fff+((u>>16)&1))>>16)<<16
    return r.astype(np.uint32).view(np.float32)
def split(x,f):
    p0=f(x); r1=(x-p0).astype(np.float32); p1=f(r1); r2=(r1-p1).astype(np.float32); p2=f(r2); return p0,p1,p2
def dot6(a,b,f,K):
    A=split(a,f); B=split(b,f)
    acc=np.zeros(a.shape[0],np.float32)
    for k0 in range(0,K,16):
        for (i,j) in [(2,0),(0,2),(1,1),(1,0),(0,1),(0,0)]:
            blk=(A[i][:,k0:k0+16].astype(np.float64)*B[j][:,k0:k0+16].astype(np.float64)).sum(1)
            acc=(acc.astype(np.float64)+blk).astype(np.float32)
    return acc
def dotf32(a,b,K):
    acc=np.zeros(a.shape[0],np.float32)
    for k in range(K):
        acc=(acc.astype(np.float64)+a[:,k].astype(np.float64)*b[:,k].astype(np.float64)).astype(np.float32)  # fma: one rounding
    return acc
for K in (512,3136):
    n=4000
    a=rng.standard_normal((n,K)).astype(np.float32); b=rng.standard_normal((n,K)).astype(np.float32)
    ref=(a.astype(np.float64)*b.astype(np.float64)).sum(1); mag=(np.abs(a.astype(np.float64)*b)).sum(1)
    for name,res in (("f32 fma chain",dotf32(a,b,K)),("bf16x3 trunc",dot6(a,b,trunc_bf16,K)),("bf16x3 rne",dot6(a,b,rne_bf16,K))):
        e=np.abs(res-ref)/mag
        print(K,name,"mean %.2e max %.2e"%(e.mean(),e.max()))
