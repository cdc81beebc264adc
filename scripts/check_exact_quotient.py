import numpy as np, sys
n=int(sys.argv[1])
y=np.float64(1.0)/np.float64(n)
bad=0
for c in range(256):
    bits=np.arange(c<<24,(c+1)<<24,dtype=np.uint64).astype(np.uint32)
    x=bits.view(np.float32)
    with np.errstate(all='ignore'):
        a=(x/np.float32(n))
        b=(x.astype(np.float64)*y).astype(np.float32)
    ne=(a.view(np.uint32)!=b.view(np.uint32))&~(np.isnan(a)&np.isnan(b))
    bad+=int(ne.sum())
    if ne.any() and bad<20: print(n,'mismatch',x[ne][:3],a[ne][:3],b[ne][:3])
print('n',n,'mismatches',bad)
