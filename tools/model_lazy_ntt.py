# integer model of the lazy-range radix-4 butterfly (29-bit limbs, N = 9): checks limb ranges, column sums, value bounds
import random
MASK=(1<<29)-1
N=9
R=1<<(29*N)
def limbs(v):  # normalised
    out=[(v>>(29*i))&MASK for i in range(N-1)]
    out.append(v>>(29*(N-1)))
    return out
def val(l): return sum(x<<(29*i) for i,x in enumerate(l))
class F:
    def __init__(s,p):
        s.p=p; s.M=limbs(p); s.INV=(-pow(p,-1,1<<29))%(1<<29)
        s.maxcol=0
    def mul(s,a,b):
        # a, b limb lists (possibly un-normalised); product scanning as in field.hip.h
        acc=0; m=[0]*N; r=[0]*N
        for k in range(N):
            for i in range(k+1): acc+=a[i]*b[k-i]
            for i in range(k): acc+=m[i]*s.M[k-i]
            s.maxcol=max(s.maxcol,acc)
            m[k]=((acc&0xFFFFFFFF)*s.INV)&MASK
            acc+=m[k]*s.M[0]
            s.maxcol=max(s.maxcol,acc)
            assert acc < 1<<64
            assert acc&MASK==0
            acc>>=29
        for k in range(N,2*N-1):
            for i in range(k-N+1,N): acc+=a[i]*b[k-i]
            for i in range(k-N+1,N): acc+=m[i]*s.M[k-i]
            s.maxcol=max(s.maxcol,acc)
            assert acc < 1<<64
            r[k-N]=acc&MASK; acc>>=29
        r[N-1]=acc
        assert acc < 1<<32
        return r
    def kp(s,k): return limbs(k*s.p)
    def bp(s,k,bits):
        c=s.kp(k)
        return [c[i]+((1<<bits) if i<N-1 else 0)-((1<<(bits-29)) if i>0 else 0) for i in range(N)]
def lz_add(a,b):
    r=[x+y for x,y in zip(a,b)]
    assert all(x< 1<<32 for x in r); return r
def lz_sub(f,a,b,k,bits):
    c=f.bp(k,bits)
    r=[x+y-z for x,y,z in zip(a,c,b)]
    assert all(0<=x< 1<<32 for x in r),(r,); return r
def lz_norm(a):
    c=0; r=[]
    for i in range(N-1):
        s=a[i]+c; assert s< 1<<32; r.append(s&MASK); c=s>>29
    r.append(a[N-1]+c); assert r[-1] < 1<<32
    return r
def lz_reduce(f,a,unit):
    U=f.kp(unit); Q=U[N-1]; MAGIC=(1<<32)//(Q+1)
    t=a[N-1]+(a[N-2]>>29)
    assert t < 1<<32
    k=(t*MAGIC)>>32
    assert k<=4,k
    r=[]; c=0
    for i in range(N):
        d=a[i]-k*U[i]
        assert -(1<<31) <= d < (1<<31), d
        s=d+c
        assert -(1<<31) <= s < (1<<31)
        if i<N-1: r.append(s&MASK); c=s>>29
        else: r.append(s)
    assert r[-1]>=0, r
    return r,k
def run(p,trials=20000,seed=1):
    f=F(p); rnd=random.Random(seed)
    B=9
    worst=0
    for t in range(trials):
        def rv(bound):
            mode=rnd.random()
            if mode<0.3: return bound*p-1-rnd.randrange(1<<20)
            if mode<0.4: return rnd.randrange(1<<20)
            return rnd.randrange(bound*p)
        xs=[rv(B) for _ in range(4)]
        ws=[rnd.choice([p-1,rnd.randrange(p),0,1]) for _ in range(3)]
        x00,x01,x10,x11=[limbs(v) for v in xs]
        w0,w1,w2=[limbs(v) for v in ws]
        a0=lz_add(x00,x10)
        a1=f.mul(w0,lz_sub(f,x00,x10,2*B,29))
        b0=lz_add(x01,x11)
        b1=f.mul(w1,lz_sub(f,x01,x11,2*B,29))
        assert val(a1)<2*p and val(b1)<2*p
        n00,k=lz_reduce(f,lz_add(a0,b0),8)
        n01=f.mul(w2,lz_sub(f,a0,b0,4*B,30))
        n10=lz_norm(lz_add(a1,b1))
        n11=f.mul(w2,lz_sub(f,a1,b1,4,29))
        for l in (n00,n01,n10,n11):
            assert all(0<=x<=MASK for x in l[:-1]) and val(l) < B*p, (val(l)/p)
        worst=max(worst,val(n00)/p)
        Rinv=pow(R,-1,p)
        assert val(n00)%p==(sum(xs))%p
        assert val(n01)%p==((xs[0]+xs[2]-xs[1]-xs[3])*ws[2]*Rinv)%p
        e1=((xs[0]-xs[2])*ws[0]*Rinv)%p; e2=((xs[1]-xs[3])*ws[1]*Rinv)%p
        assert val(n10)%p==(e1+e2)%p
        assert val(n11)%p==((e1-e2)*ws[2]*Rinv)%p
        # radix-2 odd stage
        s2,_=lz_reduce(f,lz_add(x00,x10),8); assert val(s2)<B*p and val(s2)%p==(xs[0]+xs[2])%p
        # final canonicalisation: unit 2p then two conditional subtractions
        fin,_=lz_reduce(f,x00,2); v=val(fin); assert v< 2.2*p, v/p
    print("ok p bits",p.bit_length(),"max col 2^%.3f"%( __import__('math').log2(f.maxcol)),"worst reduce out %.3f p"%worst)
run(21888242871839275222246405745257275088548364400416034343698204186575808495617)
run(52435875175126190479447740508185965837690552500527637822603658699938581184513)
