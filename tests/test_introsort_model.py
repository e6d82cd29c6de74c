"""klib's ks_introsort (reference src/ksort.h; what mem_chain_flt sorts chains with, src/bwamem.cpp:80, 631) against the wave-parallel
formulation the LDS chaining tier runs (bwa-meme_amd/csrc/meme_chain.hip k_chain_lds): each Hoare partition step from the two lists of
scan stops (up-scan stops at weight <= pivot, down-scan at weight >= pivot; the k-th swap pairs the k-th stop from the left with the k-th
from the right while the left one lies before the right one; the loop ends on a position that follows from the lists), the closing
insertion sort as a stable sort.  Both are modelled here in Python on (weight, id) pairs; chains of EQUAL weight must come out in the
same order, because the filter that follows depends on it.  (The device kernel itself is checked against the reference's chains in
tests/test_gpu_chain.py.)"""
import random
def lt(a,b): return a[0] > b[0]
def insertsort(a,s,t):
    for i in range(s+1,t):
        j=i
        while j> s and lt(a[j],a[j-1]):
            a[j],a[j-1]=a[j-1],a[j]; j-=1
def combsort(a,s,n):
    shrink=1.2473309501039786540366528676643
    gap=n
    while True:
        if gap>2:
            gap=int(gap/shrink)
            if gap in (9,10): gap=11
        do_swap=False
        for i in range(s, s+n-gap):
            j=i+gap
            if lt(a[j],a[i]): a[i],a[j]=a[j],a[i]; do_swap=True
        if not (do_swap or gap>2): break
    if gap!=1: insertsort(a,s,s+n)
def klib(a):
    a=a[:]; n=len(a)
    if n<1: return a
    if n==2:
        if lt(a[1],a[0]): a[0],a[1]=a[1],a[0]
        return a
    d=2
    while (1<<d) < n: d+=1
    d<<=1
    s=0;t=n-1;stack=[]
    while True:
        if s<t:
            d-=1
            if d==0:
                combsort(a,s,t-s+1); t=s; continue
            i=s;j=t;k=i+((j-i)>>1)+1
            if lt(a[k],a[i]):
                if lt(a[k],a[j]): k=j
            else: k = i if lt(a[j],a[i]) else j
            rp=a[k]
            if k!=t: a[k],a[t]=a[t],a[k]
            while True:
                i+=1
                while lt(a[i],rp): i+=1
                j-=1
                while i<=j and lt(rp,a[j]): j-=1
                if j<=i: break
                a[i],a[j]=a[j],a[i]
            a[i],a[t]=a[t],a[i]
            if i-s > t-i:
                if i-s>16: stack.append((s,i-1,d))
                s = i+1 if t-i>16 else t
            else:
                if t-i>16: stack.append((i+1,t,d))
                t = i-1 if i-s>16 else s
        else:
            if not stack:
                insertsort(a,0,n); return a
            s,t,d=stack.pop()
def par(a):
    a=a[:]; n=len(a)
    if n<1: return a
    if n==2:
        if a[1][0]>a[0][0]: a[0],a[1]=a[1],a[0]
        return a
    d=2
    while (1<<d) < n: d+=1
    d<<=1
    s=0;t=n-1;stack=[]
    while True:
        if s<t:
            d-=1
            if d==0:
                combsort(a,s,t-s+1); t=s; continue
            k=s+((t-s)>>1)+1
            wi,wj,wk=a[s][0],a[t][0],a[k][0]
            if wk>wi:
                if wk>wj: k=t
            else: k = s if wj>wi else t
            rpw=a[k][0]
            if k!=t: a[k],a[t]=a[t],a[k]
            L=[p for p in range(s+1,t+1) if a[p][0]<=rpw]
            R=[p for p in range(t-1,s,-1) if a[p][0]>=rpw]
            mm=0
            while mm<min(len(L),len(R)) and L[mm]<R[mm]: mm+=1
            for q in range(mm): a[L[q]],a[R[q]]=a[R[q]],a[L[q]]
            i=L[mm]
            if mm>=1 and R[mm-1]<i: i=R[mm-1]
            a[i],a[t]=a[t],a[i]
            if i-s > t-i:
                if i-s>16: stack.append((s,i-1,d))
                s = i+1 if t-i>16 else t
            else:
                if t-i>16: stack.append((i+1,t,d))
                t = i-1 if i-s>16 else s
        else:
            if not stack:
                # stable sort by descending weight
                order=sorted(range(n), key=lambda e:(-a[e][0], e))
                return [a[e] for e in order]
            s,t,d=stack.pop()


def test_parallel_formulation_of_klib_introsort_keeps_the_order_of_ties():
    random.seed(3)
    for it in range(3000):
        n = random.randint(1, 400)
        rng = random.choice([1, 2, 4, 20, 1000])
        arr = [(random.randint(0, rng), i) for i in range(n)]
        assert klib(arr) == par(arr), (n, rng)
