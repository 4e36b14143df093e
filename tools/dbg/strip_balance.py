"""Diagnostic: which SIMD carries how many live rows of a tile over the sub-steps of a tick, for the round-robin placement of a
workgroup's waves and for the best of 200 000 random assignments of the strips to the four SIMDs (4 + 4 + 4 + 3 strips).
Fenton 512x512: K = 10, 45 rows, 3-row strips; Beeler-Reuter: K = 5, 29 rows, 2-row strips.  Round robin is as good as any:
the early sub-steps, in which every strip is whole, set the figure.  (profiles/r04_issue_bound.txt)"""
import itertools, sys
def run(K, ROWS, R, NW, sizes):
    strips=[list(range(R*i, min(R*i+R, ROWS))) for i in range(NW)]
    def live(i,s): return sum(1 for y in strips[i] if s<=y<ROWS-s)
    L=[[live(i,s) for s in range(K)] for i in range(NW)]
    def cost(groups): return sum(max(sum(L[i][s] for i in g) for g in groups) for s in range(K))
    cur=[[i for i in range(NW) if i%4==j] for j in range(4)]
    print('round robin', cost(cur), 'ideal', sum(sum(L[i][s] for i in range(NW)) for s in range(K))/4, 'whole-wave', sum(max(sum(R if L[i][s] else 0 for i in g) for g in cur) for s in range(K)))
    best=None
    import random
    random.seed(1)
    idx=list(range(NW))
    for trial in range(200000):
        random.shuffle(idx)
        g=[];p=0
        for n in sizes: g.append(idx[p:p+n]); p+=n
        c=cost(g)
        if best is None or c<best[0]: best=(c,[sorted(x) for x in g])
    print('best random', best)
run(10,45,3,15,[4,4,4,3])
run(5,29,2,15,[4,4,4,3])
