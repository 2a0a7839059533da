"""LDS bank conflicts of the chain phase of pm_push.h (ds_read_b32: two groups of 32 lanes, bank =
dword address mod 32): worst / mean multiplicity over the 64 window terms, for candidate strides."""
import itertools, sys
N, R, REACH = 8, 7, 5
FWH = N + REACH
NF = N * FWH
cons = [(0, 1), (0, -1), (1, 0), (-1, 0), (0, 5), (0, -5), (5, 0), (-5, 0)]  # consumer c - producer

def conflicts(addrs):
    banks = {}
    for a in set(addrs):
        banks.setdefault(a % 32, set()).add(a)
    return max(len(v) for v in banks.values())

def dis_addr(g, c, i, j, dstride, hbase, vs=N, hs=FWH):
    dx, dy = cons[c]
    if dx == 0:
        return g * dstride + ((dy + 5) // 2 + j) * vs + i
    return g * dstride + hbase + j * hs + (dx + 5) // 2 + i

def ipl_addr(tnx, tny, c, i, j, twc):
    dx, dy = cons[c]
    tpx, tpy = tnx + dx, tny + dy
    return (tpy - R + 2 * j) * twc + ((tpx - R) >> 1) + i

def score_dis(dstride, hbase, vs=N, hs=FWH):
    tot, worst = 0, 0
    for i, j in itertools.product(range(N), range(N)):
        for half in (0, 1):
            a = [dis_addr(g, c, i, j, dstride, hbase, vs, hs) for g in range(4 * half, 4 * half + 4) for c in range(8)]
            k = conflicts(a)
            tot += k
            worst = max(worst, k)
    return tot / (2 * N * N), worst

def score_ipl(twc):
    tot, worst, n = 0, 0, 0
    for tny in (13, 14):
        for t0 in (13, 14, 21, 22):
            tnx0 = t0 if (t0 + tny) % 2 == 0 else t0 + 1
            for i, j in itertools.product(range(N), range(N)):
                a = [ipl_addr(tnx0 + 2 * g, tny, c, i, j, twc) for g in range(4) for c in range(8)]
                k = conflicts(a)
                tot += k; worst = max(worst, k); n += 1
    return tot / n, worst

if __name__ == "__main__":
    best = []
    for dstride in range(2 * NF, 2 * NF + 34):
        for hbase in range(NF, NF + 9):
            if hbase + NF > dstride: continue
            m, w = score_dis(dstride, hbase)
            best.append((m, w, dstride, hbase))
    best.sort()
    print("dis:", best[:8])
    print("dis plain:", score_dis(2 * NF, NF))
    print("dis shipped (rows 9 / 17 words apart, horizontal family at 124, groups 264 apart):", score_dis(264, 124, 9, 17))
    for twc in (29, 30, 31, 32, 33, 41):
        print("iplane twc", twc, score_ipl(twc))
