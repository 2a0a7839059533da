"""Index model of pm_push.h (box 15): every address formula of the kernel restated in Python and
checked against the plain definition -- the stencil points a group of 8 lanes evaluates per step,
where their dis values land, which slots the chain of consumer c reads for window term (i, j), and
the checkerboard-compressed reference tile.  Run on the CPU; no GPU needed."""
import itertools, random

N, R, REACH = 8, 7, 5
FWH = N + REACH           # 13
NF = N * FWH              # 104
HALO = R + REACH + 1      # 13
TILE_W, TILE_H = 32, 16
TW, TH = TILE_W + 2 * HALO, TILE_H + 2 * HALO  # 58, 42
TWC = (TW + 1) // 2       # 29
VS, HS = 9, 17            # row strides of the two families in the group's sample buffer (PushLayout<15>)
HBASE = 124               # horizontal family behind the vertical one
IPS = 41                  # row stride of the compressed I plane
# consumer c = the pixel that sees the producer as its neighbour slot c (pm::neighbour):
# slot 0 up, 1 down, 2 left, 3 right (distance 1), 4..7 the same at distance 5
def consumer_offset(c):
    d = 1 if c < 4 else 5
    k = c & 3
    # neighbour offset of slot k: up (0,-d), down (0,+d), left (-d,0), right (+d,0); consumer = producer - offset
    return [(0, d), (0, -d), (d, 0), (-d, 0)][k]

def eval_point(s, l):
    """step s (0..25) of lane l (0..7): offset of the sample from the producer, and its dis slot"""
    if s < FWH:  # vertical family: x = l, y = s
        return (2 * l - R, 2 * s - (R + REACH)), s * VS + l
    t = s - FWH
    j0, r0 = (8 * t) // FWH, (8 * t) % FWH
    w = 1 if l >= FWH - r0 else 0
    x, y = r0 + l - FWH * w, j0 + w
    return (2 * x - (R + REACH), 2 * y - R), HBASE + j0 * HS + r0 + l + w * (HS - FWH)

def eval_tile_index(s, l, tnx, tny):
    """compressed-tile index the kernel reads for that point"""
    if s < FWH:
        base_v = tny * TWC + ((tnx - R) >> 1) + l
        return base_v + (2 * s - (R + REACH)) * TWC
    t = s - FWH
    j0, r0 = (8 * t) // FWH, (8 * t) % FWH
    w = 1 if l >= FWH - r0 else 0
    base_h = tny * TWC + ((tnx - (R + REACH)) >> 1)
    return base_h + (2 * j0 - R) * TWC + r0 + l + w * (2 * TWC - FWH)

def chain_slot(c, i, j):
    dx, dy = consumer_offset(c)
    if dx == 0:
        return ((dy + REACH) // 2) * VS + j * VS + i
    return HBASE + (dx + REACH) // 2 + j * HS + i

def chain_tile_index(c, i, j, tnx, tny):
    dx, dy = consumer_offset(c)
    tpx, tpy = tnx + dx, tny + dy
    return (tpy - R) * IPS + ((tpx - R) >> 1) + 2 * j * IPS + i

def centre_index(c, tnx, tny):
    dx, dy = consumer_offset(c)
    return (tny + dy) * IPS + ((tnx + dx) >> 1)

def main():
    rnd = random.Random(1)
    for colour in (0, 1):
        cpar = 1 - colour
        # the compressed tile as the kernel stages it: entry k = (ty, cx) holds texel tx = 2 cx + ((cpar + ty) & 1)
        comp, compi = {}, {}
        for ty in range(TH):
            for cx in range(TWC):
                tx = 2 * cx + ((cpar + ty) & 1)
                comp[ty * TWC + cx] = (tx, ty) if tx < TW else None
                compi[ty * IPS + cx] = (tx, ty) if tx < TW else None  # the I plane: rows IPS words apart
        for ly in range(TILE_H):
            for lxh in range(16):
                lx = 2 * lxh + ((ly + colour) & 1)
                tnx, tny = lx + HALO, ly + HALO
                slots = {}
                for s in range(2 * FWH):
                    for l in range(8):
                        (dx, dy), e = eval_point(s, l)
                        assert e not in slots, "slot written twice"
                        slots[e] = (dx, dy)
                        k = eval_tile_index(s, l, tnx, tny)
                        assert comp[k] == (tnx + dx, tny + dy), (s, l, comp[k], (tnx + dx, tny + dy))
                        assert 1 <= tnx + dx <= TW - 2 and 1 <= tny + dy <= TH - 2  # gradients stay inside the tile
                assert len(slots) == 2 * NF and max(slots) < 264
                for c in range(8):
                    cdx, cdy = consumer_offset(c)
                    assert compi[centre_index(c, tnx, tny)] == (tnx + cdx, tny + cdy)
                    for i, j in itertools.product(range(N), range(N)):
                        q = (cdx + 2 * i - R, cdy + 2 * j - R)  # window sample of the consumer, relative to the producer
                        assert slots[chain_slot(c, i, j)] == q, (c, i, j)
                        assert compi[chain_tile_index(c, i, j, tnx, tny)] == (tnx + q[0], tny + q[1])
    # the neighbour relation: consumer p = n + consumer_offset(c) has n at slot c
    for c in range(8):
        d = 1 if c < 4 else 5
        off = [(0, -d), (0, d), (-d, 0), (d, 0)][c & 3]
        co = consumer_offset(c)
        assert (co[0] + off[0], co[1] + off[1]) == (0, 0)
    print("push index model: ok")

# ---- PushEval::family, the loop used for boxes other than 15 (and checked for 15 as well) ----
def family_points(W, HH, OX, OY, twc, tnx, tny):
    """what the kernel's stream computes: {slot: (offset from the producer, compressed tile index)}"""
    NFp = W * HH
    SF = (NFp + 7) // 8
    out = {}
    r0 = j0 = 0
    for s in range(SF):
        for l in range(8):
            w1 = 1 if l >= W - r0 else 0
            w2 = 1 if (W < 8 and l >= 2 * W - r0) else 0
            dqx = (2 * l - OX) + 2 * r0 - 2 * W * w1 - 2 * W * w2
            dqy = -OY + 2 * j0 + 2 * w1 + 2 * w2
            tbase = (tny - OY) * twc + ((tnx - OX) >> 1) + l
            tix = tbase + r0 + 2 * twc * j0 + (2 * twc - W) * (w1 + w2)
            e = 8 * s + l
            if e < NFp:
                assert e not in out
                out[e] = ((dqx, dqy), tix)
        r0 += 8
        if r0 >= W:
            r0 -= W
            j0 += 1
        if W < 8 and r0 >= W:
            r0 -= W
            j0 += 1
    assert sorted(out) == list(range(NFp))
    return out


def check_generic(box):
    n = (box + 1) // 2
    r = n - 1
    fwh = n + REACH
    nf = n * fwh
    halo = r + REACH + 1
    tw, th = TILE_W + 2 * halo, TILE_H + 2 * halo
    twc = ((tw + 1) // 2) | 1
    two_pass = box > 15
    hbase = 0 if two_pass else nf
    for colour in (0, 1):
        cpar = 1 - colour
        comp = {}
        for ty in range(th):
            for cx in range(twc):
                tx = 2 * cx + ((cpar + ty) & 1)
                comp[ty * twc + cx] = (tx, ty) if tx < tw else None
        for ly in range(TILE_H):
            for lxh in range(16):
                lx = 2 * lxh + ((ly + colour) & 1)
                tnx, tny = lx + halo, ly + halo
                fam_v = family_points(n, fwh, r, r + REACH, twc, tnx, tny)
                fam_h = family_points(fwh, n, r + REACH, r, twc, tnx, tny)
                for fam in (fam_v, fam_h):
                    for e, ((dx, dy), tix) in fam.items():
                        assert comp[tix] == (tnx + dx, tny + dy), (box, e)
                        assert 1 <= tnx + dx <= tw - 2 and 1 <= tny + dy <= th - 2
                for c in range(8):
                    cdx, cdy = consumer_offset(c)
                    tpx, tpy = tnx + cdx, tny + cdy
                    assert comp[tpy * twc + (tpx >> 1)] == (tpx, tpy)
                    dbase = ((cdy + REACH) // 2) * n if cdx == 0 else hbase + (cdx + REACH) // 2
                    jstride = n if cdx == 0 else fwh
                    fam, off = (fam_v, 0) if cdx == 0 else (fam_h, hbase)
                    for i, j in itertools.product(range(n), range(n)):
                        q = (cdx + 2 * i - r, cdy + 2 * j - r)
                        slot = dbase + j * jstride + i
                        assert fam[slot - off][0] == q, (box, c, i, j)
                        assert comp[(tpy - r) * twc + ((tpx - r) >> 1) + 2 * j * twc + i] == (tnx + q[0], tny + q[1])
    print("push family model, box %d: ok" % box)


def main_all():
    main()
    for box in (11, 15, 25):
        check_generic(box)


if __name__ == "__main__":
    main_all()
