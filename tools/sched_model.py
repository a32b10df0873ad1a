"""Host-side model of how a tile list runs on the chip: 8 XCDs x 64 workgroup slots (32 CUs x 2),
block b goes to XCD b % 8 in order, a slot takes the next block of its XCD when it frees up; a tile
costs (k-blocks + overhead).  Reports makespan / ideal for the tri-inverse and K^-1 launches.
Development aid for choosing tile orders; no GPU needed."""
import heapq, sys

def xcd_map(n, chunk):
    """block index -> list position, as gemm_tiles_kernel does it"""
    pos = []
    if chunk == 0:
        q, r = n >> 3, n & 7
        for b in range(n):
            x, yy = b & 7, b >> 3
            pos.append((x * (q + 1) if x < r else r * (q + 1) + (x - r) * q) + yy)
    else:
        C = chunk; full = (n // (8 * C)) * (8 * C)
        for b in range(n):
            if b < full:
                x, y = b & 7, b >> 3; rnd = y // C
                pos.append((rnd * 8 + ((7 - x) if rnd & 1 else x)) * C + (y % C))
            else:
                pos.append(b)
    return pos

def simulate(costs, chunk, slots=64, overhead=0.35):
    n = len(costs); pos = xcd_map(n, chunk)
    finish = []
    for x in range(8):
        q = [costs[pos[b]] + overhead for b in range(x, n, 8)]
        h = [0.0] * slots
        heapq.heapify(h)
        end = 0.0
        for c in q:
            t = heapq.heappop(h) + c
            end = max(end, t)
            heapq.heappush(h, t)
        finish.append(end)
    ideal = (sum(costs) + overhead * n) / (8 * slots)
    return max(finish), ideal, finish

def tri_levels(nb):
    levels = []
    def build(lo, hi):
        if hi - lo <= 1: return 0
        mid = lo + (hi - lo + 1) // 2
        ht = 1 + max(build(lo, mid), build(mid, hi))
        while len(levels) < ht: levels.append([])
        levels[ht - 1].append((lo, mid, hi))
        return ht
    build(0, nb)
    return levels

def rect_patch(r0, r1, c0, c1, rows_desc, col_major, kr):
    out = []
    nrg, ncg = (r1 - r0 + 7) // 8, (c1 - c0 + 7) // 8
    nouter, ninner = (ncg, nrg) if col_major else (nrg, ncg)
    for a in range(nouter):
        for bq in range(ninner):
            ig, jg = (bq, a) if col_major else (a, bq)
            if rows_desc: ig = nrg - 1 - ig
            for i in range(r0 + ig * 8, min(r1, r0 + ig * 8 + 8)):
                for j in range(c0 + jg * 8, min(c1, c0 + jg * 8 + 8)):
                    out.append(kr(i, j))
    return out

if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    for lv in tri_levels(nb):
        T, X = [], []
        for lo, mid, hi in lv:
            T += rect_patch(mid, hi, lo, mid, False, True, lambda i, j: mid - j)
            X += rect_patch(mid, hi, lo, mid, True, False, lambda i, j: i + 1 - mid)
        for name, c in (("T", T), ("X", X)):
            if len(c) <= 256: continue
            mk, ideal, fin = simulate(c, 64)
            print("tri %s nodes=%d tiles=%d: makespan/ideal = %.3f  (xcd spread %.3f)" % (name, len(lv), len(c), mk / ideal, (max(fin) - min(fin)) / ideal))
            mk2, _, _ = simulate(sorted(c, reverse=True), 64)
            print("      sorted globally longest-first: %.3f" % (mk2 / ideal))
    la = []
    PR, PC = 2, 32
    for ig in range((nb - 1) // PR + 1):
        jg = 0
        while jg * PC <= min(nb - 1, ig * PR + PR - 1):
            for i in range(ig * PR, min(nb, ig * PR + PR)):
                for j in range(jg * PC, min(nb, jg * PC + PC)):
                    if j <= i: la.append(nb - i)
            jg += 1
    mk, ideal, fin = simulate(la, 64)
    print("lauum tiles=%d: makespan/ideal = %.3f (xcd spread %.3f)" % (len(la), mk / ideal, (max(fin) - min(fin)) / ideal))
