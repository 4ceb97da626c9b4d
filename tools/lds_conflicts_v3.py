#!/usr/bin/env python3
"""Search a conflict-free LDS swizzle for the 1024-thread / 16-points-per-thread K1 (16384 = 16*16*16*4).
Layout family: element e, group G = e>>4, slot s = (e>>1)&7:  group' = G ^ swapmask(G), slot' = s ^ key(G),
key bit b = parity of (G & maskb).  Banking rules: MI355X_MICROARCH.md LDS section."""
import itertools, random, sys
from lds_conflicts import conflicts

def make_layout(masks, swap_src, pad_every=None):
    def par(x): return bin(x).count("1") & 1
    def phys(e):
        G, s, lo = e >> 4, (e >> 1) & 7, e & 1
        key = par(G & masks[0]) | (par(G & masks[1]) << 1) | (par(G & masks[2]) << 2)
        g2 = G ^ (par(G & swap_src) if swap_src else 0)
        if pad_every: g2 += g2 // pad_every
        return (g2 * 16 + ((s ^ key) << 1) + lo) * 8
    return phys

def check(phys, full=False):
    res = {}
    waves = range(16) if full else (0, 5, 10, 15)
    for w in waves:
        tids = [64 * w + l for l in range(64)]
        for i in range(16):
            a0 = [phys(i * 1024 + t) for t in tids]
            a1 = [phys(w * 1024 + i * 64 + l) for l in range(64)]
            a2 = [phys((t >> 2) * 64 + i * 4 + (t & 3)) for t in tids]
            for name, a in (('P0', a0), ('P1', a1), ('P2', a2)):
                res[name + ' r64'] = max(res.get(name + ' r64', 1), conflicts('r64', a))
                res[name + ' w64'] = max(res.get(name + ' w64', 1), conflicts('w64', a))
        for j in range(8):
            aj = [phys(16 * t + 2 * j) for t in tids]
            res['J r128'] = max(res.get('J r128', 1), conflicts('r128', aj))
            res['J w128'] = max(res.get('J w128', 1), conflicts('w128', aj))
    return res

if __name__ == "__main__":
    random.seed(1)
    cands = [1 << b for b in range(10)] + [(1 << a) | (1 << b) for a in range(10) for b in range(a + 1, 10)]
    best = None
    tried = 0
    for swap in (0, 1 << 4, 1 << 3, 1 << 5, 1 << 2, (1 << 2) | (1 << 4)):
        for _ in range(6000):
            masks = [random.choice(cands) for _ in range(3)]
            r = check(make_layout(masks, swap))
            score = sum(v - 1 for v in r.values())
            tried += 1
            if best is None or score < best[0]:
                best = (score, masks, swap, r)
                print(tried, best, flush=True)
            if score == 0:
                rf = check(make_layout(masks, swap), full=True)
                if sum(v - 1 for v in rf.values()) == 0:
                    print("FOUND", [bin(m) for m in masks], bin(swap), rf)
                    sys.exit(0)
