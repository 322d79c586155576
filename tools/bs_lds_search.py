"""XOR line layouts of the chirp-z kernels (bluestein.h: BsLds): the exchanges of ZPlan<M, 8> with a line's lanes CONSECUTIVE in the
wave, enumerated against the LDS bank rules of MI355X_MICROARCH.md (cf. tools/lds_conflicts.py for the interleaved x pass).
Build-time aid: no GPU.  usage: python tools/bs_lds_search.py"""
import sys
sys.path.insert(0,'tools')
PLANS = {64: [8, 8], 128: [8, 4, 4], 256: [8, 8, 4], 512: [8, 8, 8]}
def conflicts(LEN, R, lpos, RS, f32, lmul=0):
    E = 8; P = LEN // E; LPWV = max(1, 64 // P)
    rd = wr = nrd = nwr = 0; NS = 1
    def grp(lp, gsize, nb):
        tot = n = 0
        for g0 in range(0, 64, gsize):
            banks = {}
            for lane, pos in lp:
                if g0 <= lane < g0 + gsize:
                    banks.setdefault(pos % nb, set()).add(pos)
            if banks:
                tot += max(len(v) for v in banks.values()); n += 1
        return tot, n
    def at(l, i): return l * RS + (lpos(i) ^ ((l * lmul) & 31))
    for s in range(len(R) - 1):
        r, r2 = R[s], R[s + 1]
        NBF, NBF2 = LEN // r, LEN // r2
        NB, NB2 = max(1, NBF // P), max(1, NBF2 // P)
        for b in range(NB):
            for u in range(r):
                lp = []
                for lane in range(64):
                    l, j = lane // P, lane % P            # blocked: a line's lanes are consecutive
                    jb = j + b * P
                    if jb < NBF: lp.append((lane, at(l, (jb // NS) * (NS * r) + jb % NS + u * NS)))
                t, n = grp(lp, 32 if f32 else 16, 32 if f32 else 16)
                wr += t; nwr += n
        for b in range(NB2):
            for t_ in range(r2):
                lp = [(lane, at(lane // P, lane % P + b * P + t_ * NBF2)) for lane in range(64) if lane % P + b * P < NBF2]
                t, n = grp(lp, 32, 32)
                rd += t; nrd += n
        NS *= r
    return rd / nrd, wr / nwr
def xor_layout(xs, xm, mul): return lambda i: i ^ ((mul * ((i >> xs) & xm)) & 31)
for f32 in (False, True):
    for LEN, R in PLANS.items():
        old = conflicts(LEN, R, lambda i: i + (i >> 4), LEN + (LEN >> 4) + 2, f32)
        res = []
        for xs in (1, 2, 3, 4, 5, 6):
            for xm in (1, 3, 7, 15, 31):
                for mul in (1, 2, 3, 4, 5, 8, 9, 16, 17):
                    f = xor_layout(xs, xm, mul)
                    img = [f(i) for i in range(LEN)]
                    if len(set(img)) != LEN: continue
                    lo = (max(img) | 31) + 1
                    for lmul in ((0,) if LEN == 512 else (0, 1, 2, 4, 8, 16, 3, 5, 9, 17, 24, 12)):
                        for RS in (range(lo, lo + 1) if LEN == 512 else range(lo, lo + 33)):
                            c = conflicts(LEN, R, f, RS, f32, lmul)
                            res.append((c[0] + c[1], c, xs, xm, mul, lmul, RS))
        res.sort()
        print('f32' if f32 else 'f64', LEN, R, 'padded', old, 'best', res[:2], flush=True)
