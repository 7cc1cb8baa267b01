"""TEST INFRASTRUCTURE ONLY -- never imported by the product.

CPU restatement of the reference's redundancy filter for one (A contig, B contig, strand) group of accepted
alignments: the two elimination sweeps of align_contigs (FastGA.c:3440-3585) and entwine (FastGA.c:2818-2941), written
as a direct, slow, record-by-record walk.  tests/test_filter_oracle.py checks the product's host filter
(fastga_amd/csrc/fga_filter.c) against it on randomly generated overlapping records.  Parity note: the reference does
not expose its filter as a callable, so this file is pinned to the reference only through the end-to-end `.1aln`
comparisons (tests/test_end_to_end_gpu.py, tests/test_full_size_gpu.py), which pass through the product filter.
"""
TSPACE = 100
BOX_FUZZ = 10


class Rec:
    __slots__ = ("abpos", "bbpos", "aepos", "bepos", "diffs", "trace", "dead", "ord")

    def __init__(self, abpos, bbpos, aepos, bepos, diffs, trace, ord_):
        self.abpos, self.bbpos, self.aepos, self.bepos, self.diffs = abpos, bbpos, aepos, bepos, diffs
        self.trace = list(trace)
        self.dead = False
        self.ord = ord_


def entwine(jp, kp):
    """FastGA.c:2818-2941: (min signed separation, clamped to 0 on a crossing; last common trace point or -1)"""
    jt, kt = jp.trace, kp.trace
    where = -1
    y2, b2 = jp.bbpos, kp.bbpos
    j, k = jp.abpos // TSPACE, kp.abpos // TSPACE
    ac = k * TSPACE
    j = 1 + 2 * (k - j)
    k = 1
    for i in range(1, j, 2):
        y2 += jt[i]
    if j == 1:
        yp = y2 + (jt[j] * (kp.abpos - jp.abpos)) // (ac + TSPACE - jp.abpos)
    else:
        yp = y2 + (jt[j] * (kp.abpos - ac)) // TSPACE
    mn = b2 - yp

    def upd(mn, i):
        if mn < 0 and mn < i:
            return 0 if i >= 0 else i
        if mn > 0 and mn > i:
            return 0 if i <= 0 else i
        return mn

    ae = min(jp.aepos, kp.aepos)
    ac += TSPACE
    while ac < ae:
        y2 += jt[j]
        b2 += kt[k]
        j += 2
        k += 2
        i = b2 - y2
        mn = upd(mn, i)
        if i == 0:
            where = ac
        ac += TSPACE
    ac -= TSPACE
    if ae == jp.aepos:
        y2 = jp.bepos
        b2 += (kt[k] * (ae - ac)) // (TSPACE if kp.aepos >= ac else kp.aepos - ac)
    else:
        b2 = kp.bepos
        y2 += (jt[j] * (ae - ac)) // (TSPACE if jp.aepos >= ac else jp.aepos - ac)
    return upd(mn, b2 - y2), where


def filter_group(recs):
    """recs: Rec list of one contig pair and strand in discovery order.  Returns the survivors in abpos order
    (stable: the reference's qsort is unstable, ties on abpos are resolved by discovery order here as in the product)."""
    perm = sorted(recs, key=lambda r: (r.abpos, r.ord))
    n = len(perm)
    for j in range(n - 1, -1, -1):                                   # FastGA.c:3441-3491
        o = perm[j]
        for k in range(j + 1, n):
            w = perm[k]
            if o.aepos <= w.abpos:
                break
            if w.dead:
                continue
            if o.abpos == w.abpos and o.bbpos == w.bbpos:
                if o.aepos == w.aepos and o.bepos == w.bepos:
                    if o.diffs < w.aepos:                            # sic, FastGA.c:3456
                        w.dead = True
                        continue
                    o.dead = True
                    break
                if o.aepos > w.aepos:
                    w.dead = True
                    continue
                o.dead = True
                break
            if o.aepos == w.aepos and o.bepos == w.bepos:
                if o.abpos < w.abpos:
                    w.dead = True
                    continue
                o.dead = True
                break
    for j in range(n - 1, -1, -1):                                   # FastGA.c:3493-3585
        o = perm[j]
        if o.dead:
            continue
        for k in range(j + 1, n):
            w = perm[k]
            if o.aepos <= w.abpos:
                break
            if w.dead:
                continue
            if o.bepos <= w.bbpos or o.bbpos >= w.bepos:
                continue
            dist, where = entwine(o, w)
            if where != -1:
                ocut = 2 * (((where - o.abpos) - 1) // TSPACE + 1)
                wcut = 2 * (((where - w.abpos) - 1) // TSPACE + 1)
                o.trace = o.trace[:ocut] + w.trace[wcut:]
                o.diffs = sum(o.trace[0::2])
                o.aepos, o.bepos = w.aepos, w.bepos
                w.dead = True
                continue
            if dist != 0:
                if (o.aepos - o.abpos) + BOX_FUZZ >= w.aepos - w.abpos:
                    if (w.aepos <= o.aepos + BOX_FUZZ and w.bbpos >= o.bbpos - BOX_FUZZ and
                            w.bepos <= o.bepos + BOX_FUZZ):
                        w.dead = True
                elif (o.aepos <= w.aepos + BOX_FUZZ and o.bbpos >= w.bbpos - BOX_FUZZ and
                      o.bepos <= w.bepos + BOX_FUZZ and o.abpos >= w.abpos - BOX_FUZZ):
                    o.dead = True
    return [r for r in perm if not r.dead]
