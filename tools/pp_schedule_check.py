"""CPU model of the ping-pong GEMM kernel's LDS plumbing (mafed_amd/csrc/gemm_pp.hip).  No GPU needed.

Three independent checks of the hand-derived tables, each mirroring the kernel's formulas:
  1. index maps   -- every fragment read returns the operand elements the MFMA expects (DMA source permutation x LDS image
                     x read address x transposing-read semantics), for every configuration / layout / output type;
  2. bank conflicts of every fragment read instruction (lane groups and bank rule of MI355X_MICROARCH.md, LDS section);
  3. the DMA schedule -- RAW (a region is read only after every wave's covering counted wait and a barrier) and WAR (a region
     is re-filled at least two barrier intervals after its last reader), over both wave groups' barrier intervals, including
     the prologue, the stores of an epilogue between two tiles and the tile switch.

Run: python tools/pp_schedule_check.py  (also imported by tests/test_pp_schedule.py)
"""
import itertools
import sys

CONFIGS = {  # name: (WM, WN, MT, NT)
    "256x256": (2, 4, 8, 4),
    "192x256": (2, 4, 6, 4),
    "144x256": (1, 8, 9, 2),
}


def ks_f(k):
    return (k & 3) | (((k >> 3) & 1) << 2)


def pp_f2(k):
    return ((k >> 1) & 1) | (((k >> 3) & 1) << 1)


def pp_fpair(row):
    return ((row >> 1) & 1) | (((row >> 3) & 3) << 1)


class Cfg:
    def __init__(self, name, a_ks, b_ks, pair):
        self.WM, self.WN, self.MT, self.NT = CONFIGS[name]
        self.TM, self.TN = self.WM * self.MT * 16, self.WN * self.NT * 16
        self.NPH = 4 if self.MT == 8 else 3
        self.MTP = self.MT // self.NPH
        self.A_BYTES, self.B_BYTES = self.TM * 128, self.TN * 128
        self.A_PW = self.NPH if self.WM == 2 else 3
        self.B_PW = 4
        self.RBB = self.TN * 2
        self.a_ks, self.b_ks, self.pair = a_ks, b_ks, pair
        self.name = name


# ---------------------------------------------------------------------------------------------------------------------
# 1. index maps.  LDS modelled as a dict byte-address -> (operand, row, k) per 2-byte element; operands: A[row][k], B[row][k]
#    (row = output row / column index, k = reduction index inside the 64-deep K-tile).
# ---------------------------------------------------------------------------------------------------------------------
def dma_fill(c):
    lds = {}
    for wave in range(8):
        for lane in range(64):
            for i in range(c.A_PW):
                if not c.a_ks:
                    if c.WM == 2:
                        pj = (wave >> 2) * (c.MT * 2) + i * 4 + (wave & 3)
                    else:
                        pj = wave if i == 0 else (wave + 8 if i == 1 else 16 + (wave & 1))
                    dst = pj * 1024
                    row = 8 * pj + (lane >> 3)
                    logical = (lane & 7) ^ ((row >> 1) & 7)
                    elems = [("A", row, logical * 8 + e) for e in range(8)]
                else:
                    dst = i * 8192 + wave * 1024
                    k = 8 * wave + (lane >> 3)
                    ph = lane & 7
                    l32 = (ph >> 1) ^ pp_f2(k)
                    row = (l32 >> 1) * (c.MT * 16) + (i * 2 + (l32 & 1)) * 16 + (ph & 1) * 8
                    elems = [("A", row + e, k) for e in range(8)]
                for e, v in enumerate(elems):
                    a = dst + lane * 16 + 2 * e
                    assert a not in lds or lds[a] == v, "two pieces write different data to one address"
                    lds[a] = v
            for i in range(c.B_PW):
                pj = 4 * wave + i
                dst = c.A_BYTES + pj * 1024
                if not c.b_ks:
                    row = 8 * pj + (lane >> 3)
                    f = pp_fpair(row) if c.pair else ((row >> 1) & 7)
                    logical = (lane & 7) ^ f
                    elems = [("B", row, logical * 8 + e) for e in range(8)]
                else:
                    pb = pj * 1024 + lane * 16
                    k, within = pb // c.RBB, pb % c.RBB
                    l32 = (within >> 5) ^ ks_f(k)
                    col = l32 * 16 + ((within >> 4) & 1) * 8
                    elems = [("B", col + e, k) for e in range(8)]
                for e, v in enumerate(elems):
                    a = dst + lane * 16 + 2 * e
                    assert a not in lds, "B pieces overlap"
                    lds[a] = v
    return lds


def read_b128(lds, addr):
    assert addr % 16 == 0
    return [lds[addr + 2 * e] for e in range(8)]


def read_tr(lds, addrs):
    """ds_read_b64_tr_b16 for one 16-lane group: addrs[l] = byte address supplied by lane l (8-byte aligned); returns per lane the
    4 elements it receives (cdna_hip_programming T10: lane i gets column i of the 4 rows, row q in element q)."""
    out = []
    for i in range(16):
        out.append([lds[addrs[4 * q + (i >> 2)] + 2 * (i & 3)] for q in range(4)])
    return out


def frag_addresses(c, wave, kind, t, ks):
    """Per-lane LDS byte addresses of one fragment read; kind 'A' (t = mt) or 'B' (t = nt).  Returns (instr, [addr per lane]) lists:
    one b128 read, or two tr reads (lo, hi)."""
    wm, wn = wave // c.WN, wave % c.WN
    res = []
    if kind == "A":
        if not c.a_ks:
            addrs = []
            for lane in range(64):
                li, q4 = lane & 15, lane >> 4
                a_rd = (wm * c.MT * 16 + li) * 128 + (((ks * 4 + q4) ^ ((li >> 1) & 7)) << 4)
                addrs.append(a_rd + t * 2048)
            res.append(("b128", addrs))
        else:
            p, jf = t // c.MTP, t % c.MTP
            for h in range(2):
                addrs = []
                for lane in range(64):
                    li, q4 = lane & 15, lane >> 4
                    F2 = ((li >> 3) & 1) | ((q4 & 1) << 1)
                    a_rd = (8 * q4 + (li >> 2)) * 128 + (((wm * 2 + jf) ^ F2) << 5) + (li & 3) * 8
                    addrs.append(a_rd + p * 8192 + (32 * ks + 4 * h) * 128)
                res.append(("tr", addrs))
    else:
        nt = t
        if not c.b_ks:
            addrs = []
            for lane in range(64):
                li, q4 = lane & 15, lane >> 4
                if not c.pair:
                    b_rd = c.A_BYTES + (wn * c.NT * 16 + li) * 128 + (((ks * 4 + q4) ^ ((li >> 1) & 7)) << 4)
                    imm = nt * 2048
                else:
                    rowb = wn * c.NT * 16 + 8 * (li >> 2) + (li & 3)
                    fp = ((li >> 1) & 1) | (((li >> 2) & 3) << 1)
                    b_rd = c.A_BYTES + rowb * 128 + (((ks * 4 + q4) ^ fp) << 4)
                    imm = (32 * (nt >> 1) + 4 * (nt & 1)) * 128
                addrs.append(b_rd + imm)
            res.append(("b128", addrs))
        else:
            for h in range(2):
                addrs = []
                for lane in range(64):
                    li, q4 = lane & 15, lane >> 4
                    F = (li >> 2) | ((q4 & 1) << 2)
                    kq = (8 * q4 + (li >> 2)) * c.RBB
                    if not c.pair:
                        base = c.A_BYTES + kq + (((wn * c.NT + nt) ^ F) << 5) + (li & 3) * 8
                    else:
                        base = c.A_BYTES + kq + (((wn * c.NT + 2 * (nt >> 1) + ((li & 3) >> 1)) ^ F) << 5) + 16 * (li & 1) + 8 * (nt & 1)
                    addrs.append(base + (32 * ks + 4 * h) * c.RBB)
                res.append(("tr", addrs))
    return res


def frag_elements(c, lds, wave, kind, t, ks):
    """Per lane the 8 (operand, row, k) elements of the fragment, in MFMA element order j = 0..7."""
    reads = frag_addresses(c, wave, kind, t, ks)
    if reads[0][0] == "b128":
        return [read_b128(lds, a) for a in reads[0][1]]
    lo, hi = [], []
    for g in range(4):
        lo += read_tr(lds, reads[0][1][16 * g:16 * g + 16])
        hi += read_tr(lds, reads[1][1][16 * g:16 * g + 16])
    return [lo[l] + hi[l] for l in range(64)]


def out_col(c, wn, nt, i):
    """Tile column held by B-fragment nt's row i (the MFMA's D row) -- must match the epilogue's store addressing."""
    if c.pair:
        return wn * c.NT * 16 + 32 * (nt >> 1) + 8 * (i >> 2) + 4 * (nt & 1) + (i & 3)
    return wn * c.NT * 16 + 16 * nt + i


def check_index_maps(c):
    lds = dma_fill(c)
    for wave in range(8):
        wm, wn = wave // c.WN, wave % c.WN
        for ks in range(2):
            for mt in range(c.MT):
                fr = frag_elements(c, lds, wave, "A", mt, ks)
                for lane in range(64):
                    for j in range(8):
                        want = ("A", (wm * c.MT + mt) * 16 + (lane & 15), ks * 32 + 8 * (lane >> 4) + j)
                        assert fr[lane][j] == want, (c.name, "A", wave, mt, ks, lane, j, fr[lane][j], want)
            for nt in range(c.NT):
                fr = frag_elements(c, lds, wave, "B", nt, ks)
                for lane in range(64):
                    for j in range(8):
                        want = ("B", out_col(c, wn, nt, lane & 15), ks * 32 + 8 * (lane >> 4) + j)
                        assert fr[lane][j] == want, (c.name, "B", wave, nt, ks, lane, j, fr[lane][j], want)
    # epilogue addressing: lane (li, q4) holds D rows 4*q4 + r of fragment nt -> columns must be consecutive as stored
    for wn in range(c.WN):
        for q4 in range(4):
            if c.pair:
                for pr in range(c.NT // 2):
                    cols = [out_col(c, wn, 2 * pr + (e >> 2), 4 * q4 + (e & 3)) for e in range(8)]
                    assert cols == list(range(wn * c.NT * 16 + 32 * pr + 8 * q4, wn * c.NT * 16 + 32 * pr + 8 * q4 + 8)), cols
            else:
                for nt in range(c.NT):
                    cols = [out_col(c, wn, nt, 4 * q4 + r) for r in range(4)]
                    assert cols == list(range(wn * c.NT * 16 + 16 * nt + 4 * q4, wn * c.NT * 16 + 16 * nt + 4 * q4 + 4)), cols


# ---------------------------------------------------------------------------------------------------------------------
# 2. bank conflicts (MI355X_MICROARCH.md, LDS): bank = (addr / 4) % 64; ds_read_b128 is serviced in four 16-lane groups,
#    ds_read_b64_tr_b16 in two 32-lane halves; extra cycles = (max distinct addresses on one bank) - 1 per group.
# ---------------------------------------------------------------------------------------------------------------------
B128_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]


def conflict_ways(instr, addrs):
    worst = 1
    if instr == "b128":
        groups, width = B128_GROUPS, 16
    else:
        groups, width = [list(range(0, 32)), list(range(32, 64))], 8
    for grp in groups:
        per_bank = {}
        for l in grp:
            for b in range(0, width, 4):
                per_bank.setdefault(((addrs[l] + b) // 4) % 64, set()).add((addrs[l] + b) // 4)
        worst = max(worst, max(len(v) for v in per_bank.values()))
    return worst


def check_bank_conflicts(c):
    worst = {}
    for wave in range(8):
        for ks in range(2):
            for kind, n in (("A", c.MT), ("B", c.NT)):
                for t in range(n):
                    for instr, addrs in frag_addresses(c, wave, kind, t, ks):
                        key = kind + ":" + instr
                        worst[key] = max(worst.get(key, 1), conflict_ways(instr, addrs))
    return worst


# ---------------------------------------------------------------------------------------------------------------------
# 3. the DMA schedule.  One barrier per phase; interval J = the span between barriers J-1 and J.  In interval J every wave reads
#    the fragments of phase J+1 (prefetch, double-buffered), issues DMA, takes its counted wait and runs the MFMA cluster of phase
#    J (a staggered group 1 runs that cluster after barrier J, but retires the reads of phase J before it, like group 0).  So the
#    fragments of phase P are read in interval P-1 and have retired by barrier P: the region may be re-filled (DMA issued) from
#    interval P+1 on.  Per wave the VMEM queue is in issue order; a counted wait vmcnt(N) guarantees everything but that wave's
#    N youngest operations.  Data is readable by any wave in an interval strictly after the interval of EVERY issuing wave's
#    covering wait.
# ---------------------------------------------------------------------------------------------------------------------
def groups_of(c):
    """Issue groups in stream order: lists of (kind, index) per wave and K-tile."""
    if c.NPH == 4:
        return [[("B", 0), ("B", 1)], [("B", 2), ("B", 3)], [("A", 0), ("A", 1)], [("A", 2), ("A", 3)]]
    return [[("B", 0), ("B", 1), ("B", 2)], [("B", 3), ("A", 0)], [("A", 1), ("A", 2)]]


def piece_regions(c, wave, kind, i):
    """Regions (('B',) or ('A', phase)) that the bytes of this wave's piece belong to."""
    if kind == "B":
        return {("B",)}
    if c.a_ks:
        return {("A", i)}
    if c.WM == 2:
        return {("A", i)}          # piece of group i's rows
    pj = wave if i == 0 else (wave + 8 if i == 1 else 16 + (wave & 1))
    return {("A", (pj * 8) // (c.MTP * 16))}


def waits_of(c, post, nst):
    """vmcnt immediates after the issue of phase slot p (None = no wait)."""
    if c.NPH == 4:
        return {0: min(63, 8 + nst) if post else 8, 2: min(63, 6 + nst) if post else 6}
    return {1: min(63, 5 + nst) if post else 5, 2: 5}


def issue_of(c, p):
    """(K-tile offset relative to the computing K-tile, issue group) of phase slot p."""
    if c.NPH == 4:
        return [(1, 3), (2, 0), (2, 1), (2, 2)][p]
    return [(1, 2), (2, 0), (2, 1)][p]


def check_schedule(c, nkt=4, ntiles=3, nst=9, extra_epilogue_ops=7):
    NPH = c.NPH
    grps = groups_of(c)
    # global stream K-tile index s = tile * nkt + kt; stage = s & 1
    issue_bi = {}     # (wave, s, kind, i) -> interval of issue
    cover_bi = {}     # (wave, s, kind, i) -> interval of the first wait that guarantees it
    read_bi = {}      # (s, region) -> interval in which its fragments are read
    for wave in range(8):
        queue = []    # issue order: entries are DMA keys or ('st',)

        def do_wait(n, at_bi):
            done = queue[:len(queue) - n] if n < len(queue) else []
            for key in done:
                if key[0] != "st" and key not in cover_bi:
                    cover_bi[key] = at_bi

        def issue(s, grp_idx, at_bi):
            for kind, i in grps[grp_idx]:
                key = (wave, s, kind, i)
                queue.append(key)
                issue_bi[key] = at_bi

        # prologue ("interval -1"): K-tile 0 complete, NPH - 1 groups of K-tile 1, one counted wait, a barrier, phase 0's fragment reads
        for gi in range(NPH):
            issue(0, gi, -2)
        for gi in range(NPH - 1):
            issue(1, gi, -2)
        do_wait(8 if NPH == 4 else 5, -2)
        base = 0
        for t in range(ntiles):
            for kt in range(nkt):
                s = t * nkt + kt
                post = t > 0 and kt == 0
                w = waits_of(c, post, nst)
                for p in range(NPH):
                    f_bi = base + kt * NPH + p                    # this interval: cluster of phase p, reads of phase p + 1
                    regs = [("A", p)] + ([("B",)] if p == 0 else [])
                    for r in regs:
                        read_bi[(s, r)] = f_bi - 1                # ... whose fragments were read one interval earlier
                    dk, gi = issue_of(c, p)
                    issue(s + dk, gi, f_bi)
                    if p in w:
                        do_wait(w[p], f_bi)
            # epilogue: its loads are waited for inside it; at least nst stores stay queued
            for _ in range(nst + extra_epilogue_ops):
                queue.append(("st",))
            base += nkt * NPH
    errors = []
    total_s = ntiles * nkt
    for (s, r), rb in read_bi.items():
        for wave in range(8):
            for gi, grp in enumerate(grps):
                for kind, i in grp:
                    if r in piece_regions(c, wave, kind, i) or (r == ("B",) and kind == "B"):
                        key = (wave, s, kind, i)
                        if key not in cover_bi or cover_bi[key] >= rb:
                            errors.append(("RAW", c.name, "ktile", s, r, "wave", wave, kind, i, cover_bi.get(key), rb))
    for key, ib in issue_bi.items():
        wave, s, kind, i = key
        if s < 2 or s >= total_s:
            continue
        for r in piece_regions(c, wave, kind, i):
            rb = read_bi.get((s - 2, r))
            if rb is not None and ib < rb + 2:   # read in interval rb, retired by barrier rb + 1, re-fill from interval rb + 2
                errors.append(("WAR", c.name, key, "issued", ib, "fragments read in", rb))
    return errors


def all_cfgs():
    out = []
    for name in CONFIGS:
        for a_ks, b_ks, pair in itertools.product((False, True), (False, True), (False, True)):
            if a_ks and name != "256x256":
                continue     # the regional [k][64] A image exists for the 2 x 2-fragment regions of the 256-row tile only
            out.append(Cfg(name, a_ks, b_ks, pair))
    return out


def main():
    ok = True
    for c in all_cfgs():
        check_index_maps(c)
        bc = check_bank_conflicts(c)
        errs = []
        for nst in (9, 12, 18, 24, 32):
            errs += check_schedule(c, nst=nst, extra_epilogue_ops=0)
            errs += check_schedule(c, nst=nst, extra_epilogue_ops=13)
        tag = f"{c.name} A_{'KS' if c.a_ks else 'KC'} B_{'KS' if c.b_ks else 'KC'} C_{'bf16' if c.pair else 'f32'}"
        print(f"{tag:34s} index maps ok; worst bank conflict ways {bc}; schedule {'ok' if not errs else 'ERRORS'}")
        for e in errs[:10]:
            print("   ", e)
        ok = ok and not errs
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
