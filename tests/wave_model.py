"""Lane-level numpy model of fincflow_amd/csrc/finc_mfma.hip (TEST INFRASTRUCTURE).

It re-enacts, step by step and lane by lane, the schedule of `finc_wave_kernel`
for ONE (image, group) problem in canonical orientation: the skewed row-per-lane
wavefront, the z / x LDS rings with their 4-step I/O cadence (16-byte pieces, or the lane-pair
streams of 32-byte pieces: every lane moves one half of a due piece per window and LDS
redistributes it to the owner lane), the DPP row_shr
neighbour exchange, the band hand-over FIFO, the accumulator -> operand packing (16-row
tiles as they are, 4-row blocks through the permlane transpose-reduce) and
the fragment layout produced by `pack_kernel`.  The MFMA itself is modelled as
an exact fp64 matrix product, so any disagreement with the oracle is a schedule
or indexing bug, not rounding.  It exists so that the kernel's bookkeeping can
be checked on a machine without a GPU.
"""
import numpy as np

LANES = 64


def cfg(CQP, KH, KW, fwd):
    MTB, NSM = CQP // 16, (CQP % 16) // 4   # 16-row tiles (16x16x4) + 4-row blocks (4x4x1, 16 blocks)
    MT = MTB + NSM
    NKZ = CQP // 4
    NKD = CQP // 4
    NK = NKZ if fwd else NKD
    return dict(MT=MT, MTB=MTB, NSM=NSM, NKZ=NKZ, NKD=NKD, NK=NK)


def chan_d(MTB, j, q):
    if j < 4 * MTB:
        return 16 * (j >> 2) + 4 * q + (j & 3)
    return 16 * MTB + 4 * (j - 4 * MTB) + q


def zterm_is_zero(MTB, j, mt):
    """finc_tile.h finc_zterm_is_zero: every column 4j..4j+3 lies right of every row of tile mt."""
    return 4 * j > 16 * mt + 15 if mt < MTB else 4 * j > 16 * MTB + 4 * (mt - MTB) + 3


def pack_fragments(wc, CQP, fwd, scale=None, shift=None):
    """pack_kernel: returns {('z', j, mt) | ((a,b), j, mt): array[64]} in fp64, plus ('bias', mt): array[64, 4]
    (the affine map z = scale*y + shift folded in: z-term Linv*diag(scale), accumulators start from Linv*shift)."""
    Cq, _, KH, KW = wc.shape
    c = cfg(CQP, KH, KW, fwd)
    MT, MTB, NKZ, NK = c["MT"], c["MTB"], c["NKZ"], c["NK"]
    w = wc.astype(np.float64)
    L = w[:, :, KH - 1, KW - 1]
    Linv = np.linalg.inv(L) if not fwd else None
    frags = {}

    def frag(mat, colfn, nk, key):
        for j in range(nk):
            for mt in range(MT):
                v = np.zeros(LANES)
                for lane in range(LANES):
                    q, i = lane >> 4, lane & 15
                    row = 16 * mt + i if mt < MTB else 16 * MTB + 4 * (mt - MTB) + (i & 3)
                    col = colfn(j, q)
                    if row < Cq and col < Cq:
                        v[lane] = mat[row, col]
                frags[(key, j, mt)] = v

    if fwd:
        for a in range(KH):
            for b in range(KW):
                frag(w[:, :, KH - 1 - a, KW - 1 - b], lambda j, q: 4 * j + q, NKZ, (a, b))
    else:
        frag(Linv * (np.ones(Cq) if scale is None else np.asarray(scale, np.float64))[None, :], lambda j, q: 4 * j + q, NKZ, "z")
        bvec = Linv @ (np.zeros(Cq) if shift is None else np.asarray(shift, np.float64))
        for mt in range(MT):
            bias = np.zeros((LANES, 4))
            for lane in range(LANES):
                q = lane >> 4
                for r in range(4):
                    row = 16 * mt + 4 * q + r if mt < MTB else (16 * MTB + 4 * (mt - MTB) + r if q == 0 else Cq)
                    if row < Cq:
                        bias[lane, r] = bvec[row]
            frags[("bias", mt)] = bias
        for a in range(KH):
            for b in range(KW):
                if (a, b) != (0, 0):
                    frag(-(Linv @ w[:, :, KH - 1 - a, KW - 1 - b]), lambda j, q: chan_d(MTB, j, q), NK, (a, b))
    return frags


def mfma(a, b, c):
    """v_mfma_f32_16x16x4_f32 lane maps: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=4*(l>>4)+r][col=l&15]."""
    A = np.zeros((16, 4))
    Bm = np.zeros((4, 16))
    for lane in range(LANES):
        A[lane & 15, lane >> 4] = a[lane]
        Bm[lane >> 4, lane & 15] = b[lane]
    Dm = A @ Bm
    out = c.copy()
    for lane in range(LANES):
        for r in range(4):
            out[lane, r] += Dm[4 * (lane >> 4) + r, lane & 15]
    return out


def row_shr(n, old, src):
    out = old.copy()
    for lane in range(LANES):
        if (lane & 15) - n >= 0:
            out[lane] = src[lane - n]
    return out


def mfma4(a, b, c):
    """v_mfma_f32_4x4x1_16B_f32: block = lane // 4; D reg i of lane 4*blk + j += A[4*blk + i] * B[4*blk + j]
    (layout pinned on the hardware by scripts/micro/mfma4x4.hip)."""
    out = c.copy()
    for lane in range(LANES):
        blk = lane // 4
        for i in range(4):
            out[lane, i] += a[4 * blk + i] * b[lane]
    return out


def permlane32_swap(vdst, src):
    """v_permlane32_swap: new vdst = [vdst lanes 0-31, src lanes 0-31], new src = [vdst lanes 32-63, src lanes 32-63]."""
    return np.concatenate([vdst[:32], src[:32]]), np.concatenate([vdst[32:], src[32:]])


def permlane16_swap(vdst, src):
    """v_permlane16_swap: odd rows of vdst <-> even rows of src (rows of 16 lanes)."""
    v, s_ = vdst.reshape(4, 16), src.reshape(4, 16)
    nv = np.stack([v[0], s_[0], v[2], s_[2]]).reshape(64)
    ns = np.stack([v[1], s_[1], v[3], s_[3]]).reshape(64)
    return nv, ns


def pack_d(acc, c):
    MTB, NSM, NKD = c["MTB"], c["NSM"], c["NKD"]
    xpk = np.zeros((NKD, LANES))
    for mt in range(MTB):
        for r in range(4):
            xpk[4 * mt + r] = acc[mt][:, r]
    for sb in range(NSM):
        a = acc[MTB + sb]
        a0, a1 = permlane32_swap(a[:, 0], a[:, 2])
        b0, b1 = permlane32_swap(a[:, 1], a[:, 3])
        c0, c1 = permlane16_swap(a0 + a1, b0 + b1)
        xpk[4 * MTB + sb] = c0 + c1
    return xpk


def run(inp, wc, fwd=False, scale=None, shift=None):
    """inp [Cq,H,W] (z for inverse, x for forward), wc [Cq,Cq,KH,KW] canonical -> out [Cq,H,W] (fp64)."""
    CQ, H, W = inp.shape
    KH, KW = wc.shape[2:]
    CQP = (CQ + 3) // 4 * 4
    c = cfg(CQP, KH, KW, fwd)
    MT, MTB, NKZ, NKD, NK = c["MT"], c["MTB"], c["NKZ"], c["NKD"], c["NK"]
    assert W % 4 == 0
    P = min(16, W)
    assert P >= KH - 1
    NB = (H + P - 1) // P
    Tend = (NB * W + P - 1 + 7) // 8 * 8 if W % 8 == 0 else (NB * W + P - 1 + 3) // 4 * 4   # SEC: the loop is unrolled x8
    D = W - P + 1
    fr = pack_fragments(wc, CQP, fwd, scale, shift)
    out = np.full((CQ, H, W), np.nan)
    lanes = np.arange(LANES)
    q, p = lanes >> 4, lanes & 15

    SS = NK * 4 * max(KH - 1, 1)
    zring = np.full((NKZ, 12, LANES), np.nan)
    xring = np.full((NKD, 8, LANES), np.nan)
    fifo = np.zeros((D, NK, 4, max(KH - 1, 1)))
    fl4 = -((p + 3) >> 2)
    SEC = W % 8 == 0                      # 32-byte pieces (finc_wave_kernel<..., SEC=true>)
    lcol = 8 * ((fl4 - (fl4 & 1)) // 2 + (fl4 & 1)) if SEC else 4 * fl4
    lrow = p.copy()
    lph = fl4 & 1
    scol, srow = 4 * (fl4 - 2), p.copy()
    lslot = ((4 * (fl4 - 2 + (fl4 & 1))) % 12 + 12) % 12 if SEC else ((4 * fl4) % 12 + 12) % 12
    sslot = (4 * (fl4 - 2)) & 7
    zin = np.zeros((NKZ, 4, LANES))
    zb = np.zeros((NKZ, 2, 4, LANES))     # SEC: [j][piece][element][lane]
    sv = np.zeros((NKD, 4, LANES))
    st = dict(ok=np.zeros(LANES, bool), row=p.copy(), col=p.copy(), fire=np.zeros(LANES, bool))

    def io_land():                        # 16-byte pieces (W % 8 != 0)
        nonlocal lslot
        for j in range(NKZ):
            for k in range(4):
                zring[j, lslot + k, lanes] = zin[j, k]
        lslot = np.where(lslot == 8, 0, lslot + 4)

    def io_issue():                       # 16-byte pieces
        nonlocal lcol, lrow
        for lane in range(LANES):
            ok = lcol[lane] >= 0 and lrow[lane] < H and p[lane] < P
            for j in range(NKZ):
                ch = 4 * j + q[lane]
                for k in range(4):
                    zin[j, k, lane] = inp[ch, lrow[lane], lcol[lane] + k] if (ok and ch < CQ) else 0.0
        lcol = lcol + 4
        wrap = lcol == W
        lcol = np.where(wrap, 0, lcol)
        lrow = np.where(wrap, lrow + P, lrow)

    # ---- SEC (32-byte pieces), lane-pair I/O ---------------------------------------------------------------------
    # Rows of a band fall into two classes by the parity of ceil(row/4): a row's 32-byte piece (two 4-column groups)
    # is due every other window, class c rows in windows of parity c.  In such a window EVERY lane moves one 16-byte
    # half: the lanes of a class-c row its first group, the lanes of the partner row (the i-th row of the other
    # class) its second group -- so each memory instruction has all 64 lanes busy on 32 whole pieces, and the two
    # register sets (loads: Z[parity], stores: XS[parity]) alternate with the window (the loop is unrolled x8).
    rows0 = [r for r in range(P) if (((r + 3) >> 2) & 1) == 0]
    rows1 = [r for r in range(P) if (((r + 3) >> 2) & 1) == 1]
    assert not SEC or len(rows0) == len(rows1)
    partner = {}
    for a_, b_ in zip(rows0, rows1):
        partner[a_], partner[b_] = b_, a_
    cls = ((p + 3) >> 2) & 1
    svrow = np.full((2, LANES), -1)
    half = np.zeros((2, LANES), int)
    for lane in range(LANES):
        if SEC and p[lane] < P:
            for wp in (0, 1):
                own = cls[lane] == wp
                svrow[wp, lane] = p[lane] if own else partner[int(p[lane])]
                half[wp, lane] = 0 if own else 1
    fl4r = -((svrow + 3) >> 2)                       # of the service row
    e0 = np.array([-4, -3])                          # first event window of class 0 / 1
    lcolS = np.stack([4 * (e0[wp] + 4 + fl4r[wp]) for wp in (0, 1)])
    lrowS = svrow.copy()
    lslotS = np.stack([(4 * ((e0[wp] + 2 + fl4r[wp]) % 3)) for wp in (0, 1)])
    Z = np.zeros((2, NKZ, 4, LANES))
    # stores: stream wp = the class-wp row a lane serves; pair base column at its first fire window (class 1: -1, class 0: 0)
    f0 = np.array([0, -1])
    scolS = np.stack([4 * (f0[wp] - 2 + fl4r[wp]) for wp in (0, 1)])
    srowS = svrow.copy()
    XS = np.full((2, NKD, 4, LANES), np.nan)
    st2 = dict(ok=np.zeros(LANES, bool), ok2=np.zeros(LANES, bool), park=np.zeros(LANES, bool), row=np.zeros(LANES, int),
               col=np.zeros(LANES, int), par=0)
    win = [-4]                                       # window counter of the event stream (the pre-loop runs -4, -3, -2)

    # Sector pairing (W % 8 == 0): HBM is touched in whole 64-byte sectors.  Loads: a lane asks for the lower AND the upper
    # 32-byte piece of a sector in the same event (two instructions back to back: the second is an L2 hit on the line the
    # first one fetches); the lower piece lands one event later as before, the upper one waits in a second register set
    # (ZU, accumulation registers) and lands two events later -- exactly when the old schedule landed it.  Nothing is
    # requested at the event of an upper piece.  `lphi` = phase of the previous event's piece = which set lands now.
    ZU = np.zeros((2, NKZ, 4, LANES))
    lphi = np.zeros((2, LANES), int)

    def io_event():
        u = win[0]
        wp = u & 1
        for lane in range(LANES):
            R_ = svrow[wp, lane]
            if R_ < 0:
                continue
            h = half[wp, lane]
            sl = (lslotS[wp, lane] + 4 * h) % 12
            dst = q[lane] * 16 + R_
            src = Z if lphi[wp, lane] == 0 else ZU       # previous event issued a lower piece -> it lands now
            for j in range(NKZ):
                for k in range(4):
                    zring[j, sl + k, dst] = src[wp, j, k, lane]
            lslotS[wp, lane] = (lslotS[wp, lane] + 8) % 12
            c0 = lcolS[wp, lane]
            phi = ((c0 + 64) >> 3) & 1
            lphi[wp, lane] = phi
            if phi == 0:
                ok = c0 >= 0 and lrowS[wp, lane] < H
                oku = ok and c0 + 8 < W
                for j in range(NKZ):
                    ch = 4 * j + q[lane]
                    for k in range(4):
                        Z[wp, j, k, lane] = inp[ch, lrowS[wp, lane], c0 + 4 * h + k] if (ok and ch < CQ) else 0.0
                        ZU[wp, j, k, lane] = inp[ch, lrowS[wp, lane], c0 + 8 + 4 * h + k] if (oku and ch < CQ) else 0.0
            lcolS[wp, lane] += 8
            if lcolS[wp, lane] == W:
                lcolS[wp, lane] = 0
                lrowS[wp, lane] += P
        win[0] += 1

    swin = [-1]                                      # window counter of the store stream (first: the prologue, -1)

    def io_sread():
        if not SEC:
            return io_sread16()
        u = swin[0]
        c_ = u & 1                                   # the class that fires in this window
        for lane in range(LANES):
            if p[lane] >= P:
                st2["ok"][lane] = st2["ok2"][lane] = st2["park"][lane] = False
                continue
            T = svrow[c_, lane]
            h = half[c_, lane]
            if cls[lane] != c_:
                own = cls[lane]                      # own row: stream `own`, first group of its NEXT pair -> held
                oc = scolS[own, lane]
                for j in range(NKD):
                    for k in range(4):
                        XS[(u + 1) & 1, j, k, lane] = xring[j, (oc + k) & 7, q[lane] * 16 + p[lane]]
                        XS[u & 1, j, k, lane] = xring[j, (scolS[c_, lane] + 4 + k) & 7, q[lane] * 16 + T]
            # sector pairing: a lower piece (phase 0) is parked, the upper piece (phase 1) is stored together with it --
            # two store instructions back to back write one whole 64-byte sector; the odd last piece of a row
            # (W % 16 == 8) has no upper half and leaves at once
            sc0 = scolS[c_, lane]
            ok = sc0 >= 0 and srowS[c_, lane] < H
            phi = ((sc0 + 64) >> 3) & 1
            last = sc0 + 8 >= W
            st2["ok"][lane] = ok and (phi == 1 or last)
            st2["ok2"][lane] = ok and phi == 1 and sc0 - 8 >= 0
            st2["park"][lane] = phi == 0 and not last
            st2["row"][lane] = srowS[c_, lane]
            st2["col"][lane] = sc0 + 4 * h
            scolS[c_, lane] += 8
            if scolS[c_, lane] == W:
                scolS[c_, lane] = 0
                srowS[c_, lane] += P
        st2["par"] = u & 1
        swin[0] += 1

    PK = np.full((2, NKD, 4, LANES), np.nan)

    def io_swrite():
        if not SEC:
            return io_swrite16()
        par = st2["par"]
        for lane in range(LANES):
            for j in range(NKD):
                ch = chan_d(MTB, j, q[lane])
                if ch < CQ:
                    for k in range(4):
                        if st2["ok2"][lane]:
                            out[ch, st2["row"][lane], st2["col"][lane] - 8 + k] = PK[par, j, k, lane]
                        if st2["ok"][lane]:
                            out[ch, st2["row"][lane], st2["col"][lane] + k] = XS[par, j, k, lane]
            if st2["park"][lane]:
                PK[par, :, :, lane] = XS[par, :, :, lane]

    sph = fl4 & 1

    def io_sread16():
        nonlocal scol, srow, sslot
        st["ok"] = (scol >= 0) & (srow < H) & (p < P)
        st["row"], st["col"] = srow.copy(), scol
        for j in range(NKD):
            for k in range(4):
                sv[j, k] = xring[j, sslot + k, lanes]
        sslot = sslot ^ 4
        scol = scol + 4
        wrap = scol == W
        scol = np.where(wrap, 0, scol)
        srow = np.where(wrap, srow + P, srow)

    def io_swrite16():
        for lane in range(LANES):
            if st["ok"][lane]:
                for j in range(NKD):
                    ch = chan_d(MTB, j, q[lane])
                    if ch < CQ:
                        for k in range(4):
                            out[ch, st["row"][lane], st["col"][lane] + k] = sv[j, k, lane]

    def io_phase(ph):
        if SEC:
            [io_sread, io_swrite, lambda: None, io_event][ph]()
        else:
            [io_sread, io_swrite, io_land, io_issue][ph]()

    R = np.zeros((KH, KW, NK, LANES))
    DL = np.zeros((KH, KH, NK, LANES))
    fslot = 0
    push_l = p - (P - (KH - 1))
    do_push = (KH > 1) & (push_l >= 0) & (p < P)

    def fifo_push(v):
        for lane in range(LANES):
            if do_push[lane]:
                for j in range(NK):
                    fifo[fslot, j, q[lane], push_l[lane]] = v[j, lane]

    def fifo_pop(a):
        ps = 0 if fslot + 1 == D else fslot + 1
        v = np.zeros((NK, LANES))
        for lane in range(LANES):
            if p[lane] < a:
                for j in range(NK):
                    v[j, lane] = fifo[ps, j, q[lane], KH - 1 - a + p[lane]]
        return v

    def shift(a, fv, src):
        return np.stack([row_shr(a, fv[j], src[j]) for j in range(NK)])

    def shift_all(fvs, src, fwd_delay):
        for a in range(1, KH):
            sn = shift(a, fvs[a], src)
            nd = a if fwd_delay else a - 1
            if nd == 0:
                R[a, 0] = sn
            else:
                if fwd_delay:
                    R[a, 0] = DL[a, nd - 1]
                for k in range(nd - 1, 0, -1):
                    DL[a, k] = DL[a, k - 1]
                DL[a, 0] = sn

    def mm(key, j, mt, b, acc):
        return (mfma if mt < MTB else mfma4)(fr[(key, j, mt)], b, acc)

    if SEC:
        io_event(); io_event(); io_event()       # windows -4, -3, -2
    else:
        io_issue(); io_land(); io_issue()
    xs = (-4 - p) & 7
    nslot = ((-3 - p) % 12 + 12) % 12
    cn = -3 - p
    assert not fwd, 'the forward is finc_conv.hip (no skewed wavefront); only the inverse is modelled'
    if True:
        acc = [np.zeros((LANES, 4)) for _ in range(MT)]   # idle lanes stay at exact zero (the bias starts with the lane)
        for t in range(-4, Tend):
            wrapn = cn == 0
            zv = np.stack([np.where(cn >= 0, zring[j, nslot, lanes], 0.0) for j in range(NKZ)])
            for a in range(KH):
                for b in range(KW - 1, 0, -1):
                    if a + b >= 2:
                        R[a, b] = np.where(wrapn, 0.0, R[a, b - 1])
                if a >= 2:
                    R[a, 0] = DL[a, a - 2]
            for j in range(NK):
                for mt in range(MT):
                    if KW > 1:
                        acc[mt] = mm((0, 1), j, mt, R[0, 1, j], acc[mt])
                    if KH > 1:
                        acc[mt] = mm((1, 0), j, mt, R[1, 0, j], acc[mt])
            accn = [np.where((cn >= 0)[:, None], fr[("bias", mt)], 0.0) for mt in range(MT)]
            for j in range(NKZ):
                for mt in range(MT):
                    if zterm_is_zero(MTB, j, mt):      # Linv is lower triangular: the kernel skips these MFMAs
                        assert not fr[("z", j, mt)].any()
                        continue
                    accn[mt] = mm("z", j, mt, zv[j], accn[mt])
            for a in range(KH):
                for b in range(KW):
                    if a + b >= 2:
                        for j in range(NK):
                            for mt in range(MT):
                                accn[mt] = mm((a, b), j, mt, R[a, b, j], accn[mt])
            xpk = pack_d(acc, c)
            for j in range(NKD):
                xring[j, xs, lanes] = xpk[j]
            fvs = {}
            if KH > 1:
                fifo_push(xpk)
                fvs = {a: fifo_pop(a) for a in range(1, KH)}
            if KW > 1:
                R[0, 1] = np.where(wrapn, 0.0, xpk)
            io_phase(t & 3)
            shift_all(fvs, xpk, False)
            acc = accn
            cn = np.where(cn + 1 == W, 0, cn + 1)
            nslot = np.where(nslot + 1 == 12, 0, nslot + 1)
            xs = (xs + 1) & 7
            fslot = 0 if fslot + 1 == D else fslot + 1
        io_sread(); io_swrite()
    return out
