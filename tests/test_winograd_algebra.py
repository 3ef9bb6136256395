"""Host-only check of the algebra behind finc_wino.hip: the transform matrices the two Winograd kernels implement (the comments
at the top of wino_walk / wino4_walk and the constants in their `transform` lambdas, output transforms and wino_pack_kernel) are
restated here with exact rational arithmetic and must reproduce the direct correlation y[m] = sum_k g[k] d[m + k] -- the row
convolution of layers/conv.py:102-107 -- for every basis pair.  A typo in a constant of the kernel's comment block shows up
here without a GPU; the GPU tests (test_forward_forms_agree_with_fp64_conv) hold the kernels themselves to fp64 conv2d."""
from fractions import Fraction as Fr

import numpy as np

F23 = dict(
    BT=[[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],                       # V = (d0-d2, d1+d2, d2-d1, d1-d3)
    G=[[1, 0, 0], [Fr(1, 2), Fr(1, 2), Fr(1, 2)], [Fr(1, 2), Fr(-1, 2), Fr(1, 2)], [0, 0, 1]],
    AT=[[1, 1, 1, 0], [0, 1, -1, -1]],                                                    # y0 = M0+M1+M2, y1 = M1-M2-M3
)
b = Fr(3, 2)                                                                              # points 0, +-1, +-3/2, infinity
F43 = dict(
    BT=[[b * b, 0, -(1 + b * b), 0, 1, 0],                                                # 2.25 d0 - 3.25 d2 + d4
        [0, -b * b, -b * b, 1, 1, 0],                                                     # (d4 - 2.25 d2) + (d3 - 2.25 d1)
        [0, b * b, -b * b, -1, 1, 0],                                                     # (d4 - 2.25 d2) - (d3 - 2.25 d1)
        [0, -b, -1, b, 1, 0],                                                             # (d4 - d2) + 1.5 (d3 - d1)
        [0, b, -1, -b, 1, 0],                                                             # (d4 - d2) - 1.5 (d3 - d1)
        [0, b * b, 0, -(1 + b * b), 0, 1]],                                               # 2.25 d1 - 3.25 d3 + d5
    G=[[1 / (b * b), 0, 0],                                                               # g0 / 2.25
       [Fr(-2, 5), Fr(-2, 5), Fr(-2, 5)], [Fr(-2, 5), Fr(2, 5), Fr(-2, 5)],               # -(g0 +- g1 + g2) / 2.5
       [Fr(8, 45), Fr(8, 45) * b, Fr(8, 45) * b * b], [Fr(8, 45), -Fr(8, 45) * b, Fr(8, 45) * b * b],   # (g0 +- 1.5 g1 + 2.25 g2) / 5.625
       [0, 0, 1]],
    AT=[[1, 1, 1, 1, 1, 0], [0, 1, -1, b, -b, 0], [0, 1, 1, b * b, b * b, 0], [0, 1, -1, b ** 3, -b ** 3, 1]],
)


def _check(form, m, r):
    BT, G, AT = ([[Fr(x) for x in row] for row in form[k]] for k in ("BT", "G", "AT"))
    n = m + r - 1
    assert len(BT) == n and len(G) == n and len(AT) == m
    for k in range(r):                                    # filter = e_k
        U = [G[f][k] for f in range(n)]
        for j in range(n):                                # data = e_j
            V = [BT[f][j] for f in range(n)]
            for i in range(m):
                y = sum(AT[i][f] * U[f] * V[f] for f in range(n))
                assert y == (1 if j == i + k else 0), (k, j, i, y)


def test_f23_matrices_are_exact():
    _check(F23, 2, 3)


def test_f43_matrices_are_exact():
    _check(F43, 4, 3)
    # every constant of the kernel's transforms is a dyadic rational: exact in fp32
    for row in F43["BT"] + F43["AT"]:
        for x in row:
            x = Fr(x)
            assert x.denominator & (x.denominator - 1) == 0 and np.float32(float(x)) == float(x)
    # frequency 1 (the point +1) enters all four outputs with weight 1: the folded shift rides in as its start value
    assert [Fr(F43["AT"][i][1]) for i in range(4)] == [1, 1, 1, 1]


def test_f43_fp32_error_model_stays_within_the_margin_the_gpu_test_asserts():
    """The kernel's arithmetic in numpy fp32 (K = 72 accumulations per frequency, as at the c3 bank): the error of the shipped
    points must stay far inside BASELINE.json's 1e-5 and below the textbook points' (0, +-1, +-2)."""
    rng = np.random.default_rng(0)
    Cq, N = 24, 2048
    w = (rng.standard_normal((Cq, Cq, 3, 3)) * 0.05).astype(np.float32)
    x = rng.standard_normal((Cq, 3, N + 2)).astype(np.float32)
    ref = np.zeros((Cq, N))
    for a in range(3):
        for k in range(3):
            ref += w[:, :, a, k].astype(np.float64) @ x[:, a, k:k + N].astype(np.float64)

    def run(form):
        BT, G, AT = (np.array([[float(Fr(v)) for v in row] for row in form[k]]) for k in ("BT", "G", "AT"))
        U = np.einsum("fk,oiak->oiaf", G, w.astype(np.float64)).astype(np.float32)
        idx = (np.arange(N // 4) * 4)[:, None] + np.arange(6)[None, :]
        V = np.einsum("fr,iatr->iatf", BT.astype(np.float32), x[:, :, idx]).astype(np.float32)
        M = np.zeros((Cq, N // 4, 6), np.float32)
        for a in range(3):
            for i in range(Cq):
                M += U[:, i, a, None, :] * V[None, i, a, :, :]
        Y = np.einsum("mf,otf->otm", AT.astype(np.float32), M).astype(np.float32).reshape(Cq, N)
        return np.abs(Y - ref).max() / np.abs(ref).max()

    two = Fr(2)
    textbook = dict(
        BT=[[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
        G=[[Fr(1, 4), 0, 0], [Fr(-1, 6)] * 3, [Fr(-1, 6), Fr(1, 6), Fr(-1, 6)], [Fr(1, 24), Fr(1, 12), Fr(1, 6)], [Fr(1, 24), Fr(-1, 12), Fr(1, 6)], [0, 0, 1]],
        AT=[[1, 1, 1, 1, 1, 0], [0, 1, -1, two, -two, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]])
    _check(textbook, 4, 3)
    e_ship, e_text = run(F43), run(textbook)
    assert e_ship < 2.5e-6 and e_ship < e_text, (e_ship, e_text)


h = Fr(1, 2)                                                                              # points 0, +-1, +-1/2, infinity
F25 = dict(
    BT=[[1, 0, -5, 0, 4, 0],                                                              # d0 - 5 d2 + 4 d4
        [0, -1, -1, 4, 4, 0],                                                             # (4 d4 - d2) + (4 d3 - d1)
        [0, 1, -1, -4, 4, 0],                                                             # (4 d4 - d2) - (4 d3 - d1)
        [0, 1, 2, -1, -2, 0],                                                             # 2 (d2 - d4) + (d1 - d3)
        [0, -1, 2, 1, -2, 0],                                                             # 2 (d2 - d4) - (d1 - d3)
        [0, 1, 0, -5, 0, 4]],                                                             # d1 - 5 d3 + 4 d5
    G=[[1, 0, 0, 0, 0],
       [Fr(1, 6)] * 5, [Fr(1, 6), Fr(-1, 6), Fr(1, 6), Fr(-1, 6), Fr(1, 6)],              # (g0 +- g1 + g2 +- g3 + g4) / 6
       [Fr(16, 12), Fr(8, 12), Fr(4, 12), Fr(2, 12), Fr(1, 12)],                          # (16 g0 +- 8 g1 + 4 g2 +- 2 g3 + g4) / 12
       [Fr(16, 12), Fr(-8, 12), Fr(4, 12), Fr(-2, 12), Fr(1, 12)],
       [0, 0, 0, 0, Fr(1, 4)]],                                                           # g4 / 4
    AT=[[1, 1, 1, 1, 1, 0], [0, 1, -1, h, -h, 1]],                                        # y0 = M0+..+M4, y1 = (M1-M2) + (M3-M4)/2 + M5
)


def test_f25_matrices_are_exact():
    """finc_wino5.hip (5x5 banks): the comment block at the top of the file, wino5_walk's `transform`, its output transform and
    wino5_pack_kernel."""
    _check(F25, 2, 5)
    for row in F25["BT"] + F25["AT"]:
        for x in row:
            x = Fr(x)
            assert x.denominator & (x.denominator - 1) == 0 and np.float32(float(x)) == float(x)
    # frequency 1 (the point +1) enters both outputs with weight 1: the folded shift rides in as its start value
    assert [Fr(F25["AT"][i][1]) for i in range(2)] == [1, 1]


def test_f25_fp32_error_model():
    """The kernel's arithmetic in numpy fp32 at the c5 bank (Cq = 48, K = 240 accumulations per frequency, weights N(0, 0.02^2) as
    bench.py's c5): far inside BASELINE.json's 1e-5."""
    rng = np.random.default_rng(0)
    Cq, N, r, m = 48, 1024, 5, 2
    w = (rng.standard_normal((Cq, Cq, r, r)) * 0.02).astype(np.float32)
    x = rng.standard_normal((Cq, r, N + r - 1)).astype(np.float32)
    ref = np.zeros((Cq, N))
    for a in range(r):
        for k in range(r):
            ref += w[:, :, a, k].astype(np.float64) @ x[:, a, k:k + N].astype(np.float64)
    BT, G, AT = (np.array([[float(Fr(v)) for v in row] for row in F25[k]]) for k in ("BT", "G", "AT"))
    n = m + r - 1
    U = np.einsum("fk,oiak->oiaf", G, w.astype(np.float64)).astype(np.float32)
    idx = (np.arange(N // m) * m)[:, None] + np.arange(n)[None, :]
    V = np.einsum("fr,iatr->iatf", BT.astype(np.float32), x[:, :, idx]).astype(np.float32)
    M = np.zeros((Cq, N // m, n), np.float32)
    for a in range(r):
        for i in range(Cq):
            M += U[:, i, a, None, :] * V[None, i, a, :, :]
    Y = np.einsum("mf,otf->otm", AT.astype(np.float32), M).astype(np.float32).reshape(Cq, N)
    e = np.abs(Y - ref).max() / np.abs(ref).max()
    assert e < 4e-6, e


def test_f43_transposed_is_the_filter_gradient():
    """finc_gradw_wino_kernel (finc_gradw.hip): the same F(4,3) matrices read as the trilinear form T(y', g, d) = sum_f (A y')_f
    (G g)_f (B^T d)_f give the filter's gradient dg_k = sum_i y'_i d_{i+k} = sum_f G[f][k] (A y')_f (B^T d)_f with A = (A^T)^T --
    exact for every basis pair, and with the constants the kernel and gradw_wino_reduce_kernel spell out."""
    BT, G, AT = ([[Fr(x) for x in row] for row in F43[k]] for k in ("BT", "G", "AT"))
    for i in range(4):                                    # gz tile = e_i
        U = [AT[i][f] for f in range(6)]
        for j in range(6):                                # x tile = e_j
            V = [BT[f][j] for f in range(6)]
            for k in range(3):
                dg = sum(G[f][k] * U[f] * V[f] for f in range(6))
                assert dg == (1 if j == i + k else 0), (i, j, k, dg)
    # the kernel's U = A gz: g0 | e + o | e - o | e' + o' | e' - o' | g3 with e' = g0 + 2.25 g2, o' = 1.5 g1 + 3.375 g3
    assert [[AT[i][f] for i in range(4)] for f in range(6)] == [
        [1, 0, 0, 0], [1, 1, 1, 1], [1, -1, 1, -1], [1, b, b * b, b ** 3], [1, -b, b * b, -b ** 3], [0, 0, 0, 1]]
    assert (b * b, b ** 3) == (Fr(9, 4), Fr(27, 8))
    # the reduce kernel's G^T: k = 0: M0/2.25 - 0.4 (M1 + M2) + (8/45)(M3 + M4); k = 1: -0.4 (M1 - M2) + (4/15)(M3 - M4);
    # k = 2: 0.4 ((M3 + M4) - (M1 + M2)) + M5
    assert [G[f][0] for f in range(6)] == [Fr(4, 9), Fr(-2, 5), Fr(-2, 5), Fr(8, 45), Fr(8, 45), 0]
    assert [G[f][1] for f in range(6)] == [0, Fr(-2, 5), Fr(2, 5), Fr(4, 15), Fr(-4, 15), 0]
    assert [G[f][2] for f in range(6)] == [0, Fr(-2, 5), Fr(-2, 5), Fr(2, 5), Fr(2, 5), 1]


def test_f43_transposed_fp32_error_model():
    """The Winograd grad-weight's arithmetic in numpy fp32 -- frequency-domain sums over 8,192 tiles, G^T at the end -- against
    fp64: inside the 2e-5 the GPU test allows a B*H*W-term fp32 reduction."""
    rng = np.random.default_rng(1)
    T, Co, Ci = 8192, 6, 5
    gz = rng.standard_normal((Co, T, 4)).astype(np.float32)
    x = rng.standard_normal((Ci, 4 * T + 2)).astype(np.float32)
    BT, G, AT = (np.array([[float(Fr(v)) for v in row] for row in F43[k]]) for k in ("BT", "G", "AT"))
    idx = (np.arange(T) * 4)[:, None] + np.arange(6)[None, :]
    d = x[:, idx]                                                          # [Ci, T, 6]: d_j = x[4t - 2 + j] (x carries the 2-column halo)
    ref = np.stack([np.einsum("oti,cti->oc", gz.astype(np.float64), d[:, :, k:k + 4].astype(np.float64)) for k in range(3)], -1)
    U = np.einsum("if,oti->otf", AT.astype(np.float32), gz).astype(np.float32)
    V = np.einsum("fj,ctj->ctf", BT.astype(np.float32), d).astype(np.float32)
    M = np.zeros((Co, Ci, 6), np.float32)
    for t in range(T):                                                     # fp32 accumulation, tile by tile
        M += U[:, None, t, :] * V[None, :, t, :]
    got = np.einsum("fk,ocf->ock", G.astype(np.float32), M).astype(np.float32)
    e = np.abs(got - ref).max() / np.abs(ref).max()
    assert e < 1e-5, e


def test_f25_transposed_is_the_filter_gradient():
    """finc_gradw_winot_kernel<5,5> (finc_gradw.hip): F(2,5) read as a trilinear form gives the 5-tap filter's gradient from tiles of
    two columns; the constants of the kernel's U = A gz and of gradw_winot_reduce_kernel<5>'s G^T."""
    BT, G, AT = ([[Fr(x) for x in row] for row in F25[k]] for k in ("BT", "G", "AT"))
    for i in range(2):
        U = [AT[i][f] for f in range(6)]
        for j in range(6):
            V = [BT[f][j] for f in range(6)]
            for k in range(5):
                dg = sum(G[f][k] * U[f] * V[f] for f in range(6))
                assert dg == (1 if j == i + k else 0), (i, j, k, dg)
    assert [[AT[i][f] for i in range(2)] for f in range(6)] == [[1, 0], [1, 1], [1, -1], [1, h], [1, -h], [0, 1]]
    # k0 = M0 + S/6 + 4T/3; k1 = D/6 + 2E/3; k2 = S/6 + T/3; k3 = D/6 + E/6; k4 = S/6 + T/12 + M5/4  (S, D = M1 +- M2; T, E = M3 +- M4)
    assert [G[f][0] for f in range(6)] == [1, Fr(1, 6), Fr(1, 6), Fr(4, 3), Fr(4, 3), 0]
    assert [G[f][1] for f in range(6)] == [0, Fr(1, 6), Fr(-1, 6), Fr(2, 3), Fr(-2, 3), 0]
    assert [G[f][2] for f in range(6)] == [0, Fr(1, 6), Fr(1, 6), Fr(1, 3), Fr(1, 3), 0]
    assert [G[f][3] for f in range(6)] == [0, Fr(1, 6), Fr(-1, 6), Fr(1, 6), Fr(-1, 6), 0]
    assert [G[f][4] for f in range(6)] == [0, Fr(1, 6), Fr(1, 6), Fr(1, 12), Fr(1, 12), Fr(1, 4)]
