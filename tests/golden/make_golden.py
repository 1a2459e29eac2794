#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.json from the REAL reference.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Every expected value in the fixture is produced by oracle/_ref/libscl_ref.so,
i.e. by the reference's own code compiled from /root/reference (see
oracle/Makefile, oracle/ref_harness.cc).  Inputs are stored next to outputs, so
the fixture does not depend on this script's random generator.  Elements are
hex strings of the integer value; byte strings are hex.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def hx(arr):
    return [format(v, "x") for v in O.to_ints(arr)]


def gen_montgomery_field(ref, f):
    """secp256k1_order = FF<Secp256k1Scalar> (src/scl/math/fields/secp256k1_scalar.cc): elements are the
    4-limb Montgomery images of FF::m_value, written as hex of the 256-bit little-endian limb integer.
    Inputs come from FF::read over fixed bytes / FF(int), never from integers assumed to be values.
    The same sections serve the two-limb instance of the family (field tag 2, make_golden_mont128.py): E = 8 L bytes."""
    L = O.LIMBS[f]
    E = 8 * L
    rng = np.random.default_rng(4141)
    fd = {"limbs": L}
    I = lambda v: ref.from_int(f, v)
    edge = np.stack([I(0), I(1), I(2), I(-1), I(-2), I(2 ** 31 - 1), I(-(2 ** 31 - 1)), I(65536)])
    rnd = ref.from_bytes(f, rng.bytes(E * 60))
    a = np.concatenate([edge, rnd])
    b = np.concatenate([rnd[::-1], edge[::-1]])
    zero = I(0)
    nz = np.stack([x for x in a if not np.array_equal(x, zero)])
    nzb = np.stack([x for x in b if not np.array_equal(x, zero)])[: nz.shape[0]]
    fd["ew"] = {"a": hx(a), "b": hx(b), "add": hx(ref.ew(f, O.ADD, a, b)), "sub": hx(ref.ew(f, O.SUB, a, b)),
                "mul": hx(ref.ew(f, O.MUL, a, b)), "neg": hx(ref.ew(f, O.NEG, a)), "nz": hx(nz), "nzb": hx(nzb),
                "inv": hx(ref.ew(f, O.INV, nz)), "div": hx(ref.ew(f, O.DIV, nz[: nzb.shape[0]], nzb))}
    try:
        ref.ew(f, O.INV, zero.reshape(1, L))
        fd["inv0_error"] = None
    except O.OracleError as e:
        fd["inv0_error"] = e.message
    ints = [0, 1, -1, 5, -5, 123, 2 ** 31 - 1, -(2 ** 31 - 1), 65536, -65536]
    fd["from_int"] = {"in": ints, "out": [hx(I(v).reshape(1, L))[0] for v in ints]}
    raw = b"\x00" * E + b"\xff" * E + bytes(range(E)) + rng.bytes(E * 40)
    fd["from_bytes"] = {"raw": raw.hex(), "out": hx(ref.from_bytes(f, raw))}
    hexes = ["7b", "41621e", "00", "ffffffffffffffffffffffffffffffff",
             "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141",
             "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364140",
             "ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff", "0123456789ABCDEFabcdef"]
    if L == 2:   # (beyond 32 digits the reference's template writes past a two-limb value: not exercised)
        hexes = ["7b", "41621e", "00", "", "ffffffffffffffffffffffffffffffff", "ffffffffffffffffffffffffffffff61",
                 "ffffffffffffffffffffffffffffff60", "7fffffffffffffffffffffffffffffff", "0123456789ABCDEFabcdef",
                 "123456789abcdef0f", "0000000000000001", "fedcba9876543210fedcba987654321"]
    fh = {"in": hexes, "out": [], "to_hex": [], "errors": {}}
    for h in hexes:
        v = ref.from_hex(f, h)
        fh["out"].append(hx(v.reshape(1, L))[0])
        fh["to_hex"].append(ref.to_hex(f, v))
    for bad in ("abc", "zz"):
        try:
            ref.from_hex(f, bad)
            fh["errors"][bad] = None
        except O.OracleError as e:
            fh["errors"][bad] = e.message
    fd["hex"] = fh
    fd["to_hex"] = {"in": hx(a[:30]), "out": [ref.to_hex(f, a[i]) for i in range(30)]}
    fd["exp"] = [{"base": hx(a[i].reshape(1, L))[0], "e": e, "out": hx(ref.exp(f, a[i], e).reshape(1, L))[0]}
                 for i, e in ((9, 0), (10, 1), (11, 2), (12, 65537), (13, 2 ** 63 + 12345))]
    fd["vector_random"] = [{"seed": s.hex(), "n": n, "out": hx(ref.vector_random(f, s, n))}
                           for s, n in ((b"shamir passive", 4), (b"", 1), (b"vec", 9))]
    sh = []
    for (n, t, N, seed) in ((4, 3, 6, b"shamir passive"), (10, 3, 12, b"scl-bench-c2"), (10, 0, 3, b"t0"),
                            (40, 13, 4, b"scl-bench-c4"), (7, 6, 3, b"full-degree")):
        secrets = np.concatenate([I(123).reshape(1, L), ref.from_bytes(f, rng.bytes(E * (N - 2))), I(-1).reshape(1, L)])
        shares = ref.shamir_share(f, seed, secrets, t, n)
        alph = np.stack([I(i + 1) for i in range(n)])
        sh.append({"n": n, "t": t, "seed": seed.hex(), "secrets": hx(secrets), "shares": hx(shares),
                   "recovered_all_n": hx(ref.shamir_recover(f, shares)),
                   "lambda_1_to_n_at_0": hx(ref.lagrange_basis(f, alph, I(0)))})
    fd["shamir"] = sh
    lb = []
    for nodes, x in (([4, 5, 6, 7, 8, 9], 0), ([4, 5, 6, 7, 8, 9], 27), ([1, 2, 3], 2), ([1], 0), ([5, 3, 9, 1 << 20], -7)):
        nd = np.stack([I(v) for v in nodes])
        lb.append({"nodes": nodes, "x": x, "out": hx(ref.lagrange_basis(f, nd, I(x)))})
    fd["lagrange"] = lb
    try:
        ref.lagrange_basis(f, np.stack([I(1), I(2), I(2)]), I(0))
        fd["lagrange_dup_error"] = None
    except O.OracleError as e:
        fd["lagrange_dup_error"] = e.message
    coeffs = np.concatenate([I(123).reshape(1, L), rnd[:3]])
    xs = np.stack([I(v) for v in range(42, 49)])
    ys = ref.poly_eval(f, coeffs, xs)
    fd["poly_eval"] = {"coeffs": hx(coeffs), "xs": hx(xs), "ys": hx(ys)}
    fd["recover_at"] = [{"alphas": hx(xs), "x": hx(xv.reshape(1, L))[0], "shares": hx(ys),
                         "out": hx(ref.shamir_recover_at(f, ys.reshape(1, 7, L), xs, xv))[0]}
                        for xv in (I(0), xs[0], I(1000))]
    secrets = np.stack([I(123), I(5), I(-1), I(77), I(0), I(9)])
    shares = ref.shamir_share(f, b"shamir detect", secrets, 4, 9)
    shares[1, 2] = I(4)
    shares[2, 5] = I(4)
    shares[3, 8] = I(4)
    shares[4, 7] = I(1)
    out, status = ref.shamir_recover_d(f, shares, 4)
    fd["recover_d"] = {"t": 4, "n": 9, "shares": hx(shares), "out": hx(out), "status": status.tolist()}
    ad = []
    for (n, N, seed) in ((3, 8, b""), (10, 3, b"additive"), (1, 2, b"one")):
        secrets = np.concatenate([I(12345).reshape(1, L), ref.from_bytes(f, rng.bytes(E * (N - 1)))])
        shares = ref.additive_share(f, seed, secrets, n)
        ad.append({"n": n, "seed": seed.hex(), "secrets": hx(secrets), "shares": hx(shares),
                   "sum": hx(ref.additive_recover(f, shares))})
    fd["additive"] = ad
    fd["dot"] = {"a": hx(a), "b": hx(b), "out": hx(ref.dot(f, a, b).reshape(1, L))[0]}
    fd["sum"] = {"a": hx(a), "out": hx(ref.sum(f, a).reshape(1, L))[0]}
    fd["scalar_mul"] = {"a": hx(a), "scalar": hx(b[3].reshape(1, L))[0], "out": hx(ref.scalar_mul(f, a, b[3]))}
    xs5 = np.stack([I(7), I(-1), I(0), I(1), rnd[5]])
    fd["vandermonde"] = [{"n": 3, "m": 3, "xs": None, "out": hx(ref.vandermonde(f, 3, 3))},
                         {"n": 10, "m": 4, "xs": None, "out": hx(ref.vandermonde(f, 10, 4))},
                         {"n": 5, "m": 6, "xs": hx(xs5), "out": hx(ref.vandermonde(f, 5, 6, xs5))}]
    A = rnd[:20].reshape(2, 10, L)
    B = rnd[20:50].reshape(10, 3, L)
    fd["matmul"] = [{"n": 2, "k": 10, "m": 3, "A": hx(A), "B": hx(B), "C": hx(ref.matmul(f, A, B))}]
    V = ref.vandermonde(f, 10, 4)
    Cm = rnd[30:50].reshape(4, 5, L)
    fd["vandermonde_eval"] = {"n": 10, "m": 4, "N": 5, "C": hx(Cm), "out": hx(ref.matmul(f, V, Cm))}
    fd["wire"] = [{"elems": hx(a[:12]), "bytes": ref.wire_vector(f, a[:12]).hex()},
                  {"elems": [], "bytes": ref.wire_vector(f, np.zeros((0, L), np.uint64)).hex()}]
    fd["wire_matrix"] = [{"rows": 2, "cols": 10, "elems": hx(A), "bytes": ref.wire_matrix(f, A).hex()},
                         {"rows": 0, "cols": 0, "elems": [], "bytes": ref.wire_matrix(f, np.zeros((0, 0, L), np.uint64)).hex()}]
    fd["frame"] = gen_frames(ref, f, a[:12], A)
    return fd


def gen_recover_c(ref, f, rng):
    """shamirRecoverC (Berlekamp-Welch, shamir.h:202-259; test/scl/ss/test_shamir.cc:111-160 restated at more
    shapes): shares with 0 .. t+2 corrupted positions, so both the correcting and the failing regime are held"""
    L = O.LIMBS[f]
    out = []
    for (n, t) in ((4, 1), (7, 2), (10, 3), (13, 4), (12, 3)):
        N = 2 * (t + 3)
        secrets = ref.from_bytes(f, rng.bytes(8 * L * N))
        shares = ref.shamir_share(f, b"recover-c", secrets, t, n)
        for s in range(N):
            for i in rng.choice(n, size=min(s % (t + 3), n), replace=False):
                shares[s, i] = ref.from_bytes(f, rng.bytes(8 * L))[0]
        shares[N - 1] = 0
        fo, eo, st, ne = ref.shamir_recover_c(f, shares)
        out.append({"n": n, "t": t, "secrets": hx(secrets), "shares": hx(shares), "f": hx(fo), "err": hx(eo),
                    "status": st.tolist(), "nerr": ne.tolist()})
    al = np.stack([ref.from_int(f, v) for v in (42, 43, 44, 45, 46, 47, 48)])
    sec = ref.from_bytes(f, rng.bytes(8 * L * 5))
    co = ref.from_bytes(f, rng.bytes(8 * L * 10)).reshape(5, 2, L)
    sh = np.stack([ref.poly_eval(f, np.concatenate([sec[s:s + 1], co[s]]), al) for s in range(5)])
    sh[1, 2] = ref.from_int(f, 7)
    sh[2, 0] = ref.from_int(f, 9)
    sh[2, 6] = ref.from_int(f, 9)
    fo, eo, st, ne = ref.shamir_recover_c(f, sh, al)
    out.append({"n": 7, "t": 2, "alphas": hx(al), "secrets": hx(sec), "shares": hx(sh), "f": hx(fo), "err": hx(eo),
                "status": st.tolist(), "nerr": ne.tolist()})
    return out


def gen_frames(ref, f, vec, mat):
    """TcpChannel frames (tcp_channel.h:125-160): u32 packet size || Packet bytes, for a Packet with one Vector / Matrix"""
    L = O.LIMBS[f]
    return [{"kind": "vector", "elems": hx(vec), "bytes": ref.frame(f, vec).hex()},
            {"kind": "vector", "elems": [], "bytes": ref.frame(f, np.zeros((0, L), np.uint64)).hex()},
            {"kind": "matrix", "rows": int(mat.shape[0]), "cols": int(mat.shape[1]), "elems": hx(mat),
             "bytes": ref.frame(f, mat, as_matrix=True).hex()}]


def gen_ring(ref, K):
    """Z2k<K> (include/scl/math/z2k.h:39-320; cases of test/scl/math/test_z2k.cc restated): values cross as the
    masked little-endian word Z2k::write emits, zero-extended to 1 (K <= 64) or 2 limbs"""
    f, L = O.Z2K(K), O.LIMBS[O.Z2K(K)]
    rng = np.random.default_rng(1000 + K)
    I = lambda v: ref.from_int(f, v)
    fd = {"bits": K, "limbs": L, "byte_size": O.byte_size(f)}
    raw = rng.bytes(O.byte_size(f) * 40)
    a = ref.from_bytes(f, raw)
    b = ref.from_bytes(f, rng.bytes(O.byte_size(f) * 40))
    a[:6] = np.stack([I(0), I(1), I(-1), I(2), I(-2), I(3)])
    b[:6] = np.stack([I(-1), I(-1), I(-1), I(5), I(7), I(0)])
    fd["from_bytes"] = {"bytes": raw.hex(), "out": hx(ref.from_bytes(f, raw))}
    ew = {"a": hx(a), "b": hx(b)}
    for name, op in (("add", O.ADD), ("sub", O.SUB), ("mul", O.MUL)):
        ew[name] = hx(ref.ew(f, op, a, b))
    ew["neg"] = hx(ref.ew(f, O.NEG, a))
    fd["ew"] = ew
    odd = a.copy()
    odd[:, 0] |= np.uint64(1)
    fd["inverse"] = {"in": hx(odd), "out": hx(ref.ew(f, O.INV, odd)), "div_b_by_in": hx(ref.ew(f, O.DIV, b, odd))}
    try:
        ref.ew(f, O.INV, I(2).reshape(1, L))
        fd["inverse_even_error"] = None
    except O.OracleError as e:
        fd["inverse_even_error"] = e.message
    fd["vector_random"] = [{"seed": s.hex(), "n": n, "out": hx(ref.vector_random(f, s, n))}
                           for s, n in ((b"shamir passive", 4), (b"", 1), (b"vec", 33))]
    ad = []
    for (n, N, seed) in ((3, 8, b""), (10, 3, b"additive"), (1, 2, b"one")):
        secrets = np.concatenate([I(12345).reshape(1, L), ref.from_bytes(f, rng.bytes(O.byte_size(f) * (N - 1)))])
        shares = ref.additive_share(f, seed, secrets, n)
        ad.append({"n": n, "seed": seed.hex(), "secrets": hx(secrets), "shares": hx(shares),
                   "sum": hx(ref.additive_recover(f, shares))})
    fd["additive"] = ad
    fd["dot"] = {"a": hx(a), "b": hx(b), "out": hx(ref.dot(f, a, b).reshape(1, L))[0]}
    fd["sum"] = {"a": hx(a), "out": hx(ref.sum(f, a).reshape(1, L))[0]}
    fd["scalar_mul"] = {"a": hx(a), "scalar": hx(b[7].reshape(1, L))[0], "out": hx(ref.scalar_mul(f, a, b[7]))}
    A, B = a[:20].reshape(2, 10, L), b[:30].reshape(10, 3, L)
    fd["matmul"] = [{"n": 2, "k": 10, "m": 3, "A": hx(A), "B": hx(B), "C": hx(ref.matmul(f, A, B))}]
    return fd


def main():
    ref = O.Ref()
    rng = np.random.default_rng(20261003)
    doc = {"generator": "tests/golden/make_golden.py", "source": "oracle/_ref/libscl_ref.so (reference 0.1.0)",
           "fields": {}, "prg": []}

    # ---- PRG streams (src/scl/util/prg.cc) ----
    for seed in (b"", b"shamir passive", b"0123456789abcdefXYZ-longer-than-16", b"\x00\x01\x02"):
        for sizes in ([16], [8, 8, 8], [1, 15, 16, 17, 31, 32, 33, 0, 100], [4096]):
            doc["prg"].append({"seed": seed.hex(), "sizes": sizes, "out": ref.prg(seed, sizes).hex()})

    for f, name in ((O.M61, "Mersenne61"), (O.M127, "Mersenne127")):
        L, p = O.LIMBS[f], O.P[f]
        fd = {"limbs": L, "p": format(p, "x")}
        edge = [0, 1, 2, 3, p - 1, p - 2, p - 3, (p - 1) // 2, (p + 1) // 2, 1 << 32, (1 << 32) - 1,
                (1 << 60), (1 << 60) + 1, 0xFFFFFFFF00000000 % p, 0x123456789ABCDEF]
        rnd = [int.from_bytes(rng.bytes(16), "little") % p for _ in range(100)]
        va = edge + rnd
        vb = rnd[::-1] + edge[::-1]
        a, b = O.from_ints(va, L), O.from_ints(vb, L)
        nz = O.from_ints([v for v in va if v], L)
        nzb = O.from_ints([v for v in vb if v][: nz.shape[0]], L)
        fd["ew"] = {
            "a": hx(a), "b": hx(b),
            "add": hx(ref.ew(f, O.ADD, a, b)), "sub": hx(ref.ew(f, O.SUB, a, b)),
            "mul": hx(ref.ew(f, O.MUL, a, b)), "neg": hx(ref.ew(f, O.NEG, a)),
            "nz": hx(nz), "nzb": hx(nzb[: nz.shape[0]]),
            "inv": hx(ref.ew(f, O.INV, nz)),
            "div": hx(ref.ew(f, O.DIV, nz[: nzb.shape[0]], nzb)),
        }
        # all-pairs products of the edge values
        ea = O.from_ints([x for x in edge for _ in edge], L)
        eb = O.from_ints([y for _ in edge for y in edge], L)
        fd["ew_edge_pairs"] = {"a": hx(ea), "b": hx(eb), "mul": hx(ref.ew(f, O.MUL, ea, eb)),
                               "add": hx(ref.ew(f, O.ADD, ea, eb)), "sub": hx(ref.ew(f, O.SUB, ea, eb))}
        try:
            ref.ew(f, O.INV, O.from_ints([0], L))
            fd["inv0_error"] = None
        except O.OracleError as e:
            fd["inv0_error"] = e.message

        ints = [0, 1, -1, 5, -5, 123, 2 ** 31 - 1, -(2 ** 31), 65536, -65536]
        fd["from_int"] = {"in": ints, "out": [hx(ref.from_int(f, v).reshape(1, L))[0] for v in ints]}

        # fromBytes: "% p" of the raw little-endian word, incl. pre-reduction edge words
        words = [0, 1, p, p + 1, p - 1, (1 << (64 * L)) - 1, 1 << 61, (1 << 64) - 1, 2 * p, 2 * p + 5]
        words = [w % (1 << (64 * L)) for w in words] + [int.from_bytes(rng.bytes(8 * L), "little") for _ in range(40)]
        raw = b"".join(w.to_bytes(8 * L, "little") for w in words)
        fd["from_bytes"] = {"raw": raw.hex(), "out": hx(ref.from_bytes(f, raw))}

        hexes = ["7b", "41621e", "00", "0000000000000001", "ffffffffffffffff", "1fffffffffffffff",
                 "2000000000000000", "ABCDEF0123456789", "58797a14d0653d22a05c11c60e1aacf4",
                 "80000000000000000000000000000000", "7fffffffffffffffffffffffffffffff",
                 "ffffffffffffffffffffffffffffffff", "0123456789abcdef0123456789abcdef01"]
        fh = {"in": hexes, "out": [], "to_hex": []}
        for h in hexes:
            v = ref.from_hex(f, h)
            fh["out"].append(hx(v.reshape(1, L))[0])
            fh["to_hex"].append(ref.to_hex(f, v))
        errs = {}
        for bad in ("abc", "zz", "0g"):
            try:
                ref.from_hex(f, bad)
                errs[bad] = None
            except O.OracleError as e:
                errs[bad] = e.message
        fh["errors"] = errs
        fd["hex"] = fh
        fd["to_hex"] = {"in": hx(a[:40]), "out": [ref.to_hex(f, a[i]) for i in range(40)]}

        fd["exp"] = [{"base": hx(a[i].reshape(1, L))[0], "e": e, "out": hx(ref.exp(f, a[i], e).reshape(1, L))[0]}
                     for i, e in ((15, 0), (16, 1), (17, 2), (18, 65537), (19, 2 ** 61 - 3), (20, 2 ** 63 + 12345))]

        fd["vector_random"] = [{"seed": s.hex(), "n": n, "out": hx(ref.vector_random(f, s, n))}
                               for s, n in ((b"shamir passive", 4), (b"", 1), (b"vec", 43), (b"vec", 7))]

        # ---- Shamir ----
        sh = []
        for (n, t, N, seed) in ((4, 3, 8, b"shamir passive"), (3, 1, 8, b"s31"), (10, 3, 32, b"scl-bench-c2"),
                                (10, 0, 4, b"t0"), (40, 13, 12, b"scl-bench-c4"), (128, 42, 6, b"scl-bench-c5"),
                                (100, 5, 3, b"shamir recons"), (7, 6, 5, b"full-degree")):
            secrets = O.from_ints([123] + [int.from_bytes(rng.bytes(16), "little") % p for _ in range(N - 2)] + [p - 1], L)
            shares = ref.shamir_share(f, seed, secrets, t, n)
            rec = ref.shamir_recover(f, shares)
            alph = O.from_ints(list(range(1, n + 1)), L)
            lam = ref.lagrange_basis(f, alph, ref.from_int(f, 0))
            sh.append({"n": n, "t": t, "seed": seed.hex(), "secrets": hx(secrets), "shares": hx(shares),
                       "recovered_all_n": hx(rec), "lambda_1_to_n_at_0": hx(lam)})
        fd["shamir"] = sh

        # Lagrange bases with explicit nodes / evaluation points (test/scl/ss/test_shamir.cc:42-66)
        lb = []
        for nodes, x in (([4, 5, 6, 7, 8, 9], 0), ([4, 5, 6, 7, 8, 9], 27), ([42, 43, 44, 45], 0), ([1, 2, 3], 2),
                         ([1], 0), ([5, 3, 9, 1 << 20], -7)):
            nd = O.from_ints(nodes, L)
            lb.append({"nodes": nodes, "x": x, "out": hx(ref.lagrange_basis(f, nd, ref.from_int(f, x)))})
        fd["lagrange"] = lb
        try:
            ref.lagrange_basis(f, O.from_ints([1, 2, 2], L), ref.from_int(f, 0))
            fd["lagrange_dup_error"] = None
        except O.OracleError as e:
            fd["lagrange_dup_error"] = e.message

        # recover at other nodes/x: polynomial evaluated at 42..48 (test_shamir.cc:81-109)
        coeffs = O.from_ints([123] + rnd[:3], L)
        xs = O.from_ints(list(range(42, 49)), L)
        ys = ref.poly_eval(f, coeffs, xs)
        fd["poly_eval"] = {"coeffs": hx(coeffs), "xs": hx(xs), "ys": hx(ys)}
        fd["recover_at"] = [{"alphas": hx(xs), "x": hx(xv.reshape(1, L))[0], "shares": hx(ys),
                             "out": hx(ref.shamir_recover_at(f, ys.reshape(1, 7, L), xs, xv))[0]}
                            for xv in (ref.from_int(f, 0), xs[0], ref.from_int(f, 1000))]

        # error detection, short overload (test_shamir.cc:68-79)
        secrets = O.from_ints([123, 5, p - 1, 77, 0, 9], L)
        shares = ref.shamir_share(f, b"shamir detect", secrets, 4, 9)
        shares[1, 2] = ref.from_int(f, 4)      # checked? (no: index 2 < d+1 defines the polynomial)
        shares[2, 5] = ref.from_int(f, 4)      # index 5 = t+1 is checked
        shares[3, 8] = ref.from_int(f, 4)      # index 8 = 2t is NOT checked (quirk, SURVEY note E)
        shares[4, 7] = ref.from_int(f, 1)      # index 7 = 2t-1 is checked
        out, status = ref.shamir_recover_d(f, shares, 4)
        fd["recover_d"] = {"t": 4, "n": 9, "shares": hx(shares), "out": hx(out), "status": status.tolist()}

        # ---- additive ----
        ad = []
        for (n, N, seed) in ((3, 16, b""), (10, 6, b"additive"), (1, 3, b"one"), (2, 5, b"two")):
            secrets = O.from_ints([12345] + [int.from_bytes(rng.bytes(16), "little") % p for _ in range(N - 1)], L)
            shares = ref.additive_share(f, seed, secrets, n)
            ad.append({"n": n, "seed": seed.hex(), "secrets": hx(secrets), "shares": hx(shares),
                       "sum": hx(ref.additive_recover(f, shares))})
        fd["additive"] = ad

        # ---- vector / matrix ----
        fd["dot"] = {"a": hx(a), "b": hx(b), "out": hx(ref.dot(f, a, b).reshape(1, L))[0]}
        fd["sum"] = {"a": hx(a), "out": hx(ref.sum(f, a).reshape(1, L))[0]}
        fd["scalar_mul"] = {"a": hx(a), "scalar": hx(b[3].reshape(1, L))[0], "out": hx(ref.scalar_mul(f, a, b[3]))}
        xs = O.from_ints([7, p - 1, 0, 1, rnd[5]], L)
        fd["vandermonde"] = [{"n": 3, "m": 3, "xs": None, "out": hx(ref.vandermonde(f, 3, 3))},
                             {"n": 10, "m": 4, "xs": None, "out": hx(ref.vandermonde(f, 10, 4))},
                             {"n": 5, "m": 6, "xs": hx(xs), "out": hx(ref.vandermonde(f, 5, 6, xs))},
                             {"n": 128, "m": 43, "xs": None, "sha": None, "out": hx(ref.vandermonde(f, 128, 43))}]
        A = O.from_ints(rnd[:20], L).reshape(2, 10, L)
        B = O.from_ints(rnd[20:50], L).reshape(10, 3, L)
        A2 = O.from_ints([1, 2, 3, 4], L).reshape(2, 2, L)
        B2 = O.from_ints([5, 6, 7, 8], L).reshape(2, 2, L)
        fd["matmul"] = [{"n": 2, "k": 10, "m": 3, "A": hx(A), "B": hx(B), "C": hx(ref.matmul(f, A, B))},
                        {"n": 2, "k": 2, "m": 2, "A": hx(A2), "B": hx(B2), "C": hx(ref.matmul(f, A2, B2))}]
        # Vandermonde evaluation == sharing (test/scl/math/test_matrix.cc:342-365)
        V = ref.vandermonde(f, 10, 4)
        Cm = O.from_ints(rnd[50:50 + 4 * 5], L).reshape(4, 5, L)
        fd["vandermonde_eval"] = {"n": 10, "m": 4, "N": 5, "C": hx(Cm), "out": hx(ref.matmul(f, V, Cm))}
        # ---- wire image (seri::Serializer<Vector<FF>>, test/scl/serialization/test_serializer.cc:106-123) ----
        wv = O.from_ints([1, 2, 3], L)
        fd["wire"] = [{"elems": hx(wv), "bytes": ref.wire_vector(f, wv).hex()},
                      {"elems": hx(a[:40]), "bytes": ref.wire_vector(f, a[:40]).hex()},
                      {"elems": [], "bytes": ref.wire_vector(f, np.zeros((0, L), np.uint64)).hex()}]
        # Serializer<Matrix> (matrix.h:910-963; test/scl/serialization/test_serializer.cc matrix case)
        fd["wire_matrix"] = [{"rows": 2, "cols": 10, "elems": hx(A), "bytes": ref.wire_matrix(f, A).hex()},
                             {"rows": 2, "cols": 2, "elems": hx(A2), "bytes": ref.wire_matrix(f, A2).hex()},
                             {"rows": 0, "cols": 0, "elems": [],
                              "bytes": ref.wire_matrix(f, np.zeros((0, 0, L), np.uint64)).hex()}]
        fd["frame"] = gen_frames(ref, f, a[:40], A)
        doc["fields"][name] = fd

    doc["fields"]["secp256k1_order"] = gen_montgomery_field(ref, O.SECP256K1_SCALAR)

    for name, f in (("Mersenne61", O.M61), ("Mersenne127", O.M127), ("secp256k1_order", O.SECP256K1_SCALAR)):
        doc["fields"][name]["recover_c"] = gen_recover_c(ref, f, np.random.default_rng(77 + f))
        # shamirSecretShare over math::Array<FF, W> (pedersen.h:138: W = 2): secrets [N][W], shares [N][n][W]
        L = O.LIMBS[f]
        rng = np.random.default_rng(99 + f)
        pk = []
        for (W, n, t, N, seed) in ((2, 4, 3, 5, b"pedersen"), (2, 10, 3, 4, b"pedersen-2"), (3, 5, 2, 3, b"array3")):
            sec = ref.from_bytes(f, rng.bytes(8 * L * W * N)).reshape(N, W, L)
            pk.append({"W": W, "n": n, "t": t, "seed": seed.hex(), "secrets": hx(sec.reshape(-1, L)),
                       "shares": hx(ref.shamir_share_packed(f, seed, sec, t, n).reshape(-1, L))})
        doc["fields"][name]["shamir_packed"] = pk
    doc["rings"] = {f"Z2k<{K}>": gen_ring(ref, K) for K in O.REF_RING_BITS}

    path = os.path.join(HERE, "golden_v1.json")
    with open(path, "w") as fh_:
        json.dump(doc, fh_, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")

    # FF<Secp256k1Field> (src/scl/math/fields/secp256k1_field.cc:43-135), the second N = 4 instance of the Montgomery
    # family: its own small fixture file, same sections as secp256k1_order
    f = O.SECP256K1_FIELD
    fd = gen_montgomery_field(ref, f)
    fd["recover_c"] = gen_recover_c(ref, f, np.random.default_rng(77 + f))
    L = O.LIMBS[f]
    rng = np.random.default_rng(99 + f)
    pk = []
    for (W, n, t, N, seed) in ((2, 4, 3, 5, b"pedersen"), (3, 5, 2, 3, b"array3")):
        sec = ref.from_bytes(f, rng.bytes(8 * L * W * N)).reshape(N, W, L)
        pk.append({"W": W, "n": n, "t": t, "seed": seed.hex(), "secrets": hx(sec.reshape(-1, L)),
                   "shares": hx(ref.shamir_share_packed(f, seed, sec, t, n).reshape(-1, L))})
    fd["shamir_packed"] = pk
    doc2 = {"generator": doc["generator"], "source": doc["source"], "fields": {"secp256k1_field": fd}}
    path2 = os.path.join(HERE, "golden_secp256k1_field.json")
    with open(path2, "w") as fh_:
        json.dump(doc2, fh_, indent=0, separators=(",", ":"))
    print("wrote", path2, os.path.getsize(path2), "bytes")


if __name__ == "__main__":
    main()
