#!/usr/bin/env python3
"""Differential fuzzing of the batch entry points against the CPU oracle (oracle/scl_oracle.c through tests/oracle_lib.py) on the
GPU box: random field, entry point, sizes (0, 1, odd, tile edges +-1), party counts, thresholds, zeros planted in inverse
operands, in-place outputs.  Test infrastructure like tests/: prints one line per mismatch with the seed that reproduces it and
a summary; exit code 1 if anything differed.

    python3 tests/fuzz_abi.py [seconds=240] [seed=1]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "secure-computation-library_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))     # oracle_lib: the oracle is for tests/ only, and so is this file
import torch  # noqa: E402
import scl_amd as scl  # noqa: E402
import oracle_lib as O  # noqa: E402

FIELDS = [scl.M61, scl.M127, scl.MONT128, scl.GF2_128, scl.SECP256K1_SCALAR, scl.SECP256K1_FIELD]
EDGES = [0, 1, 2, 3, 63, 64, 65, 127, 255, 256, 257, 511, 512, 513, 1023, 1025, 4095, 4096, 4097, 8191, 8193, 16385, 65537]


# tuning knobs (scl_hip_set_tuning) and the values the library documents for them; defaults in KNOB_DEFAULTS
KNOBS = {"force_table": [1, 2], "prg_two_pass": [-1, 1], "nontemporal": [0], "force_scalar": [1], "stream_block": [256],
         "stream_waves": [0, 4, 8], "share_waves": [0, 6, 12], "share_waves128": [0, 8, 16], "max_blocks": [1, 7, 300],
         "gf_tiles": [0], "mfma": [-1, 1, 2], "mfma_areg": [0], "mfma_pipe": [0, 1], "inv_batch": [-1, 8, 16, 32, 64, 128], "inv_two_level": [-1, 4, 8],
         "transpose_tile": [64, 128, 256], "gemm_slab_mib": [1, 4], "matmul_lds_min": [1024, 4096], "aes_blocks": [1, 64], "prg_t3": [0]}
KNOB_DEFAULTS = {"force_table": 0, "prg_two_pass": 0, "nontemporal": 1, "force_scalar": 0, "stream_block": 64, "stream_waves": -1,
                 "share_waves": 9, "share_waves128": 12, "max_blocks": 0, "gf_tiles": 1, "mfma": 0, "mfma_areg": 1, "mfma_pipe": 2,
                 "inv_batch": 0, "inv_two_level": 0, "transpose_tile": 0, "gemm_slab_mib": 0, "matmul_lds_min": 0, "aes_blocks": 0, "prg_t3": 1}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    port = O.Port()
    scl.set_mont128_prime(2 ** 128 - 159)
    port.mont128_set_prime(2 ** 128 - 159)
    rng = np.random.default_rng(seed0)
    t_end = time.time() + budget
    runs, bad, by_kind, secs, by_knob = 0, 0, {}, {}, {}

    def size(cap):
        r = rng.random()
        if r < 0.5:
            return int(min(cap, EDGES[rng.integers(len(EDGES))]))
        return int(rng.integers(0, cap + 1))

    def rand(f, n, tag):
        return port.vector_random(f, b"fuzz-%d-%s" % (runs, tag), n) if n else np.zeros((0, scl.limbs(f)), dtype=np.uint64)

    def dev(a):
        a = np.ascontiguousarray(a)
        if a.ndim == 2 and a.shape[1] == 1 and a.shape[0] and rng.random() < 0.3:
            # a Mersenne61 vector that starts 8 bytes into a 16-byte pack (legal: elements are 8-byte aligned): the kernels must take
            # their one-element-per-lane forms
            wide = scl.to_device(np.concatenate([np.zeros((1, 1), dtype=np.uint64), a]))
            return wide[1:]
        return scl.to_device(a)

    def report(kind, detail):
        nonlocal bad
        bad += 1
        print(f"MISMATCH {kind}: {detail} knob {knob} (run {runs}, seed {seed0})", flush=True)

    t_note = time.time() + 60
    while time.time() < t_end:
        runs += 1
        if time.time() > t_note:       # a line a minute: a silent GPU job is taken for a hung one
            print(f"fuzz_abi: {runs} cases so far, {bad} mismatches", flush=True)
            t_note = time.time() + 60
        f = FIELDS[rng.integers(len(FIELDS))]
        L = scl.limbs(f)
        slow = L == 4 or f in (scl.MONT128, scl.GF2_128)   # Fermat / bit-serial oracle arithmetic
        KINDS = ["ew", "inv", "scalar", "dotsum", "shamir", "recover_at", "additive", "matmul", "layout", "detect",
                 "coeffs", "correct", "wire", "vdm", "matmul_big", "ring", "inv_big", "random", "packed", "misc"]
        kind = KINDS[rng.integers(len(KINDS))]
        by_kind[kind] = by_kind.get(kind, 0) + 1
        t_case = time.time()
        # one case in three runs with a tuning knob off its default: the alternative kernels and launch geometries behind every
        # entry point (the knobs pin paths for A/B runs; none of them may change a result)
        knob = None
        if rng.random() < 0.33 and kind != "matmul_big":
            key = list(KNOBS)[rng.integers(len(KNOBS))]
            knob = (key, int(KNOBS[key][rng.integers(len(KNOBS[key]))]))
            scl.set_tuning(*knob)
            by_knob[key] = by_knob.get(key, 0) + 1
        try:
            if kind == "ew":
                n = size(70000)
                op = [scl.ADD, scl.SUB, scl.MUL, scl.NEG][rng.integers(4)]
                a, b = rand(f, n, b"a"), rand(f, n, b"b")
                want = port.ew(f, {scl.ADD: O.ADD, scl.SUB: O.SUB, scl.MUL: O.MUL, scl.NEG: O.NEG}[op], a, None if op == scl.NEG else b)
                if n == 0:
                    continue
                da, db = dev(a), dev(b)
                inplace = rng.random() < 0.3
                got = scl.ew(f, op, da, None if op == scl.NEG else db, out=da if inplace else None)
                if not np.array_equal(scl.to_host(got), want):
                    report(kind, f"field {f} op {op} n {n} inplace {inplace}")
            elif kind == "inv":
                n = size(700 if slow else 40000)
                if n == 0:
                    continue
                a, b = rand(f, n, b"a"), rand(f, n, b"b")
                div = rng.random() < 0.5
                zeros = rng.random() < 0.3
                tgt = b if div else a
                tgt[(tgt == 0).all(axis=1)] = port.from_int(f, 1)
                want_err = False
                if zeros:
                    idx = rng.integers(0, n, size=min(n, 3))
                    tgt[idx] = 0
                    want_err = True
                if want_err:
                    keep = np.ones(n, dtype=bool)
                    keep[idx] = False
                    safe = tgt.copy()
                    safe[idx] = port.from_int(f, 1)
                    want = port.ew(f, O.DIV if div else O.INV, a if div else safe, safe if div else None)
                    want[idx] = 0
                else:
                    want = port.ew(f, O.DIV if div else O.INV, a, b if div else None)
                da, db = dev(a), dev(b)
                inplace = rng.random() < 0.3
                out = (db if div else da) if inplace else scl.empty(f, n)
                err = False
                use_status = rng.random() < 0.5      # scl_hip_ew_status: the same kernels, the report left in a device word
                if use_status:
                    status = scl.ew_status_buffer()
                    scl.ew_status(f, scl.DIV if div else scl.INV, da, db if div else None, status, out=out)
                    err = int(status.item()) != 0
                else:
                    try:
                        scl.ew(f, scl.DIV if div else scl.INV, da, db if div else None, out=out)
                    except scl.SclError as e:
                        err = "0 not invertible" in str(e)
                if err != want_err or not np.array_equal(scl.to_host(out), want):
                    report(kind, f"field {f} div {div} n {n} zeros {zeros} inplace {inplace} err {err} status-form {use_status}")
            elif kind == "scalar":
                n = size(50000)
                if n == 0:
                    continue
                a, s = rand(f, n, b"a"), rand(f, 1, b"s")
                want = port.scalar_mul(f, a, s[0])
                got = scl.scalar_mul(f, dev(a), s[0])
                if not np.array_equal(scl.to_host(got), want):
                    report(kind, f"field {f} n {n}")
            elif kind == "dotsum":
                n = size(50000)
                a, b = rand(f, n, b"a"), rand(f, n, b"b")
                if n == 0:
                    continue
                if not np.array_equal(np.asarray(scl.dot(f, dev(a), dev(b))).reshape(-1), np.asarray(port.dot(f, a, b)).reshape(-1)):
                    report(kind, f"dot field {f} n {n}")
                if not np.array_equal(np.asarray(scl.vsum(f, dev(a))).reshape(-1), np.asarray(port.sum(f, a)).reshape(-1)):
                    report(kind, f"sum field {f} n {n}")
            elif kind == "shamir":
                N = max(1, size(600 if slow else 5000))
                n = int(rng.integers(1, 70 if not slow else 24))
                t = int(rng.integers(0, n))
                secrets = rand(f, N, b"s")
                sd = b"fuzz-seed-%d" % runs
                got = scl.shamir_share_prg(f, dev(secrets), t, n, sd)
                back = scl.shamir_recover(f, got)
                if knob is None and t <= 48:
                    # the fused kernel and the two passes draw the same coefficients from the same blocks: identical shares
                    # (for GF(2^128), whose sharing the oracle cannot check, this is what ties the two forms together)
                    both = []
                    for mode in (-1, 1):
                        scl.set_tuning("prg_two_pass", mode)
                        try:
                            both.append(scl.to_host(scl.shamir_share_prg(f, dev(secrets), t, n, sd)))
                        finally:
                            scl.set_tuning("prg_two_pass", 0)
                    if not (np.array_equal(both[0], both[1]) and np.array_equal(both[0], scl.to_host(got))):
                        report(kind, f"fused and two-pass sharing differ: field {f} N {N} n {n} t {t}")
                if f == scl.GF2_128:
                    # the oracle walks the reference's x++ over the nodes, which in characteristic 2 alternates 1, 0, 1, ..: the
                    # library shares GF(2^128) at the bit patterns 1..n (tests/test_gpu_parity.py); checked by the round trip
                    if not np.array_equal(scl.to_host(back), secrets):
                        report(kind, f"round trip field {f} N {N} n {n} t {t}")
                else:
                    want = port.shamir_share(f, sd, secrets, t, n)           # [N][n][L]: one Vector of n shares per secret
                    if not np.array_equal(scl.to_host(got).transpose(1, 0, 2), np.asarray(want).reshape(N, n, L)):
                        report(kind, f"share_prg field {f} N {N} n {n} t {t}")
                    want_back = port.shamir_recover(f, np.asarray(want).reshape(N, n, L))
                    if not np.array_equal(scl.to_host(back), np.asarray(want_back).reshape(N, L)):
                        report(kind, f"recover field {f} N {N} n {n} t {t}")
            elif kind == "recover_at":
                N = max(1, size(40 if slow else 400))
                m = int(rng.integers(1, 30 if not slow else 12))
                shares = rand(f, N * m, b"sh").reshape(N, m, L)
                nodes = port.vector_random(f, b"fuzz-nodes-%d" % runs, m + 1)
                if len({tuple(r) for r in nodes.tolist()}) != m + 1:
                    continue
                x = nodes[m]
                want = port.shamir_recover_at(f, shares, nodes[:m], x)
                lam = scl.lagrange_basis(f, m, alphas=nodes[:m], x=x)
                got = scl.shamir_recover(f, dev(np.ascontiguousarray(shares.transpose(1, 0, 2))), lam)
                if not np.array_equal(scl.to_host(got), np.asarray(want).reshape(N, L)):
                    report(kind, f"field {f} N {N} m {m}")
            elif kind == "additive":
                N = max(1, size(5000))
                n = int(rng.integers(1, 40))
                secrets = rand(f, N, b"s")
                sd = b"fuzz-add-%d" % runs
                want = np.asarray(port.additive_share(f, sd, secrets, n)).reshape(N, n, L)
                got = scl.additive_share_prg(f, dev(secrets), n, sd)
                if not np.array_equal(scl.to_host(got).transpose(1, 0, 2), want):
                    report(kind, f"share field {f} N {N} n {n}")
                if not np.array_equal(scl.to_host(scl.additive_recover(f, got)), secrets):
                    report(kind, f"recover field {f} N {N} n {n}")
            elif kind == "matmul":
                cap = 24 if slow else 90
                M, K, N = int(rng.integers(1, cap)), int(rng.integers(0, cap * 3)), int(rng.integers(1, cap))
                if rng.random() < 0.2:
                    N = 1
                A, B = rand(f, M * K, b"A").reshape(M, K, L), rand(f, K * N, b"B").reshape(K, N, L)
                want = np.asarray(port.matmul(f, A, B)).reshape(M, N, L) if K else np.zeros((M, N, L), dtype=np.uint64)
                got = scl.matmul(f, dev(A) if K else scl.empty(f, M, 0), dev(B) if K else scl.empty(f, 0, N))
                if not np.array_equal(scl.to_host(got), want):
                    report(kind, f"field {f} M {M} K {K} N {N}")
            elif kind == "layout":
                N, n = max(1, size(5000)), int(rng.integers(1, 50))
                aos = rand(f, N * n, b"l").reshape(N, n, L)
                soa = scl.aos_to_soa(f, dev(aos))
                if not np.array_equal(scl.to_host(soa), aos.transpose(1, 0, 2)):
                    report(kind, f"aos_to_soa field {f} N {N} n {n}")
                if not np.array_equal(scl.to_host(scl.soa_to_aos(f, soa)), aos):
                    report(kind, f"soa_to_aos field {f} N {N} n {n}")
            elif kind == "detect":
                if f == scl.GF2_128:
                    continue               # (the oracle's sharing has no usable nodes in characteristic 2, see above)
                t = int(rng.integers(1, 9 if not slow else 5))
                n = 2 * t + 1
                N = max(1, size(300 if slow else 2000))
                secrets = rand(f, N, b"s")
                sh = np.asarray(port.shamir_share(f, b"fuzz-det-%d" % runs, secrets, t, n)).reshape(N, n, L).copy()
                hit = rng.random(N) < 0.1
                for s in np.nonzero(hit)[0]:
                    j = int(rng.integers(n))
                    sh[s, j] = port.ew(f, O.ADD, sh[s, j:j + 1], port.from_int(f, 1).reshape(1, L))[0]
                out, status = scl.shamir_recover_detect(f, dev(np.ascontiguousarray(sh.transpose(1, 0, 2))), t)[:2]
                st = status.cpu().numpy().reshape(-1).astype(bool)
                want_val, want_st = port.shamir_recover_d(f, sh, t)
                want_bad = np.asarray(want_st).astype(bool)
                if not np.array_equal(st, want_bad) or not np.array_equal(scl.to_host(out)[~want_bad], want_val[~want_bad]):
                    report(kind, f"field {f} N {N} t {t}")
            elif kind == "coeffs":
                # explicit coefficients, custom nodes, the share matrix a window of a wider one (row pitch > N)
                N = max(1, size(300 if slow else 3000))
                n = int(rng.integers(1, 50 if not slow else 14))
                t = int(rng.integers(0, min(n, 24 if not slow else 8)))
                secrets = rand(f, N, b"s")
                coeffs = rand(f, t * N, b"c").reshape(N, t, L)
                nodes = port.vector_random(f, b"fuzz-cn-%d" % runs, n)
                if len({tuple(r) for r in nodes.tolist()}) != n or (nodes == 0).all(axis=1).any():
                    continue
                want = np.stack([port.poly_eval(f, np.concatenate([secrets[s_:s_ + 1], coeffs[s_]]), nodes) for s_ in range(N)])
                dco = dev(np.ascontiguousarray(coeffs.transpose(1, 0, 2))) if t else None
                pitch = N + int(rng.integers(0, 9)) * 2
                wide = scl.empty(f, n, pitch)
                out = scl.shamir_share(f, dev(secrets), dco, n, alphas=nodes, out=None)
                if not np.array_equal(scl.to_host(out).transpose(1, 0, 2), want):
                    report(kind, f"share field {f} N {N} n {n} t {t}")
                wide[:, :N] = out
                lam = scl.lagrange_basis(f, n, alphas=nodes)
                back = scl.shamir_recover(f, wide[:, :N], lam)
                if not np.array_equal(scl.to_host(back), secrets):
                    report(kind, f"recover through a pitched window field {f} N {N} n {n} t {t} pitch {pitch}")
            elif kind == "correct":
                if f == scl.GF2_128:
                    continue
                t = int(rng.integers(1, 6 if not slow else 3))
                n = 3 * t + 1 + int(rng.integers(0, 3))
                N = max(1, size(60 if slow else 600))
                secrets = rand(f, N, b"s")
                sh = np.asarray(port.shamir_share(f, b"fuzz-bw-%d" % runs, secrets, t, n)).reshape(N, n, L).copy()
                for s_ in np.nonzero(rng.random(N) < 0.3)[0]:
                    for j in rng.choice(n, size=int(rng.integers(1, t + 2)), replace=False):   # up to t + 1 errors: some uncorrectable
                        sh[s_, j] = port.ew(f, O.ADD, sh[s_, j:j + 1], port.from_int(f, int(rng.integers(1, 99))).reshape(1, L))[0]
                fo, eo, st, ne = port.shamir_recover_c(f, sh)
                r = scl.shamir_recover_correct(f, dev(np.ascontiguousarray(sh.transpose(1, 0, 2))))
                st_g = r["status"].cpu().numpy().astype(bool)
                ok = np.array_equal(st_g, np.asarray(st).astype(bool))
                good = ~np.asarray(st).astype(bool)
                ok = ok and np.array_equal(scl.to_host(r["f"]).transpose(1, 0, 2)[good], np.asarray(fo)[good])
                ok = ok and np.array_equal(scl.to_host(r["err"]).transpose(1, 0, 2)[good], np.asarray(eo)[good])
                ok = ok and np.array_equal(r["nerr"].cpu().numpy()[good], np.asarray(ne)[good].astype(np.int32))
                if not ok:
                    report(kind, f"field {f} N {N} n {n} t {t}")
            elif kind == "wire":
                n = size(5000)
                a = rand(f, n, b"w")
                img = port.wire_vector(f, a)
                got = scl.wire_pack(f, dev(a) if n else scl.empty(f, 0))
                if bytes(got.cpu().numpy().tobytes()) != img:
                    report(kind, f"wire_pack field {f} n {n}")
                back = scl.wire_unpack(f, got)
                if n and not np.array_equal(scl.to_host(back), a):
                    report(kind, f"wire_unpack field {f} n {n}")
                fr = port.frame(f, a)
                gfr = scl.frame_pack(f, dev(a) if n else scl.empty(f, 0))
                if bytes(gfr.cpu().numpy().tobytes()) != fr:
                    report(kind, f"frame_pack field {f} n {n}")
                rows, cols = int(rng.integers(1, 40)), int(rng.integers(1, 40))
                mat = rand(f, rows * cols, b"wm").reshape(rows, cols, L)
                if bytes(scl.wire_pack_matrix(f, dev(mat)).cpu().numpy().tobytes()) != port.wire_matrix(f, mat):
                    report(kind, f"wire_pack_matrix field {f} {rows}x{cols}")
            elif kind == "vdm":
                n, m = int(rng.integers(1, 60 if not slow else 16)), int(rng.integers(1, 40 if not slow else 12))
                xs = rand(f, n, b"x") if rng.random() < 0.5 else None
                if f == scl.GF2_128 and xs is None:
                    continue
                want = port.vandermonde(f, n, m, xs)
                got = scl.vandermonde(f, n, m, xs)
                if not np.array_equal(scl.to_host(got), want):
                    report(kind, f"field {f} n {n} m {m} xs {xs is not None}")
            elif kind == "matmul_big":
                # Mersenne61 through every matmul path: the matrix cores (one tile, row blocks, k chunks, the general kernel) and the
                # vector-ALU kernels, picked by the shape or pinned by the tuning knob
                M, K, N = int(rng.integers(1, 200)), int(rng.integers(1, 1200)), int(rng.integers(1, 300))
                if rng.random() < 0.3:
                    N = int(rng.integers(4096, 9000))
                    M, K = int(rng.integers(1, 130)), int(rng.integers(1, 70))
                A, B = rand(scl.M61, M * K, b"A").reshape(M, K, 1), rand(scl.M61, K * N, b"B").reshape(K, N, 1)
                want = np.asarray(port.matmul(scl.M61, A, B)).reshape(M, N, 1)
                mode = int(rng.integers(-1, 3))
                scl.set_tuning("mfma", mode)
                try:
                    got = scl.matmul(scl.M61, dev(A), dev(B))
                finally:
                    scl.set_tuning("mfma", 0)
                if not np.array_equal(scl.to_host(got), want):
                    report(kind, f"M {M} K {K} N {N} mode {mode}")
            elif kind == "ring":
                # Z2k<K>: every width; additive sharing and the matrix product are what the reference runs over rings
                bits = int(rng.integers(1, 129))
                fr_ = scl.Z2K(bits)
                Lr, bs = scl.limbs(fr_), O.byte_size(fr_)
                n = max(1, size(20000))
                raw = port.prg(b"fuzz-ring-%d" % runs, [2 * n * bs])
                a, b = port.from_bytes(fr_, raw[: n * bs]), port.from_bytes(fr_, raw[n * bs:])
                op = [scl.ADD, scl.SUB, scl.MUL, scl.NEG][rng.integers(4)]
                want = port.ew(fr_, {scl.ADD: O.ADD, scl.SUB: O.SUB, scl.MUL: O.MUL, scl.NEG: O.NEG}[op], a, None if op == scl.NEG else b)
                got = scl.ew(fr_, op, dev(a), None if op == scl.NEG else dev(b))
                if not np.array_equal(scl.to_host(got), want):
                    report(kind, f"Z2k<{bits}> op {op} n {n}")
                odd = a.copy()
                odd[:, 0] |= np.uint64(1)
                if not np.array_equal(scl.to_host(scl.ew(fr_, scl.DIV, dev(b), dev(odd))), port.ew(fr_, O.DIV, b, odd)):
                    report(kind, f"Z2k<{bits}> divide n {n}")
                if not np.array_equal(np.asarray(scl.dot(fr_, dev(a), dev(b))).reshape(-1), np.asarray(port.dot(fr_, a, b)).reshape(-1)):
                    report(kind, f"Z2k<{bits}> dot n {n}")
                nn = int(rng.integers(1, 20))
                sec = a[: min(n, 700)]
                sd = b"fuzz-radd-%d" % runs
                want = np.asarray(port.additive_share(fr_, sd, sec, nn)).reshape(sec.shape[0], nn, Lr)
                got = scl.additive_share_prg(fr_, dev(sec), nn, sd)
                if not np.array_equal(scl.to_host(got).transpose(1, 0, 2), want):
                    report(kind, f"Z2k<{bits}> additive_share_prg N {sec.shape[0]} n {nn}")
                M_, K_, N_ = int(rng.integers(1, 30)), int(rng.integers(1, 60)), int(rng.integers(1, 80))
                rawm = port.prg(b"fuzz-rm-%d" % runs, [(M_ * K_ + K_ * N_) * bs])
                A_ = port.from_bytes(fr_, rawm[: M_ * K_ * bs]).reshape(M_, K_, Lr)
                B_ = port.from_bytes(fr_, rawm[M_ * K_ * bs:]).reshape(K_, N_, Lr)
                if not np.array_equal(scl.to_host(scl.matmul(fr_, dev(A_), dev(B_))), np.asarray(port.matmul(fr_, A_, B_)).reshape(M_, N_, Lr)):
                    report(kind, f"Z2k<{bits}> matmul {M_}x{K_}x{N_}")
            elif kind == "inv_big":
                if rng.random() < 0.9:
                    continue               # (a few seconds of oracle each: one in ten)
                # sizes at which the chained inversion picks its longer chains (n >= L * 64 * 4096); the fast-oracle fields only
                fb = [scl.M61, scl.M127][rng.integers(2)]
                n = int(rng.integers(1_000_000, 9_000_000))
                a = port.vector_random(fb, b"fuzz-ib-%d" % runs, n)
                a[(a == 0).all(axis=1)] = port.from_int(fb, 1)
                want = port.ew(fb, O.INV, a)
                got = scl.ew(fb, scl.INV, dev(a))
                if not np.array_equal(scl.to_host(got), want):
                    report(kind, f"field {fb} n {n}")
            elif kind == "random":
                # the PRG discipline: blocks at a counter, FF::read of raw bytes, Vector::random (seeds of any length: the reference
                # zero-pads or truncates to 16 bytes)
                sd = bytes(rng.integers(0, 256, size=int(rng.integers(0, 40)), dtype=np.uint8))
                c0 = int(rng.integers(0, 2 ** 40)) if rng.random() < 0.5 else int(2 ** 32 - rng.integers(0, 50))
                nb = int(rng.integers(0, 5000))
                want = port.prg_blocks(sd, c0, nb)
                got = scl.prg_blocks(nb, sd, c0)
                if bytes(got.cpu().numpy().tobytes()) != want:
                    report(kind, f"prg_blocks seed {len(sd)} bytes counter {c0} blocks {nb}")
                n = size(20000)
                bs = 8 * L
                raw = port.prg(b"fuzz-raw-%d" % runs, [max(1, n) * bs])[: n * bs]
                if n:
                    rt = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
                    if not np.array_equal(scl.to_host(scl.from_bytes(f, rt)), port.from_bytes(f, raw)):
                        report(kind, f"from_bytes field {f} n {n}")
                    if not np.array_equal(scl.to_host(scl.vector_random(f, n, sd)), port.vector_random(f, sd, n)):
                        report(kind, f"vector_random field {f} n {n} seed {len(sd)} bytes")
            elif kind == "packed":
                # shamirSecretShare over Array<FF, W> (what Pedersen shares): W interleaved sharings on one PRG draw
                if f == scl.GF2_128:
                    continue
                W = int(rng.integers(1, 5))
                N = max(1, size(200 if slow else 1500))
                n = int(rng.integers(1, 20 if not slow else 9))
                t = int(rng.integers(0, n))
                sec = rand(f, N * W, b"p").reshape(N, W, L)
                sd = b"fuzz-packed-%d" % runs
                want = np.asarray(port.shamir_share_packed(f, sd, sec, t, n))                  # [N][n][W][L]
                got = scl.shamir_share_prg_packed(f, dev(np.ascontiguousarray(sec.transpose(1, 0, 2))), t, n, sd)   # [W][n][N][L]
                if not np.array_equal(scl.to_host(got).transpose(2, 1, 0, 3), want):
                    report(kind, f"field {f} N {N} n {n} t {t} W {W}")
            elif kind == "misc":
                n = max(1, size(30000))
                a = rand(f, n, b"a")
                b = a.copy()
                da = dev(a)
                if not scl.equals(f, da, dev(b)):
                    report(kind, f"equals(a, a) field {f} n {n}")
                j = int(rng.integers(n))
                b[j, int(rng.integers(L))] ^= np.uint64(1) << np.uint64(rng.integers(0, 60))
                if scl.equals(f, da, dev(b)):
                    report(kind, f"equals misses a flipped bit: field {f} n {n} at {j}")
                m = int(rng.integers(1, 30 if not slow else 10))
                nodes = port.vector_random(f, b"fuzz-lb-%d" % runs, m + 1)
                if len({tuple(r) for r in nodes.tolist()}) == m + 1:
                    if not np.array_equal(scl.lagrange_basis(f, m, alphas=nodes[:m], x=nodes[m]), port.lagrange_basis(f, nodes[:m], nodes[m])):
                        report(kind, f"lagrange_basis field {f} m {m}")
                nn = int(rng.integers(2, 12))
                N = min(n, 2000)
                rnd = rand(f, (nn - 1) * N, b"r").reshape(nn - 1, N, L)
                sh = scl.additive_share(f, dev(a[:N]), dev(rnd), nn)
                hs = scl.to_host(sh)
                if not np.array_equal(hs[: nn - 1], rnd) or not np.array_equal(scl.to_host(scl.additive_recover(f, sh)), a[:N]):
                    report(kind, f"additive_share with explicit randomness field {f} N {N} n {nn}")
                rows, cols = int(rng.integers(1, 30)), int(rng.integers(1, 30))
                mat = rand(f, rows * cols, b"m").reshape(rows, cols, L)
                if not np.array_equal(scl.to_host(scl.wire_unpack_matrix(f, scl.wire_pack_matrix(f, dev(mat)))), mat):
                    report(kind, f"matrix wire round trip field {f} {rows}x{cols}")
                if not np.array_equal(scl.to_host(scl.frame_unpack(f, scl.frame_pack(f, da))), a):
                    report(kind, f"frame round trip field {f} n {n}")
        except Exception as e:  # an exception the oracle did not raise too is a finding
            report(kind, f"field {f}: {type(e).__name__}: {e}")
        finally:
            if knob is not None:
                scl.set_tuning(knob[0], KNOB_DEFAULTS[knob[0]])
        secs[kind] = secs.get(kind, 0.0) + time.time() - t_case
    torch.cuda.synchronize()
    sys.path.insert(0, ROOT)
    from bench_legs.pmc import kernel_source_hash     # one campaign per change of the kernel sources: the line says which
    print(f"fuzz_abi: kernel sources {kernel_source_hash()}, seed {seed0}")
    print(f"fuzz_abi: {runs} cases in {budget:g} s, {bad} mismatches; by kind {by_kind}; seconds by kind { {k: round(v, 1) for k, v in secs.items()} }; cases by knob {by_knob}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
