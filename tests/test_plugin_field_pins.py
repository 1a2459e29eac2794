"""Independent pins for the two plug-in fields the reference does not have ("parity unpinned" ->
pinned here against arithmetic that shares no code with the oracle or the kernels):

* Mont128 -- the N = 2 instance of the reference's Montgomery mpn family (include/scl/math/fields/ff_ops_gmp.h:44-260):
  every operation against Python big integers (x * y * R^-1 mod p with R = 2^128, pow(x, -1, p), ...), for the default
  prime 2^128 - 159, the degenerate p = 7 (the reference's GF(7) test field, test/scl/gf7.h), the Mersenne prime 2^127 - 1
  and two random 128-bit primes found by a Miller-Rabin search in this file.
* GF(2^128) -- multiplication and inversion against a bit-serial shift-xor multiplier with the reduction x^128 = x^7+x^2+x+1
  written here in Python, AND against published known answers that nobody here wrote: GCM's GHASH is multiplication in
  exactly this field (same polynomial; GCM writes the coefficient of x^0 in the most significant bit of a block, this field
  in the least significant bit of the integer), so the intermediate values X_i and the GHASH outputs of test cases 2 and 3 of
  the GCM specification (McGrew & Viega, "The Galois/Counter Mode of Operation", appendix B; the vectors NIST SP 800-38D
  validation uses) are products the oracle and the kernels must reproduce after reversing the bits of each block.
  Mont128's reference pin is tests/golden/golden_mont128.json (the reference's own Montgomery templates at two limbs).

The CPU half checks the oracle; the `gpu` half sends the same vectors through the HIP kernels (scl_hip_ew), so the GPU path
is pinned against the big-integer model directly, not only against the oracle."""
import random

import numpy as np
import pytest

import oracle_lib as O

R = 1 << 128
MASK = R - 1


def _is_prime(n: int) -> bool:
    if n < 2:
        return False
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):   # deterministic far beyond 2^128 for these bases? no: also
        x = pow(a, d, n)                                       # 40 random bases below make an error < 4^-40
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    rng = random.Random(n)
    for _ in range(40):
        a = rng.randrange(2, n - 1)
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _random_prime_128(seed: int) -> int:
    rng = random.Random(seed)
    while True:
        c = rng.getrandbits(128) | (1 << 127) | 1
        if _is_prime(c):
            return c


PRIMES = [R - 159, 7, (1 << 127) - 1, _random_prime_128(1), _random_prime_128(2)]


def test_the_moduli_are_what_they_claim():
    assert all(_is_prime(p) for p in PRIMES)
    assert PRIMES[3] != PRIMES[4] and PRIMES[3].bit_length() == 128 and PRIMES[4].bit_length() == 128


def mont_vectors(p: int, n: int = 200):
    """operand pairs as integers in [0, p): edge values first, then uniformly random ones"""
    rng = random.Random(p ^ 0x5EED)
    xs = [0, 1, p - 1, 2 % p, (p - 1) // 2, (R % p), (p - 2) % p]
    ys = [1, p - 1, 0, (p - 1) // 2, 2 % p, (R * R) % p, 3 % p]
    while len(xs) < n:
        xs.append(rng.randrange(p))
        ys.append(rng.randrange(p))
    return xs, ys


def to_mont(v, p):
    return [(x * R) % p for x in v]


def mont_expected(p, xs, ys):
    """what each element-wise op must return, in Montgomery form (the in-memory image, like FF::m_value)"""
    inv = [pow(x, -1, p) if x % p else None for x in xs]
    return {
        O.ADD: to_mont([(x + y) % p for x, y in zip(xs, ys)], p),
        O.SUB: to_mont([(x - y) % p for x, y in zip(xs, ys)], p),
        O.MUL: to_mont([(x * y) % p for x, y in zip(xs, ys)], p),
        O.NEG: to_mont([(-x) % p for x in xs], p),
    }, inv


@pytest.fixture(scope="module")
def port():
    return O.Port()


@pytest.mark.parametrize("p", PRIMES, ids=lambda p: f"p={p:#x}")
def test_oracle_mont128_against_python_big_integers(port, p):
    f = O.MONT128
    port.mont128_set_prime(p)
    try:
        assert port.mont128_get_prime() == p
        xs, ys = mont_vectors(p)
        a, b = O.from_ints(to_mont(xs, p), 2), O.from_ints(to_mont(ys, p), 2)
        want, inv = mont_expected(p, xs, ys)
        for op in (O.ADD, O.SUB, O.MUL, O.NEG):
            got = O.to_ints(port.ew(f, op, a, b if op != O.NEG else None))
            assert got == want[op], f"op {op}"
        # the raw Montgomery product of the images: a_m * b_m * R^-1 mod p
        rinv = pow(R, -1, p)
        assert O.to_ints(port.ew(f, O.MUL, a, b)) == [(am * bm * rinv) % p for am, bm in zip(to_mont(xs, p), to_mont(ys, p))]
        nz = [i for i, x in enumerate(xs) if x % p]
        got = O.to_ints(port.ew(f, O.INV, a[nz]))
        assert got == [(inv[i] * R) % p for i in nz]
        got = O.to_ints(port.ew(f, O.DIV, b[nz], a[nz]))
        assert got == [(ys[i] * inv[i] * R) % p for i in nz]
        with pytest.raises(O.OracleError) as e:
            port.ew(f, O.INV, O.from_ints([0], 2))
        assert e.value.message == "0 not invertible modulo prime"
        # montyInFromInt (ff_ops_gmp.h:108-114): |v|, and for a negative v the N-limb subtraction prime - |v| (mpn_sub_n: it
        # wraps modulo 2^128 when |v| > p, which only the toy modulus 7 can show), then montyIn's reduction modulo p
        for v in (0, 1, 2, 5, 6, 123, 2 ** 31 - 1, -1, -5, -123, -(2 ** 31) + 1):
            x = abs(v) if v >= 0 else (p - abs(v)) % R
            assert O.to_ints(port.from_int(f, v)) == [((x % p) * R) % p], v
        # montyFromBytes (ff_ops_gmp.h:279-290): 16 bytes, big-endian, reduced modulo p
        rng = random.Random(99)
        raw = [bytes([0] * 16), bytes([255] * 16), (p - 1).to_bytes(16, "big"), p.to_bytes(16, "big")] + [
            rng.getrandbits(128).to_bytes(16, "big") for _ in range(50)]
        got = O.to_ints(port.from_bytes(f, b"".join(raw)))
        assert got == [((int.from_bytes(r, "big") % p) * R) % p for r in raw]
    finally:
        port.mont128_set_prime(R - 159)


# ---- GF(2^128) --------------------------------------------------------------------------------------------------
def gf_mul(a: int, b: int) -> int:
    """bit-serial: for each set bit of b add a * x^i, a * x reduced by x^128 = x^7 + x^2 + x + 1 (0x87)"""
    r = 0
    while b:
        if b & 1:
            r ^= a
        b >>= 1
        a <<= 1
        if a >> 128:
            a = (a & MASK) ^ 0x87
    return r


def gf_inv(a: int) -> int:
    """a^(2^128 - 2) by square-and-multiply on the bit-serial multiplier"""
    r, e, base = 1, (1 << 128) - 2, a
    while e:
        if e & 1:
            r = gf_mul(r, base)
        base = gf_mul(base, base)
        e >>= 1
    return r


def gf_vectors(n: int = 120):
    rng = random.Random(0x6F2128)
    xs = [0, 1, 2, MASK, 1 << 127, 0x87, (1 << 127) | 1, 3]
    ys = [MASK, MASK, 1 << 127, MASK, 1 << 127, 1 << 121, 2, 0]
    while len(xs) < n:
        xs.append(rng.getrandbits(128))
        ys.append(rng.getrandbits(128))
    return xs, ys


def test_python_gf_model_sanity():
    assert gf_mul(1 << 127, 2) == 0x87 and gf_mul(3, 3) == 5
    for x in (1, 2, 0x87, MASK, 1 << 127):
        assert gf_mul(x, gf_inv(x)) == 1


def test_oracle_gf2_128_against_python_shift_xor(port):
    f = O.GF2_128
    xs, ys = gf_vectors()
    a, b = O.from_ints(xs, 2), O.from_ints(ys, 2)
    assert O.to_ints(port.ew(f, O.MUL, a, b)) == [gf_mul(x, y) for x, y in zip(xs, ys)]
    assert O.to_ints(port.ew(f, O.ADD, a, b)) == [x ^ y for x, y in zip(xs, ys)]
    assert O.to_ints(port.ew(f, O.SUB, a, b)) == [x ^ y for x, y in zip(xs, ys)]
    assert O.to_ints(port.ew(f, O.NEG, a)) == xs
    nz = [i for i, x in enumerate(xs) if x][:40]
    assert O.to_ints(port.ew(f, O.INV, a[nz])) == [gf_inv(xs[i]) for i in nz]
    assert O.to_ints(port.ew(f, O.DIV, b[nz], a[nz])) == [gf_mul(ys[i], gf_inv(xs[i])) for i in nz]
    with pytest.raises(O.OracleError) as e:
        port.ew(f, O.INV, O.from_ints([0], 2))
    assert e.value.message == "0 not invertible modulo prime"


# ---- GF(2^128) against published vectors: GHASH of the GCM specification ---------------------------------------------------------
# A GCM block B (16 bytes, big-endian integer here) stands for sum_i b_i x^i with b_0 the MOST significant bit; an element of
# this field is the integer with b_i at bit i.  Same field, same polynomial (GCM's R = 11100001 || 0^120 is 1 + x + x^2 + x^7):
# element = the block's 128 bits reversed.
def _refl(v: int) -> int:
    return int(format(v, "0128b")[::-1], 2)


GCM_CASES = [
    # test case 2: K = 0^128, P = 0^128, IV = 0^96
    {"H": 0x66e94bd4ef8a2c3b884cfa59ca342b2e, "blocks": [0x0388dace60b6a392f328c2b971b2fe78, 0x80],
     "X": [0x5e2ec746917062882c85b0685353deb7, 0xf38cbb1ad69223dcc3457ae5b6b0f885]},
    # test case 3: K = feffe9928665731c6d6a8f9467308308, 64 bytes of plaintext: four ciphertext blocks, then len(A) || len(C)
    {"H": 0xb83b533708bf535d0aa6e52980d53b78,
     "blocks": [0x42831ec2217774244b7221b784d0d49c, 0xe3aa212f2c02a4e035c17e2329aca12e, 0x21d514b25466931c7d8f6a5aac84aa05,
                0x1ba30b396a0aac973d58e091473f5985, 0x200],
     "X": [0x59ed3f2bb1a0aaa07c9f56c6a504647b, 0xb714c9048389afd9f9bc5c1d4378e052, 0x47400c6577b1ee8d8f40b2721e86ff10,
           0x4796cf49464704b5dd91f159bb1b7f95, 0x7f1b32b81b820d02614f8895ac1d4eac]},
]


def _gcm_products():
    """(a, b, a * b) as elements of this field, one per GHASH step X_i = (X_{i-1} xor C_i) * H of the published vectors"""
    out = []
    for c in GCM_CASES:
        prev = 0
        for blk, x in zip(c["blocks"], c["X"]):
            out.append((_refl(prev ^ blk), _refl(c["H"]), _refl(x)))
            prev = x
    return out


def test_the_gcm_vectors_are_the_published_ones():
    """The constants above, checked two ways that share nothing with the engine: (1) the multiplication of the GCM
    specification as it is written there (Algorithm 1: right shifts, R = e1 || 0^120) reproduces every X_i from H and the
    blocks; (2) where the image has openssl, H is AES_K(0^128) for the keys of the two test cases."""
    R_ = 0xe1 << 120

    def spec_mul(X, Y):
        Z, V = 0, X
        for i in range(128):
            if (Y >> (127 - i)) & 1:
                Z ^= V
            V = (V >> 1) ^ R_ if V & 1 else V >> 1
        return Z
    for c in GCM_CASES:
        x = 0
        for blk, want in zip(c["blocks"], c["X"]):
            x = spec_mul(x ^ blk, c["H"])
            assert x == want
    import shutil
    import subprocess
    if shutil.which("openssl"):
        for key, c in (("00" * 16, GCM_CASES[0]), ("feffe9928665731c6d6a8f9467308308", GCM_CASES[1])):
            r = subprocess.run(["openssl", "enc", "-aes-128-ecb", "-K", key, "-nopad"], input=bytes(16), capture_output=True)
            if r.returncode == 0:
                assert int.from_bytes(r.stdout, "big") == c["H"]
    # and the bit reversal is the right dictionary: the shift-xor model gives the same products
    for a, b, ab in _gcm_products():
        assert gf_mul(a, b) == ab


def test_oracle_gf2_128_against_published_gcm_vectors(port):
    f = O.GF2_128
    prods = _gcm_products()
    a, b = O.from_ints([p[0] for p in prods], 2), O.from_ints([p[1] for p in prods], 2)
    assert O.to_ints(port.ew(f, O.MUL, a, b)) == [p[2] for p in prods]
    assert O.to_ints(port.ew(f, O.MUL, b, a)) == [p[2] for p in prods]
    # division undoes it: X_i / H = X_{i-1} xor C_i  (H is invertible: the inverse enters through the same vectors)
    ab = O.from_ints([p[2] for p in prods], 2)
    assert O.to_ints(port.ew(f, O.DIV, ab, b)) == [p[0] for p in prods]


def test_oracle_gf2_128_shamir_against_python(port):
    """Horner evaluation, the Lagrange basis and reconstruction at explicit nodes (the bit patterns of 1..n: the reference's
    x++ walk, which the oracle's shamirSecretShare restates literally, cycles 1, 0, 1, .. in characteristic 2) against the
    Python model"""
    f = O.GF2_128
    n, t = 7, 3
    xs, ys = gf_vectors(40)
    secrets, coeffs = xs[:6], [ys[k * 6: k * 6 + 6] for k in range(t)]
    nodes = list(range(1, n + 1))
    shares = []
    for s in range(6):
        c = [secrets[s]] + [coeffs[k][s] for k in range(t)]
        got = O.to_ints(port.poly_eval(f, O.from_ints(c, 2), O.from_ints(nodes, 2)))
        want = []
        for x in nodes:
            y = 0
            for ck in reversed(c):
                y = gf_mul(y, x) ^ ck
            want.append(y)
        assert got == want
        shares.append(want)
    lam = []
    for i in nodes:
        num, den = 1, 1
        for j in nodes:
            if j != i:
                num, den = gf_mul(num, j), gf_mul(den, i ^ j)
        lam.append(gf_mul(num, gf_inv(den)))
    assert O.to_ints(port.lagrange_basis(f, O.from_ints(nodes, 2), O.from_ints([0], 2)[0])) == lam
    sh = np.stack([O.from_ints(row, 2) for row in shares])                       # [N][n][2]
    assert O.to_ints(port.shamir_recover_at(f, sh, O.from_ints(nodes, 2), O.from_ints([0], 2)[0])) == secrets
    # from the first t + 1 shares alone, evaluated at the node of a later party: that party's share
    sub = O.from_ints(nodes[: t + 1], 2)
    got = O.to_ints(port.shamir_recover_at(f, np.ascontiguousarray(sh[:, : t + 1]), sub, O.from_ints([n], 2)[0]))
    assert got == [row[n - 1] for row in shares]


# ---- the same vectors through the HIP kernels ------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def scl():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import scl_amd
    return scl_amd


@pytest.mark.gpu
@pytest.mark.parametrize("p", PRIMES, ids=lambda p: f"p={p:#x}")
def test_gpu_mont128_against_python_big_integers(scl, p):
    f = scl.MONT128
    scl.set_mont128_prime(p)
    try:
        xs, ys = mont_vectors(p, 1000)
        a, b = scl.to_device(O.from_ints(to_mont(xs, p), 2)), scl.to_device(O.from_ints(to_mont(ys, p), 2))
        want, inv = mont_expected(p, xs, ys)
        for op in (scl.ADD, scl.SUB, scl.MUL, scl.NEG):
            got = O.to_ints(scl.to_host(scl.ew(f, op, a, b if op != scl.NEG else None)))
            assert got == want[op], f"op {op}"
        nz = [i for i, x in enumerate(xs) if x % p]
        an = scl.to_device(O.from_ints(to_mont([xs[i] for i in nz], p), 2))
        bn = scl.to_device(O.from_ints(to_mont([ys[i] for i in nz], p), 2))
        assert O.to_ints(scl.to_host(scl.ew(f, scl.INV, an))) == [(inv[i] * R) % p for i in nz]
        assert O.to_ints(scl.to_host(scl.ew(f, scl.DIV, bn, an))) == [(ys[i] * inv[i] * R) % p for i in nz]
        with pytest.raises(scl.SclError) as e:
            scl.ew(f, scl.INV, a)                 # xs[0] = 0
        assert e.value.reference_message == "0 not invertible modulo prime"
        # dot / sum / scalar multiply in the big-integer model
        assert O.to_ints(scl.dot(f, a, b)[None]) == [(sum(x * y for x, y in zip(xs, ys)) % p) * R % p]
        assert O.to_ints(scl.vsum(f, a)[None]) == [(sum(xs) % p) * R % p]
        k = (R - 12345) % p
        assert O.to_ints(scl.to_host(scl.scalar_mul(f, a, O.from_ints([k * R % p], 2)[0]))) == to_mont([k * x % p for x in xs], p)
        if p > 64:
            # share + reconstruct: Horner at the nodes 1..n in the model (values, then Montgomery images)
            import torch
            n, t = 10, 3
            cs = [ys[1 + k * 50: 1 + k * 50 + 40] for k in range(t)]
            sec = xs[:40]
            shares = scl.shamir_share(f, scl.to_device(O.from_ints(to_mont(sec, p), 2)),
                                      torch.stack([scl.to_device(O.from_ints(to_mont(c, p), 2)) for c in cs]), n)
            for i in range(n):
                x = i + 1
                want_row = [(s + sum(c[j] * pow(x, k + 1, p) for k, c in enumerate(cs))) % p for j, s in enumerate(sec)]
                assert O.to_ints(scl.to_host(shares[i])) == to_mont(want_row, p), f"party {i}"
            assert O.to_ints(scl.to_host(scl.shamir_recover(f, shares))) == to_mont(sec, p)
            # the small-node kernels (one Barrett step per share for a full-width modulus; the blocked form at t = 13):
            # extreme coefficients -- every residue p - 1, 0, 1 or a top-heavy value -- so that the lazy sums reach their
            # bound p * sum(v) and the quotient estimate its corner cases
            edge = [p - 1, 0, 1, p - 2, (p >> 1) + 1, (1 << 127) % p, (p - 1) ^ 0xFFFFFFFF, p - (1 << 96) if p > (1 << 96) else 3]
            for n2, t2 in ((10, 7), (10, 3), (40, 13), (64, 16), (6, 5)):
                N2 = 64
                sec2 = [edge[j % len(edge)] % p for j in range(N2)]
                cs2 = [[edge[(j * 3 + k) % len(edge)] % p if (j // 8) % 2 else p - 1 for j in range(N2)] for k in range(t2)]
                sh2 = scl.shamir_share(f, scl.to_device(O.from_ints(to_mont(sec2, p), 2)),
                                       torch.stack([scl.to_device(O.from_ints(to_mont(c, p), 2)) for c in cs2]), n2)
                for i in range(n2):
                    x = i + 1
                    want_row = [(s0 + sum(c[j] * pow(x, k + 1, p) for k, c in enumerate(cs2))) % p for j, s0 in enumerate(sec2)]
                    assert O.to_ints(scl.to_host(sh2[i])) == to_mont(want_row, p), f"({n2},{t2}) party {i}"
    finally:
        scl.set_mont128_prime(R - 159)


@pytest.mark.gpu
def test_gpu_gf2_128_against_published_gcm_vectors(scl):
    """the GHASH steps of the GCM specification's test cases 2 and 3 through the kernels: the LDS-window product
    (k_ew_gf128_mul), the register product (inv_batch = -1), scalar multiplication by H, the dot product of the step
    operands with H, and division by H"""
    f = scl.GF2_128
    prods = _gcm_products()
    a = scl.to_device(O.from_ints([p[0] for p in prods], 2))
    b = scl.to_device(O.from_ints([p[1] for p in prods], 2))
    want = [p[2] for p in prods]
    assert O.to_ints(scl.to_host(scl.ew(f, scl.MUL, a, b))) == want
    scl.set_tuning("inv_batch", -1)
    try:
        assert O.to_ints(scl.to_host(scl.ew(f, scl.MUL, a, b))) == want
    finally:
        scl.set_tuning("inv_batch", 0)
    assert O.to_ints(scl.to_host(scl.ew(f, scl.DIV, scl.to_device(O.from_ints(want, 2)), b))) == [p[0] for p in prods]
    for c in GCM_CASES:
        h = O.from_ints([_refl(c["H"])], 2)[0]
        steps = [p for p in prods if p[1] == _refl(c["H"])]
        xs = scl.to_device(O.from_ints([p[0] for p in steps], 2))
        assert O.to_ints(scl.to_host(scl.scalar_mul(f, xs, h))) == [p[2] for p in steps]
        acc = 0
        for p in steps:
            acc ^= p[2]
        hs = scl.to_device(O.from_ints([_refl(c["H"])] * len(steps), 2))
        assert O.to_ints(scl.dot(f, xs, hs)[None]) == [acc]


@pytest.mark.gpu
def test_gpu_gf2_128_against_python_shift_xor(scl):
    f = scl.GF2_128
    xs, ys = gf_vectors(400)
    a, b = scl.to_device(O.from_ints(xs, 2)), scl.to_device(O.from_ints(ys, 2))
    assert O.to_ints(scl.to_host(scl.ew(f, scl.MUL, a, b))) == [gf_mul(x, y) for x, y in zip(xs, ys)]
    assert O.to_ints(scl.to_host(scl.ew(f, scl.ADD, a, b))) == [x ^ y for x, y in zip(xs, ys)]
    nz = [i for i, x in enumerate(xs) if x][:60]
    an = scl.to_device(O.from_ints([xs[i] for i in nz], 2))
    assert O.to_ints(scl.to_host(scl.ew(f, scl.INV, an))) == [gf_inv(xs[i]) for i in nz]
    # reconstruct (nibble-table kernels) and share (shift-xor Horner) in the Python model, C4's shape
    import torch
    n, t, N = 40, 13, 64
    sec = xs[:N]
    cs = [[gf_mul(y, k + 2) ^ k for y in ys[:N]] for k in range(t)]
    shares = scl.shamir_share(f, scl.to_device(O.from_ints(sec, 2)), torch.stack([scl.to_device(O.from_ints(c, 2)) for c in cs]), n)
    for i in (0, 1, 2, 7, 31, 39):
        x, want_row = i + 1, []
        for j in range(N):
            y = cs[t - 1][j]
            for k in range(t - 2, -1, -1):
                y = gf_mul(y, x) ^ cs[k][j]
            want_row.append(gf_mul(y, x) ^ sec[j])
        assert O.to_ints(scl.to_host(shares[i])) == want_row, f"party {i}"
    assert O.to_ints(scl.to_host(scl.shamir_recover(f, shares))) == sec
