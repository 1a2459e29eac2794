"""GPU parity: every entry point of the C ABI (through scl_amd) against the CPU oracle on the same
seeded inputs, against the committed golden vectors, and -- at BASELINE sizes -- through
size-independent properties.  Bit-exact everywhere (integer arithmetic)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
with open(os.path.join(HERE, "golden", "golden_v1.json")) as fh:
    GOLD = json.load(fh)
with open(os.path.join(HERE, "golden", "golden_secp256k1_field.json")) as fh:   # FF<Secp256k1Field>, same generator
    GOLD["fields"].update(json.load(fh)["fields"])

GOLDEN_NAME = {O.M61: "Mersenne61", O.M127: "Mersenne127", O.MONT128: "Mont128", O.SECP256K1_SCALAR: "secp256k1_order",
               O.SECP256K1_FIELD: "secp256k1_field"}     # the reference-emitted fixture of a field (Mont128: its default modulus)
# the two-limb instance of the reference's Montgomery family (field tag 2; golden_mont128.json, emitted by the reference's
# own templates compiled at N = 2): BASELINE configs[2]'s field, one fixture per modulus
with open(os.path.join(HERE, "golden", "golden_mont128.json")) as fh:
    MONT = json.load(fh)["fields"]
GOLD["fields"].update(MONT)
FIELDS = [(O.M61, "Mersenne61"), (O.M127, "Mersenne127"), (O.SECP256K1_SCALAR, "secp256k1_order"),
          (O.SECP256K1_FIELD, "secp256k1_field")] + [(O.MONT128, name) for name in sorted(MONT)]
ALL_FIELDS = [O.M61, O.M127, O.MONT128, O.GF2_128, O.SECP256K1_SCALAR, O.SECP256K1_FIELD]
SLOW_ORACLE = (O.MONT128, O.GF2_128, O.SECP256K1_SCALAR, O.SECP256K1_FIELD)  # bit-serial / Fermat inversions in the C oracle


@pytest.fixture(scope="module")
def scl():
    if not torch.cuda.is_available():
        pytest.fail("gpu tests need a GPU")
    import scl_amd
    return scl_amd


@pytest.fixture(scope="module")
def port():
    return O.Port()


@pytest.fixture(autouse=True)
def _mont128_modulus(request):
    """a Mont128 golden fixture is taken with ITS modulus: set on the library and on the oracle, 2^128 - 159 again after"""
    name = getattr(request.node, "callspec", None) and request.node.callspec.params.get("name")
    if not (isinstance(name, str) and name in MONT):
        yield
        return
    scl_, port_ = request.getfixturevalue("scl"), request.getfixturevalue("port")
    p = int(MONT[name]["prime"], 16)
    scl_.set_mont128_prime(p)
    port_.mont128_set_prime(p)
    try:
        yield
    finally:
        scl_.set_mont128_prime((1 << 128) - 159)
        port_.mont128_set_prime((1 << 128) - 159)


def rand_elems(port, f, n, seed):
    """uniform canonical elements via the oracle's own FF::read over PRG bytes"""
    return port.vector_random(f, seed, n)


def dev(scl, a):
    return scl.to_device(a)


def host(scl, t):
    return scl.to_host(t)


def soa(aos):  # [N][n][L] -> [n][N][L]
    return np.ascontiguousarray(np.transpose(aos, (1, 0, 2)))


def ints(hexes):
    return [int(h, 16) for h in hexes]


# ---------------------------------------------------------------------------------------------- element-wise
@pytest.mark.parametrize("f", ALL_FIELDS)
@pytest.mark.parametrize("n", [0, 1, 2, 3, 255, 257, 4099])
def test_elementwise_vs_oracle(scl, port, f, n):
    L = O.LIMBS[f]
    a = rand_elems(port, f, n, b"ew-a") if n else np.zeros((0, L), np.uint64)
    b = rand_elems(port, f, n, b"ew-b") if n else np.zeros((0, L), np.uint64)
    if n > 2:  # edge values
        a[0] = port.from_int(f, 0)
        a[1] = port.from_int(f, -1)
        b[0] = port.from_int(f, -1)
        b[1] = port.from_int(f, -1)
    da, db = dev(scl, a), dev(scl, b)
    for op in (O.ADD, O.SUB, O.MUL, O.NEG):
        got = host(scl, scl.ew(f, op, da, db))
        want = port.ew(f, op, a, b) if n else a
        assert np.array_equal(got, want), (f, op, n)
    if n:
        nz = a.copy()
        zero = port.from_int(f, 0)
        for i in range(n):
            if np.array_equal(nz[i], zero):
                nz[i] = port.from_int(f, 7)
        nzb = b.copy()
        for i in range(n):
            if np.array_equal(nzb[i], zero):
                nzb[i] = port.from_int(f, 9)
        if f in SLOW_ORACLE and n > 300:
            nz, nzb = nz[:300], nzb[:300]  # bit-serial oracle inversions are slow
        assert np.array_equal(host(scl, scl.ew(f, O.INV, dev(scl, nz))), port.ew(f, O.INV, nz))
        assert np.array_equal(host(scl, scl.ew(f, O.DIV, dev(scl, nz), dev(scl, nzb))), port.ew(f, O.DIV, nz, nzb))


def _plant_zeros(port, f, a, chain):
    """zeros where a lane's chain, a wave, a workgroup tile (64 lanes x chain) and the batch begin and end"""
    n, tile = len(a), 64 * chain
    spots = {0, 1, 63, 64, 65, tile - 1, tile, tile + 1, 2 * tile - 1, n - 1, n - 2, n // 2}
    spots |= {(7919 * k) % n for k in range(1, 40)}
    spots = sorted(s_ for s_ in spots if 0 <= s_ < n)
    a[spots] = port.from_int(f, 0)
    return np.array(spots)


@pytest.mark.parametrize("f", [O.M61, O.M127])
@pytest.mark.parametrize("n", [1_000_003, 1 << 20])
def test_batched_inverse_against_the_oracle_at_a_million(scl, port, f, n):
    """FF::invert / operator/ over 10^6 elements (k_ew_inv / k_ew_inv_rolled: one Fermat chain per lane-chain instead of one per
    element, ff.h:203-246, small_ff.h:61-92) against the oracle's extended Euclid, element for element; then the same batch with zeros
    at every lane / wave / tile / batch boundary: the reference's error, every other slot still the inverse, the zeros' slots 0."""
    a, b = rand_elems(port, f, n, b"inv-1e6-a"), rand_elems(port, f, n, b"inv-1e6-b")
    zero = port.from_int(f, 0)
    for x, sub in ((a, 7), (b, 9)):
        x[np.all(x == zero, axis=1)] = port.from_int(f, sub)
    want_inv, want_div = port.ew(f, O.INV, a), port.ew(f, O.DIV, b, a)
    for chain in ([0] if f == O.M61 else [0, 8, 16, 32, 64, 128]):
        scl.set_tuning("inv_batch", chain)
        try:
            da = dev(scl, a)
            assert np.array_equal(host(scl, scl.ew(f, O.INV, da)), want_inv), (f, n, chain)
            assert np.array_equal(host(scl, scl.ew(f, O.DIV, dev(scl, b), da)), want_div), (f, n, chain)
            assert np.array_equal(host(scl, scl.ew(f, O.INV, da, out=da)), want_inv), "in place"
            z = a.copy()
            spots = _plant_zeros(port, f, z, chain or (32 if f == O.M61 else 8))
            out = scl.empty(f, n)
            with pytest.raises(scl.SclError) as ei:
                scl.ew(f, O.INV, dev(scl, z), out=out)
            assert ei.value.status == scl.ERR_ZERO_INVERSE
            got, keep = host(scl, out), np.ones(n, bool)
            keep[spots] = False
            assert np.array_equal(got[keep], want_inv[keep]) and not got[spots].any()
        finally:
            scl.set_tuning("inv_batch", 0)
    scl.set_tuning("inv_batch", -1)  # the per-element kernels of rounds 1-4 give the same bits
    try:
        assert np.array_equal(host(scl, scl.ew(f, O.INV, dev(scl, a[:4099]))), want_inv[:4099])
    finally:
        scl.set_tuning("inv_batch", 0)


@pytest.mark.parametrize("f", [O.MONT128, O.GF2_128, O.SECP256K1_SCALAR, O.SECP256K1_FIELD])
@pytest.mark.parametrize("chain", [0, 8, 16, 32, 64, 128])
def test_batched_inverse_slow_oracle_fields(scl, port, f, chain):
    """The fields whose oracle inversion is slow: x * x^-1 = 1 and (b / x) * x = b over 2^20 + 5 elements (products are pinned by the
    oracle on their own; an inverse is unique), an oracle window of 200 elements across the first tile boundary, the per-element
    kernel on a prefix, zeros planted at the chain / wave / tile boundaries."""
    n = (1 << 20) + 5
    a, b = rand_elems(port, f, n, b"inv-slow-a"), rand_elems(port, f, n, b"inv-slow-b")
    zero = port.from_int(f, 0)
    a[np.all(a == zero, axis=1)] = port.from_int(f, 7)
    one = np.broadcast_to(port.from_int(f, 1), a.shape)
    scl.set_tuning("inv_batch", chain)
    try:
        da, db = dev(scl, a), dev(scl, b)
        inv = scl.ew(f, O.INV, da)
        assert np.array_equal(host(scl, scl.ew(f, O.MUL, inv, da)), one)
        quo = scl.ew(f, O.DIV, db, da)
        assert np.array_equal(host(scl, scl.ew(f, O.MUL, quo, da)), b)
        tile = 64 * (chain or 8)
        lo = max(0, tile - 100)
        assert np.array_equal(host(scl, inv)[lo:lo + 200], port.ew(f, O.INV, a[lo:lo + 200]))
        assert np.array_equal(host(scl, quo)[lo:lo + 200], port.ew(f, O.DIV, b[lo:lo + 200], a[lo:lo + 200]))
        assert np.array_equal(host(scl, inv)[-100:], port.ew(f, O.INV, a[-100:]))
        scl.set_tuning("inv_batch", -1)
        assert np.array_equal(host(scl, scl.ew(f, O.INV, dev(scl, a[:8191]))), host(scl, inv)[:8191])
        scl.set_tuning("inv_batch", chain)
        z = a.copy()
        spots = _plant_zeros(port, f, z, chain or 8)
        out = scl.empty(f, n)
        with pytest.raises(scl.SclError) as ei:
            scl.ew(f, O.INV, dev(scl, z), out=out)
        assert ei.value.status == scl.ERR_ZERO_INVERSE
        got, keep = host(scl, out), np.ones(n, bool)
        keep[spots] = False
        assert np.array_equal(got[keep], host(scl, inv)[keep]) and not got[spots].any()
    finally:
        scl.set_tuning("inv_batch", 0)


@pytest.mark.parametrize("f", [O.M127, O.MONT128])
@pytest.mark.parametrize("chain,block", [(32, 4), (32, 8), (64, 4), (64, 8), (128, 4), (128, 8), (256, 4), (256, 8)])
def test_inverse_in_two_levels(scl, port, f, chain, block):
    """k_ew_inv_blocked -- a lane's chain as checkpoints plus recomputed blocks (3 E instead of 5 E bytes per element through
    HBM) -- against the rolled kernel word for word (an inverse is unique) and against the oracle: inverse, divide, in place,
    ragged sizes around the tile (64 lanes x chain) and the block, zeros planted at every block / chain / wave / tile boundary."""
    tile = 64 * chain
    zero, one = port.from_int(f, 0), port.from_int(f, 1)
    for n in (1, 63, 64 * block - 1, 64 * block + 1, tile - 1, tile, tile + 1, 3 * tile + 64 * block + 5, 200_003):
        a, b = rand_elems(port, f, n, b"two-a"), rand_elems(port, f, n, b"two-b")
        a[np.all(a == zero, axis=1)] = port.from_int(f, 7)
        z = a.copy()
        spots = sorted({s_ for s_ in (0, 63, 64, 64 * block - 1, 64 * block, 64 * (chain - 1), tile - 1, tile, n // 2, n - 1) if 0 <= s_ < n})
        z[spots] = zero
        da, db, dz = dev(scl, a), dev(scl, b), dev(scl, z)
        res = {}
        for two in (-1, block):
            scl.set_tuning("inv_batch", chain)
            scl.set_tuning("inv_two_level", two)
            try:
                status = scl.ew_status_buffer()
                inv = scl.ew(f, O.INV, da)
                quo = scl.ew(f, O.DIV, db, da)
                zs = scl.ew_status(f, O.INV, dz, None, status)
                flagged = int(status.item())
                inplace = da.clone()
                scl.ew(f, O.INV, inplace, out=inplace)
                res[two] = (host(scl, inv), host(scl, quo), host(scl, zs), flagged, host(scl, inplace))
            finally:
                scl.set_tuning("inv_batch", 0)
                scl.set_tuning("inv_two_level", 0)
        r0, r1 = res[-1], res[block]
        assert all(np.array_equal(x, y) for x, y in zip(r0[:3], r1[:3])) and r0[3] == r1[3] == 1, (f, chain, block, n)
        assert np.array_equal(r1[4], r1[0]) and not r1[2][spots].any()
        w = slice(0, min(n, 300 if f == O.M127 else 40))
        assert np.array_equal(r1[0][w], port.ew(f, O.INV, a[w])) and np.array_equal(r1[1][w], port.ew(f, O.DIV, b[w], a[w]))
        assert np.array_equal(host(scl, scl.ew(f, O.MUL, dev(scl, r1[0]), da)), np.broadcast_to(one, a.shape))


def test_inverse_in_two_levels_is_what_a_large_mersenne127_batch_takes(scl, port):
    """the automatic choice: Mersenne127 from chains of 64 (6 * 10^6 elements) on; the oracle's Euclid on windows, x * x^-1 = 1 over
    the batch, the rolled kernel word for word"""
    f, n = O.M127, 6_200_011
    a = scl.vector_random(f, n, b"two-auto")
    inv = scl.ew(f, O.INV, a)
    scl.set_tuning("inv_two_level", -1)
    try:
        assert scl.equals(f, scl.ew(f, O.INV, a), inv)
    finally:
        scl.set_tuning("inv_two_level", 0)
    for lo in (0, 64 * 64 - 50, n - 100):
        assert np.array_equal(host(scl, inv[lo:lo + 100]), port.ew(f, O.INV, host(scl, a[lo:lo + 100])))
    ones = scl.to_device(np.broadcast_to(port.from_int(f, 1), (n, 2)).copy())
    assert scl.equals(f, scl.ew(f, O.MUL, inv, a), ones)


def test_gf2_128_products_on_the_lds_table(scl, port):
    """multiplyEntryWise over GF(2^128): the comb product on the per-lane window table in LDS (k_ew_gf128_mul) against the oracle's
    shift-xor product and against the register-only kernel, sparse / dense corner operands included."""
    f, n = O.GF2_128, 100_003
    a, b = rand_elems(port, f, n, b"gfmul-a"), rand_elems(port, f, n, b"gfmul-b")
    corners = [0, 1, 2, 0x87, 1 << 127, (1 << 128) - 1, (1 << 127) | 1, 0xF << 124, 0xFFFFFFFF << 96]
    k = 0
    for x in corners:
        for y in corners:
            a[k], b[k] = O.from_ints([x], 2)[0], O.from_ints([y], 2)[0]
            k += 1
    got = host(scl, scl.ew(f, O.MUL, dev(scl, a), dev(scl, b)))
    assert np.array_equal(got[:3000], port.ew(f, O.MUL, a[:3000], b[:3000]))
    assert np.array_equal(got[-500:], port.ew(f, O.MUL, a[-500:], b[-500:]))
    # Vector::dot (per-lane tables, xor reduction) and Vector::scalarMultiply (one shared table of the scalar) the same way
    da, db = dev(scl, a), dev(scl, b)
    k = 3000
    d_small = scl.dot(f, da[:k], db[:k])
    assert np.array_equal(d_small.reshape(1, 2), port.dot(f, a[:k], b[:k]).reshape(1, 2))
    d_all, sm = scl.dot(f, da, db), host(scl, scl.scalar_mul(f, da, b[7]))
    assert np.array_equal(sm[:2000], port.scalar_mul(f, a[:2000], b[7])) and np.array_equal(sm[-300:], port.scalar_mul(f, a[-300:], b[7]))
    scl.set_tuning("inv_batch", -1)
    try:
        assert np.array_equal(host(scl, scl.ew(f, O.MUL, dev(scl, a), dev(scl, b))), got)
        assert np.array_equal(scl.dot(f, da, db), d_all) and np.array_equal(host(scl, scl.scalar_mul(f, da, b[7])), sm)
    finally:
        scl.set_tuning("inv_batch", 0)


@pytest.mark.parametrize("f", ALL_FIELDS)
def test_batched_inverse_ragged_sizes(scl, port, f):
    """Thirty batch sizes between 1 and 70 000 (around every chain length x 64 lanes, and random ones) x every chain length: the last
    lane's chain, the last wave and the last tile are all partial somewhere.  x * x^-1 = 1 and (b / x) * x = b over the batch; the
    fast-oracle fields also against the oracle's Euclid element for element."""
    rng = np.random.default_rng(77 + f)
    sizes = sorted({1, 2, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 8191, 8192, 8193, 65535, 65537} | {int(x) for x in rng.integers(1, 70_000, 14)})
    a_all, b_all = rand_elems(port, f, 70_000, b"rag-a"), rand_elems(port, f, 70_000, b"rag-b")
    zero = port.from_int(f, 0)
    a_all[np.all(a_all == zero, axis=1)] = port.from_int(f, 3)
    want_all = port.ew(f, O.INV, a_all) if f not in SLOW_ORACLE else None
    one = port.from_int(f, 1)
    for chain in ([0] if f == O.M61 else [0, 8, 16, 32, 64, 128]):
        scl.set_tuning("inv_batch", chain)
        try:
            for n in sizes:
                da, db = dev(scl, a_all[:n]), dev(scl, b_all[:n])
                inv, quo = scl.ew(f, O.INV, da), scl.ew(f, O.DIV, db, da)
                if want_all is not None:
                    assert np.array_equal(host(scl, inv), want_all[:n]), (f, chain, n)
                else:
                    assert np.array_equal(host(scl, scl.ew(f, O.MUL, inv, da)), np.broadcast_to(one, (n, O.LIMBS[f]))), (f, chain, n)
                assert np.array_equal(host(scl, scl.ew(f, O.MUL, quo, da)), b_all[:n]), (f, chain, n)
        finally:
            scl.set_tuning("inv_batch", 0)


@pytest.mark.parametrize("f", ALL_FIELDS)
def test_inverse_of_zero_is_the_reference_error(scl, port, f):
    a = rand_elems(port, f, 100, b"z")
    a[37] = port.from_int(f, 0)
    with pytest.raises(scl.SclError) as ei:
        scl.ew(f, O.INV, dev(scl, a))
    assert ei.value.status == scl.ERR_ZERO_INVERSE
    assert ei.value.reference_message == "0 not invertible modulo prime"  # test_ff.cc:168-171
    with pytest.raises(scl.SclError):
        scl.ew(f, O.DIV, dev(scl, a), dev(scl, a))


def test_size_mismatch_message(scl, port):
    a = dev(scl, rand_elems(port, O.M61, 4, b"x"))
    b = dev(scl, rand_elems(port, O.M61, 5, b"x"))
    with pytest.raises(scl.SclError) as ei:
        scl.ew(O.M61, O.ADD, a, b)
    assert ei.value.reference_message == "Vec sizes mismatch"  # vector.h:481-485
    assert not scl.equals(O.M61, a, b)


@pytest.mark.parametrize("f,name", FIELDS)
def test_elementwise_golden(scl, f, name):
    g, L = GOLD["fields"][name], O.LIMBS[f]
    for key in ("ew", "ew_edge_pairs"):
        if key not in g:
            continue
        e = g[key]
        a, b = dev(scl, O.from_ints(ints(e["a"]), L)), dev(scl, O.from_ints(ints(e["b"]), L))
        assert O.to_ints(host(scl, scl.ew(f, O.ADD, a, b))) == ints(e["add"])
        assert O.to_ints(host(scl, scl.ew(f, O.SUB, a, b))) == ints(e["sub"])
        assert O.to_ints(host(scl, scl.ew(f, O.MUL, a, b))) == ints(e["mul"])
    e = g["ew"]
    assert O.to_ints(host(scl, scl.ew(f, O.NEG, dev(scl, O.from_ints(ints(e["a"]), L))))) == ints(e["neg"])
    nz, nzb = dev(scl, O.from_ints(ints(e["nz"]), L)), dev(scl, O.from_ints(ints(e["nzb"]), L))
    assert O.to_ints(host(scl, scl.ew(f, O.INV, nz))) == ints(e["inv"])
    assert O.to_ints(host(scl, scl.ew(f, O.DIV, nz, nzb))) == ints(e["div"])
    sm = g["scalar_mul"]
    got = scl.scalar_mul(f, dev(scl, O.from_ints(ints(sm["a"]), L)), O.from_ints([int(sm["scalar"], 16)], L)[0])
    assert O.to_ints(host(scl, got)) == ints(sm["out"])
    d = g["dot"]
    a, b = dev(scl, O.from_ints(ints(d["a"]), L)), dev(scl, O.from_ints(ints(d["b"]), L))
    assert O.to_ints(scl.dot(f, a, b).reshape(1, L)) == [int(d["out"], 16)]
    assert O.to_ints(scl.vsum(f, a).reshape(1, L)) == [int(g["sum"]["out"], 16)]


@pytest.mark.parametrize("f", ALL_FIELDS)
@pytest.mark.parametrize("n", [0, 1, 63, 1000, 100003])
def test_sum_dot_equals(scl, port, f, n):
    L = O.LIMBS[f]
    if f in SLOW_ORACLE and n > 1000:
        n = 5001
    a = rand_elems(port, f, n, b"sd-a") if n else np.zeros((0, L), np.uint64)
    b = rand_elems(port, f, n, b"sd-b") if n else np.zeros((0, L), np.uint64)
    da, db = dev(scl, a), dev(scl, b)
    assert np.array_equal(scl.vsum(f, da), port.sum(f, a) if n else np.zeros(L, np.uint64))
    assert np.array_equal(scl.dot(f, da, db), port.dot(f, a, b) if n else np.zeros(L, np.uint64))
    assert scl.equals(f, da, da.clone())
    if n:
        c = a.copy()
        c[n // 2, 0] ^= np.uint64(1)
        assert not scl.equals(f, da, dev(scl, c))
    s = rand_elems(port, f, 1, b"scalar")[0]
    if n:
        assert np.array_equal(host(scl, scl.scalar_mul(f, da, s)), port.scalar_mul(f, a, s))


# ---------------------------------------------------------------------------------------------- PRG
def test_prg_golden_and_oracle(scl, port):
    for case in GOLD["prg"]:
        if case["sizes"] == [4096]:
            got = bytes(scl.prg_blocks(256, bytes.fromhex(case["seed"])).cpu().numpy())
            assert got.hex() == case["out"]
    # (the last two launches cross a 2^32 / 2^48 boundary of the counter: the upper word of the input block changes inside them)
    for seed, c0, nb in ((b"shamir passive", 0, 2), (b"", 5, 1000), (b"x" * 40, 2 ** 40, 4097), (b"k", 2 ** 63, 3),
                         (b"wrap", 2 ** 32 - 777, 5000), (b"wrap48", 2 ** 48 - 3, 64)):
        got = bytes(scl.prg_blocks(nb, seed, c0).cpu().numpy())
        assert got == port.prg_blocks(seed, c0 % 2 ** 64, nb)
    assert scl.prg_blocks(0, b"s").numel() == 0


@pytest.mark.parametrize("f", ALL_FIELDS)
def test_from_bytes_and_vector_random(scl, port, f):
    L = O.LIMBS[f]
    raw = port.prg(b"raw-bytes", [8 * L * 1000]) + b"\xff" * (8 * L) + b"\x00" * (8 * L)
    rt = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    assert np.array_equal(host(scl, scl.from_bytes(f, rt)), port.from_bytes(f, raw))
    for n in (1, 2, 3, 43, 1001):
        assert np.array_equal(host(scl, scl.vector_random(f, n, b"vec")), port.vector_random(f, b"vec", n))
    # a PRG that has already produced `c0` blocks
    c0 = 11
    skip = port.prg(b"vec", [16 * c0, 8 * L * 9])[16 * c0:]
    assert np.array_equal(host(scl, scl.vector_random(f, 9, b"vec", c0)), port.from_bytes(f, skip))


@pytest.mark.parametrize("name", sorted(MONT))
def test_mont128_c3_shapes_golden(scl, port, name):
    """BASELINE configs[2] -- Shamir (10,3) over the 128-bit Montgomery prime field -- and (40,13): PRG-driven shares and
    reconstructions emitted by the reference's shamirSecretShare / shamirRecoverP over FF<the reference's own Montgomery
    templates at N = 2> (golden_mont128.json), for three moduli; the kernels must give those bits."""
    f, L = O.MONT128, 2
    for c in MONT[name]["shamir_c3"]:
        secrets = O.from_ints(ints(c["secrets"]), L)
        shares = scl.shamir_share_prg(f, dev(scl, secrets), c["t"], c["n"], bytes.fromhex(c["seed"]))
        want = O.from_ints(ints(c["shares"]), L).reshape(len(secrets), c["n"], L)
        assert np.array_equal(host(scl, shares), soa(want)), (name, c["n"], c["t"])
        assert O.to_ints(host(scl, scl.shamir_recover(f, shares))) == ints(c["recovered_all_n"]) == ints(c["secrets"])


@pytest.mark.parametrize("f,name", FIELDS)
def test_from_bytes_golden(scl, f, name):
    g = GOLD["fields"][name]["from_bytes"]
    raw = bytes.fromhex(g["raw"])
    rt = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    assert O.to_ints(host(scl, scl.from_bytes(f, rt))) == ints(g["out"])


# ---------------------------------------------------------------------------------------------- Shamir
@pytest.mark.parametrize("f,name", FIELDS)
def test_shamir_golden(scl, f, name):
    """shares / reconstructions emitted by the real reference (one PRG over all secrets)"""
    g, L = GOLD["fields"][name], O.LIMBS[f]
    for c in g["shamir"]:
        n, t = c["n"], c["t"]
        secrets = O.from_ints(ints(c["secrets"]), L)
        N = secrets.shape[0]
        want = soa(O.from_ints(ints(c["shares"]), L).reshape(N, n, L))
        got = scl.shamir_share_prg(f, dev(scl, secrets), t, n, bytes.fromhex(c["seed"]))
        assert np.array_equal(host(scl, got), want), (name, n, t)
        lam = scl.lagrange_basis(f, n)
        assert O.to_ints(lam) == ints(c["lambda_1_to_n_at_0"])
        rec = scl.shamir_recover(f, got)
        assert O.to_ints(host(scl, rec)) == ints(c["recovered_all_n"])
        # same through the AoS image the reference holds
        aos = scl.soa_to_aos(f, got)
        assert O.to_ints(host(scl, aos)) == ints(c["shares"])
        assert np.array_equal(host(scl, scl.aos_to_soa(f, aos)), want)


@pytest.mark.parametrize("f", ALL_FIELDS)
@pytest.mark.parametrize("n,t,N", [(10, 3, 1000), (4, 3, 257), (3, 1, 2), (10, 0, 65), (40, 13, 300), (128, 42, 130),
                                   (17, 16, 64), (1, 0, 5), (5, 4, 1),
                                   # 49..63 coefficients: one tile of the matrix-core kernel over Mersenne61 (since round 5; the
                                   # chunked Horner kernel for the other fields and for fewer than 1024 multiply-adds per secret)
                                   (128, 63, 130), (70, 49, 90), (64, 50, 70), (60, 59, 33)])
def test_shamir_share_recover_vs_oracle(scl, port, f, n, t, N):
    L = O.LIMBS[f]
    if f in SLOW_ORACLE:
        N = min(N, 40)  # bit-serial / Fermat oracle is slow
    secrets = rand_elems(port, f, N, b"secrets")
    seed = b"share-seed"
    # PRG-driven: bit-identical to per-secret shamirSecretShare on one PRG
    if f != O.GF2_128:  # x++ nodes do not exist in characteristic 2; explicit-node path below covers it
        want = soa(port.shamir_share(f, seed, secrets, t, n))
        got = scl.shamir_share_prg(f, dev(scl, secrets), t, n, seed)
        assert np.array_equal(host(scl, got), want)
        # continuing a batch at secret index 7
        if N > 9:
            tail = scl.shamir_share_prg(f, dev(scl, secrets[7:]), t, n, seed, first_secret=7)
            assert np.array_equal(host(scl, tail), want[:, 7:])
        # both forms of the PRG-driven sharing whatever the automatic choice is: the fused kernels (coefficients drawn into
        # registers) and the two passes (rows drawn into a temporary, then the explicit-coefficient kernel of the shape)
        for mode in (1, -1):
            scl.set_tuning("prg_two_pass", mode)
            try:
                assert np.array_equal(host(scl, scl.shamir_share_prg(f, dev(scl, secrets), t, n, seed)), want), mode
                if N > 9:
                    tail = scl.shamir_share_prg(f, dev(scl, secrets[7:]), t, n, seed, first_secret=7)
                    assert np.array_equal(host(scl, tail), want[:, 7:]), mode
            finally:
                scl.set_tuning("prg_two_pass", 0)
    # explicit coefficients
    coeffs = rand_elems(port, f, t * N, b"coeffs").reshape(N, t, L) if t else np.zeros((N, 0, L), np.uint64)
    if f == O.GF2_128:
        al = O.from_ints(list(range(1, n + 1)), L)
        xs = al
        want2 = np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), xs) for s in range(N)])
        want2 = soa(want2)
    else:
        want2 = soa(port.shamir_share_coeffs(f, secrets, coeffs, n))
    dco = dev(scl, np.ascontiguousarray(np.transpose(coeffs, (1, 0, 2)))) if t else None
    got2 = scl.shamir_share(f, dev(scl, secrets), dco, n)
    assert np.array_equal(host(scl, got2), want2)
    # the Horner kernels on the same inputs: 1 = small-constant nodes, 2 = full-width nodes
    for mode in (1, 2):
        scl.set_tuning("force_table", mode)
        try:
            assert np.array_equal(host(scl, scl.shamir_share(f, dev(scl, secrets), dco, n)), want2), mode
            if f != O.GF2_128:
                assert np.array_equal(host(scl, scl.shamir_share_prg(f, dev(scl, secrets), t, n, seed)), want), mode
        finally:
            scl.set_tuning("force_table", 0)
    # reconstruct from all n shares (reference semantics) and from the first t+1
    lam = scl.lagrange_basis(f, n)
    assert np.array_equal(lam, port.lagrange_basis(f, O.from_ints(list(range(1, n + 1)), L) if f == O.GF2_128 else
                                                   np.stack([port.from_int(f, i + 1) for i in range(n)]),
                                                   port.from_int(f, 0)))
    rec = scl.shamir_recover(f, got2, lam)
    assert np.array_equal(host(scl, rec), secrets)
    rec_t = scl.shamir_recover(f, got2[: t + 1].contiguous())
    assert np.array_equal(host(scl, rec_t), secrets)
    # oracle's per-secret recompute-the-basis path gives the same values
    if f != O.GF2_128:
        assert np.array_equal(host(scl, rec), port.shamir_recover(f, np.ascontiguousarray(np.transpose(want2, (1, 0, 2)))))


@pytest.mark.parametrize("f,n,t,N", [(O.M61, 300, 5, 33), (O.M61, 12, 100, 40), (O.M61, 300, 60, 9), (O.M61, 257, 49, 3),
                                     (O.M127, 150, 50, 10), (O.MONT128, 130, 2, 7), (O.GF2_128, 140, 3, 6),
                                     (O.GF2_128, 9, 55, 5), (O.SECP256K1_SCALAR, 70, 49, 5), (O.SECP256K1_SCALAR, 3, 30, 4)])
def test_no_bound_on_parties_or_threshold(scl, port, f, n, t, N):
    """shamirSecretShare / shamirRecoverP bound neither n nor t (include/scl/ss/shamir.h:51-104; the reference's own test
    shares to 100 parties): beyond one launch's node table (256 / limbs parties) the share kernels go block by block over
    the parties and the reconstruct kernels add canonical partial sums; beyond 48 coefficients the polynomial is evaluated
    in chunks (k_share_chunk), PRG-driven coefficients first drawn into rows (k_prg_coeff_rows)."""
    L = O.LIMBS[f]
    secrets = rand_elems(port, f, N, b"big-secrets")
    coeffs = rand_elems(port, f, t * N, b"big-coeffs").reshape(N, t, L)
    dco = dev(scl, np.ascontiguousarray(np.transpose(coeffs, (1, 0, 2))))
    if f == O.GF2_128:
        xs = O.from_ints(list(range(1, n + 1)), L)
        want = soa(np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), xs) for s in range(N)]))
    else:
        want = soa(port.shamir_share_coeffs(f, secrets, coeffs, n))
    got = scl.shamir_share(f, dev(scl, secrets), dco, n)
    assert np.array_equal(host(scl, got), want)
    if f != O.GF2_128:
        want_prg = soa(port.shamir_share(f, b"big-seed", secrets, t, n))
        assert np.array_equal(host(scl, scl.shamir_share_prg(f, dev(scl, secrets), t, n, b"big-seed")), want_prg)
        k = N // 2
        tail = scl.shamir_share_prg(f, dev(scl, secrets[k:]), t, n, b"big-seed", first_secret=k)
        assert np.array_equal(host(scl, tail), want_prg[:, k:])
    # explicit nodes far from 1..n
    nodes = rand_elems(port, f, n, b"big-nodes")
    want_x = soa(np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)]))
    got_x = scl.shamir_share(f, dev(scl, secrets), dco, n, alphas=nodes)
    assert np.array_equal(host(scl, got_x), want_x)
    if n <= t:
        return                                  # fewer shares than coefficients: nothing to reconstruct
    # reconstruct from ALL n shares (the reference's semantics), default and explicit nodes
    assert np.array_equal(host(scl, scl.shamir_recover(f, got)), secrets)
    lam = port.lagrange_basis(f, nodes, port.from_int(f, 0))
    assert np.array_equal(scl.lagrange_basis(f, n, nodes), lam)
    assert np.array_equal(host(scl, scl.shamir_recover(f, got_x, lam)), secrets)
    if f == O.GF2_128:   # the shared-shift nibble kernel on the same blocks
        scl.set_tuning("force_table", 3)
        try:
            assert np.array_equal(host(scl, scl.shamir_recover(f, got_x, lam)), secrets)
        finally:
            scl.set_tuning("force_table", 0)


@pytest.mark.parametrize("f", ALL_FIELDS)
def test_share_with_explicit_nodes(scl, port, f):
    """nodes 42..48 (test_shamir.cc:81-109), worst-case small nodes, and arbitrary field elements"""
    L, N, t = O.LIMBS[f], 333, 3
    secrets = rand_elems(port, f, N, b"xs")
    coeffs = rand_elems(port, f, t * N, b"xc").reshape(N, t, L)
    if N > 3:  # p-1 everywhere in one column: the lazy per-limb accumulators at their bound
        secrets[0] = port.from_int(f, -1)
        coeffs[0] = port.from_int(f, -1)
    dco = dev(scl, np.ascontiguousarray(np.transpose(coeffs, (1, 0, 2))))
    node_sets = [np.stack([port.from_int(f, v) for v in range(42, 49)]),
                 np.stack([port.from_int(f, v) for v in (812, 1, 811, 3)]),   # 812^3 just below 2^29
                 np.stack([port.from_int(f, v) for v in (813, 2)]),           # 813^3 just above 2^29
                 rand_elems(port, f, 5, b"nodes")]
    if f == O.GF2_128:
        node_sets = [O.from_ints(list(range(42, 49)), L), rand_elems(port, f, 5, b"nodes")]
    for nodes in node_sets:
        want = soa(np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)]))
        got = scl.shamir_share(f, dev(scl, secrets), dco, nodes.shape[0], alphas=nodes)
        assert np.array_equal(host(scl, got), want)
        if nodes.shape[0] > t:
            lam = scl.lagrange_basis(f, nodes.shape[0], nodes)
            assert np.array_equal(host(scl, scl.shamir_recover(f, got, lam)), secrets)


@pytest.mark.parametrize("f", [O.SECP256K1_SCALAR, O.SECP256K1_FIELD])
@pytest.mark.parametrize("n", [1, 2, 3, 10, 11])
def test_share_on_lane_pairs_256_bit_fields(scl, port, f, n):
    """k_share_small_pair (32-byte elements: two lanes per secret, the parties two at a time with the reductions shared out
    between the lanes): every threshold 1..7 it is compiled for, odd and even party counts (the odd party out of the last pair),
    odd N, p - 1 / 0 operands, against the oracle; the lane-per-element kernel ("share_waves128" 0) gives the same matrix; a
    window of a wider share matrix through the raw ABI stays inside its rows."""
    import ctypes as C
    L, N = 4, 777
    secrets = rand_elems(port, f, N, b"pair-s")
    allc = rand_elems(port, f, 7 * N, b"pair-c").reshape(N, 7, L)
    secrets[0] = port.from_int(f, -1)
    allc[0] = port.from_int(f, -1)
    allc[N - 1] = port.from_int(f, 0)
    for t in range(1, 8):
        coeffs = np.ascontiguousarray(allc[:, :t])
        dco = dev(scl, np.ascontiguousarray(np.transpose(coeffs, (1, 0, 2))))
        want = soa(port.shamir_share_coeffs(f, secrets, coeffs, n))
        got = scl.shamir_share(f, dev(scl, secrets), dco, n)
        assert np.array_equal(host(scl, got), want), (n, t)
        scl.set_tuning("share_waves128", 0)
        try:
            assert np.array_equal(host(scl, scl.shamir_share(f, dev(scl, secrets), dco, n)), want), (n, t, "lane per element")
        finally:
            scl.set_tuning("share_waves128", 12)
    stride = N + 5
    buf = torch.full((n * stride * L,), -1, dtype=torch.int64, device="cuda")
    dsec = dev(scl, secrets)
    st = scl.lib.scl_hip_shamir_share(f, C.c_void_p(buf.data_ptr()), C.c_size_t(stride), C.c_void_p(dsec.data_ptr()), C.c_void_p(dco.data_ptr()),
                                      C.c_size_t(N), C.c_size_t(N), C.c_size_t(7), C.c_size_t(n), None, None)
    assert st == 0, scl.lib.scl_hip_last_error()
    torch.cuda.synchronize()
    flat = buf.cpu().numpy().view(np.uint64).reshape(n, stride, L)
    assert np.array_equal(flat[:, :N], want) and (flat[:, N:] == np.uint64(2 ** 64 - 1)).all()


@pytest.mark.parametrize("f", [O.M61, O.M127, O.GF2_128])
@pytest.mark.parametrize("n", [5, 10, 40, 128])
def test_share_every_threshold_up_to_16(scl, port, f, n):
    """k_share_blocked (Mersenne61, group size 8 / 8 / 6 / 4 for these n) and the per-threshold Horner bodies:
    every t in 1..16, odd N (vector head + scalar tail), p-1 everywhere in one secret, against the oracle"""
    L, N = O.LIMBS[f], 131
    secrets = rand_elems(port, f, N, b"bl-s")
    allc = rand_elems(port, f, 16 * N, b"bl-c").reshape(N, 16, L)
    secrets[0] = port.from_int(f, -1)
    allc[0] = port.from_int(f, -1)
    allc[N - 1] = port.from_int(f, 0)
    nodes_big = O.from_ints([811 + 7 * i for i in range(n)], L)   # small nodes whose 4th powers pass 2^29
    for t in range(1, 17):
        coeffs = np.ascontiguousarray(allc[:, :t])
        dco = dev(scl, np.ascontiguousarray(np.transpose(coeffs, (1, 0, 2))))
        if f == O.GF2_128:   # default nodes are the bit patterns of 1..n (the x++ walk of shamir.h:62-65 cycles 1,0,1,.. here)
            nodes = O.from_ints(list(range(1, n + 1)), L)
            want = soa(np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)]))
        else:
            want = soa(port.shamir_share_coeffs(f, secrets, coeffs, n))
        for mode in ((-1, 0), (-1, 1)):      # blocked / small-node kernels, then plain Horner
            scl.set_tuning("mfma", mode[0])
            scl.set_tuning("force_table", mode[1])
            try:
                got = scl.shamir_share(f, dev(scl, secrets), dco, n)
            finally:
                scl.set_tuning("mfma", 0)
                scl.set_tuning("force_table", 0)
            assert np.array_equal(host(scl, got), want), (t, mode)
        if t in (5, 9, 16) and n <= 40:
            want2 = soa(np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes_big) for s in range(N)]))
            got2 = scl.shamir_share(f, dev(scl, secrets), dco, n, alphas=nodes_big)
            assert np.array_equal(host(scl, got2), want2), t


@pytest.mark.parametrize("n", [47, 57, 63, 64])
def test_gf_tile_share_upper_tiles(scl, port, n):
    """k_share_gf_tiles at the nodes the smaller cases never reach: the full tiles H = 5..7 (nodes 40..63), the one-node tile
    H = 8 (node 64: fold cadence 2, 12-bit shifts) and ragged last tiles (n = 47, 57, 63), every threshold 5..16, against the
    oracle's Polynomial::evaluate at the bit-pattern nodes -- and word for word against the per-node kernel ("gf_tiles" 0)."""
    f, L, N = O.GF2_128, 2, 67
    secrets = rand_elems(port, f, N, b"gft-s")
    allc = rand_elems(port, f, 16 * N, b"gft-c").reshape(N, 16, L)
    secrets[0] = port.from_int(f, -1)
    allc[0] = np.uint64(0xFFFFFFFFFFFFFFFF)     # all-ones coefficients: every bit pushed past x^127 gets folded
    allc[N - 1] = port.from_int(f, 0)
    nodes = O.from_ints(list(range(1, n + 1)), L)
    for t in range(5, 17):
        coeffs = np.ascontiguousarray(allc[:, :t])
        dco = dev(scl, np.ascontiguousarray(np.transpose(coeffs, (1, 0, 2))))
        got = {}
        for tiles in (1, 0):
            scl.set_tuning("gf_tiles", tiles)
            try:
                got[tiles] = host(scl, scl.shamir_share(f, dev(scl, secrets), dco, n))
            finally:
                scl.set_tuning("gf_tiles", 1)
        assert np.array_equal(got[1], got[0]), t
        if t in (5, 8, 13, 16):   # (the bit-serial oracle multiplier: four thresholds per n keep the case to seconds)
            want = soa(np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)]))
            assert np.array_equal(got[1], want), t


@pytest.mark.parametrize("n,t,N", [(128, 42, 300), (40, 13, 257), (10, 3, 1000), (33, 8, 31), (64, 31, 65), (65, 32, 96),
                                   (5, 1, 7), (128, 48, 129), (1, 1, 40), (32, 31, 64)])
def test_share_on_matrix_cores_vs_oracle(scl, port, n, t, N):
    """the i8-limb MFMA formulation of V * C (csrc/share_mfma.hpp) == per-secret Horner in the oracle"""
    f, L = O.M61, 1
    secrets = rand_elems(port, f, N, b"mf-s")
    coeffs = rand_elems(port, f, t * N, b"mf-c").reshape(N, t, L)
    secrets[0] = port.from_int(f, -1)          # p-1 everywhere: every 7-bit limb at its maximum
    coeffs[0] = port.from_int(f, -1)
    coeffs[N - 1] = port.from_int(f, 0)
    dco = dev(scl, np.ascontiguousarray(np.transpose(coeffs, (1, 0, 2))))
    want = soa(port.shamir_share_coeffs(f, secrets, coeffs, n))
    scl.set_tuning("mfma", 1)
    try:
        got = scl.shamir_share(f, dev(scl, secrets), dco, n)
        assert np.array_equal(host(scl, got), want)
        # arbitrary (full-width) nodes too
        nodes = rand_elems(port, f, n, b"mf-nodes")
        want2 = soa(np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)]))
        got2 = scl.shamir_share(f, dev(scl, secrets), dco, n, alphas=nodes)
        assert np.array_equal(host(scl, got2), want2)
    finally:
        scl.set_tuning("mfma", 0)
    # and the path chosen automatically agrees with the forced VALU path
    auto = scl.shamir_share(f, dev(scl, secrets), dco, n)
    assert np.array_equal(host(scl, auto), want)


@pytest.mark.parametrize("f", [O.M61, O.M127])
def test_recover_fixed_and_table_kernels_agree(scl, port, f):
    L = O.LIMBS[f]
    for m in (1, 2, 7, 8, 9, 15, 16, 17, 24, 33, 100, 128):
        N = 515
        shares = rand_elems(port, f, m * N, b"tbl").reshape(m, N, L)
        lam = rand_elems(port, f, m, b"lam")
        want = port.shamir_recover_lambda(f, np.ascontiguousarray(np.transpose(shares, (1, 0, 2))), lam)
        ds = dev(scl, shares)
        assert np.array_equal(host(scl, scl.shamir_recover(f, ds, lam)), want), m
        scl.set_tuning("force_table", 1)
        try:
            assert np.array_equal(host(scl, scl.shamir_recover(f, ds, lam)), want), m
        finally:
            scl.set_tuning("force_table", 0)
        scl.set_tuning("force_scalar", 1)
        scl.set_tuning("nontemporal", 0)
        try:
            assert np.array_equal(host(scl, scl.shamir_recover(f, ds, lam)), want), m
        finally:
            scl.set_tuning("force_scalar", 0)
            scl.set_tuning("nontemporal", 1)


def test_worst_case_lazy_accumulation(scl, port):
    """all operands p-1: the lazy 128-bit accumulators must not wrap (M61: 64-term bound); for the Montgomery
    fields the unreduced column sums and their single reduction at the largest residue"""
    for f in (O.M61, O.M127, O.MONT128, O.SECP256K1_SCALAR, O.SECP256K1_FIELD):
        L = O.LIMBS[f]
        for m in (16, 64, 65, 128, 200 if f == O.M61 else 128):
            if m > 256 // L:
                continue
            N = 130
            pm1 = port.from_int(f, -1)
            shares = np.tile(pm1, (m, N, 1))
            lam = np.tile(pm1, (m, 1))
            want = port.shamir_recover_lambda(f, np.ascontiguousarray(np.transpose(shares, (1, 0, 2))), lam)
            assert np.array_equal(host(scl, scl.shamir_recover(f, dev(scl, shares), lam)), want), (f, m)
        n = 100000
        a = np.tile(port.from_int(f, -1), (n, 1))
        assert np.array_equal(scl.dot(f, dev(scl, a), dev(scl, a)), port.dot(f, a, a))
        assert np.array_equal(scl.vsum(f, dev(scl, a)), port.sum(f, a))
        # sharing with every coefficient and every node at p-1 (full-width nodes: Horner / Vandermonde-table kernels)
        N, t, n = 70, 3, 6
        pm1 = port.from_int(f, -1)
        secrets = np.tile(pm1, (N, 1))
        coeffs = np.tile(pm1, (N, t, 1))
        nodes = np.stack([port.from_int(f, -1 - i) for i in range(n)])
        want = soa(np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)]))
        got = scl.shamir_share(f, dev(scl, secrets), dev(scl, np.ascontiguousarray(np.transpose(coeffs, (1, 0, 2)))), n,
                               alphas=nodes)
        assert np.array_equal(host(scl, got), want), f


@pytest.mark.parametrize("f,name", FIELDS)
def test_lagrange_golden_and_errors(scl, f, name):
    g, L = GOLD["fields"][name], O.LIMBS[f]
    port = O.Port()
    for c in g["lagrange"]:
        nodes = np.stack([port.from_int(f, v) for v in c["nodes"]])
        x = port.from_int(f, c["x"])
        assert O.to_ints(scl.lagrange_basis(f, len(c["nodes"]), nodes, x)) == ints(c["out"])
    with pytest.raises(scl.SclError) as ei:
        scl.lagrange_basis(f, 3, np.stack([port.from_int(f, v) for v in (1, 2, 2)]))
    assert ei.value.reference_message == g["lagrange_dup_error"]
    for c in g["recover_at"]:
        al = O.from_ints(ints(c["alphas"]), L)
        lam = scl.lagrange_basis(f, al.shape[0], al, O.from_ints([int(c["x"], 16)], L)[0])
        sh = O.from_ints(ints(c["shares"]), L).reshape(-1, 1, L)
        assert O.to_ints(host(scl, scl.shamir_recover(f, dev(scl, sh), lam))) == [int(c["out"], 16)]


@pytest.mark.parametrize("f,name", FIELDS)
def test_recover_detect(scl, port, f, name):
    g, L = GOLD["fields"][name], O.LIMBS[f]
    rd = g["recover_d"]
    aos = O.from_ints(ints(rd["shares"]), L).reshape(-1, rd["n"], L)
    out, status, bad = scl.shamir_recover_detect(f, dev(scl, soa(aos)), rd["t"])
    assert status.cpu().tolist() == rd["status"]
    assert bad == sum(rd["status"])
    assert O.to_ints(host(scl, out)) == ints(rd["out"])
    # bigger batch vs the oracle, random corruption
    N, t = 2000, 5
    n = 2 * t + 1
    secrets = rand_elems(port, f, N, b"det")
    aos = port.shamir_share(f, b"det-seed", secrets, t, n)
    rng = np.random.default_rng(3)
    for s in rng.choice(N, 100, replace=False):
        aos[s, rng.integers(0, n)] = port.from_int(f, int(rng.integers(0, 1000)))
    want_out, want_st = port.shamir_recover_d(f, aos, t)
    out, status, bad = scl.shamir_recover_detect(f, dev(scl, soa(aos)), t)
    assert np.array_equal(status.cpu().numpy(), want_st)
    assert np.array_equal(host(scl, out), want_out)
    assert bad == int(want_st.sum())
    with pytest.raises(scl.SclError) as ei:
        scl.shamir_recover_detect(f, dev(scl, soa(aos))[:5].contiguous(), t)
    assert ei.value.reference_message == "not enough shares provided to detect errors"


@pytest.mark.parametrize("t,d,extra,N", [(42, 42, 0, 5000), (23, 23, 1, 4100), (30, 20, 3, 4500), (63, 63, 0, 4200), (10, 60, 0, 4097)])
def test_recover_detect_on_matrix_cores(scl, port, t, d, extra, N):
    """shamirRecoverD over Mersenne61 with many check rows over many shares: the rows-times-shares product runs on the matrix
    cores (k_share_mfma_* as a matmul) and k_detect_compare does the checks.  Same statuses, values and count as the oracle
    and as the vector-ALU kernel ("mfma" -1), errors in checked rows, in the interpolated rows and in ignored shares."""
    f, L = O.M61, 1
    m = d + t + extra
    nodes = np.stack([port.from_int(f, i + 1) for i in range(m)])
    coeffs = rand_elems(port, f, (d + 1) * N, b"mcd-c").reshape(N, d + 1, L)
    # shares of N polynomials of degree d at 1..m, through the library's own matmul (Vandermonde rows times coefficients)
    V = O.from_ints([pow(i + 1, k, (1 << 61) - 1) for i in range(m) for k in range(d + 1)], L).reshape(m, d + 1, L)
    soa_sh = scl.matmul(f, dev(scl, V), dev(scl, soa(coeffs)))
    aos = np.ascontiguousarray(host(scl, soa_sh).transpose(1, 0, 2))
    some = [0, 1, N // 2, N - 1]
    for s in some:
        assert np.array_equal(aos[s], port.poly_eval(f, coeffs[s], nodes))
    junk = rand_elems(port, f, N, b"mcd-j")
    rng = np.random.default_rng(t * 100 + d)
    hit = rng.choice(N, N // 6, replace=False)
    for k, s in enumerate(hit):
        aos[s, k % m] = junk[s]
    shares = dev(scl, soa(aos))
    x = port.from_int(f, 0)
    res = {}
    for mode in (1, -1):
        scl.set_tuning("mfma", mode)
        try:
            res[mode] = scl.shamir_recover_detect(f, shares, t, d=d, alphas=nodes, x=x) if (d != t or extra) else scl.shamir_recover_detect(f, shares, t)
        finally:
            scl.set_tuning("mfma", 0)
    (o1, s1, b1), (o0, s0, b0) = res[1], res[-1]
    assert b1 == b0 and np.array_equal(s1.cpu().numpy(), s0.cpu().numpy()) and np.array_equal(host(scl, o1), host(scl, o0))
    st = s1.cpu().numpy()
    sub = np.concatenate([hit[:40], np.setdiff1d(np.arange(N), hit)[:40]])      # against the oracle on a sample
    if d == t and extra == 0:
        want_out, want_st = port.shamir_recover_d(f, aos[sub], t)
        assert np.array_equal(st[sub], want_st) and np.array_equal(host(scl, o1)[sub], want_out)
    for k, s in enumerate(hit):
        i = k % m
        if d + 1 <= i < d + t and not np.array_equal(junk[s], port.poly_eval(f, coeffs[s], nodes[i:i + 1])[0]):
            assert st[s] == 1
        if i >= d + t:
            assert st[s] == 0 and np.array_equal(host(scl, o1)[s], coeffs[s, 0])
    clean = np.setdiff1d(np.arange(N), hit)
    assert not st[clean].any() and np.array_equal(host(scl, o1)[clean], coeffs[clean, 0])
    assert b1 == int(st.sum())


@pytest.mark.parametrize("f", ALL_FIELDS)
@pytest.mark.parametrize("t,N", [(1, 7), (2, 300), (4, 513), (5, 1001), (8, 257), (9, 600), (13, 333), (17, 64), (20, 129), (42, 70)])
def test_recover_detect_row_blocks(scl, port, f, t, N):
    """shamirRecoverD over one, two and three row blocks of the check table (t rows: RB = 4 up to t = 4, 8 above; two
    rows per pass for the 256-bit field), odd batch sizes (the scalar tail of the 2-secrets-per-lane launch), errors in
    every share position -- the last share (index 2t) is never checked by the reference's short overload."""
    slow = f in SLOW_ORACLE
    if slow and t > 5:
        N = min(N, 40)
    L = O.LIMBS[f]
    n = 2 * t + 1
    secrets = rand_elems(port, f, N, b"detb")
    if f == O.GF2_128:   # the oracle's shamirSecretShare walks x++ like the reference (meaningless in characteristic 2)
        nodes = O.from_ints(list(range(1, n + 1)), L)
        coeffs = rand_elems(port, f, t * N, b"detb-c").reshape(N, t, L)
        aos = np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)])
    else:
        aos = port.shamir_share(f, b"detb-seed-%d" % t, secrets, t, n)
    clean = aos.copy()
    rng = np.random.default_rng(100 * t + N)
    junk = rand_elems(port, f, N, b"detb-junk")
    hit = rng.choice(N, max(1, N // 5), replace=False)
    for k, s in enumerate(hit):
        aos[s, k % n] = junk[s]                      # walks over every party index, check rows of every block included
    out, status, bad = scl.shamir_recover_detect(f, dev(scl, soa(aos)), t)
    if not (slow and t > 5):                          # the C oracle's per-call bases take minutes for these
        want_out, want_st = port.shamir_recover_d(f, aos, t)
        assert np.array_equal(status.cpu().numpy(), want_st)
        assert np.array_equal(host(scl, out), want_out)
        assert bad == int(want_st.sum())
    # what the algebra fixes without any oracle: a junk value at index 0..2t-1 is caught unless it equals the share,
    # index 2t is never looked at, flagged secrets come back as zero, the others as the shared secret
    st = status.cpu().numpy()
    got = host(scl, out)
    for k, s in enumerate(hit):
        i = k % n
        changed = not np.array_equal(aos[s, i], clean[s, i])
        if i < 2 * t and changed and not (t == 1):    # t = 1 has no check row at all (shamir.h:129: i in [d+1, d+t))
            assert st[s] == 1 and not got[s].any()
        if i == 2 * t:
            assert st[s] == 0 and np.array_equal(got[s], secrets[s])
    untouched = np.setdiff1d(np.arange(N), hit)
    assert not st[untouched].any() and np.array_equal(got[untouched], secrets[untouched])
    assert bad == int(st.sum())


# ---------------------------------------------------------------------------------------------- additive
@pytest.mark.parametrize("f,name", FIELDS)
def test_additive_golden(scl, f, name):
    g, L = GOLD["fields"][name], O.LIMBS[f]
    for c in g["additive"]:
        secrets = O.from_ints(ints(c["secrets"]), L)
        N, n = secrets.shape[0], c["n"]
        got = scl.additive_share_prg(f, dev(scl, secrets), n, bytes.fromhex(c["seed"]))
        assert np.array_equal(host(scl, got), soa(O.from_ints(ints(c["shares"]), L).reshape(N, n, L)))
        assert O.to_ints(host(scl, scl.additive_recover(f, got))) == ints(c["sum"])


@pytest.mark.parametrize("f", ALL_FIELDS)
@pytest.mark.parametrize("n,N", [(3, 1001), (10, 64), (1, 7), (2, 1), (70, 33)])
def test_additive_vs_oracle(scl, port, f, n, N):
    L = O.LIMBS[f]
    secrets = rand_elems(port, f, N, b"add-secrets")
    want = soa(port.additive_share(f, b"add", secrets, n))
    got = scl.additive_share_prg(f, dev(scl, secrets), n, b"add")
    assert np.array_equal(host(scl, got), want)
    assert np.array_equal(host(scl, scl.additive_recover(f, got)), secrets)
    if N > 4:
        tail = scl.additive_share_prg(f, dev(scl, secrets[3:]), n, b"add", first_secret=3)
        assert np.array_equal(host(scl, tail), want[:, 3:])
    rnd = rand_elems(port, f, (n - 1) * N, b"rnd").reshape(n - 1, N, L) if n > 1 else None
    got2 = scl.additive_share(f, dev(scl, secrets), dev(scl, rnd) if rnd is not None else None, n)
    h = host(scl, got2)
    if n > 1:
        assert np.array_equal(h[: n - 1], rnd)
    assert np.array_equal(host(scl, scl.additive_recover(f, got2)), secrets)
    assert np.array_equal(host(scl, scl.additive_recover(f, got2)),
                          port.additive_recover(f, np.ascontiguousarray(np.transpose(h, (1, 0, 2)))))


def test_additive_n0_rejected(scl, port):
    with pytest.raises(scl.SclError):
        scl.additive_share_prg(O.M61, dev(scl, rand_elems(port, O.M61, 4, b"s")), 0, b"")


# ---------------------------------------------------------------------------------------------- matrices
@pytest.mark.parametrize("f,name", FIELDS)
def test_matrix_golden(scl, f, name):
    g, L = GOLD["fields"][name], O.LIMBS[f]
    for c in g["vandermonde"]:
        xs = O.from_ints(ints(c["xs"]), L) if c["xs"] else None
        assert O.to_ints(host(scl, scl.vandermonde(f, c["n"], c["m"], xs))) == ints(c["out"])
    for c in g["matmul"]:
        A = dev(scl, O.from_ints(ints(c["A"]), L).reshape(c["n"], c["k"], L))
        B = dev(scl, O.from_ints(ints(c["B"]), L).reshape(c["k"], c["m"], L))
        assert O.to_ints(host(scl, scl.matmul(f, A, B))) == ints(c["C"])
    ve = g["vandermonde_eval"]
    V = scl.vandermonde(f, ve["n"], ve["m"])
    Cm = dev(scl, O.from_ints(ints(ve["C"]), L).reshape(ve["m"], ve["N"], L))
    assert O.to_ints(host(scl, scl.matmul(f, V, Cm))) == ints(ve["out"])
    with pytest.raises(scl.SclError) as ei:
        scl.matmul(f, V, V)
    assert ei.value.reference_message == "matmul: this->cols() != that->rows()"


@pytest.mark.parametrize("f", ALL_FIELDS)
@pytest.mark.parametrize("M,K,N", [(1, 1, 1), (10, 4, 1000), (128, 43, 300), (3, 70, 65), (129, 5, 2)])
def test_matmul_vs_oracle(scl, port, f, M, K, N):
    L = O.LIMBS[f]
    if f in SLOW_ORACLE:
        M, N = min(M, 20), min(N, 30)
    A = rand_elems(port, f, M * K, b"A").reshape(M, K, L)
    B = rand_elems(port, f, K * N, b"B").reshape(K, N, L)
    assert np.array_equal(host(scl, scl.matmul(f, dev(scl, A), dev(scl, B))), port.matmul(f, A, B))


@pytest.mark.parametrize("M,K,N", [(128, 43, 5000), (64, 22, 4099), (100, 64, 300), (1, 1, 70), (33, 5, 129), (16, 40, 4500), (32, 64, 700)])
def test_matmul_on_matrix_cores(scl, port, M, K, N):
    """Matrix::multiply with a small left factor through the i8-digit MFMA kernel == the oracle's i-k-j loop"""
    f, L = O.M61, 1
    A = rand_elems(port, f, M * K, b"mmA").reshape(M, K, L)
    B = rand_elems(port, f, K * N, b"mmB").reshape(K, N, L)
    A[0, 0] = port.from_int(f, -1)
    B[0, 0] = port.from_int(f, -1)
    want = port.matmul(f, A, B)
    scl.set_tuning("mfma", 1)
    try:
        assert np.array_equal(host(scl, scl.matmul(f, dev(scl, A), dev(scl, B))), want)
    finally:
        scl.set_tuning("mfma", 0)
    assert np.array_equal(host(scl, scl.matmul(f, dev(scl, A), dev(scl, B))), want)  # automatic choice


def _matmul_window_check(scl, port, f, A, B, got, rows, cols):
    """the product's entries at rows x cols against the oracle's i-k-j loop over exactly those rows of A and columns of B"""
    want = port.matmul(f, np.ascontiguousarray(A[rows]), np.ascontiguousarray(B[:, cols]))
    assert np.array_equal(got[np.ix_(rows, cols)], want)


@pytest.mark.parametrize("f", ALL_FIELDS)
@pytest.mark.parametrize("M,K,N", [(200, 7000, 300), (300, 300, 300), (257, 10_000, 1), (3, 20_000, 70), (65, 129, 33), (17, 6145, 16)])
def test_matmul_has_no_bound_on_its_shape(scl, port, f, M, K, N):
    """Matrix::multiply(Matrix) is an unbounded i-k-j loop and multiply(Vector) one innerProd per row (matrix.h:477-513): no shape
    is refused -- K beyond the 48 KiB LDS image the column-per-thread kernel keeps of the left factor goes through k_matmul_tiled
    (both factors tiled, K in steps), one column through k_matvec (a wavefront per row).  Against the oracle on a window of rows
    and columns that takes in every edge of the tiling (whole product for the fast-oracle fields at 300^3)."""
    L = O.LIMBS[f]
    A = rand_elems(port, f, M * K, b"nb-A").reshape(M, K, L)
    B = rand_elems(port, f, K * N, b"nb-B").reshape(K, N, L)
    A[0, 0] = A[M - 1, K - 1] = port.from_int(f, -1)
    B[0, 0] = B[K - 1, N - 1] = port.from_int(f, -1)
    got = host(scl, scl.matmul(f, dev(scl, A), dev(scl, B)))
    assert got.shape == (M, N, L)
    if f not in SLOW_ORACLE and M * K * N <= 30_000_000:
        assert np.array_equal(got, port.matmul(f, A, B))
    rows = sorted({0, 1, M // 2, M - 1} | {r for r in (15, 16, 31, 32, 63, 64, 127, 128, 199) if r < M})[:7 if f in SLOW_ORACLE else 12]
    cols = sorted({0, N // 2, N - 1} | {c for c in (15, 16, 17, 63, 64, 255, 256, 299) if c < N})[:5 if f in SLOW_ORACLE else 10]
    if f in SLOW_ORACLE and K > 7000:
        rows, cols = rows[:3], cols[:3]
    _matmul_window_check(scl, port, f, A, B, got, rows, cols)


@pytest.mark.parametrize("M,K,N", [(300, 200, 5000), (129, 65, 4100), (128, 130, 4096), (1000, 64, 4500), (260, 7000, 4200),
                                   (300, 300, 300), (65, 8300, 70), (33, 16500, 33), (97, 33, 131)])
def test_matmul_on_matrix_cores_beyond_one_tile(scl, port, M, K, N):
    """The matrix-core product beyond one 128 x 64 tile of the left factor == the oracle: the general kernel (gemm_mfma.hpp: digit
    planes of both factors in fragment order, K looped inside the kernel, super-steps of 8192 inner columns -- "mfma" 1), the
    (row block, k-chunk) form on the sharing kernels ("mfma" 2: later chunks add to C in the kernel's epilogue), the automatic
    choice, and the vector-ALU kernels."""
    f, L = O.M61, 1
    A = rand_elems(port, f, M * K, b"mmb-A").reshape(M, K, L)
    B = rand_elems(port, f, K * N, b"mmb-B").reshape(K, N, L)
    A[M - 1, K - 1] = B[K - 1, N - 1] = port.from_int(f, -1)
    rows = sorted({0, 1, 31, 32, 63, 64, 127, 128, 129, 255, 256, M // 2, M - 1} & set(range(M)))
    cols = sorted({0, 1, 31, 32, 63, 64, 4095, N // 2, N - 2, N - 1} & set(range(N)))
    first = None
    for mode in (1, 2, 0):
        scl.set_tuning("mfma", mode)
        try:
            got = host(scl, scl.matmul(f, dev(scl, A), dev(scl, B)))
        finally:
            scl.set_tuning("mfma", 0)
        _matmul_window_check(scl, port, f, A, B, got, rows, cols)
        first = got if first is None else first
        assert np.array_equal(got, first), mode
    scl.set_tuning("mfma", -1)   # and the vector-ALU kernels give the same matrix
    try:
        assert np.array_equal(host(scl, scl.matmul(f, dev(scl, A), dev(scl, B))), got)
    finally:
        scl.set_tuning("mfma", 0)
    # factors whose digit planes outgrow one launch's budget go slab by slab (rows of A outside, columns of B inside): 1 MiB here
    scl.set_tuning("mfma", 1)
    scl.set_tuning("gemm_slab_mib", 1)
    try:
        assert np.array_equal(host(scl, scl.matmul(f, dev(scl, A), dev(scl, B))), got)
    finally:
        scl.set_tuning("mfma", 0)
        scl.set_tuning("gemm_slab_mib", 0)


@pytest.mark.parametrize("f", ALL_FIELDS)
@pytest.mark.parametrize("n,N", [(10, 5000), (10, 1027), (3, 513), (7, 64), (1, 100), (40, 1301), (128, 300), (10, 1)])
def test_layout_bridge_both_ways(scl, port, f, n, N):
    """AoS [secret][party] (the reference's Vector per secret, shamir.h:52-68) <-> SoA [party][secret] through the 16-byte kernel
    (k_transpose16: ragged last tiles, odd secret counts, one party) and the 8-byte one ("force_scalar"), against numpy's
    transpose; a window of a wider SoA matrix through the raw ABI (row stride > N, rows 16-byte aligned or not)."""
    import ctypes as C
    L = O.LIMBS[f]
    aos = rand_elems(port, f, N * n, b"layout").reshape(N, n, L)
    want = soa(aos)
    for scalar in (0, 1):
        scl.set_tuning("force_scalar", scalar)
        try:
            got = scl.aos_to_soa(f, dev(scl, aos))
            assert np.array_equal(host(scl, got), want), (f, n, N, scalar)
            assert np.array_equal(host(scl, scl.soa_to_aos(f, got)), aos), (f, n, N, scalar)
        finally:
            scl.set_tuning("force_scalar", 0)
    for stride, off in ((N + 6, 0), (N + 3, 0), (N + 4, 1)):      # even stride, odd stride, rows that start 8 bytes into a line
        buf = torch.full((n * stride * L + 4,), -1, dtype=torch.int64, device="cuda")
        base = C.c_void_p(buf.data_ptr() + 8 * off)
        src = dev(scl, aos)
        st = scl.lib.scl_hip_aos_to_soa(f, base, C.c_size_t(stride), C.c_void_p(src.data_ptr()), C.c_size_t(N), C.c_size_t(n), None)
        assert st == 0, scl.lib.scl_hip_last_error()
        torch.cuda.synchronize()
        flat = buf.cpu().numpy().view(np.uint64)[off:off + n * stride * L].reshape(n, stride, L)
        assert np.array_equal(flat[:, :N], want) and (flat[:, N:] == np.uint64(2 ** 64 - 1)).all()
        back = torch.zeros((N, n, L), dtype=torch.int64, device="cuda")
        st = scl.lib.scl_hip_soa_to_aos(f, C.c_void_p(back.data_ptr()), base, C.c_size_t(stride), C.c_size_t(N), C.c_size_t(n), None)
        assert st == 0, scl.lib.scl_hip_last_error()
        assert np.array_equal(host(scl, back), aos)


@pytest.mark.parametrize("f", [O.MONT128, O.SECP256K1_SCALAR, O.SECP256K1_FIELD])
@pytest.mark.parametrize("n", [1, 2, 3, 10, 16])
def test_reconstruct_with_small_integer_coefficients(scl, port, f, n):
    """At the default nodes 1..n and x = 0 the Lagrange coefficients are the signed binomials (-1)^(i-1) C(n, i): over the
    Montgomery fields the reconstruct kernel then multiplies residues by plain small integers (k_recover_small) instead of
    running Montgomery products -- the same residues as the oracle's shamirRecoverP (shamir.h:81-104) and as the table kernel
    ("force_table"), ragged sizes and a window of a wider matrix included; coefficients that are not small take the table kernel."""
    from math import comb
    L, N, t = O.LIMBS[f], 4099, min(3, n - 1)
    lam = scl.lagrange_basis(f, n)
    for i in range(n):   # lambda_i = (-1)^(i) C(n, i + 1) for party i + 1
        assert np.array_equal(lam[i], port.from_int(f, (-1) ** i * comb(n, i + 1))), (n, i)
    secrets = rand_elems(port, f, N, b"small-lam-s")
    shares = scl.shamir_share_prg(f, dev(scl, secrets), t, n, b"small-lam")
    aos = np.ascontiguousarray(np.transpose(host(scl, shares), (1, 0, 2)))
    want = port.shamir_recover(f, aos[:300])
    got = host(scl, scl.shamir_recover(f, shares, lam))
    assert np.array_equal(got, secrets) and np.array_equal(got[:300], want)
    scl.set_tuning("force_table", 1)
    try:
        assert np.array_equal(host(scl, scl.shamir_recover(f, shares, lam)), got)
    finally:
        scl.set_tuning("force_table", 0)
    if n >= 3:   # explicit nodes / an evaluation point whose coefficients are full-size residues: the general kernel
        nodes = rand_elems(port, f, n, b"small-lam-nodes")
        x = rand_elems(port, f, 1, b"small-lam-x")[0]
        lam2 = scl.lagrange_basis(f, n, nodes, x)
        sh2 = scl.shamir_share(f, dev(scl, secrets), dev(scl, rand_elems(port, f, t * N, b"small-lam-c").reshape(t, N, L)), n, alphas=nodes)
        got2 = host(scl, scl.shamir_recover(f, sh2, lam2))
        aos2 = np.ascontiguousarray(np.transpose(host(scl, sh2), (1, 0, 2)))
        assert np.array_equal(got2[:200], port.shamir_recover_lambda(f, aos2[:200], lam2))


@pytest.mark.parametrize("f", [O.M61, O.M127, O.SECP256K1_SCALAR])
def test_matmul_random_shapes_every_path(scl, port, f):
    """Forty pseudo-random shapes up to 260 x 700 x 260 (ones, primes, powers of two and tile sizes +- 1 among them) through every
    matmul path the field has -- automatic choice, matrix cores forced (general kernel, then the block / chunk form), vector ALU --
    the whole product against the oracle's i-k-j loop."""
    rng = np.random.default_rng(20261004 + f)
    L = O.LIMBS[f]
    special = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 256, 257]
    shapes = []
    for i in range(40 if f != O.SECP256K1_SCALAR else 14):
        pick = lambda hi: int(rng.choice(special)) if rng.random() < 0.5 else int(rng.integers(1, hi))  # noqa: E731
        M, K, N = min(pick(260), 260), min(pick(700), 700), min(pick(260), 260)
        if f == O.SECP256K1_SCALAR:
            M, K, N = min(M, 40), min(K, 200), min(N, 40)
        shapes.append((M, K, N))
    for (M, K, N) in shapes:
        A = rand_elems(port, f, M * K, b"rs-A%d" % (M * 1000 + K)).reshape(M, K, L)
        B = rand_elems(port, f, K * N, b"rs-B%d" % (N * 1000 + K)).reshape(K, N, L)
        want = port.matmul(f, A, B)
        dA, dB = dev(scl, A), dev(scl, B)
        for mode in ((0, 1, 2, -1) if f == O.M61 else (0,)):
            scl.set_tuning("mfma", mode)
            try:
                got = host(scl, scl.matmul(f, dA, dB))
            finally:
                scl.set_tuning("mfma", 0)
            assert np.array_equal(got, want), (f, M, K, N, mode)


@pytest.mark.parametrize("f", ALL_FIELDS)
@pytest.mark.parametrize("M,K,N", [(10, 4, 5001), (3, 16, 2049), (40, 8, 1500), (1, 1, 1024), (128, 7, 4097)])
def test_matmul_thin_inner_dimension(scl, port, f, M, K, N):
    """A Vandermonde-sized left factor against long coefficient rows (test_matrix.cc:342-365's way of sharing) through
    k_matmul_thin -- B's rows in registers, 16-byte packs, odd tails -- against the oracle (a window of columns for the slow-oracle
    fields), against the column-per-thread kernel ("force_table"), and as a window of wider matrices through the raw ABI."""
    import ctypes as C
    L = O.LIMBS[f]
    A = rand_elems(port, f, M * K, b"thin-A").reshape(M, K, L)
    B = rand_elems(port, f, K * N, b"thin-B").reshape(K, N, L)
    A[0, 0] = B[0, 0] = B[K - 1, N - 1] = port.from_int(f, -1)
    # "matmul_lds_min" 1024 pins the thread-per-column kernels at these sizes (the automatic choice takes them from 65536 columns
    # on unless the left factor is tiny: below that the tiled kernel is ahead)
    scl.set_tuning("matmul_lds_min", 1024)
    scl.set_tuning("mfma", -1)
    try:
        got = host(scl, scl.matmul(f, dev(scl, A), dev(scl, B)))
        cols = list(range(N)) if f not in SLOW_ORACLE else sorted({0, 1, 2, 3, 255, 256, 257, N // 2, N - 2, N - 1})
        assert np.array_equal(got[:, cols], port.matmul(f, A, np.ascontiguousarray(B[:, cols])))
        scl.set_tuning("force_table", 1)
        assert np.array_equal(host(scl, scl.matmul(f, dev(scl, A), dev(scl, B))), got)
    finally:
        scl.set_tuning("force_table", 0)
        scl.set_tuning("mfma", 0)
        scl.set_tuning("matmul_lds_min", 0)
    assert np.array_equal(host(scl, scl.matmul(f, dev(scl, A), dev(scl, B))), got)      # whatever the automatic choice is
    scl.set_tuning("matmul_lds_min", 1024)      # (the raw-ABI window below: the thin kernel's odd pitches)
    ldb, ldc = N + 3, N + 5      # odd pitches: the 8-byte path of the one-limb fields
    Bw = torch.full((K, ldb, L), -1, dtype=torch.int64, device="cuda")
    Bw[:, :N] = dev(scl, B)
    Cw = torch.full((M, ldc, L), -1, dtype=torch.int64, device="cuda")
    dA = dev(scl, A)
    st = scl.lib.scl_hip_matmul(f, C.c_void_p(Cw.data_ptr()), C.c_size_t(ldc), C.c_void_p(dA.data_ptr()), C.c_size_t(K),
                                C.c_void_p(Bw.data_ptr()), C.c_size_t(ldb), C.c_size_t(M), C.c_size_t(K), C.c_size_t(N), None)
    assert st == 0, scl.lib.scl_hip_last_error()
    scl.set_tuning("matmul_lds_min", 0)
    hw = host(scl, Cw)
    assert np.array_equal(hw[:, :N], got) and (hw[:, N:] == np.uint64(2 ** 64 - 1)).all()


@pytest.mark.parametrize("f,mode", [(O.M61, 1), (O.M61, -1), (O.M61, 0), (O.M127, 0), (O.SECP256K1_SCALAR, 0)])
def test_matmul_on_windows_of_wider_matrices(scl, port, f, mode):
    """scl_hip_matmul with row pitches beyond the matrices' widths on all three operands (lda > K, ldb > N, ldc > N: windows of
    wider matrices, odd pitches) through the general matrix-core kernel, the tiled kernel and split-K's fallback: the product
    against the oracle, the padding of C untouched."""
    import ctypes as C
    L = O.LIMBS[f]
    M, K, N = (130, 700, 140) if f != O.SECP256K1_SCALAR else (40, 300, 36)
    lda, ldb, ldc = K + 3, N + 5, N + 7
    A = rand_elems(port, f, M * K, b"win-A").reshape(M, K, L)
    B = rand_elems(port, f, K * N, b"win-B").reshape(K, N, L)
    Aw = torch.full((M, lda, L), -1, dtype=torch.int64, device="cuda")
    Bw = torch.full((K, ldb, L), -1, dtype=torch.int64, device="cuda")
    Cw = torch.full((M, ldc, L), -1, dtype=torch.int64, device="cuda")
    Aw[:, :K], Bw[:, :N] = dev(scl, A), dev(scl, B)
    scl.set_tuning("mfma", mode)
    try:
        st = scl.lib.scl_hip_matmul(f, C.c_void_p(Cw.data_ptr()), C.c_size_t(ldc), C.c_void_p(Aw.data_ptr()), C.c_size_t(lda),
                                    C.c_void_p(Bw.data_ptr()), C.c_size_t(ldb), C.c_size_t(M), C.c_size_t(K), C.c_size_t(N), None)
    finally:
        scl.set_tuning("mfma", 0)
    assert st == 0, scl.lib.scl_hip_last_error()
    hw = host(scl, Cw)
    assert np.array_equal(hw[:, :N], port.matmul(f, A, B)) and (hw[:, N:] == np.uint64(2 ** 64 - 1)).all()


def test_vandermonde_matmul_is_sharing(scl, port):
    """test_matrix.cc:342-365: V(n, t+1) x coefficient matrix == Shamir shares at nodes 1..n"""
    f, L, n, t, N = O.M61, 1, 10, 3, 500
    secrets = rand_elems(port, f, N, b"s")
    coeffs = rand_elems(port, f, t * N, b"c").reshape(t, N, L)
    Cm = np.concatenate([secrets.reshape(1, N, L), coeffs])
    V = scl.vandermonde(f, n, t + 1)
    via_matmul = scl.matmul(f, V, dev(scl, Cm))
    via_share = scl.shamir_share(f, dev(scl, secrets), dev(scl, coeffs), n)
    assert scl.equals(f, via_matmul, via_share)


# ---------------------------------------------------------------------------------------------- wire image
@pytest.mark.parametrize("f", ALL_FIELDS)
def test_wire_image(scl, port, f):
    """seri::Serializer<Vector<FF>>: u32 count || FF::write images; golden bytes from the reference"""
    L = O.LIMBS[f]
    name = GOLDEN_NAME.get(f)
    if name:
        for c in GOLD["fields"][name]["wire"]:
            el = O.from_ints(ints(c["elems"]), L) if c["elems"] else np.zeros((0, L), np.uint64)
            got = bytes(scl.wire_pack(f, dev(scl, el) if len(el) else scl.empty(f, 0)).cpu().numpy())
            assert got.hex() == c["bytes"]
    for n in (0, 1, 2, 1000, 4097):
        el = rand_elems(port, f, n, b"wire") if n else np.zeros((0, L), np.uint64)
        raw = scl.wire_pack(f, dev(scl, el) if n else scl.empty(f, 0))
        assert bytes(raw.cpu().numpy()) == port.wire_vector(f, el)
        back = scl.wire_unpack(f, raw)
        assert np.array_equal(host(scl, back), el)
    # unpack reduces non-canonical words like FF::read does
    junk = port.prg(b"wire-junk", [8 * L * 50])
    raw = (50).to_bytes(4, "little") + junk
    rt = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    assert np.array_equal(host(scl, scl.wire_unpack(f, rt)), port.from_bytes(f, junk))
    with pytest.raises(scl.SclError):
        scl.wire_unpack(f, rt[: 4 + 8 * L * 10].clone())  # count says 50, only 10 present


@pytest.mark.parametrize("f", ALL_FIELDS)
def test_wire_matrix_image(scl, port, f):
    """seri::Serializer<Matrix<FF>>: u32 rows || u32 cols || vector image; golden bytes from the reference"""
    L = O.LIMBS[f]
    name = GOLDEN_NAME.get(f)
    if name:
        for c in GOLD["fields"][name]["wire_matrix"]:
            m = O.from_ints(ints(c["elems"]), L).reshape(c["rows"], c["cols"], L) if c["elems"] else None
            got = scl.wire_pack_matrix(f, dev(scl, m) if m is not None else scl.empty(f, 0))
            assert bytes(got.cpu().numpy()).hex() == c["bytes"]
            back = scl.wire_unpack_matrix(f, got)
            assert tuple(back.shape[:2]) == (c["rows"], c["cols"])
            if m is not None:
                assert np.array_equal(host(scl, back), m)
    for rows, cols in ((1, 1), (3, 7), (43, 128), (128, 43)):
        m = rand_elems(port, f, rows * cols, b"wm").reshape(rows, cols, L)
        raw = scl.wire_pack_matrix(f, dev(scl, m))
        assert bytes(raw.cpu().numpy()) == port.wire_matrix(f, m)
        assert np.array_equal(host(scl, scl.wire_unpack_matrix(f, raw)), m)
        # into a wider destination (pitch > cols) and out of it again
        wide = scl.wire_unpack_matrix(f, raw, capacity=(rows + 2, cols + 5))
        assert np.array_equal(host(scl, wide), m)
    bad = bytearray(port.wire_matrix(f, rand_elems(port, f, 4, b"wm").reshape(2, 2, L)))
    bad[0] = 3
    with pytest.raises(scl.SclError):
        scl.wire_unpack_matrix(f, torch.frombuffer(bad, dtype=torch.uint8).cuda(), capacity=(4, 4))
    with pytest.raises(scl.SclError):
        scl.wire_unpack_matrix(f, raw, capacity=(1, 1))


@pytest.mark.parametrize("f", ALL_FIELDS)
def test_shamir_over_arrays(scl, port, f):
    """shamirSecretShare<Array<FF, W>> (pedersen.h:138): golden shares from the reference, then the oracle at more
    shapes; every component reconstructs on its own"""
    L = O.LIMBS[f]
    name = GOLDEN_NAME.get(f)
    cases = []
    if name:
        for c in GOLD["fields"][name]["shamir_packed"]:
            sec = O.from_ints(ints(c["secrets"]), L).reshape(-1, c["W"], L)
            want = O.from_ints(ints(c["shares"]), L).reshape(sec.shape[0], c["n"], c["W"], L)
            cases.append((sec, c["t"], c["n"], bytes.fromhex(c["seed"]), want))
    if f != O.GF2_128:     # (the oracle's GF(2^128) nodes follow the x++ walk, see test_share_every_threshold_up_to_16)
        for W, n, t, N in ((2, 10, 3, 777), (2, 40, 13, 65), (3, 7, 0, 20), (5, 4, 3, 33), (2, 10, 9, 50)):
            sec = rand_elems(port, f, N * W, b"pk-s").reshape(N, W, L)
            cases.append((sec, t, n, b"pk-seed", port.shamir_share_packed(f, b"pk-seed", sec, t, n)))
    for sec, t, n, seed, want in cases:
        got = scl.shamir_share_prg_packed(f, dev(scl, np.ascontiguousarray(sec.transpose(1, 0, 2))), t, n, seed)
        assert np.array_equal(host(scl, got), want.transpose(2, 1, 0, 3)), (t, n)     # [W][n][N][L]
        for mode in (1, -1):     # two passes / the fused kernels, whatever the automatic choice was
            scl.set_tuning("prg_two_pass", mode)
            try:
                again = scl.shamir_share_prg_packed(f, dev(scl, np.ascontiguousarray(sec.transpose(1, 0, 2))), t, n, seed)
            finally:
                scl.set_tuning("prg_two_pass", 0)
            assert scl.equals(f, again.reshape(-1, L), got.reshape(-1, L)), (t, n, mode)
        lam = scl.lagrange_basis(f, n)
        for j in range(sec.shape[1]):
            assert np.array_equal(host(scl, scl.shamir_recover(f, got[j], lam)), sec[:, j])


@pytest.mark.parametrize("f", ALL_FIELDS)
def test_tcp_frames(scl, port, f):
    """TcpChannel frame = u32 packet size || Packet bytes (tcp_channel.h:125-160); golden frames from the reference"""
    L = O.LIMBS[f]
    name = GOLDEN_NAME.get(f)
    if name:
        for c in GOLD["fields"][name]["frame"]:
            if c["kind"] == "vector":
                el = O.from_ints(ints(c["elems"]), L) if c["elems"] else None
                got = scl.frame_pack(f, dev(scl, el) if el is not None else scl.empty(f, 0))
                assert bytes(got.cpu().numpy()).hex() == c["bytes"]
                back = scl.frame_unpack(f, got)
                assert np.array_equal(host(scl, back), el if el is not None else np.zeros((0, L), np.uint64))
            else:
                m = O.from_ints(ints(c["elems"]), L).reshape(c["rows"], c["cols"], L)
                got = scl.frame_pack(f, dev(scl, m), as_matrix=True)
                assert bytes(got.cpu().numpy()).hex() == c["bytes"]
                assert np.array_equal(host(scl, scl.wire_unpack_matrix(f, got[4:].clone())), m)
    for n in (1, 1000):
        el = rand_elems(port, f, n, b"frame")
        raw = scl.frame_pack(f, dev(scl, el))
        assert bytes(raw.cpu().numpy()) == port.frame(f, el)
        # a receive buffer with trailing bytes of the next frame: only packet-size bytes are looked at
        longer = torch.cat([raw, raw[:12]])
        assert np.array_equal(host(scl, scl.frame_unpack(f, longer)), el)
    with pytest.raises(scl.SclError):
        scl.frame_unpack(f, raw[: raw.numel() - 8].clone())     # frame says more bytes than present


# ------------------------------------------------------------------------------------ Berlekamp-Welch (shamirRecoverC)
def _check_recover_c(scl, port, f, shares_aos, alphas=None):
    """GPU against the oracle, output for output"""
    L = O.LIMBS[f]
    fo, eo, st, ne = port.shamir_recover_c(f, shares_aos, alphas)
    r = scl.shamir_recover_correct(f, dev(scl, soa(shares_aos)), alphas)
    assert np.array_equal(host(scl, r["f"]), soa(fo))
    assert np.array_equal(host(scl, r["err"]), soa(eo))
    assert r["status"].cpu().numpy().tolist() == st.tolist()
    assert r["nerr"].cpu().numpy().tolist() == ne.tolist()
    assert r["failed"] == int(st.sum())
    return r, (fo, eo, st, ne)


@pytest.mark.parametrize("f,name", FIELDS)
def test_recover_correct_golden(scl, port, f, name):
    """shamirRecoverC: the reference's own outputs (correcting and failing regimes) through the GPU path"""
    L = O.LIMBS[f]
    for c in GOLD["fields"][name]["recover_c"]:
        n, N = c["n"], len(c["status"])
        shares = O.from_ints(ints(c["shares"]), L).reshape(N, n, L)
        al = O.from_ints(ints(c["alphas"]), L) if "alphas" in c else None
        r = scl.shamir_recover_correct(f, dev(scl, soa(shares)), al)
        t = c["t"]
        assert O.to_ints(host(scl, r["f"]).transpose(1, 0, 2)) == ints(c["f"])
        assert O.to_ints(host(scl, r["err"]).transpose(1, 0, 2)) == ints(c["err"])
        assert r["status"].cpu().numpy().tolist() == c["status"]
        assert r["nerr"].cpu().numpy().tolist() == c["nerr"]


@pytest.mark.parametrize("f", ALL_FIELDS)
@pytest.mark.parametrize("n,t,N", [(4, 1, 300), (10, 3, 1000), (13, 4, 200), (12, 3, 77), (1, 0, 5), (3, 0, 9), (40, 13, 40), (64, 21, 6)])
def test_recover_correct_vs_oracle(scl, port, f, n, t, N):
    if f in SLOW_ORACLE and n >= 40:
        N = min(N, 6 if n == 40 else 2)
    L = O.LIMBS[f]
    gold64 = None
    if L == 4 and n > 40:
        # the oracle port takes minutes at n = 64 over the 256-bit fields: the expected outputs of these two cases were
        # produced once by the reference itself (tests/golden/make_recover_c_n64.py -> golden_recover_c_n64.json)
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_recover_c_n64.json")) as fh:
            gold64 = next(c for c in json.load(fh)["cases"]
                          if c["field"] == {O.SECP256K1_SCALAR: "secp256k1_order", O.SECP256K1_FIELD: "secp256k1_field"}[f])
        assert (gold64["n"], gold64["t"], gold64["N"]) == (n, t, N)
    rng = np.random.default_rng(n * 100 + t)
    secrets = rand_elems(port, f, N, b"bw-s")
    coeffs = rand_elems(port, f, max(t, 1) * N, b"bw-c").reshape(N, max(t, 1), L)[:, :t]
    nodes = O.from_ints(list(range(1, n + 1)), L) if f == O.GF2_128 else np.stack([port.from_int(f, i + 1) for i in range(n)])
    shares = np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)])
    junk = rand_elems(port, f, N * n, b"bw-j").reshape(N, n, L)
    nbad = np.zeros(N, dtype=int)
    for s in range(N):
        k = 0 if s % 3 == 0 else int(rng.integers(0, t + 3))     # a third clean, the rest 0 .. t+2 corrupted shares
        nbad[s] = k
        for i in rng.choice(n, size=min(k, n), replace=False):
            shares[s, i] = junk[s, i]
    if gold64 is not None:
        import hashlib
        assert hashlib.sha256(shares.tobytes()).hexdigest() == gold64["shares_sha"], "inputs drifted from the generator's"
        r = scl.shamir_recover_correct(f, dev(scl, soa(shares)), None)
        assert O.to_ints(host(scl, r["f"]).transpose(1, 0, 2)) == [int(v, 16) for v in gold64["f"]]
        assert O.to_ints(host(scl, r["err"]).transpose(1, 0, 2)) == [int(v, 16) for v in gold64["err"]]
        assert r["status"].cpu().numpy().tolist() == gold64["status"] and r["nerr"].cpu().numpy().tolist() == gold64["nerr"]
        assert nbad.tolist() == gold64["nbad"]
        return
    r, (fo, eo, st, ne) = _check_recover_c(scl, port, f, shares, nodes if f == O.GF2_128 else None)
    used = 3 * t + 1
    for s in range(N):
        if nbad[s] <= t and n == used:      # within the correction radius: the secret and the number of errors come back
            assert st[s] == 0 and ne[s] == nbad[s] and np.array_equal(fo[s, 0], secrets[s])
    assert r["queued"] <= int((nbad > 0).sum())


def test_recover_correct_wikipedia_gf7(scl, port):
    """test/scl/ss/test_shamir.cc:144-160: the Berlekamp-Welch example over GF(7), here through the plug-in
    Montgomery field with p = 7"""
    f, L = O.MONT128, 2
    scl.set_mont128_prime(7)
    port.mont128_set_prime(7)
    try:
        I = lambda v: port.from_int(f, v)
        bs = np.stack([I(v) for v in (1, 5, 3, 6, 3, 2, 2)]).reshape(1, 7, L)
        corrected = [1, 6, 3, 6, 1, 2, 2]
        r, (fo, eo, st, ne) = _check_recover_c(scl, port, f, bs)
        assert st[0] == 0 and ne[0] == 2
        for root in (2, 5):
            assert O.to_ints(port.poly_eval(f, eo[0], I(root).reshape(1, L))) == [0]
        got = port.poly_eval(f, fo[0], np.stack([I(i + 1) for i in range(7)]))
        assert np.array_equal(got, np.stack([I(v) for v in corrected]))
    finally:
        scl.set_mont128_prime(2 ** 128 - 159)
        port.mont128_set_prime(2 ** 128 - 159)


def test_recover_correct_edges(scl, port):
    f, L = O.M61, 1
    # explicit nodes 42..48 (test_shamir.cc:81-109 nodes), all-zero shares, every share of one secret corrupted
    al = np.stack([port.from_int(f, v) for v in range(42, 49)])
    sec = rand_elems(port, f, 50, b"bwe")
    co = rand_elems(port, f, 100, b"bwc").reshape(50, 2, L)
    sh = np.stack([port.poly_eval(f, np.concatenate([sec[s:s + 1], co[s]]), al) for s in range(50)])
    sh[1, 2] = port.from_int(f, 7)
    sh[2, 0] = sh[2, 6] = port.from_int(f, 9)
    sh[3] = 0
    sh[4] = rand_elems(port, f, 7, b"bwx")
    _check_recover_c(scl, port, f, sh, al)
    with pytest.raises(scl.SclError):      # duplicate nodes: "0 not invertible modulo prime" like computeLagrangeBasis
        scl.shamir_recover_correct(f, dev(scl, soa(sh)), np.stack([al[0]] * 7))
    with pytest.raises(scl.SclError):      # rings have no Berlekamp-Welch
        scl.shamir_recover_correct(scl.Z2K(62), dev(scl, soa(sh)))


@pytest.mark.parametrize("f,n,N", [(O.M61, 67, 9), (O.M61, 100, 8), (O.M61, 148, 5), (O.M127, 70, 6), (O.M127, 97, 4),
                                   (O.MONT128, 130, 3), (O.SECP256K1_SCALAR, 67, 3)])
def test_recover_correct_has_no_bound_on_shares(scl, port, f, n, N):
    """shamirRecoverC takes any number of shares (shamir.h:202-259).  Beyond 64 the systems of a secret are solved by a
    workgroup with threads over rows, in LDS while they fit (M61: 136 shares, 128-bit fields: 95, 256-bit: 66) and in a
    device-memory slice per workgroup after that -- every case against the oracle, output for output."""
    L = O.LIMBS[f]
    t = (n - 1) // 3
    rng = np.random.default_rng(n)
    secrets = rand_elems(port, f, N, b"bwL-s")
    coeffs = rand_elems(port, f, t * N, b"bwL-c").reshape(N, t, L)
    nodes = np.stack([port.from_int(f, i + 1) for i in range(n)])
    shares = np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)])
    junk = rand_elems(port, f, N * n, b"bwL-j").reshape(N, n, L)
    nbad = [0, 1, t, t + 1, t // 2, 2, 3, t - 1, 5][:N]      # clean, inside the radius, at it, beyond it
    for s in range(N):
        for i in rng.choice(n, size=nbad[s], replace=False):
            shares[s, i] = junk[s, i]
    r, (fo, eo, st, ne) = _check_recover_c(scl, port, f, shares)
    used = 3 * t + 1
    for s in range(N):
        if nbad[s] <= t and n == used:
            assert st[s] == 0 and ne[s] == nbad[s] and np.array_equal(fo[s, 0], secrets[s])
    assert r["queued"] == sum(1 for k in nbad if k > 0)


# ------------------------------------------------------------------------------------------------------- rings Z2k<K>
RING_BITS_REF = sorted(GOLD["rings"], key=lambda k: GOLD["rings"][k]["bits"])


@pytest.mark.parametrize("name", RING_BITS_REF)
def test_ring_against_reference_golden(scl, name):
    """Z2k<K> on the GPU against vectors emitted by the reference (include/scl/math/z2k.h, test_z2k.cc)"""
    g = GOLD["rings"][name]
    f, L = scl.Z2K(g["bits"]), g["limbs"]
    assert scl.limbs(f) == L and scl.byte_size(f) == g["byte_size"]
    A = lambda hexes: O.from_ints(ints(hexes), L)
    eq = lambda t, hexes: O.to_ints(host(scl, t)) == ints(hexes)
    raw = torch.frombuffer(bytearray(bytes.fromhex(g["from_bytes"]["bytes"])), dtype=torch.uint8).cuda()
    assert eq(scl.from_bytes(f, raw), g["from_bytes"]["out"])
    a, b = dev(scl, A(g["ew"]["a"])), dev(scl, A(g["ew"]["b"]))
    for nm, op in (("add", O.ADD), ("sub", O.SUB), ("mul", O.MUL)):
        assert eq(scl.ew(f, op, a, b), g["ew"][nm]), nm
    assert eq(scl.ew(f, O.NEG, a), g["ew"]["neg"])
    odd = dev(scl, A(g["inverse"]["in"]))
    assert eq(scl.ew(f, O.INV, odd), g["inverse"]["out"])
    assert eq(scl.ew(f, O.DIV, b, odd), g["inverse"]["div_b_by_in"])
    with pytest.raises(scl.SclError) as ei:
        scl.ew(f, O.INV, dev(scl, O.from_ints([2], L)))
    assert ei.value.reference_message == g["inverse_even_error"] == "value not invertible modulo 2^K"
    for c in g["vector_random"]:
        assert eq(scl.vector_random(f, c["n"], bytes.fromhex(c["seed"])), c["out"])
    for c in g["additive"]:
        n = c["n"]
        sec = A(c["secrets"])
        sh = scl.additive_share_prg(f, dev(scl, sec), n, bytes.fromhex(c["seed"]))
        want = soa(A(c["shares"]).reshape(len(sec), n, L))
        assert np.array_equal(host(scl, sh), want)
        assert eq(scl.additive_recover(f, sh), c["sum"])
    assert O.to_ints(scl.dot(f, a, b)) == ints([g["dot"]["out"]])
    assert O.to_ints(scl.vsum(f, a)) == ints([g["sum"]["out"]])
    assert eq(scl.scalar_mul(f, a, A([g["scalar_mul"]["scalar"]])[0]), g["scalar_mul"]["out"])
    for c in g["matmul"]:
        Cm = scl.matmul(f, dev(scl, A(c["A"]).reshape(c["n"], c["k"], L)), dev(scl, A(c["B"]).reshape(c["k"], c["m"], L)))
        assert eq(Cm, c["C"])


@pytest.mark.parametrize("K", [1, 2, 7, 8, 9, 31, 32, 33, 62, 63, 64, 65, 66, 96, 123, 127, 128])
def test_ring_vs_oracle(scl, port, K):
    """every ring width against the CPU restatement at sizes that exercise the vector paths and ragged tails"""
    f = scl.Z2K(K)
    L, bs = O.LIMBS[f], O.byte_size(f)
    for n in (1, 2, 255, 4097):
        raw = port.prg(b"ring-%d" % K, [2 * n * bs])
        a, b = port.from_bytes(f, raw[: n * bs]), port.from_bytes(f, raw[n * bs:])
        rt = torch.frombuffer(bytearray(raw[: n * bs]), dtype=torch.uint8).cuda()
        assert np.array_equal(host(scl, scl.from_bytes(f, rt)), a)
        da, db = dev(scl, a), dev(scl, b)
        for op in (O.ADD, O.SUB, O.MUL):
            assert np.array_equal(host(scl, scl.ew(f, op, da, db)), port.ew(f, op, a, b)), (K, n, op)
        assert np.array_equal(host(scl, scl.ew(f, O.NEG, da)), port.ew(f, O.NEG, a))
        odd = a.copy()
        odd[:, 0] |= np.uint64(1)
        assert np.array_equal(host(scl, scl.ew(f, O.INV, dev(scl, odd))), port.ew(f, O.INV, odd))
        assert np.array_equal(host(scl, scl.ew(f, O.DIV, db, dev(scl, odd))), port.ew(f, O.DIV, b, odd))
        assert np.array_equal(scl.dot(f, da, db), port.dot(f, a, b))
        assert np.array_equal(scl.vsum(f, da), port.sum(f, a))
        assert np.array_equal(host(scl, scl.vector_random(f, n, b"vr", counter0=3)),
                              port.from_bytes(f, port.prg_blocks(b"vr", 3, (n * bs + 15) // 16)[: n * bs]))
        assert scl.equals(f, da, da) and (n < 2 or K < 8 or not scl.equals(f, da, db))
        # equality is modulo 2^K: garbage above bit K does not matter (z2k_ops.h:97-103)
        if K not in (64, 128):
            noisy = a.copy()
            noisy[:, K // 64] |= np.uint64(1) << np.uint64(K % 64)   # bit K: outside the ring
            assert scl.equals(f, da, dev(scl, noisy))
            assert np.array_equal(host(scl, scl.ew(f, O.ADD, dev(scl, noisy), db)), port.ew(f, O.ADD, a, b))
    for n, N in ((1, 5), (3, 1000), (10, 333)):
        sec = port.from_bytes(f, port.prg(b"rsec", [N * bs]))
        want = port.additive_share(f, b"radd", sec, n)
        got = scl.additive_share_prg(f, dev(scl, sec), n, b"radd")
        assert np.array_equal(host(scl, got), soa(want)), (K, n, N)
        assert np.array_equal(host(scl, scl.additive_recover(f, got)), sec)
    A_, B_ = port.from_bytes(f, port.prg(b"rA", [5 * 9 * bs])).reshape(5, 9, L), port.from_bytes(f, port.prg(b"rB", [9 * 70 * bs])).reshape(9, 70, L)
    assert np.array_equal(host(scl, scl.matmul(f, dev(scl, A_), dev(scl, B_))), port.matmul(f, A_, B_))
    # the field-only entry points refuse ring tags
    with pytest.raises(scl.SclError):
        scl.lagrange_basis(f, 3)
    with pytest.raises(scl.SclError):
        scl.wire_pack(f, da)


def test_open_step_on_one_rank_rccl(scl, port):
    """scl_amd.dist with the HIP kernels and a one-rank RCCL group: the all-gather open and the Mersenne61
    partial-sum open (reduce-scatter + one fold) both give back the secrets"""
    import socket
    import torch.distributed as dist
    from scl_amd import dist as sd
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        prt = sck.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(prt), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        f, n, t, N = O.M61, 10, 3, 4096
        secrets = rand_elems(port, f, N, b"open1")
        secrets[0] = port.from_int(f, -1)
        shares = scl.shamir_share_prg(f, dev(scl, secrets), t, n, b"open1-seed")
        lam = scl.lagrange_basis(f, n)
        out = sd.open_and_reconstruct(f, shares, n, lam, chunk=1000)
        assert np.array_equal(host(scl, out), secrets)
        mine = sd.open_by_partial_sums(shares, lam)
        assert np.array_equal(host(scl, mine), secrets)
        # the any-field partial-sum open (partial sums by the reconstruct kernel, all-gather of one element per secret and
        # rank, Vector::sum per secret), C4's field and shape, with a ragged last chunk
        f2, n2, t2, N2 = O.GF2_128, 40, 13, 3001
        sec2 = scl.vector_random(f2, N2, b"open1-gf")
        sh2 = scl.shamir_share(f2, sec2, scl.vector_random(f2, t2 * N2, b"open1-gfc").reshape(t2, N2, -1), n2)
        lam2 = scl.lagrange_basis(f2, n2)
        out2 = sd.open_by_partial_gather(f2, sh2, lam2, chunk=1024)
        assert scl.equals(f2, out2, sec2) and scl.equals(f2, out2, sd.open_and_reconstruct(f2, sh2, n2, lam2, chunk=1024))
    finally:
        dist.destroy_process_group()


def test_six_host_threads_on_their_own_streams(scl, port):
    """include/scl_hip.h, Conventions: "callable concurrently from several host threads on different streams"; the only mutable
    state is per thread (tuning knobs, the Mont128 modulus, scratch and temporary arenas, the last error) or behind a mutex
    (device table caches).  Since round 4 that state lives in ONE translation unit and is `extern thread_local` in the six
    others -- this is its test: six threads, each on its own torch stream with its own field, shape, knobs and (for Mont128)
    modulus, run share -> reconstruct -> detect / sum / dot loops at the same time; every result must equal what the same
    calls give one after the other on the main thread, which in turn is checked against the oracle on a window.  Two of the
    threads share (128, 42) Mersenne61 on the matrix cores at once (one cached table, pinned by both), two set different
    Mont128 primes, one provokes an error and must read ITS text from scl_hip_last_error."""
    import threading
    jobs = [
        dict(f=O.M61, n=10, t=3, N=200_001, knobs={}),
        dict(f=O.M61, n=128, t=42, N=4_097, knobs={"stream_block": 256}),
        dict(f=O.M61, n=128, t=42, N=3_001, knobs={"nontemporal": 0}),
        dict(f=O.MONT128, n=10, t=3, N=50_001, knobs={}, prime=2 ** 128 - 159),
        dict(f=O.MONT128, n=7, t=2, N=40_001, knobs={"share_waves128": 0}, prime=2 ** 127 - 1),   # (below 2^127: the Vandermonde-row share kernel instead of the small-node one)
        dict(f=O.GF2_128, n=40, t=13, N=9_001, knobs={"gf_tiles": 0}),
    ]

    def run(job, seed):
        f, n, t, N = job["f"], job["n"], job["t"], job["N"]
        for k, v in job["knobs"].items():
            scl.set_tuning(k, v)
        if "prime" in job:
            scl.set_mont128_prime(job["prime"])
        out = []
        for rep in range(3):
            secrets = scl.vector_random(f, N, seed + b"-s%d" % rep)
            coeffs = scl.vector_random(f, t * N, seed + b"-c%d" % rep).reshape(t, N, -1)
            shares = scl.shamir_share(f, secrets, coeffs, n)
            rec = scl.shamir_recover(f, shares)
            assert scl.equals(f, rec, secrets)
            out.append((scl.to_host(shares[n - 1, :64]).copy(), scl.vsum(f, rec).copy(), scl.dot(f, shares[0], shares[1]).copy()))
        if f == O.M61 and n == 10:   # an error on this thread: its own text, whatever the others are doing
            with pytest.raises(scl.SclError) as ei:
                scl.ew(f, scl.INV, scl.to_device(np.zeros((4, 1), dtype=np.uint64)))
            assert "0 not invertible modulo prime" in str(ei.value)
            out.append(scl.lib.scl_hip_last_error())
        for k in job["knobs"]:
            scl.set_tuning(k, {"stream_block": 64, "nontemporal": 1, "share_waves128": 12, "gf_tiles": 1}[k])
        return out

    want = [run(job, b"thr%d" % i) for i, job in enumerate(jobs)]      # one after the other, main thread
    scl.set_mont128_prime(2 ** 128 - 159)
    # oracle window on the first job (the others' kernels have their own oracle tests)
    sec = port.vector_random(O.M61, b"thr0-s0", 64)
    co = port.vector_random(O.M61, b"thr0-c0", 3 * 200_001).reshape(3, 200_001, 1)[:, :64]
    assert np.array_equal(want[0][0][0], soa(port.shamir_share_coeffs(O.M61, sec, np.ascontiguousarray(np.transpose(co, (1, 0, 2))), 10))[9])
    got, errs = [None] * len(jobs), []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                got[i] = run(jobs[i], b"thr%d" % i)
                torch.cuda.current_stream().synchronize()
            scl.lib.scl_hip_thread_cleanup()
        except BaseException as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    for _ in range(2):
        ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))]
        for th in ths:
            th.start()
        for th in ths:
            th.join(timeout=300)
        assert not errs and not any(th.is_alive() for th in ths), errs
        for i in range(len(jobs)):
            for a, b in zip(got[i], want[i]):
                if isinstance(a, tuple):
                    assert all(np.array_equal(x, y) for x, y in zip(a, b)), i
                else:
                    assert a == b, i


def test_batch_calls_capture_into_a_hip_graph(scl, port):
    """The batch entry points that take their tables as kernel arguments neither synchronise nor allocate nor copy from pageable
    memory once their per-thread arenas exist, so a caller can capture them on its stream (hipStreamBeginCapture; here through
    torch.cuda.CUDAGraph, which captures the stream the library launches on) and replay the step: (10,3) Mersenne61 share +
    reconstruct and PRG-driven additive share + sum, replayed on NEW secrets written into the captured input buffer, equal the
    oracle.  (What replay buys is small: ~12 us against ~20 us per two-kernel step, profiles/r4_probe_graph.txt -- the floor is the
    device's own dispatch.)"""
    f, N = O.M61, 4097
    secrets = dev(scl, rand_elems(port, f, N, b"graph-s0"))
    coeffs = dev(scl, np.ascontiguousarray(np.transpose(rand_elems(port, f, 3 * N, b"graph-c").reshape(N, 3, 1), (1, 0, 2))))
    sh10, out, sh3, out3 = scl.empty(f, 10, N), scl.empty(f, N), scl.empty(f, 3, N), scl.empty(f, N)
    lam = scl.lagrange_basis(f, 10)

    def step():
        scl.shamir_share(f, secrets, coeffs, 10, out=sh10)
        scl.shamir_recover(f, sh10, lam, out=out)
        scl.additive_share_prg(f, secrets, 3, b"graph-seed", out=sh3)
        scl.additive_recover(f, sh3, out=out3)

    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.synchronize()
    for rep in range(3):
        fresh = rand_elems(port, f, N, b"graph-s%d" % (rep + 1))
        secrets.copy_(dev(scl, fresh))
        out.zero_()
        out3.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(host(scl, out), fresh) and np.array_equal(host(scl, out3), fresh)
        co = host(scl, coeffs)
        want = soa(port.shamir_share_coeffs(f, fresh, np.ascontiguousarray(np.transpose(co, (1, 0, 2))), 10))
        assert np.array_equal(host(scl, sh10), want)
        assert np.array_equal(host(scl, sh3), soa(port.additive_share(f, b"graph-seed", fresh, 3)))


@pytest.mark.parametrize("f", ALL_FIELDS + [scl_ring for scl_ring in (0x100 + 64, 0x100 + 128)])
def test_ew_status_leaves_the_zero_flag_on_the_device(scl, port, f):
    """scl_hip_ew_status: FF::inverse / operator/ (ff.h:203-246; the throw of small_ff.h:61-70, ff_ops_gmp.h:250-260,
    z2k_ops.h:81-83) without a host synchronisation -- same values as scl_hip_ew and the oracle, the "not invertible" report a
    device word the call ORs into and never clears, the offending slots 0 and every other slot computed."""
    ring = f > 0x100
    L = scl.limbs(f)
    n = 3001 if (ring or f in SLOW_ORACLE) else 20011
    if ring:
        rng = np.random.default_rng(f)
        a = rng.integers(0, 2**63, size=(n, L), dtype=np.uint64) * np.uint64(2) + np.uint64(1)
        a[:, 1:] = rng.integers(0, 2**63, size=(n, L - 1), dtype=np.uint64)
        b = a[::-1].copy()
        bad_value = np.zeros(L, np.uint64)
        bad_value[0] = 6                        # even: not invertible modulo 2^K
    else:
        a, b = rand_elems(port, f, n, b"ews-a"), rand_elems(port, f, n, b"ews-b")
        zero = port.from_int(f, 0)
        for x, sub in ((a, 7), (b, 9)):
            x[np.all(x == zero, axis=1)] = port.from_int(f, sub)
        bad_value = zero
    da, db = dev(scl, a), dev(scl, b)
    status = scl.ew_status_buffer()
    want_inv = host(scl, scl.ew(f, O.INV, da))            # (the synchronous call is checked against the oracle elsewhere)
    want_div = host(scl, scl.ew(f, O.DIV, db, da))
    if not ring:
        w = slice(0, 200)
        assert np.array_equal(want_inv[w], port.ew(f, O.INV, a[w])) and np.array_equal(want_div[w], port.ew(f, O.DIV, b[w], a[w]))
    assert np.array_equal(host(scl, scl.ew_status(f, O.INV, da, None, status)), want_inv)
    assert np.array_equal(host(scl, scl.ew_status(f, O.DIV, db, da, status)), want_div)
    assert np.array_equal(host(scl, scl.ew_status(f, O.ADD, da, db, None)), host(scl, scl.ew(f, O.ADD, da, db)))   # ops that cannot fail: no word needed
    assert int(status.item()) == 0
    spots = np.array(sorted({0, 1, 63, 64, n // 2, n - 2, n - 1}))
    z = a.copy()
    z[spots] = bad_value
    out = scl.ew_status(f, O.INV, dev(scl, z), None, status)
    assert int(status.item()) == 1
    got, keep = host(scl, out), np.ones(n, bool)
    keep[spots] = False
    assert np.array_equal(got[keep], want_inv[keep]) and not got[spots].any()
    # never cleared by the call: a clean batch afterwards leaves the word raised; the caller clears it
    scl.ew_status(f, O.INV, da, None, status)
    assert int(status.item()) == 1
    status.zero_()
    got = host(scl, scl.ew_status(f, O.DIV, db, dev(scl, z), status))
    assert int(status.item()) == 1 and np.array_equal(got[keep], want_div[keep]) and not got[spots].any()
    with pytest.raises(scl.SclError) as ei:
        scl.ew_status(f, O.INV, da, None, None)           # INV / DIV report through the word: it must be there
    assert ei.value.status == scl.ERR_BAD_ARG


def test_inverse_and_divide_capture_into_a_hip_graph(scl, port):
    """The asynchronous inverse / divide (scl_hip_ew_status) neither synchronises nor copies: a sequence of small batches is
    captured on a stream (through torch.cuda.CUDAGraph) together with the clearing of its status word, replayed on new operands,
    and the flag is read once after the replay -- clean operands leave it 0, a planted zero raises it.  The synchronous
    scl_hip_ew cannot be captured: it waits for the stream to report the reference's error."""
    for f, N in ((O.M61, 4097), (O.M127, 1025), (O.MONT128, 513), (O.GF2_128, 769), (O.SECP256K1_SCALAR, 257)):
        a, b = scl.vector_random(f, N, b"graph-inv-a"), scl.vector_random(f, N, b"graph-inv-b")
        inv, quo, status = scl.empty(f, N), scl.empty(f, N), scl.ew_status_buffer()

        def step():
            status.zero_()
            scl.ew_status(f, O.INV, a, None, status, out=inv)
            scl.ew_status(f, O.DIV, b, a, status, out=quo)

        g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(s):
            step()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                step()
        torch.cuda.synchronize()
        for rep in range(3):
            fresh = scl.vector_random(f, N, b"graph-inv-%d" % rep)
            if rep == 2:
                fresh[N // 3] = 0                       # a zero operand in the last replay
            a.copy_(fresh)
            inv.zero_()
            quo.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert int(status.item()) == (1 if rep == 2 else 0)
            keep = np.ones(N, bool)
            if rep == 2:
                keep[N // 3] = False
            w = np.flatnonzero(keep)[:128 if f in (O.M61, O.M127) else 24]
            ha, hb = host(scl, a), host(scl, b)
            assert np.array_equal(host(scl, inv)[w], port.ew(f, O.INV, ha[w]))
            assert np.array_equal(host(scl, quo)[w], port.ew(f, O.DIV, hb[w], ha[w]))
            mask = torch.from_numpy(keep).cuda()
            ones = scl.to_device(np.broadcast_to(port.from_int(f, 1), (int(keep.sum()), scl.limbs(f))).copy())
            assert bool(scl.equals(f, scl.ew(f, O.MUL, inv, a)[mask].contiguous(), ones))


def test_reference_binding_compiled_against_the_reference():
    """integration/include/scl/hip/binding.h -- the header INTEGRATION.md section 2 tells a maintainer of the reference to add --
    compiled against the REAL reference (/root/reference/include + its translation units, oracle/Makefile `binding`; the
    static_asserts on sizeof / alignment / standard layout of FF<Mersenne61 | Mersenne127 | Secp256k1Scalar | Secp256k1Field>
    held at that build) and run here: reference Vector<FF> -> C ABI -> reference Vector<FF> equals the reference's own
    multiplyEntryWise, N sequential shamirSecretShare calls on one PRG, and shamirRecoverP.  The binary is built where the
    reference is and travels like oracle/_ref/libscl_ref.so.  Round 5: the same binary also sends every element's FF::inverse and a Matrix::multiply
    with an inner dimension of 7000 through the binding, against the reference's own element-by-element results."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "binding_check")
    assert os.path.exists(exe), "oracle/_ref/binding_check is built by `make -C oracle binding` where /root/reference exists"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("[ ok ]") == 4 and "0 failure(s)" in r.stdout


@pytest.mark.parametrize("world,f,n,t,N,chunk", [
    (2, O.M61, 10, 3, 5001, 1000),      # BASELINE configs[1]'s shape: 6 chunks, the last one a single (odd) secret
    (4, O.M61, 7, 2, 3001, 700),        # 7 parties on 4 ranks: two rows per rank, one padding row on the last
    (4, O.M61, 5, 2, 2001, 512),        # 5 parties on 4 ranks: a rank with a padding row AND a rank without parties
    (8, O.M61, 10, 3, 2501, 600),       # ranks 5..7 hold nothing
    (9, O.M61, 10, 3, 1801, 600),       # nine ranks: the gathers work, the reduce-scatter form refuses (a 64-bit sum of nine could wrap)
    (3, O.M127, 10, 3, 1001, 300),      # 16-byte elements, a world that does not divide n
    (2, O.GF2_128, 40, 13, 1201, 256),  # BASELINE configs[3]'s shape (the position-table reconstruct kernel)
    (8, O.GF2_128, 40, 13, 1001, 300),  # .. on the world BASELINE quotes it on: five parties per rank
])
def test_c_abi_open_with_a_world_of_threads(scl, world, f, n, t, N, chunk):
    """scl_hip_open_all_gather / scl_hip_open_partial_gather (and, over Mersenne61, scl_hip_open_reduce_scatter in both its
    forms) with world = 2, 3, 4, 8 on this one GPU: the ranks are host threads
    of a child process and the library binds tests/cxx/_build/libfake_rccl.so (SCL_HIP_RCCL_LIBRARY), whose all-gather is a
    rendezvous + device-to-device copies.  The permuted lambda (row q = j * world + r), the grouped per-row gathers, the two
    streams and their events across >= 3 chunks and across calls from different streams, padding rows (sent as zeros, filled
    with ones here), ranks without parties and the partial-sum form all run as they would on eight GPUs; every rank's four
    outputs must equal the secrets the oracle shared (tests/open_world_check.py)."""
    import subprocess
    import sys
    fake = os.path.join(ROOT, "tests", "cxx", "_build", "libfake_rccl.so")
    assert os.path.exists(fake), "tests/cxx/_build/libfake_rccl.so is built by make -C tests/cxx (__graft_entry__.build)"
    env = dict(os.environ, SCL_HIP_RCCL_LIBRARY=fake)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "open_world_check.py")] + [str(x) for x in (world, f, n, t, N, chunk)],
                       env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-4000:]
    rep = json.loads(lines[-1])
    assert r.returncode == 0 and rep["ok"], (rep, r.stderr[-2000:])
    assert rep["chunks"] >= 3 and rep["padding_rows"] == -(-n // world) * world - n
    assert rep["reduce_scatter"] == (f == O.M61 and world <= 8)


@pytest.mark.parametrize("n,counter0", [(1003, 0), (1004, 5), (1, 0), (2, 7)])
def test_vector_random_into_an_8_byte_aligned_window_through_the_raw_abi(scl, port, n, counter0):
    """scl_hip_vector_random with a Mersenne61 destination that starts 8 bytes into a 16-byte line (a window of a larger
    vector): the kernel then stores element by element instead of in 16-byte pairs; the words either side stay untouched"""
    import ctypes as C
    f = O.M61
    buf = torch.full((n + 3,), -1, dtype=torch.int64, device="cuda")
    assert buf.data_ptr() % 16 == 0
    seed = b"vr-window"
    st = scl.lib.scl_hip_vector_random(f, C.c_void_p(buf.data_ptr() + 8), C.c_size_t(n), seed, C.c_size_t(len(seed)),
                                       C.c_uint64(counter0), None)
    assert st == 0, scl.lib.scl_hip_last_error()
    torch.cuda.synchronize()
    # Vector::random (vector.h:507-519): one draw of ceil(8 n / 16) blocks from `counter0`, FF::read per 8 bytes
    want = port.from_bytes(f, port.prg_blocks(seed, counter0, (8 * n + 15) // 16))[:n]
    got = buf.cpu().numpy().view(np.uint64)
    assert np.array_equal(got[1:1 + n], want[:, 0])
    assert got[0] == np.uint64(2 ** 64 - 1) and (got[1 + n:] == np.uint64(2 ** 64 - 1)).all()
    # and the aligned form of the same draw agrees
    assert np.array_equal(host(scl, scl.vector_random(f, n, seed, counter0=counter0))[:, 0], want[:, 0])


def test_open_step_behind_the_c_abi_one_rank_rccl(scl, port):
    """scl_hip_comm_* / scl_hip_open_all_gather / scl_hip_open_partial_gather (RCCL called from the library, no torch
    process group): a one-rank communicator holds every party; chunking with a ragged, odd last chunk, the two streams and
    their events, the permuted lambda and the partial-sum form are the code that runs on eight ranks."""
    from scl_amd import dist as sd
    comm = sd.Communicator()
    scl.set_tuning("open_gather_always", 1)     # one rank would otherwise reconstruct straight from its slab
    try:
        assert (comm.world, comm.rank) == (1, 0)
        f, n, t, N = O.M61, 10, 3, 5001
        secrets = rand_elems(port, f, N, b"copen")
        secrets[0] = port.from_int(f, -1)
        shares = scl.shamir_share_prg(f, dev(scl, secrets), t, n, b"copen-seed")
        lam = scl.lagrange_basis(f, n)
        for chunk in (2048, 1 << 24):
            assert np.array_equal(host(scl, sd.open_all_gather_c(comm, f, shares, n, lam, chunk=chunk)), secrets)
            assert np.array_equal(host(scl, sd.open_partial_gather_c(comm, f, shares, lam, chunk=chunk)), secrets)
            # the Mersenne61 reduce-scatter form through the real ncclReduceScatter / in-place ncclAllGather (one rank)
            assert np.array_equal(host(scl, sd.open_reduce_scatter_c(comm, f, shares, lam, chunk=chunk)), secrets)
            assert np.array_equal(host(scl, sd.open_reduce_scatter_c(comm, f, shares, lam, chunk=chunk, all_ranks=False)), secrets)
        f2, n2, t2, N2 = O.GF2_128, 40, 13, 6001
        sec2 = scl.vector_random(f2, N2, b"copen-gf")
        sh2 = scl.shamir_share(f2, sec2, scl.vector_random(f2, t2 * N2, b"copen-gfc").reshape(t2, N2, -1), n2)
        lam2 = scl.lagrange_basis(f2, n2)
        out2 = sd.open_all_gather_c(comm, f2, sh2, n2, lam2, chunk=2500)
        assert scl.equals(f2, out2, sec2)
        assert scl.equals(f2, sd.open_partial_gather_c(comm, f2, sh2, lam2, chunk=2500), sec2)
        # a rank without parties contributes zeros to the partial-sum form
        none = sh2[:0]
        z = sd.open_partial_gather_c(comm, f2, none, lam2[:0], chunk=2500)
        assert not torch.any(z).item()
        with pytest.raises(ValueError):
            sd.open_all_gather_c(comm, f2, sh2[:5], n2, lam2)
        with pytest.raises(scl.SclError):     # no 128-bit RCCL type: the reduce-scatter form is Mersenne61's alone
            sd.open_reduce_scatter_c(comm, f2, sh2, lam2)
        scl.set_tuning("open_gather_always", 0)  # and the one-rank shortcut gives the same secrets
        assert scl.equals(f2, sd.open_all_gather_c(comm, f2, sh2, n2, lam2, chunk=2500), sec2)
    finally:
        scl.set_tuning("open_gather_always", 0)
        comm.close()


@pytest.mark.parametrize("n,t,N,off,pad", [(128, 42, 1001, 1, 3), (100, 40, 777, 3, 0), (128, 63, 96, 0, 1)])
def test_matrix_core_share_on_unaligned_rows_through_the_raw_abi(scl, port, n, t, N, off, pad):
    """k_share_mfma_m61_p16 (n > 96, 32 <= t <= 63) stores one secret per lane, whole lines when the rows start on one; rows
    that start 8 or 24 bytes into a line, odd row strides and ragged party / secret counts go through the same kernel: the C
    ABI called directly with offset pointers, the whole share matrix against the oracle, the gaps between rows untouched."""
    import ctypes as C
    f = O.M61
    stride, cstride = N + pad, N + 5
    secrets = rand_elems(port, f, N, b"p16-s")
    coeffs = rand_elems(port, f, t * N, b"p16-c").reshape(N, t, 1)
    want = soa(port.shamir_share_coeffs(f, secrets, coeffs, n))            # [n][N][1]
    sec_buf = torch.zeros(N + off, dtype=torch.int64, device="cuda")
    sec_buf[off:] = torch.from_numpy(secrets.view(np.int64).reshape(N)).cuda()
    co_buf = torch.zeros(t * cstride + off, dtype=torch.int64, device="cuda")
    co_buf[off:].view(t, cstride)[:, :N] = torch.from_numpy(
        np.ascontiguousarray(np.transpose(coeffs, (1, 0, 2))).view(np.int64).reshape(t, N)).cuda()
    sh_buf = torch.full((n * stride + off + 64,), -1, dtype=torch.int64, device="cuda")
    p = lambda tns: C.c_void_p(tns.data_ptr() + 8 * off)
    scl.set_tuning("mfma", 1)
    try:
        st = scl.lib.scl_hip_shamir_share(f, p(sh_buf), C.c_size_t(stride), p(sec_buf), p(co_buf), C.c_size_t(cstride),
                                          C.c_size_t(N), C.c_size_t(t), C.c_size_t(n), None, None)
        assert st == 0, scl.lib.scl_hip_last_error()
    finally:
        scl.set_tuning("mfma", 0)
    torch.cuda.synchronize()
    flat = sh_buf.cpu().numpy().view(np.uint64)
    got = flat[off:off + n * stride].reshape(n, stride)
    assert np.array_equal(got[:, :N], want[:, :, 0])
    untouched = np.uint64(2 ** 64 - 1)
    assert (got[:, N:] == untouched).all() and (flat[:off] == untouched).all() and (flat[off + n * stride:] == untouched).all()


@pytest.mark.parametrize("t,mode", [(3, {}), (9, {}), (9, {"force_table": 1}), (20, {"mfma": 1}), (3, {"force_scalar": 1})])
def test_odd_strides_and_8_byte_alignment_through_the_raw_abi(scl, port, t, mode):
    """Mersenne61 rows that are only 8-byte aligned and row strides that are odd (so the 16-byte packs cannot be
    used, or only for every other row): the C ABI called directly with offset pointers, against the oracle"""
    import ctypes as C
    f, n, N = O.M61, 10, 1001
    stride, cstride = N + 3, N + 5
    secrets = rand_elems(port, f, N, b"st-s")
    coeffs = rand_elems(port, f, t * N, b"st-c").reshape(N, t, 1)
    want = soa(port.shamir_share_coeffs(f, secrets, coeffs, n))            # [n][N][1]
    lib = scl.lib
    sec_buf = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    sec_buf[1:] = torch.from_numpy(secrets.view(np.int64).reshape(N)).cuda()       # starts 8 bytes into the buffer
    co_buf = torch.zeros(t * cstride + 1, dtype=torch.int64, device="cuda")
    co_view = co_buf[1:].view(-1)[: t * cstride].view(t, cstride)
    co_view[:, :N] = torch.from_numpy(np.ascontiguousarray(np.transpose(coeffs, (1, 0, 2))).view(np.int64).reshape(t, N)).cuda()
    sh_buf = torch.full((n * stride + 1,), -1, dtype=torch.int64, device="cuda")
    p = lambda tns, off=0: C.c_void_p(tns.data_ptr() + 8 * off)
    for k, v in mode.items():
        scl.set_tuning(k, v)
    try:
        st = lib.scl_hip_shamir_share(f, p(sh_buf, 1), C.c_size_t(stride), p(sec_buf, 1), p(co_buf, 1), C.c_size_t(cstride),
                                      C.c_size_t(N), C.c_size_t(t), C.c_size_t(n), None, None)
        assert st == 0, scl.lib.scl_hip_last_error()
        got = sh_buf[1:].view(n, stride).cpu().numpy().view(np.uint64)
        assert np.array_equal(got[:, :N], want[:, :, 0])
        assert (got[:, N:] == np.uint64(2 ** 64 - 1)).all()                 # the padding between rows is untouched
        # reconstruct from the strided, 8-byte aligned matrix into an 8-byte aligned output
        lam = scl.lagrange_basis(f, n)
        out_buf = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
        st = lib.scl_hip_shamir_recover(f, p(out_buf, 1), p(sh_buf, 1), C.c_size_t(stride),
                                        lam.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_size_t(N), None)
        assert st == 0
        if t < n:    # (with t >= n the n shares do not determine the polynomial)
            assert np.array_equal(out_buf[1:].cpu().numpy().view(np.uint64), secrets[:, 0])
        else:
            assert np.array_equal(out_buf[1:].cpu().numpy().view(np.uint64),
                                  port.shamir_recover_lambda(f, np.ascontiguousarray(np.transpose(want, (1, 0, 2))), lam)[:, 0])
        # element-wise on the misaligned rows
        dst = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
        st = lib.scl_hip_ew(f, O.MUL, p(dst, 1), p(sh_buf, 1), p(sh_buf, 1 + stride), C.c_size_t(N), None)
        assert st == 0
        assert np.array_equal(dst[1:].cpu().numpy().view(np.uint64),
                              port.ew(f, O.MUL, want[0], want[1])[:, 0])
        # error correction straight from the same buffer
        if t == 3:
            fo = torch.zeros(n * stride, dtype=torch.int64, device="cuda")
            eo = torch.zeros((t + 1) * stride, dtype=torch.int64, device="cuda")
            status = torch.zeros(N, dtype=torch.uint8, device="cuda")
            nerr = torch.zeros(N, dtype=torch.int32, device="cuda")
            sh_buf[1 + 2 * stride + 5] = 12345                              # corrupt party 2's share of secret 5
            q, fl = C.c_size_t(0), C.c_size_t(0)
            st = lib.scl_hip_shamir_recover_correct(f, p(fo), C.c_size_t(stride), p(eo), C.c_size_t(stride), p(status), p(nerr),
                                                    p(sh_buf, 1), C.c_size_t(stride), C.c_size_t(n), C.c_size_t(N), None,
                                                    C.byref(q), C.byref(fl), None)
            assert st == 0 and q.value == 1 and fl.value == 0
            assert np.array_equal(fo.view(n, stride)[0, :N].cpu().numpy().view(np.uint64), secrets[:, 0])
            assert nerr.cpu().numpy().tolist() == [1 if s == 5 else 0 for s in range(N)]
    finally:
        for k in mode:
            scl.set_tuning(k, 0)


@pytest.mark.parametrize("f", ALL_FIELDS)
def test_random_shapes_share_and_recover(scl, port, f):
    """a seeded sweep over (n, t, N): PRG-mode and coefficient-mode sharing against the oracle, reconstruction of
    every shape that has more than t shares -- whatever kernel the dispatch picks for the shape"""
    L = O.LIMBS[f]
    rng = np.random.default_rng(1234 + f)
    budget = 30_000 if f in SLOW_ORACLE else 400_000          # oracle multiply-adds per case
    for case in range(24):
        n = int(rng.integers(1, 65 if L == 4 else 129))
        t = int(rng.integers(0, min(48, 2 * n) + 1))
        N = int(max(1, min(700, budget // (n * (t + 1)))))
        N = int(rng.integers(1, N + 1))
        secrets = rand_elems(port, f, N, b"rs-%d" % case)
        if f == O.GF2_128:      # (default nodes differ between the x++ walk and the bit patterns: explicit coefficients only)
            nodes = O.from_ints(list(range(1, n + 1)), L)
            coeffs = rand_elems(port, f, max(t, 1) * N, b"rc-%d" % case).reshape(N, max(t, 1), L)[:, :t]
            want = soa(np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)]))
            got = scl.shamir_share(f, dev(scl, secrets), dev(scl, np.ascontiguousarray(np.transpose(coeffs, (1, 0, 2)))) if t else None, n)
        else:
            want = soa(port.shamir_share(f, b"rs-seed", secrets, t, n))
            got = scl.shamir_share_prg(f, dev(scl, secrets), t, n, b"rs-seed")
        assert np.array_equal(host(scl, got), want), (n, t, N)
        if n > t:
            lam = scl.lagrange_basis(f, n)
            assert np.array_equal(host(scl, scl.shamir_recover(f, got, lam)), secrets), (n, t, N)


def test_prg_share_two_pass_across_slabs(scl, port):
    """PRG-driven sharing in two passes cuts a large batch into slabs of coefficient rows (at most 1 GiB each): at (128,42)
    over Mersenne61 a slab is 3 145 728 secrets.  Seven million secrets = three slabs; windows around every slab boundary
    and at both ends must equal what the same call gives for those secrets alone (first_secret = their index: no slabs
    involved), the first window also the oracle's shares, and everything reconstructs."""
    f, L, n, t, N = O.M61, 1, 128, 42, 7_000_003
    torch = __import__("torch")
    free = torch.cuda.mem_get_info()[0]
    if free < (n * N * 8) * 1.3:
        pytest.skip("not enough free device memory")
    secrets = scl.vector_random(f, N, b"slab-secrets")
    seed = b"slab-seed"
    big = scl.shamir_share_prg(f, secrets, t, n, seed)
    slab = 3_145_728
    for a in (0, slab - 3, 2 * slab - 3, N - 6):
        k = 6
        part = scl.shamir_share_prg(f, secrets[a:a + k].clone(), t, n, seed, first_secret=a)
        assert scl.equals(f, part.reshape(-1, L), big[:, a:a + k].reshape(-1, L).clone()), a
    want = soa(port.shamir_share(f, seed, host(scl, secrets[:4].clone()), t, n))
    assert np.array_equal(host(scl, big[:, :4].clone()), want)
    lam = scl.lagrange_basis(f, n)
    assert scl.equals(f, scl.shamir_recover(f, big, lam), secrets)


# ---------------------------------------------------------------------------------------------- beyond 2^32 elements
def test_more_than_2_to_the_32_elements(scl, port):
    """Vector<T>::SizeType is uint32_t in the reference (vector.h:73): a Vector holds fewer than 2^32 elements.  The batch API
    takes size_t and the card holds 288 GB, so a batch may be longer: N = 2^32 + 4097 Mersenne61 elements (34 GB per vector)
    through the element-wise kernels, sum / dot, coefficient-fed and PRG-driven sharing (3, 1) and reconstruction -- every
    index past 2^32 and every byte offset past 2^35.  Checked by properties over the whole batch and by oracle windows at the
    start, ACROSS element 2^32 and at the end (PRG blocks addressed by counter, note P of SURVEY.md section 8a)."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    assert free > 225 * 10**9, f"this case needs 215 GB of HBM, {free / 1e9:.0f} GB free"
    f, L = O.M61, 1
    N = 2**32 + 4097
    wins = [(0, 64), (2**32 - 32, 64), (N - 64, 64)]

    def window(t_, lo, w):
        return host(scl, t_[..., lo:lo + w, :].contiguous())

    a = scl.vector_random(f, N, b"2^32-a")
    b = scl.vector_random(f, N, b"2^32-b")
    # Vector::random over more than 2^32 elements: element s = bytes [8 s, 8 s + 8) of the stream, block s / 2
    for lo, w in wins:
        raw = port.prg_blocks(b"2^32-a", lo // 2, w // 2 + 1)[8 * (lo % 2): 8 * (lo % 2) + 8 * w]
        assert np.array_equal(window(a, lo, w), port.from_bytes(f, raw)), lo
    out = scl.empty(f, N)
    for op in (O.ADD, O.MUL):
        scl.ew(f, op, a, b, out=out)
        for lo, w in wins:
            assert np.array_equal(window(out, lo, w), port.ew(f, op, window(a, lo, w), window(b, lo, w))), (op, lo)
    scl.ew(f, O.ADD, a, b, out=out)
    scl.ew(f, O.SUB, out, b, out=out)
    assert scl.equals(f, out, a)
    # inverse (simultaneous inversion, 32 elements per lane: tiles past 2^32) and x * x^-1 = 1 over all of it (a uniform element
    # is zero with probability 2^-61: the status word must stay clear)
    status = scl.ew_status_buffer()
    scl.ew_status(f, O.INV, a, None, status, out=out)
    assert int(status.item()) == 0
    for lo, w in wins:
        assert np.array_equal(window(out, lo, w), port.ew(f, O.INV, window(a, lo, w))), lo
    scl.ew(f, O.MUL, out, a, out=out)
    assert int((out.view(-1) != 1).sum().item()) == 0
    # Vector::sum / dot: the whole = the part below 2^32 + the part above
    h = 2**32 - 5
    assert np.array_equal(scl.vsum(f, a), port.ew(f, O.ADD, scl.vsum(f, a[:h])[None], scl.vsum(f, a[h:])[None])[0])
    assert np.array_equal(scl.dot(f, a, b), port.ew(f, O.ADD, scl.dot(f, a[:h], b[:h])[None], scl.dot(f, a[h:], b[h:])[None])[0])
    del out
    torch.cuda.empty_cache()
    # shamirSecretShare (3, 1) from resident coefficients, shamirRecoverP from all three and from two shares
    n, t = 3, 1
    shares = scl.shamir_share(f, a, b.view(1, N, 1), n)
    for lo, w in wins:
        want = soa(port.shamir_share_coeffs(f, window(a, lo, w), window(b, lo, w).reshape(w, 1, L), n))
        assert np.array_equal(window(shares, lo, w), want), lo
    rec = scl.shamir_recover(f, shares)
    assert scl.equals(f, rec, a)
    scl.shamir_recover(f, shares[:2], out=rec)
    assert scl.equals(f, rec, a)
    # the reference's own mode: coefficients from the PRG, secret s from block s (Vector::random(2) of 8-byte elements = 1 block)
    scl.shamir_share_prg(f, a, t, n, b"2^32-seed", out=shares)
    for lo, w in wins:
        elems = port.from_bytes(f, port.prg_blocks(b"2^32-seed", lo, w)).reshape(w, 2, L)
        want = soa(port.shamir_share_coeffs(f, window(a, lo, w), np.ascontiguousarray(elems[:, 1:2]), n))
        assert np.array_equal(window(shares, lo, w), want), lo
    scl.shamir_recover(f, shares, out=rec)
    assert scl.equals(f, rec, a)
    # additiveShare n = 3 from the PRG (2 blocks per secret) and Vector::sum per secret
    scl.additive_share_prg(f, a, 3, b"2^32-add", out=shares)
    for lo, w in wins:      # share i < n - 1 of secret s: the first 8 bytes of block s (n - 1) + i; the last one = secret - the others
        blk = np.frombuffer(port.prg_blocks(b"2^32-add", 2 * lo, 2 * w), dtype=np.uint8).reshape(w, 2, 16)[:, :, :8]
        r = port.from_bytes(f, np.ascontiguousarray(blk).tobytes()).reshape(w, 2, L)
        got = window(shares, lo, w)
        assert np.array_equal(got[0], r[:, 0]) and np.array_equal(got[1], r[:, 1]), lo
        assert np.array_equal(got[2], port.ew(f, O.SUB, window(a, lo, w), port.ew(f, O.ADD, r[:, 0], r[:, 1]))), lo
    scl.additive_recover(f, shares, out=rec)
    assert scl.equals(f, rec, a)
    del a, b, shares, rec
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------- full size properties
@pytest.mark.parametrize("f,n,t,N", [(O.M61, 10, 3, 100_000_000), (O.M127, 10, 3, 10_000_000), (O.MONT128, 10, 3, 10_000_000),
                                     (O.GF2_128, 40, 13, 12_500_000)])
def test_full_size_round_trip(scl, port, f, n, t, N):
    """share -> reconstruct round trip, linearity and threshold consistency at the BASELINE sizes -- C2 (10^8), C3 (10^7,
    Mersenne127 and the literal 128-bit Montgomery prime) and C4's per-GPU shard (1.25 * 10^7 at (40,13) over GF(2^128)) --
    through size-independent properties; the oracle spot-checks a window."""
    L = O.LIMBS[f]
    secrets = scl.vector_random(f, N, b"big-secrets")
    other = scl.vector_random(f, N, b"big-secrets-2")
    sh = scl.shamir_share_prg(f, secrets, t, n, b"big-seed")
    assert scl.equals(f, scl.shamir_recover(f, sh), secrets)
    assert scl.equals(f, scl.shamir_recover(f, sh[: t + 1].contiguous()), secrets)
    # any t+1 shares: parties 3,5,6,9 (nodes 4,6,7,10) for t = 3, every third party otherwise
    idx = [3, 5, 6, 9] if t == 3 else list(range(0, 3 * (t + 1), 3))[: t + 1]
    nodes = np.stack([port.from_int(f, i + 1) for i in idx])
    lam = scl.lagrange_basis(f, len(idx), nodes)
    assert scl.equals(f, scl.shamir_recover(f, sh[idx].contiguous(), lam), secrets)
    # linearity: share(a) + share(b) reconstructs to a + b
    sh2 = scl.shamir_share_prg(f, other, t, n, b"other-seed")
    ssum = scl.ew(f, O.ADD, sh, sh2)
    assert scl.equals(f, scl.shamir_recover(f, ssum), scl.ew(f, O.ADD, secrets, other))
    del sh2, ssum
    if f == O.GF2_128:
        # oracle window at the shard size: the oracle's shamirSecretShare walks x++ like the reference (1, 0, 1, .. in
        # characteristic 2), so the window is evaluated at the bit-pattern nodes explicitly -- coefficient rows of secret s
        # from PRG blocks [s*B, (s+1)*B), B = t + 1, c_0 replaced by the secret, Polynomial::evaluate at 1 .. n
        w0, w, B = 7654321 % (N - 24), 24, t + 1
        elems = port.from_bytes(f, port.prg_blocks(b"big-seed", w0 * B, w * B)).reshape(w, -1, L)
        hs = host(scl, secrets[w0:w0 + w])
        nodes_bits = O.from_ints(list(range(1, n + 1)), L)
        want = soa(np.stack([port.poly_eval(f, np.concatenate([hs[s:s + 1], elems[s, 1:t + 1]]), nodes_bits) for s in range(w)]))
        assert np.array_equal(host(scl, sh[:, w0:w0 + w]), want)
        lam_all = port.lagrange_basis(f, nodes_bits, port.from_int(f, 0))
        rec = port.shamir_recover_lambda(f, np.ascontiguousarray(np.transpose(want, (1, 0, 2))), lam_all)
        assert np.array_equal(rec, hs)
        return
    # oracle window: coefficients of secret s come from PRG blocks [s*B, (s+1)*B)
    w0, w = 1234567 % (N - 300), 300 if f == O.M61 else 40
    B = (t + 2) // 2 if L == 1 else t + 1
    elems = port.from_bytes(f, port.prg_blocks(b"big-seed", w0 * B, w * B)).reshape(w, -1, L)
    hs = host(scl, secrets[w0:w0 + w])
    want = soa(port.shamir_share_coeffs(f, hs, np.ascontiguousarray(elems[:, 1:t + 1]), n))
    assert np.array_equal(host(scl, sh[:, w0:w0 + w]), want)
    tail = scl.shamir_share_prg(f, secrets[w0:w0 + w].contiguous(), t, n, b"big-seed", first_secret=w0)
    assert np.array_equal(host(scl, tail), want)
    # additive at size
    ad = scl.additive_share_prg(f, secrets, 3, b"big-add")
    assert scl.equals(f, scl.additive_recover(f, ad), secrets)


@pytest.mark.parametrize("f", [O.M61, O.M127])
def test_prg_share_across_a_2_32_counter_boundary(scl, port, f):
    """shamirSecretShare(secret, t, n, prg) on a PRG whose block counter passes 2^32 in the middle of the batch (prg.h:167-169:
    the counter is a 64-bit `long`; the upper word of the AES input block changes inside the launch): every share against the
    oracle's per-secret evaluation of the coefficients the oracle's PRG blocks spell, and additive sharing the same way."""
    L, n, t, N = O.LIMBS[f], 10, 3, 1500
    B = (t + 2) // 2 if L == 1 else t + 1
    c0 = 2 ** 32 - 700 * B - (1 if B > 1 else 0)      # the boundary falls inside secret 700's blocks
    secrets = rand_elems(port, f, N, b"wrap-secrets")
    sh = scl.shamir_share_prg(f, dev(scl, secrets), t, n, b"wrap-seed", counter0=c0)
    elems = port.from_bytes(f, port.prg_blocks(b"wrap-seed", c0, N * B)).reshape(N, -1, L)
    want = soa(port.shamir_share_coeffs(f, secrets, np.ascontiguousarray(elems[:, 1:t + 1]), n))
    assert np.array_equal(host(scl, sh), want)
    assert scl.equals(f, scl.shamir_recover(f, sh), dev(scl, secrets))
    v = scl.vector_random(f, 4000, b"wrap-vr", counter0=2 ** 32 - 1000)
    bs = 8 * L
    assert np.array_equal(host(scl, v), port.from_bytes(f, port.prg_blocks(b"wrap-vr", 2 ** 32 - 1000, (4000 * bs + 15) // 16)[: 4000 * bs]))
    # the two-pass form (coefficient rows first; forced) across the boundary, and a threshold that takes it by itself
    scl.set_tuning("prg_two_pass", 1)
    try:
        sh2 = scl.shamir_share_prg(f, dev(scl, secrets), t, n, b"wrap-seed", counter0=c0)
    finally:
        scl.set_tuning("prg_two_pass", 0)
    assert np.array_equal(host(scl, sh2), want)
    t9, B9 = 9, (9 + 2) // 2 if L == 1 else 10
    c9 = 2 ** 32 - 300 * B9 - 1
    sh9 = scl.shamir_share_prg(f, dev(scl, secrets), t9, n, b"wrap-seed9", counter0=c9)
    el9 = port.from_bytes(f, port.prg_blocks(b"wrap-seed9", c9, N * B9)).reshape(N, -1, L)
    assert np.array_equal(host(scl, sh9), soa(port.shamir_share_coeffs(f, secrets, np.ascontiguousarray(el9[:, 1:t9 + 1]), n)))
    # additive sharing: share i < n - 1 of secret s is the element drawn from block s (n - 1) + i (one block per FF::random)
    na, ca = 3, 2 ** 32 - 900
    ad = scl.additive_share_prg(f, dev(scl, secrets), na, b"wrap-add", counter0=ca)
    blk = np.frombuffer(port.prg_blocks(b"wrap-add", ca, N * (na - 1)), dtype=np.uint8).reshape(N * (na - 1), 16)
    rnd = port.from_bytes(f, blk[:, :bs].tobytes()).reshape(N, na - 1, L)
    had = host(scl, ad)
    assert np.array_equal(had[: na - 1], np.ascontiguousarray(np.transpose(rnd, (1, 0, 2))))
    assert scl.equals(f, scl.additive_recover(f, ad), dev(scl, secrets))


def test_c5_shard_size(scl, port):
    """BASELINE configs[4] at the size one of eight GPUs holds: (n, t) = (128, 42) over Mersenne61, 1.25 * 10^8 secrets --
    128 GB of shares, 42 GB of coefficients generated on the device (addressing beyond 2^32 elements per matrix, the
    matrix-core share kernel over thousands of trips per workgroup, the 128-row table reconstruct)."""
    f, n, t, N = O.M61, 128, 42, 125_000_000
    torch.cuda.empty_cache()
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < 190 * (1 << 30):
        pytest.skip("needs 190 GB of free HBM")
    secrets = scl.vector_random(f, N, b"c5-secrets")
    coeffs = scl.empty(f, t, N)
    per_row = (N * 8 + 15) // 16
    for k in range(t):
        scl.vector_random(f, N, b"c5-coeffs", counter0=k * per_row, out=coeffs[k])
    shares = scl.shamir_share(f, secrets, coeffs, n)
    assert scl.equals(f, scl.shamir_recover(f, shares), secrets)
    # the last t + 1 parties alone (rows past 2^32 elements into the matrix)
    idx = list(range(n - t - 1, n))
    lam = scl.lagrange_basis(f, len(idx), np.stack([port.from_int(f, i + 1) for i in idx]))
    assert scl.equals(f, scl.shamir_recover(f, shares[n - t - 1:], lam), secrets)
    # oracle windows at both ends and in the middle of the batch
    for w0 in (0, N // 2 + 12345, N - 64):
        hs = host(scl, secrets[w0:w0 + 64])
        hc = np.ascontiguousarray(np.transpose(host(scl, coeffs[:, w0:w0 + 64].contiguous()), (1, 0, 2)))
        want = soa(port.shamir_share_coeffs(f, hs, hc, n))
        assert np.array_equal(host(scl, shares[:, w0:w0 + 64].contiguous()), want), w0


def test_bench_contract_small(scl):
    """bench.py end to end on a small batch: ONE compact JSON line on stdout with the contract's keys, the roofline, kernels and
    cpu_baseline objects and the verdict over every leg -- short enough for the driver's record -- and, in the detail file it
    names, everything the legs measured: the open-step report, the element-wise path with its compute rooflines and the
    asynchronous inverse, the layout bridge, Matrix::multiply, the full CPU baseline."""
    from bench_util import assert_compact, run_bench
    r = run_bench(["--secrets", "300000", "--steps", "2", "--warmup", "1", "--cpu-sample", "2000", "--cpu-all-cores", "1",
                   "--configs", "0", "--open-secrets", "50000", "--open-chunk", "20000", "--c4-rank-secrets", "60001",
                   "--ew-elements", "300001"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line, detail = assert_compact(r), r.detail
    # the line tells the truth about its side legs: top-level verified = AND over the headline and every leg, the CPU model
    # beside the core count, where the traffic figure comes from, and (below) a failed leg -> error listed, exit code 1
    assert line["verified_headline"] is True and line["verified_legs"] and all(line["verified_legs"].values())
    assert {"open.c4_all_gather", "open.c4_all_gather.c_abi", "open.c4_all_gather.partial_gather", "open.m61_partial_sums",
            "open.c4_rank_shape", "ew", "layout", "matmul"} <= set(line["verified_legs"])
    assert "errors" not in line and detail["line"] == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "kernels", "cpu_baseline", "verified", "verified_legs", "detail"):
        assert k in line, k
    for k in ("ew", "layout", "matmul", "open", "configs", "prg_mode", "by_allocation", "c1_additive"):
        assert k not in line, k                      # the legs' own figures live in the detail file
    assert line["verified"] is True and line["value"] > 0 and line["n_gpus"] == 1 and line["steps"] == 2
    assert set(("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes",
                "traffic_over_algorithmic")) <= set(line["roofline"])
    assert set(line["kernels"]) == {"shamir_share", "shamir_recover"} and all(k["ms"] > 0 and 0 < k["frac"] < 1 for k in line["kernels"].values())
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample", "cpu_model"}
    assert line["cpu_baseline"]["kind"] in ("reference", "port") and line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["cpu_model"]
    assert line["config"]["allocations"] == 3 and "configs" not in detail and "model" not in line["config"]
    assert line["rccl"]["ranks"] == 1 and len(line["ms_per_step_by_rank"]) == 1
    # the committed PMC traffic describes BASELINE configs[1] only: any other configuration reports null
    assert line["roofline"]["traffic"] is None and line["roofline"]["traffic_source"] is None
    # the headline's steps rotate over independently allocated operand sets; the per-allocation kernel times ride along
    assert len(detail["by_allocation"]) == 2 and all(b["steps"] == 1 and b["share_ms"] > 0 for b in detail["by_allocation"])
    rs = detail["open"]["c4_rank_shape"]
    assert rs["verified"] is True and rs["partial_bytes_per_secret"] == 96 and rs["sum_bytes_per_secret"] == 144
    # the element-wise path north_star names first, and the layout bridge: add / mul / inv per field with their rooflines
    ew = detail["ew"]
    assert ew["verified"] is True and set(ew["fields"]) == {"Mersenne61", "Mersenne127", "Mont128", "GF(2^128)"}
    for name, fld in ew["fields"].items():
        assert fld["elements"] == 300001 and all(fld[op]["verified"] and fld[op]["GBps"] > 0 for op in ("add", "mul", "inv", "inv_async"))
        assert fld["add"]["bytes_per_element"] == 3 * fld["inv"]["bytes_per_element"] // 2
        # legs HBM does not bound carry the ceiling they do run against: instructions per element x rate against the issue rate
        rc = fld["inv"]["roofline_compute"]
        assert rc["bound"] == "vector ALU" and rc["per_element"] > 50 and 0 < rc["frac"] < 1.2 and rc["peak"] > 1e11
        assert set(fld["inverse_small_batches"]) == {"10000", "100000"} and fld["inverse_small_batches"]["10000"]["async_wall_us_per_call"] > 0
        assert line["legs"]["ew " + name]["inv"] > 0 and line["legs"]["ew " + name]["inv_frac_valu"] > 0
    assert "roofline_lds" in ew["fields"]["GF(2^128)"]["mul"] and ew["fields"]["Mersenne127"]["inv"]["traffic_over_algorithmic"]["expected"] == 2.5
    mm = detail["matmul"]
    assert mm["verified"] is True and any(v["path"].startswith("matrix cores") and v["frac_of_int8_peak"] > 0 for v in mm["shapes"].values())
    lay = detail["layout"]
    assert lay["verified"] is True and all(v["soa_to_aos"]["frac"] > 0 and v["aos_to_soa"]["frac"] > 0 for v in lay["fields"].values())
    cb = detail["cpu_baseline"]
    assert cb["physical_cores"] >= 1 and cb["all_cores"]["cores"] == cb["physical_cores"] and cb["cores_per_gpu"]["cores"] <= 16
    c4 = detail["open"]["c4_all_gather"]
    assert c4["verified"] is True and c4["secrets"] == 50000 and c4["chunk"] == 20000 and c4["parties_per_rank"] == 40
    assert detail["open"]["m61_partial_sums"]["verified"] is True
    bad = run_bench(["--secrets", "300000", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--configs", "0",
                     "--open-secrets", "50000", "--open-chunk", "20000", "--c4-rank-secrets", "0", "--ew", "0", "--inject-error", "c_abi"],
                    timeout=600)
    bl = assert_compact(bad)
    assert bad.returncode != 0 and bl["verified"] is False and bl["verified_headline"] is True
    assert any("c_abi" in e for e in bl["errors"]) and bl["verified_legs"]["open.c4_all_gather.c_abi"] is False
    assert "cpu_baseline" not in bl


def test_bench_default_command_prints_one_short_line(scl):
    """The driver's own command at full size -- `python3 bench.py --gpus 1 --steps 20 --warmup 5`: BASELINE configs[1], 10^8
    secrets, every leg on -- prints ONE line of less than 8 KB that parses and carries `roofline` and `cpu_baseline`
    (BENCH_r05.json: a 20 KB line, "parsed": null), and finishes in about half a minute."""
    import gc
    import time
    from bench_util import assert_compact, run_bench
    gc.collect()
    torch.cuda.empty_cache()      # this process's cached blocks go back: the (128,42) shard alone takes 172 of the 288 GB
    t0 = time.time()
    r = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"], timeout=900)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    line = assert_compact(r)
    assert len(r.lines[0]) < 6500, len(r.lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["verified"] is True
    assert "BASELINE configs[1]" in line["config"]["workload"] and line["config"]["secrets_per_gpu"] == 100_000_000
    assert line["roofline"]["bound"] == "hbm" and 0.5 < line["roofline"]["frac"] < 1 and line["roofline"]["peak"] == 8000.0
    assert line["cpu_baseline"]["value"] > 1e5 and line["cpu_baseline"]["kind"] == "reference"
    assert abs(line["value"] - 1e8 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3
    assert {"C3_mersenne127_10_3", "C3_mont128_10_3", "C4_shard_gf2_128_40_13", "C5_shard_mersenne61_128_42"} <= set(line["legs"])
    assert set(r.detail["configs"]) >= {"C3_mont128_10_3", "F3_secp256k1_scalar_10_3"} and r.detail["line"] == line
    c4 = r.detail["configs"]["C4_shard_gf2_128_40_13"]
    assert c4["recover_roofline_compute"]["bound"] == "LDS table reads" and c4["share_roofline_compute"]["bound"] == "vector ALU"
    assert r.detail["prg_mode"]["k_prg_blocks"]["roofline_compute"]["frac"] > 0
    assert wall < 120, wall


def test_bench_observes_its_hbm_traffic(scl):
    """--pmc-live 1: roofline.traffic is OBSERVED by the run that prints it: after its timed regions bench.py runs the headline
    alone twice as a child under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (no trace domain beside --pmc; the program
    itself after `--`) and reads the two kernels' counters; the doubling of FETCH_SIZE is checked on k_copy16 in the same
    pass.  At 10^7 secrets here: traffic = the algorithmic bytes within 1 %."""
    from bench_util import assert_compact, run_bench
    r = run_bench(["--secrets", "10000000", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--configs", "0", "--open", "0",
                   "--ew", "0", "--pmc-live", "1"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = assert_compact(r)
    rf = line["roofline"]
    assert rf["traffic_source"].startswith("live"), (rf["traffic_source"], r.detail.get("pmc_live"))
    live = r.detail["traffic_live"]
    assert abs(live["fetch_correction_measured_on_k_copy16"] - 2.0) < 0.02
    assert abs(live["shamir_share"]["bytes"] / (112 * 10**7) - 1) < 1e-2 and abs(live["shamir_recover"]["bytes"] / (88 * 10**7) - 1) < 1e-2
    assert rf["traffic"] == live[rf["kernel"]]["bytes"] and abs(rf["traffic_over_algorithmic"] - 1) < 1e-2
    assert rf["traffic_stamped"] is None     # the stamped file describes 10^8 secrets only
    assert r.detail["pmc_live"]["ran"] is True


def test_open_step_checker_of_the_first_contact_kit_on_one_rank(scl):
    """tools/open_rccl_check.py -- step (a) of tools/first_contact_8gpu.sh: the C ABI's three open forms and their
    torch.distributed twins against the CPU oracle over REAL RCCL, one process per GPU -- started the way the kit starts it,
    with the one rank this box has: the script's own logic (slabs, padding rows, the agreement all-reduce, the JSON line) and
    ncclAllGather / ncclReduceScatter of a one-rank communicator.  Ranks 2..8 are what the kit is for."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port_ = s_.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port_), os.path.join(ROOT, "tools", "open_rccl_check.py"), "--secrets", "20001",
                        "--chunk", "6000"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["ok"] is True and line["world"] == 1 and line["backend"] == "nccl"
    assert set(line["fields"]) == {"Mersenne61", "Mersenne127", "GF(2^128)"}
    for name, f in line["fields"].items():
        assert f["c_abi_all_gather"] and f["c_abi_partial_gather"] and f["torch_all_gather"] and f["torch_partial_gather"], (name, f)
    assert line["fields"]["Mersenne61"]["c_abi_reduce_scatter"] is True and line["fields"]["GF(2^128)"]["parties_per_rank"] == 40


def test_bench_open_mode_line(scl):
    """--mode open: the exchange step alone (reference: Network::send + Network::recv, include/scl/net/network.h:148-185),
    with the collective's bandwidth fields beside the reconstruct kernel's HBM fraction"""
    from bench_util import assert_compact, run_bench
    r = run_bench(["--mode", "open", "--secrets", "200000", "--open-secrets", "70000", "--open-chunk", "32768"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = assert_compact(r)
    assert line["metric"] == "shamir_open_reconstructions_per_sec" and line["n_gpus"] == 1 and line["value"] > 0
    c4 = r.detail["open"]["c4_all_gather"]
    for k in ("rccl_busbw_GBps", "rccl_algbw_GBps", "gather_ms_per_chunk", "reconstruct_hbm_frac", "pipeline_ms"):
        assert k in c4, k
    assert c4["verified"] and r.detail["open"]["m61_partial_sums"]["verified"] and line["verified"] is True
    assert line["roofline"]["kernel"] == "shamir_recover" and 0 < line["roofline"]["frac"] < 1


def test_bench_two_ranks_rehearsal_on_one_device(scl):
    """bench.py's multi-rank logic with the HIP kernels: two ranks, both on this box's one GPU, gloo collectives on device
    tensors (SCL_BENCH_ONE_DEVICE=1: a rehearsal, never a measurement; the driver's multi-GPU run uses RCCL, one rank per GPU).
    Started plainly with --gpus 2, so the self-launcher runs too.  Covers the sharded headline, the all-gather open
    (20 parties per rank) and the reduce-scatter partial-sum open; every round trip must verify."""
    from bench_util import assert_compact, run_bench
    r = run_bench(["--gpus", "2", "--backend", "gloo", "--secrets", "1000000", "--open-secrets", "100000", "--open-chunk", "32768",
                   "--configs", "0", "--cpu-sample", "0", "--steps", "2", "--warmup", "1"], {"SCL_BENCH_ONE_DEVICE": "1"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = assert_compact(r)
    assert line["n_gpus"] == 2 and line["verified"] is True and line["config"]["parallelism"] == "shard2"
    c4, ps = r.detail["open"]["c4_all_gather"], r.detail["open"]["m61_partial_sums"]
    assert c4["verified"] and c4["parties_per_rank"] == 20 and c4["collective"] == "all_gather_into_tensor"
    assert c4["rccl_busbw_GBps"] > 0 and ps["verified"] and ps["collective"].startswith("reduce_scatter_tensor")
    assert "cpu_baseline" not in line and "configs" not in r.detail
    assert "skipped in the one-device rehearsal" in c4["c_abi"]["skipped"] and "errors" not in line
    # what the process group reports about the job, and every rank's own time per step (the line's is their maximum)
    assert line["rccl"]["ranks"] == 2 and line["rccl"]["backend"] == "gloo" and line["rccl"]["devices"] == [0, 0]
    assert len(line["ms_per_step_by_rank"]) == 2 and abs(max(line["ms_per_step_by_rank"]) - line["ms_per_step"]) < 1e-9
    assert len(r.detail["by_allocation"]) == 2      # two timed steps on the first two of the three operand sets
    assert line["value"] > 0 and abs(line["value"] - 2 * 1e6 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3


def test_differential_fuzz_of_the_entry_points_for_twenty_seconds(scl):
    """tests/fuzz_abi.py (random field / entry point / sizes / parties / thresholds / zeros / in-place / unaligned operands
    against the CPU oracle) with a fixed seed: a regression net under the hand-picked cases above.  The campaigns behind
    profiles/r5_fuzz_abi.txt ran it for minutes per seed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_abi.py"), "20", "2024"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert " 0 mismatches" in r.stdout and "MISMATCH" not in r.stdout


def test_share_matrix_of_one_secret_from_a_transposed_array(scl, port):
    """Found by tests/fuzz_abi.py: a [m][1][L] share matrix made by transposing the reference's [1][m][L] layout keeps the
    source's stride in its dimension of extent 1; the harness took that for a non-dense row and refused the call."""
    for f in (O.M61, O.M127, O.SECP256K1_SCALAR):
        L = O.LIMBS[f]
        m = 7
        aos = rand_elems(port, f, m, b"one-secret").reshape(1, m, L)
        t = scl.to_device(np.ascontiguousarray(aos.transpose(1, 0, 2)))
        assert np.array_equal(host(scl, scl.shamir_recover(f, t)), port.shamir_recover(f, aos))
        view = scl.to_device(aos).permute(1, 0, 2)          # the same bytes as a strided torch view
        assert np.array_equal(host(scl, scl.shamir_recover(f, view)), port.shamir_recover(f, aos))


def test_bench_line_survives_a_rank_that_never_reaches_the_open_step(scl):
    """The legs after the headline are collectives; real multi-rank RCCL runs only on the driver's node.  A rank that never
    arrives (--inject-error hang: the last rank sleeps before the open step) must not cost the headline: after --side-timeout
    seconds rank 0 writes the line with the headline it measured and the missing leg as an error, and every rank exits -- with
    exit code 0 on a multi-rank run whose headline verified (the line says what failed; a scaling record must not be lost to a
    side leg), so the launcher reports a finished job."""
    from bench_util import assert_compact, run_bench
    r = run_bench(["--gpus", "2", "--backend", "gloo", "--secrets", "1000000", "--open-secrets", "100000", "--open-chunk", "32768",
                   "--configs", "0", "--cpu-sample", "0", "--steps", "2", "--warmup", "1", "--inject-error", "hang",
                   "--side-timeout", "20"], {"SCL_BENCH_ONE_DEVICE": "1"}, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "watchdog" in r.stderr
    line = assert_compact(r)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["verified_headline"] is True and line["verified"] is False
    assert "watchdog" in r.detail["open"]["error"] and any("watchdog" in e for e in line["errors"])
    assert line["roofline"]["frac"] > 0 and len(line["ms_per_step_by_rank"]) == 2 and line["verified_legs"]["open"] is False


@pytest.mark.parametrize("config,extra,check", [
    ("c4", ["--total-secrets", "100000", "--open-chunk", "32768"],
     lambda ln, d: ln["scaling"] == "strong" and ln["config"]["total_secrets"] == 100000 and ln["config"]["parallelism"] == "parties2"
     and d["open"]["c4_all_gather"]["parties_per_rank"] == 20 and ln["rccl_busbw_GBps"] > 0),
    ("c5", ["--total-secrets", "200003"],
     lambda ln, d: ln["scaling"] == "strong" and ln["config"]["total_secrets"] == 200003 and ln["config"]["n"] == 128
     and ln["config"]["secrets_per_gpu"] == 100002 and ln["dtype"] == "u64"
     and ln["roofline"]["bound"] in ("mfma", "hbm"))])      # (which of the two kernels dominates a 10^5-secret shard is timing noise)
def test_bench_configs_quoted_on_eight_gpus_rehearsal_on_one_device(scl, config, extra, check):
    """`bench.py --gpus 2 --config c4 / c5` with the HIP kernels, both ranks on this box's one GPU over gloo (a rehearsal of
    the code the driver runs on eight GPUs over RCCL, never a measurement): BASELINE configs[3] as the open step over the
    total, configs[4] as share + reconstruct of ragged shards of the total; every round trip must verify."""
    from bench_util import assert_compact, run_bench
    r = run_bench(["--gpus", "2", "--backend", "gloo", "--config", config, "--cpu-sample", "0", "--steps", "2", "--warmup", "1"] + extra,
                  {"SCL_BENCH_ONE_DEVICE": "1"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = assert_compact(r)
    assert line["n_gpus"] == 2 and line["verified"] is True and check(line, r.detail), line


@pytest.mark.parametrize("n,t,N", [(128, 42, 3 * 256 * 32 + 17), (97, 5, 2 * 256 * 32 + 31), (128, 48, 256 * 32 + 1), (100, 33, 4 * 256 * 32)])
def test_share_matrix_core_pipeline_many_trips(scl, port, n, t, N):
    """k_share_mfma_m61_p16 / k_share_mfma_m61_pipe with several trips per workgroup (the cross-block software pipeline: fetch two blocks ahead,
    recode under the matrix instructions) and a ragged last block: bit-identical to the burst kernel, to the VALU kernels
    and, on a sample of secrets, to the oracle's per-secret Horner."""
    f, L = O.M61, 1
    secrets = scl.vector_random(f, N, b"pipe-s")
    coeffs = scl.vector_random(f, t * N, b"pipe-c").reshape(t, N, L)
    outs = {}
    for name, keys in (("p16", {"mfma": 1, "mfma_pipe": 2}), ("pipe", {"mfma": 1, "mfma_pipe": 1}),
                       ("burst", {"mfma": 1, "mfma_pipe": 0}), ("valu", {"mfma": -1})):
        for k, v in keys.items():
            scl.set_tuning(k, v)
        try:
            outs[name] = scl.shamir_share(f, secrets, coeffs, n)
        finally:
            scl.set_tuning("mfma", 0)
            scl.set_tuning("mfma_pipe", 2)
    assert scl.equals(f, outs["pipe"], outs["burst"])
    assert scl.equals(f, outs["pipe"], outs["valu"])
    assert scl.equals(f, outs["pipe"], outs["p16"])
    hs, hc, got = host(scl, secrets), host(scl, coeffs), host(scl, outs["pipe"])
    idx = sorted(set([0, 1, 31, 32, 33, 8191, 8192, N - 33, N - 2, N - 1] + list(range(N // 2, N // 2 + 40))))
    sub_s = hs[idx]
    sub_c = np.ascontiguousarray(np.transpose(hc[:, idx], (1, 0, 2)))
    want = soa(port.shamir_share_coeffs(f, sub_s, sub_c, n))
    assert np.array_equal(got[:, idx], want)


@pytest.mark.parametrize("f", ALL_FIELDS)
@pytest.mark.parametrize("d,t,extra,N", [(2, 3, 0, 200), (5, 4, 2, 333), (9, 9, 1, 129), (3, 1, 0, 64)])
def test_recover_detect_general_overload(scl, port, f, d, t, extra, N):
    """shamirRecoverD(shares, alphas, t, d, x) (shamir.h:116-139) with explicit nodes, degree d != t, an evaluation point
    x != 0 and more shares than needed: the value is the interpolant of the first d+1 shares at x, shares d+1 .. d+t-1 are
    checked against it, anything beyond index d+t-1 is ignored.  Expected values from the oracle's Polynomial::evaluate."""
    L = O.LIMBS[f]
    m = d + t + extra
    nodes = rand_elems(port, f, m, b"gen-nodes-%d" % d)
    x = rand_elems(port, f, 1, b"gen-x")[0]
    coeffs = rand_elems(port, f, (d + 1) * N, b"gen-c").reshape(N, d + 1, L)
    aos = np.stack([port.poly_eval(f, coeffs[s], nodes) for s in range(N)])          # [N][m][L]
    want = np.stack([port.poly_eval(f, coeffs[s], x.reshape(1, L))[0] for s in range(N)])
    junk = rand_elems(port, f, N, b"gen-junk")
    rng = np.random.default_rng(d * 10 + t)
    hit = rng.choice(N, N // 4, replace=False)
    pos = {}
    for k, s in enumerate(hit):
        i = k % m
        if not np.array_equal(aos[s, i], junk[s]):
            aos[s, i] = junk[s]
            pos[int(s)] = i
    out, status, bad = scl.shamir_recover_detect(f, dev(scl, soa(aos)), t, d=d, alphas=nodes, x=x)
    st, got = status.cpu().numpy(), host(scl, out)
    for s in range(N):
        i = pos.get(s)
        if i is None or i >= d + t:                       # untouched, or a share the call never looks at
            assert st[s] == 0 and np.array_equal(got[s], want[s]), (s, i)
        elif i > d:                                       # a checked share disagrees with the interpolant
            assert st[s] == 1 and not got[s].any(), (s, i)
        elif t > 1:                                       # one of the d+1 interpolated shares is off: some check fails
            assert st[s] == 1 and not got[s].any(), (s, i)
    assert bad == int(st.sum())
    with pytest.raises(scl.SclError) as ei:
        scl.shamir_recover_detect(f, dev(scl, soa(aos))[: d + t - 1].contiguous(), t, d=d, alphas=nodes[: d + t - 1], x=x)
    assert ei.value.reference_message == "not enough shares provided to detect errors"
