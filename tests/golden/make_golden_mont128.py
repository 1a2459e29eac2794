#!/usr/bin/env python3
"""Generate tests/golden/golden_mont128.json from the REAL reference's Montgomery code at two limbs.

The reference's Montgomery arithmetic (include/scl/math/fields/ff_ops_gmp.h:44-392) is a family of templates on the limb
count which the reference itself instantiates at N = 4 (src/scl/math/fields/secp256k1_scalar.cc:50-135).  oracle/ref_harness.cc
instantiates the same templates at N = 2 through the reference's field plug-in boundary (field tag 2, run-time modulus), so
every expected value below is produced by the reference's own code compiled here -- BASELINE configs[2]'s "Fp (128-bit prime,
Montgomery)".  One fixture per modulus: 2^128 - 159 (the engine's default), the Mersenne prime 2^127 - 1 and a random 128-bit
prime; each from its own copy of the library (FF::one() / zero() are function-local statics that latch the first modulus,
include/scl/math/ff.h:90-101).  Run in the build container only:

    make -C oracle ref && python tests/golden/make_golden_mont128.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import oracle_lib as O  # noqa: E402
from make_golden import gen_montgomery_field, gen_recover_c, hx  # noqa: E402

PRIMES = {"Mont128": (1 << 128) - 159,
          "Mont128@2^127-1": (1 << 127) - 1,
          "Mont128@c381e88f38c0c8fd8712b8bc076f3787": 0xc381e88f38c0c8fd8712b8bc076f3787}   # tests/test_plugin_field_pins.py PRIMES[3]


def main():
    f = O.MONT128
    doc = {"generator": "tests/golden/make_golden_mont128.py", "fields": {},
           "source": "oracle/_ref/libscl_ref.so: the reference's monty*<N> templates (ff_ops_gmp.h) instantiated at N = 2 in "
                     "oracle/ref_harness.cc, elements = the two-limb Montgomery image of FF::m_value"}
    for name, p in PRIMES.items():
        ref = O.Ref(fresh=True)
        ref.mont128_set_prime(p)
        assert ref.mont128_get_prime() == p
        fd = gen_montgomery_field(ref, f)
        fd["prime"] = format(p, "x")
        fd["recover_c"] = gen_recover_c(ref, f, np.random.default_rng(77 + f))
        L = O.LIMBS[f]
        rng = np.random.default_rng(99 + f)
        pk = []
        for (W, n, t, N, seed) in ((2, 4, 3, 5, b"pedersen"), (3, 5, 2, 3, b"array3")):
            sec = ref.from_bytes(f, rng.bytes(8 * L * W * N)).reshape(N, W, L)
            pk.append({"W": W, "n": n, "t": t, "seed": seed.hex(), "secrets": hx(sec.reshape(-1, L)),
                       "shares": hx(ref.shamir_share_packed(f, seed, sec, t, n).reshape(-1, L))})
        fd["shamir_packed"] = pk
        # BASELINE configs[2]'s own shape beside the generic sections: (10,3) and (40,13) on more secrets
        rng = np.random.default_rng(128)
        c3 = []
        for (n, t, N, seed) in ((10, 3, 64, b"scl-bench-c3"), (40, 13, 16, b"c3-40-13")):
            secrets = ref.from_bytes(f, rng.bytes(16 * N))
            shares = ref.shamir_share(f, seed, secrets, t, n)
            c3.append({"n": n, "t": t, "seed": seed.hex(), "secrets": hx(secrets), "shares": hx(shares),
                       "recovered_all_n": hx(ref.shamir_recover(f, shares))})
        fd["shamir_c3"] = c3
        doc["fields"][name] = fd
    path = os.path.join(HERE, "golden_mont128.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
