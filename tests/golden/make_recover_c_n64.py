#!/usr/bin/env python3
"""tests/golden/golden_recover_c_n64.json: shamirRecoverC over the two 256-bit fields at n = 64 (t = 21), the two cases of
tests/test_gpu_parity.py::test_recover_correct_vs_oracle that the oracle needs minutes for.  The inputs are rebuilt by the
test from the same seeds; this file holds what the REFERENCE (oracle/_ref, compiled from /root/reference) returns for them --
corrected polynomial, error locator, status and error count per secret.  Run here once (needs /root/reference built into
oracle/_ref); the result is data, committed."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def case_inputs(port, f, n, t, N):
    """exactly what test_recover_correct_vs_oracle builds (keep the two in step)"""
    L = O.LIMBS[f]
    rng = np.random.default_rng(n * 100 + t)

    def rand_elems(count, seed):
        return port.vector_random(f, seed, count)
    secrets = rand_elems(N, b"bw-s")
    coeffs = rand_elems(max(t, 1) * N, b"bw-c").reshape(N, max(t, 1), L)[:, :t]
    nodes = np.stack([port.from_int(f, i + 1) for i in range(n)])
    shares = np.stack([port.poly_eval(f, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)])
    junk = rand_elems(N * n, b"bw-j").reshape(N, n, L)
    nbad = np.zeros(N, dtype=int)
    for s in range(N):
        k = 0 if s % 3 == 0 else int(rng.integers(0, t + 3))
        nbad[s] = k
        for i in rng.choice(n, size=min(k, n), replace=False):
            shares[s, i] = junk[s, i]
    return secrets, shares, nbad


def main():
    try:
        lib, kind = O.Ref(), "reference (oracle/_ref)"
    except Exception:
        lib, kind = O.Port(), "oracle port"
    port = O.Port()
    out = {"_generator": "tests/golden/make_recover_c_n64.py", "_source": kind, "cases": []}
    for f, name in ((O.SECP256K1_SCALAR, "secp256k1_order"), (O.SECP256K1_FIELD, "secp256k1_field")):
        n, t, N = 64, 21, 2
        secrets, shares, nbad = case_inputs(port, f, n, t, N)
        t0 = time.time()
        fo, eo, st, ne = lib.shamir_recover_c(f, shares, None)
        print(f"{name}: {time.time() - t0:.1f} s by the {kind}", file=sys.stderr)
        out["cases"].append({"field": name, "n": n, "t": t, "N": N, "nbad": nbad.tolist(),
                             "shares_sha": __import__("hashlib").sha256(shares.tobytes()).hexdigest(),
                             "f": [hex(v) for v in O.to_ints(fo)], "err": [hex(v) for v in O.to_ints(eo)],
                             "status": st.tolist(), "nerr": ne.tolist()})
    with open(os.path.join(HERE, "golden_recover_c_n64.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
