"""Child process of tests/test_gpu_parity.py::test_c_abi_open_with_a_world_of_threads: the open step behind the C ABI
(scl_hip_open_all_gather / scl_hip_open_partial_gather / scl_hip_open_reduce_scatter, csrc/open_rccl.inc) with world > 1 on ONE GPU.

The ranks are host threads of this process, all on device 0; the library is told (SCL_HIP_RCCL_LIBRARY, set by the
caller before this process starts) to bind tests/cxx/_build/libfake_rccl.so instead of RCCL, whose all-gather is a
rendezvous of the ranks' threads plus device-to-device copies.  Everything else is the code eight ranks run: the
permuted lambda, one grouped all-gather per party row, the communicator's stream and events, padding rows, ranks
without parties, the partial-sum form.  The CPU oracle is the checker: the shares come from its Polynomial::evaluate,
every rank's output must equal the secrets it shared, and its own shamirRecoverP (hoisted basis) must agree.

Reference of the exchange: Network::send + Network::recv to / from every party
(/root/reference/include/scl/net/network.h:148-152,178-185; test/scl/protocol/beaver.h:43-55).

usage: open_world_check.py WORLD FIELD n t N CHUNK   -> prints one JSON line, exit code 0 iff every rank agreed"""
import ctypes as C
import json
import os
import sys
import threading

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "secure-computation-library_amd"))
sys.path.insert(0, HERE)
import oracle_lib as O  # noqa: E402
import scl_amd as scl  # noqa: E402
from scl_amd import dist as sd  # noqa: E402


def main():
    world, field, n, t, N, chunk = (int(x) for x in sys.argv[1:7])
    assert os.environ.get("SCL_HIP_RCCL_LIBRARY"), "the caller names the stand-in library"
    port = O.Port()
    L = O.LIMBS[field]
    lib = scl.lib
    # --- the oracle's shares: f_s(alpha_i) for a random polynomial with f_s(0) = secret_s, nodes 1..n (FF(int) images)
    secrets = port.vector_random(field, b"world-secrets", N)
    secrets[0] = port.from_int(field, -1)
    coeffs = port.vector_random(field, b"world-coeffs", N * t).reshape(N, t, L)
    nodes = O.from_ints(list(range(1, n + 1)), L)
    aos = np.stack([port.poly_eval(field, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)])  # [N][n][L]
    full = np.ascontiguousarray(np.transpose(aos, (1, 0, 2)))                                                      # [n][N][L]
    lam = port.lagrange_basis(field, nodes, port.from_int(field, 0))
    assert np.array_equal(port.shamir_recover_lambda(field, aos, lam), secrets)
    per = sd.parties_per_rank(n, world)
    order = sd.open_row_order(n, world)
    assert len(order) == per * world and sorted(p for p in order if p >= 0) == list(range(n))

    reduce_scatter = field == O.M61 and world <= 8
    ident = (C.c_ubyte * 128)()
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")  # the context exists before the threads start
    st = lib.scl_hip_comm_unique_id(ident)
    assert st == 0, lib.scl_hip_last_error()
    results, errors = {}, []

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            h = C.c_void_p()
            assert lib.scl_hip_comm_init_rank(C.byref(h), world, r, ident) == 0, lib.scl_hip_last_error()
            w, me = C.c_int(), C.c_int()
            assert lib.scl_hip_comm_info(h, C.byref(w), C.byref(me)) == 0 and (w.value, me.value) == (world, r)
            first, cnt = sd.party_slab(n, r, world)
            # the rank's slab: its parties' rows, then padding rows full of ones (never to be read, let alone sent)
            slab = np.full((per, N, L), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
            slab[:cnt] = full[first:first + cnt]
            local = torch.from_numpy(slab.view(np.int64)).cuda()
            sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
            outs = [torch.full((N, L), -1, dtype=torch.int64, device="cuda") for _ in range(6 if reduce_scatter else 4)]
            torch.cuda.synchronize()
            lam_h = np.ascontiguousarray(lam)
            lam_p = lam_h.ctypes.data_as(C.c_void_p)
            lam_mine = np.ascontiguousarray(lam[first:first + cnt])
            mine_p = lam_mine.ctypes.data_as(C.c_void_p) if cnt else None
            dp = lambda ten: C.c_void_p(ten.data_ptr())  # noqa: E731
            # two opens back to back on DIFFERENT streams of the same handle, nothing waited for in between: the second
            # call's first gathers must not overwrite buffers the first call's reconstruct still reads
            for out, s in ((outs[0], sA), (outs[1], sB)):
                rc = lib.scl_hip_open_all_gather(h, field, dp(out), dp(local), C.c_size_t(N), C.c_size_t(n), lam_p,
                                                 C.c_size_t(N), C.c_size_t(chunk), C.c_void_p(s.cuda_stream))
                assert rc == 0, lib.scl_hip_last_error()
            for out, s in ((outs[2], sB), (outs[3], sA)):
                rc = lib.scl_hip_open_partial_gather(h, field, dp(out), dp(local) if cnt else None, C.c_size_t(N),
                                                     C.c_size_t(cnt), mine_p, C.c_size_t(N), C.c_size_t(chunk),
                                                     C.c_void_p(s.cuda_stream))
                assert rc == 0, lib.scl_hip_last_error()
            if reduce_scatter:
                # Mersenne61, world <= 8: the partial sums through ncclReduceScatter as plain 64-bit sums, slices folded mod p;
                # outs[4] every secret on every rank, outs[5] each secret only on its owner (the rest stays -1)
                for out, s, every in ((outs[4], sA, 1), (outs[5], sB, 0)):
                    rc = lib.scl_hip_open_reduce_scatter(h, field, dp(out), dp(local) if cnt else None, C.c_size_t(N),
                                                         C.c_size_t(cnt), mine_p, C.c_size_t(N), C.c_size_t(chunk), every,
                                                         C.c_void_p(s.cuda_stream))
                    assert rc == 0, lib.scl_hip_last_error()
            if field == O.M61 and world > 8:
                # nine canonical partials could wrap a 64-bit sum: the reduce-scatter form refuses, before any collective is issued
                rc = lib.scl_hip_open_reduce_scatter(h, field, dp(outs[0]), dp(local) if cnt else None, C.c_size_t(N), C.c_size_t(cnt),
                                                     mine_p, C.c_size_t(N), C.c_size_t(chunk), 1, C.c_void_p(sA.cuda_stream))
                assert rc != 0 and b"8 ranks" in lib.scl_hip_last_error(), (rc, lib.scl_hip_last_error())
            sA.synchronize()
            sB.synchronize()
            results[r] = [o.cpu().numpy().view(np.uint64) for o in outs]
            assert lib.scl_hip_comm_destroy(h) == 0
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=240)
    hung = [i for i, th in enumerate(threads) if th.is_alive()]
    ok = not errors and not hung and len(results) == world
    bad = []
    if ok:
        owner = np.array([sd.slice_owner(s_, N, world, chunk) for s_ in range(N)])
        for r in range(world):
            for k, got in enumerate(results[r]):
                want = secrets
                if k == 5:  # reconstruct-to-one-owner: this rank's slices hold the secrets, every other slot is untouched
                    want = np.where((owner == r)[:, None], secrets, np.uint64(0xFFFFFFFFFFFFFFFF))
                if not np.array_equal(got, want):
                    bad.append((r, k, int((got != want).any(axis=1).sum())))
        ok = not bad
    print(json.dumps({"ok": ok, "world": world, "field": field, "n": n, "t": t, "N": N, "chunk": chunk, "per": per,
                      "chunks": -(-N // ((min(chunk, N) + 1) & ~1)), "ranks_without_parties": sum(1 for r in range(world) if sd.party_slab(n, r, world)[1] == 0),
                      "padding_rows": per * world - n, "reduce_scatter": reduce_scatter, "errors": errors, "hung": hung, "mismatches": bad}))
    sys.stdout.flush()
    os._exit(0 if ok else 1)  # (daemon threads stuck at a rendezvous must not keep the process)


if __name__ == "__main__":
    main()
