#!/usr/bin/env python3
"""The open step behind the C ABI over REAL RCCL: one process per GPU (started by torch.distributed.run), every rank on its own
device.  scl_hip_open_all_gather / scl_hip_open_partial_gather / scl_hip_open_reduce_scatter (csrc/open_rccl.inc) and their
torch.distributed twins (scl_amd/dist.py) against the CPU oracle: the shares are the oracle's Polynomial::evaluate of seeded
polynomials (every rank computes the same ones and keeps its parties' rows), every rank's every output must equal the secrets.
The in-tree tests run this code over a stand-in (tests/cxx/fake_rccl.cc, ranks = threads on one GPU): this is the first
contact with ncclGroupStart / ncclAllGather / ncclReduceScatter of a communicator with more than one rank.

Reference of the exchange: Network::send + Network::recv to / from every party (include/scl/net/network.h:148-152,178-185;
test/scl/protocol/beaver.h:43-55).

    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P \\
        tools/open_rccl_check.py [--secrets 100000] [--chunk 30000]
    (--backend gloo --one-device: the same code with both ranks on GPU 0 and the library's RCCL replaced by the stand-in named in
     SCL_HIP_RCCL_LIBRARY -- a rehearsal of this script, not of RCCL)

Rank 0 prints one JSON line; exit code 0 iff every rank agreed on every form."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "secure-computation-library_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--secrets", type=int, default=100_000)
    ap.add_argument("--chunk", type=int, default=30_000)      # four chunks, the last one short
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--one-device", action="store_true")
    args = ap.parse_args()
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist

    dev = 0 if args.one_device else local_rank
    torch.cuda.set_device(dev)
    dist.init_process_group(args.backend, device_id=torch.device("cuda", dev) if args.backend == "nccl" else None)
    import oracle_lib as O
    import scl_amd as scl
    from scl_amd import dist as sd

    port = O.Port()
    N, chunk = args.secrets, args.chunk
    report, ok_all = {}, True
    comm = sd.Communicator()
    try:
        for field, name, n, t in ((O.M61, "Mersenne61", 10, 3), (O.M127, "Mersenne127", 10, 3), (O.GF2_128, "GF(2^128)", 40, 13)):
            L = O.LIMBS[field]
            secrets = port.vector_random(field, b"first-contact-secrets", N)
            # PRG-driven sharing on the device, identical on every rank (same seed); checked against the oracle on a window
            dsec = scl.to_device(secrets)
            full = scl.shamir_share_prg(field, dsec, t, n, b"first-contact")                 # [n][N][L]
            if field != O.GF2_128:   # (the oracle's GF(2^128) nodes follow the x++ walk; the round trip below covers that field)
                w = slice(0, 64)
                want = port.shamir_share(field, b"first-contact", secrets[w], t, n)
                assert np.array_equal(scl.to_host(full[:, w]), np.ascontiguousarray(np.transpose(want, (1, 0, 2))))
            lam = scl.lagrange_basis(field, n)
            per = sd.parties_per_rank(n, world)
            first, cnt = sd.party_slab(n, rank, world)
            local = torch.full((per, N, L), -1, dtype=torch.int64, device="cuda")            # padding rows: never to be read
            if cnt:
                local[:cnt].copy_(full[first:first + cnt])
            mine = local[:cnt].contiguous()
            del full
            forms = {"c_abi_all_gather": lambda: sd.open_all_gather_c(comm, field, local, n, lam, chunk=chunk),
                     "c_abi_partial_gather": lambda: sd.open_partial_gather_c(comm, field, mine, lam[first:first + cnt], chunk=chunk),
                     "torch_all_gather": lambda: sd.open_and_reconstruct(field, local, n, lam, chunk=chunk),
                     "torch_partial_gather": lambda: sd.open_by_partial_gather(field, mine, lam[first:first + cnt], chunk=chunk)}
            if field == O.M61 and world <= 8:
                forms["c_abi_reduce_scatter"] = lambda: sd.open_reduce_scatter_c(comm, field, mine, lam[first:first + cnt], chunk=chunk)
            res = {}
            for key, fn in forms.items():
                try:
                    out = fn()
                    torch.cuda.synchronize()
                    good = bool(np.array_equal(scl.to_host(out), secrets))
                except Exception as e:   # noqa: BLE001
                    good, res[key + "_error"] = False, f"{type(e).__name__}: {e}"
                flag = torch.tensor([1 if good else 0], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank must agree
                res[key] = bool(flag.item())
                ok_all = ok_all and res[key]
            if field == O.M61 and world <= 8 and N % world == 0:
                try:
                    sl = sd.open_by_partial_sums(mine, lam[first:first + cnt])
                    lo = rank * (N // world)
                    good = bool(np.array_equal(scl.to_host(sl.reshape(-1, 1)), secrets[lo:lo + N // world]))
                except Exception as e:   # noqa: BLE001
                    good, res["torch_reduce_scatter_error"] = False, f"{type(e).__name__}: {e}"
                flag = torch.tensor([1 if good else 0], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                res["torch_reduce_scatter"] = bool(flag.item())
                ok_all = ok_all and res["torch_reduce_scatter"]
            report[name] = dict(res, n=n, t=t, parties_per_rank=per)
    finally:
        comm.close()
    devs = [None] * world
    dist.all_gather_object(devs, int(torch.cuda.current_device()))
    if rank == 0:
        print(json.dumps({"ok": ok_all, "world": world, "backend": dist.get_backend(), "devices": devs, "secrets": N, "chunk": chunk,
                          "fields": report}), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok_all else 1)


if __name__ == "__main__":
    main()
