#!/usr/bin/env python3
"""bench.py -- Shamir reconstructions/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: shamirSecretShare of N secrets
(coefficients resident in HBM) followed by shamirRecoverP of all N from all n shares, for the
configuration BASELINE.json quotes the metric on: Shamir (n=10, t=3) over Mersenne61, 100 M
secrets per GPU.  Inputs are generated on the device before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the batch of independent secrets shards across ranks (rank r owns its own N secrets,
weak scaling, no data-path collective); time = max over ranks.

Prints ONE JSON line on rank 0 (fields described in DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "secure-computation-library_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

FIELD_TAGS = {"m61": 0, "m127": 1, "mont128": 2, "gf2_128": 3, "secp256k1": 4}
FIELD_NAMES = {"m61": "Mersenne61", "m127": "Mersenne127", "mont128": "Mont128", "gf2_128": "GF(2^128)",
               "secp256k1": "secp256k1_order"}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable)


def cpu_baseline(field_key, n, t, sample):
    """The reference CPU path on this box's host cores (single thread, like SCL itself): per secret
    shamirSecretShare + shamirRecoverP(shares).  oracle/_ref (the real reference, prebuilt) when it
    is there, else the oracle port."""
    import oracle_lib as O
    kind = "reference"
    try:
        lib = O.Ref()
    except Exception:
        lib, kind = O.Port(), "port"
    f = FIELD_TAGS[field_key]
    r = lib.time_shamir(f, sample, t, n)
    if r["mismatches"]:
        raise RuntimeError("CPU baseline failed its own round trip")
    total = r["share_s"] + r["recover_s"]
    out = {
        "value": sample / total, "unit": "reconstructions/s", "cores": 1, "kind": kind,
        "sample": f"{sample} secrets, per-secret shamirSecretShare + shamirRecoverP (n={n}, t={t}), "
                  f"share {r['share_s']:.2f}s + recover {r['recover_s']:.2f}s",
        "recover_only_per_s": sample / r["recover_s"], "share_only_per_s": sample / r["share_s"],
    }
    # SURVEY.md section 8d asks for two more figures beside the faithful single-thread run (SCL itself is single-threaded;
    # the threads below are this harness's, one PRG and one slab of secrets each):
    #   all_cores  the same per-secret path on the host cores that go with one GPU (at most 16 threads)
    #   hoisted    one thread, Lagrange basis computed once instead of per secret (reference library only)
    try:
        from concurrent.futures import ThreadPoolExecutor
        cores = min(len(os.sched_getaffinity(0)), 16)   # the CPU share that goes with one GPU on the bench boxes
        per = max(1, sample // 4)          # a quarter of the sample per thread keeps the leg to a few seconds
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:   # ctypes releases the GIL for the duration of each call
            rs = list(ex.map(lambda i: lib.time_shamir(f, per, t, n, b"scl-bench-%d" % i), range(cores)))
        wall = time.perf_counter() - t0
        if not any(x["mismatches"] for x in rs):
            out["all_cores"] = {"value": per * cores / wall, "cores": cores,
                                "sample": f"{per} secrets on each of {cores} threads, wall {wall:.2f}s"}
    except Exception as e:  # the extra legs never fail the bench line
        out["all_cores"] = {"error": str(e)}
    if kind == "reference":
        try:
            hr = lib.time_shamir_hoisted(f, sample, t, n)
            if not hr["mismatches"]:
                out["hoisted_basis"] = {"value": sample / (hr["share_s"] + hr["recover_s"]), "cores": 1,
                                        "recover_only_per_s": sample / hr["recover_s"]}
        except Exception as e:
            out["hoisted_basis"] = {"error": str(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--field", default="m61", choices=sorted(FIELD_TAGS))
    ap.add_argument("--n", type=int, default=10)
    ap.add_argument("--t", type=int, default=3)
    ap.add_argument("--secrets", type=int, default=100_000_000, help="secrets per GPU")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="secrets timed on the CPU baseline (0 = skip)")
    ap.add_argument("--placement-probes", type=int, default=10,
                    help="positions of the share matrix inside one HBM arena to try before the warm-up, each with its best "
                         "output and input slots (0 = plain allocations)")
    ap.add_argument("--share-mode", default="coeffs", choices=["coeffs", "prg"],
                    help="coeffs: polynomial coefficients resident in HBM; prg: AES-CTR PRG inside the share kernel")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import scl_amd as scl

    f = FIELD_TAGS[args.field]
    L = scl.limbs(f)
    E = 8 * L
    n, t, N = args.n, args.t, args.secrets

    # ---- synthetic inputs, generated on the device (uniform field elements from the AES-CTR PRG) ----
    seed = f"scl-bench-{args.field}-{rank}".encode()
    lam = scl.lagrange_basis(f, n)
    tc = t if args.share_mode == "coeffs" else 0
    rowN = N * L                                   # int64 words per row of N elements
    blocks_per_row = (N * E + 15) // 16

    def fill_inputs(secrets, coeffs):
        secrets.copy_(scl.vector_random(f, N, seed + b"-secrets"))
        for k in range(tc):
            coeffs[k].copy_(scl.vector_random(f, N, seed + b"-coeffs", counter0=k * blocks_per_row))

    def run_share(secrets, coeffs, dst):
        if args.share_mode == "coeffs":
            scl.shamir_share(f, secrets, coeffs if tc else None, n, out=dst)
        else:
            scl.shamir_share_prg(f, secrets, t, n, seed, out=dst)

    # ---- where the operands land in HBM.  The same two kernels run up to 10 % apart depending on which physical
    # region the share matrix, the inputs and the reconstruct output occupy (tools/probe_placement*.py: it follows
    # the region, not the row stride or small offsets; read-heavy and write-heavy kernels prefer different
    # arrangements).  So the operands are carved out of one arena at a few spread-out arrangements, each is timed for
    # two passes, and the fastest is kept -- all before the warm-up; --placement-probes 0 allocates plainly.
    placement = None
    GiB = 1 << 30
    up = lambda x: (x + 4095) // 4096 * 4096       # operands start on 4 KiB boundaries and never overlap
    m_bytes, i_bytes, o_bytes = up(n * rowN * 8), up((1 + tc) * rowN * 8), up(rowN * 8)
    need = m_bytes + i_bytes + o_bytes
    arena = None
    if args.placement_probes > 0:
        free_b, _total = torch.cuda.mem_get_info()
        arena_bytes = min(int(free_b * 0.6), max(4 * need, 44 * GiB)) // 4096 * 4096
        if arena_bytes >= 2 * need + 3 * GiB:
            try:
                arena = torch.empty(arena_bytes // 8, dtype=torch.int64, device="cuda")
            except RuntimeError:   # no room for the arena (another tenant on the card): plain allocations below
                arena = None
                torch.cuda.empty_cache()

    def carve(off_bytes, rows):
        assert off_bytes % 4096 == 0
        off = off_bytes // 8
        return arena[off: off + rows * rowN].view(rows, N, L)

    if arena is not None:
        top = arena.numel() * 8
        tm = scl.Timer()

        def timed(fn):
            fn()
            tm.start()
            fn()
            fn()
            tm.stop()
            return tm.elapsed_ms() / 2

        def slots(size, taken, count):
            """up to `count` evenly spread offsets for `size` bytes that avoid the `taken` (offset, size) ranges"""
            out_ = []
            for k in range(count + 2):
                c = min(up(int(top * k / (count + 1))), top - size)
                if all(c + size <= a_ or c >= a_ + b_ for a_, b_ in taken) and c not in out_:
                    out_.append(c)
            return out_

        # The two kernels care about different pairs: reconstruct about (matrix, output), share about (inputs, matrix).
        # For a few positions of the matrix, the best output slot is found with the reconstruct kernel alone and then
        # the best input slot with the share kernel alone; the matrix position with the smallest sum is kept.
        trials = []
        src = scl.empty(f, 1 + tc, N)
        fill_inputs(src[0], src[1:])
        n_m = max(2, min(12, args.placement_probes))
        for mo in slots(m_bytes, [], n_m)[:n_m + 2]:
            M = carve(mo, n)
            run_share(src[0], src[1:], M)                      # valid shares for the reconstruct trials
            best_o = best_i = None
            for oo in slots(o_bytes, [(mo, m_bytes)], 6):
                O_ = carve(oo, 1)[0]
                x = timed(lambda: scl.shamir_recover(f, M, lam, out=O_))
                if best_o is None or x < best_o[0]:
                    best_o = (x, oo)
            for io in slots(i_bytes, [(mo, m_bytes), (best_o[1], o_bytes)], 6):
                inp = carve(io, 1 + tc)
                inp.copy_(src)
                x = timed(lambda: run_share(inp[0], inp[1:], M))
                if best_i is None or x < best_i[0]:
                    best_i = (x, io)
            trials.append((best_o[0] + best_i[0], mo, best_i[1], best_o[1], best_i[0], best_o[0]))
        trials.sort()
        _, mo, io, oo, _, _ = trials[0]
        inp = carve(io, 1 + tc)
        inp.copy_(src)
        del src
        secrets, coeffs, shares, out = inp[0], (inp[1:] if tc else None), carve(mo, n), carve(oo, 1)[0]
        placement = {"arena_GiB": round(top / GiB, 1),
                     "matrix_positions": [{"matrix_GiB": round(t_[1] / GiB, 1), "inputs_GiB": round(t_[2] / GiB, 1),
                                           "output_GiB": round(t_[3] / GiB, 1), "share_ms": round(t_[4], 4),
                                           "recover_ms": round(t_[5], 4)} for t_ in trials],
                     "chosen": 0}
    else:
        secrets = scl.empty(f, N)
        coeffs = scl.empty(f, tc, N) if tc else None
        fill_inputs(secrets, coeffs)
        shares = scl.empty(f, n, N)
        out = scl.empty(f, N)

    def step(timers=None):
        if timers:
            timers[0].start()
        run_share(secrets, coeffs, shares)
        if timers:
            timers[0].stop()
            timers[1].start()
        scl.shamir_recover(f, shares, lam, out=out)
        if timers:
            timers[1].stop()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    timers = [(scl.Timer(), scl.Timer()) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(timers[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- per-kernel durations from the HIP events recorded inside the timed region -------------------
    share_ms = sum(tm[0].elapsed_ms() for tm in timers) / max(1, args.steps)
    rec_ms = sum(tm[1].elapsed_ms() for tm in timers) / max(1, args.steps)
    verified = bool(scl.equals(f, out, secrets))

    # measured copy bandwidth of the same device (read+write bytes / time), for context
    probe_bytes = min(4 << 30, shares.numel() * 8 // 2) & ~15
    src = shares.view(-1)[: probe_bytes // 8]
    dst = shares.view(-1)[probe_bytes // 8: 2 * (probe_bytes // 8)]
    scl.stream_copy(dst, src)
    tm = scl.Timer()
    tm.start()
    for _ in range(5):
        scl.stream_copy(dst, src)
    tm.stop()
    copy_gbps = 2 * probe_bytes * 5 / (tm.elapsed_ms() * 1e-3) / 1e9

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    rec_bytes = (n + 1) * E            # n shares in, 1 secret out            
    share_bytes = (1 + t) * E + n * E if args.share_mode == "coeffs" else E + n * E
    kernels = {
        "shamir_recover": {"ms": rec_ms, "bytes_per_secret": rec_bytes, "GBps": rec_bytes * N / (rec_ms * 1e-3) / 1e9},
        "shamir_share": {"ms": share_ms, "bytes_per_secret": share_bytes,
                         "GBps": share_bytes * N / (share_ms * 1e-3) / 1e9},
    }
    dom = "shamir_share" if share_ms >= rec_ms else "shamir_recover"
    ach = kernels[dom]["GBps"]
    # HBM traffic per launch of the dominant kernel: PMC counters collected offline with rocprofv3 on this
    # exact configuration (bench.py cannot read PMCs itself); null for any other configuration.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        c = pmc["config"]
        if (c["field"], c["n"], c["t"], c["secrets_per_gpu"], c["share_mode"]) == (args.field, n, t, N, args.share_mode):
            traffic = pmc[dom]["bytes"]
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes": kernels[dom]["bytes_per_secret"] * N,
                "measured_copy_GBps": copy_gbps, "frac_of_measured_copy": ach / copy_gbps}
    total = N * world * args.steps
    line = {
        "metric": "shamir_reconstructions_per_sec", "value": total / elapsed, "unit": "reconstructions/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {1: "u64", 2: "u128", 4: "u256"}[L], "data": "synthetic",
        "config": {"workload": f"shamir_share+reconstruct n={n} t={t} {FIELD_NAMES[args.field]} "
                               f"{N} secrets/GPU (BASELINE configs[1])" if (n, t, args.field, N) == (10, 3, "m61", 100_000_000)
                   else f"shamir_share+reconstruct n={n} t={t} {FIELD_NAMES[args.field]} {N} secrets/GPU",
                   "field": FIELD_NAMES[args.field], "n": n, "t": t, "secrets_per_gpu": N,
                   "share_mode": args.share_mode, "layout": "SoA [party][secret]", "parallelism": f"shard{world}"},
        "roofline": roofline, "kernels": kernels, "verified": verified, "placement_probe": placement,
        "reconstruct_only_per_s": N * world / (rec_ms * 1e-3), "share_only_per_s": N * world / (share_ms * 1e-3),
    }
    if world == 1 and args.cpu_sample > 0:
        line["cpu_baseline"] = cpu_baseline(args.field, n, t, args.cpu_sample)
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
