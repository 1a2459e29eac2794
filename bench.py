#!/usr/bin/env python3
"""bench.py -- Shamir reconstructions/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: shamirSecretShare of N secrets
(coefficients resident in HBM) followed by shamirRecoverP of all N from all n shares, for the
configuration BASELINE.json quotes the metric on: Shamir (n=10, t=3) over Mersenne61, 100 M
secrets per GPU.  Inputs are generated on the device before the timed region, in plain allocations.

    python bench.py                       one GPU, one JSON line
    python bench.py --gpus N              starts N ranks itself (a child `python -m torch.distributed.run`,
                                          before this process has touched torch or the GPU) and relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N     (what the driver does)
    python bench.py --mode open [--gpus N]   times only the cross-party "open" step (RCCL all-gather + reconstruct)
    python bench.py --gpus 8 --config c4     BASELINE configs[3]: 10^8 GF(2^128) secrets (40,13) in all, the open step
                                             (all-gather of the party slabs + reconstruct on every rank) is the timed step
    python bench.py --gpus 8 --config c5     BASELINE configs[4]: 10^9 Mersenne61 secrets (128,42) in all, split over the
                                             ranks; share (matrix cores) + reconstruct of the rank's shard per step

Multi-GPU: the batch of independent secrets shards across ranks (rank r owns its own N secrets,
weak scaling, no data-path collective); time = max over ranks.  The one exchange step of the path, the MPC
"open" (reference: Network::send + Network::recv, include/scl/net/network.h:148-152,178-185), is timed after
the headline region.

Output: ONE compact JSON line on rank 0's stdout (the contract's keys, `roofline`, `kernels`, `cpu_baseline`, `verified`,
`verified_legs`, a one-number-per-leg digest `legs`; < 8 KB, asserted) and everything the legs measured in the DETAIL FILE
(--detail, default bench_detail.json next to this script): the element-wise path, the layout bridge, Matrix::multiply, the
other BASELINE configurations, the open step, the PRG-driven mode, additive sharing, the full CPU baseline, the compute
rooflines of the legs HBM does not bound.  The legs live in bench_legs/ (one module each, explicit context); this file owns
the arguments, the launcher, the headline's timed region and the line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bench_legs.common import (C5_PER_GPU_CAP, DTYPES, FIELD_NAMES, FIELD_TAGS, HBM_PEAK_GBPS, I8_PEAK_TOPS, Ctx,  # noqa: E402,F401
                               mfma_share_roofline, on_matrix_cores, side_legs, sig)
from bench_legs.cpu import cgroup_cpu_limit, cpu_baseline, cpu_model, physical_cores  # noqa: E402,F401
from bench_legs.pmc import (KERNEL_SOURCES, PMC_CONFIGS, PMC_NEEDLES, kernel_source_hash, live_pmc_traffic,  # noqa: E402,F401
                            pmc_config_traffic, pmc_means, pmc_report, pmc_traffic)

LINE_LIMIT = 8000        # bytes of the result line; the driver's record of round 5 lost a 20 KB line
DEFAULT_DETAIL = os.path.join(ROOT, "bench_detail.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="path", choices=["path", "open"],
                    help="path: share + reconstruct per step (the headline); open: only the cross-party open step")
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5"],
                    help="c2 (default): BASELINE configs[1], weak scaling; c4 / c5: the configurations BASELINE quotes on 8 GPUs, "
                         "their total split over the ranks (strong scaling)")
    ap.add_argument("--field", default="m61", choices=sorted(FIELD_TAGS))
    ap.add_argument("--n", type=int, default=10)
    ap.add_argument("--t", type=int, default=3)
    ap.add_argument("--secrets", type=int, default=100_000_000, help="secrets per GPU")
    ap.add_argument("--cpu-sample", type=int, default=40_000_000,
                    help="most secrets the CPU baseline may time (0 = skip); a pilot bounds it to about 10 s of work")
    ap.add_argument("--cpu-all-cores", type=int, default=0,
                    help="1: beside the single-thread reference run, the harness-threaded (all physical cores, 16 per GPU) and "
                         "hoisted-basis variants -- three more bounded legs, into the detail file")
    ap.add_argument("--share-mode", default="coeffs", choices=["coeffs", "prg"],
                    help="coeffs: polynomial coefficients resident in HBM; prg: AES-CTR PRG inside the share kernel")
    ap.add_argument("--configs", type=int, default=1,
                    help="1: after the headline, time the other BASELINE configurations at per-GPU shard size (one GPU only)")
    ap.add_argument("--open", type=int, default=1, help="1: after the headline, time the open step (all ranks)")
    ap.add_argument("--open-secrets", type=int, default=0,
                    help="secrets opened per step in the C4 shape (default 12.5 M x ranks: 10^8 at 8 ranks)")
    ap.add_argument("--open-chunk", type=int, default=1 << 24, help="secrets per all-gather")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --dry-run: the launcher / rendezvous / timing skeleton on CPU (tests)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: steps are empty (launcher test)")
    ap.add_argument("--total-secrets", type=int, default=0, help="--config c4 / c5: total over all ranks (default 10^8 / 10^9)")
    ap.add_argument("--c4-rank-secrets", type=int, default=100_000_000,
                    help="open.c4_rank_shape: secrets of the one-rank-of-eight shape of BASELINE configs[3] (0 = skip)")
    ap.add_argument("--allocations", type=int, default=0,
                    help="independently allocated operand sets the headline's steps rotate over (0 = three for BASELINE configs[1] at "
                         "up to 10^8 secrets per GPU, one otherwise): where the operands land moves the kernels by up to 10 percent")
    ap.add_argument("--ew", type=int, default=1,
                    help="1: after the headline, the element-wise add / mul / inverse path (Mersenne61 10^8, the 16-byte fields 10^7), "
                         "the layout bridge and Matrix::multiply")
    ap.add_argument("--ew-elements", type=int, default=0,
                    help="tests only: run the element-wise and layout legs on this many elements per field instead of 10^8 / 10^7")
    ap.add_argument("--pmc-live", type=int, default=0,
                    help="1: roofline.traffic observed in THIS run -- two child runs of the headline under rocprofv3 --pmc FETCH_SIZE / "
                         "WRITE_SIZE after everything else (one GPU, BASELINE configs[1] only; about 10 s).  Default: the stamped "
                         "figures of profiles/pmc_traffic.json, valid while the kernel sources hash the same")
    ap.add_argument("--side-timeout", type=float, default=300.0,
                    help="multi-rank runs: seconds the legs after the headline (RCCL through torch and through the C ABI) may take "
                         "before rank 0 writes the line with what it has and every rank exits (0 = no limit)")
    ap.add_argument("--detail", default=DEFAULT_DETAIL,
                    help="where rank 0 writes the detail file (everything the legs measured); '' = nowhere")
    ap.add_argument("--inject-error", default="", choices=["", "c_abi", "hang"],
                    help="tests only: make the named side leg fail, to see the line report it and the exit code follow")
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N > 1 started plainly: start the N ranks as a CHILD process and relay rank 0's line.  Runs before torch is
    imported, so this process never initialises the GPU (a process that has must not start or become another)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver supports only dmabuf IPC; without it RCCL's peer-buffer exchange
    # fails with "hipIpcGetMemHandle: invalid argument".  The image exports it already (so a driver that starts torchrun
    # itself has it too); it is set here only so that a bare environment behaves the same.  An explicit value wins.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = []
    for ln in child.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln:
            lines.append(ln)
        elif ln:
            print(ln, file=sys.stderr, flush=True)
    rc = child.wait()
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: expected one result line from rank 0, got {len(lines)}", file=sys.stderr)
        rc = 1
    for ln in lines[-1:]:
        print(ln, flush=True)
    sys.exit(rc)


def plan(args, world, rank):
    """What `--config` asks of this rank: field, shape, this rank's secrets, the total, and the strings of the result line.
    c2 is weak scaling (--secrets per GPU); c4 and c5 split BASELINE's totals over the ranks (strong scaling)."""
    from math import ceil
    if args.config == "c4":
        total = args.total_secrets or 100_000_000
        per = ceil(40 / world)
        return {"key": "c4", "field": "gf2_128", "n": 40, "t": 13, "total": total, "mine": total, "scaling": "strong",
                "dtype": "u128", "parallelism": f"parties{world}",
                "workload": f"open step of Shamir (n=40,t=13) over GF(2^128), {total} secrets in all: all-gather of the ranks' "
                            f"party slabs ({per} parties per rank) + reconstruct on every rank (BASELINE configs[3])"}
    if args.config == "c5":
        full = args.total_secrets or 1_000_000_000
        total = min(full, C5_PER_GPU_CAP * world)      # fewer than 8 GPUs cannot hold 10^9 x (128 + 42 + 2) x 8 bytes
        base, rem = divmod(total, world)
        mine = base + (1 if rank < rem else 0)
        return {"key": "c5", "field": "m61", "n": 128, "t": 42, "total": total, "mine": mine,
                "scaling": "strong" if total == full else "weak", "dtype": "u64", "parallelism": f"shard{world}",
                "workload": f"shamir_share (Vandermonde x coefficient matrix on the matrix cores) + reconstruct n=128 t=42 "
                            f"Mersenne61, {total} secrets in all split over {world} GPU(s) (BASELINE configs[4]"
                            + ("" if total == full else f": {full} in all needs 8 GPUs; {C5_PER_GPU_CAP} per GPU here") + ")"}
    n, t, N = args.n, args.t, args.secrets
    headline = (n, t, args.field, N) == (10, 3, "m61", 100_000_000)
    return {"key": "c2", "field": args.field, "n": n, "t": t, "total": N * world, "mine": N, "scaling": "weak",
            "dtype": None, "parallelism": f"shard{world}",
            "workload": f"shamir_share+reconstruct n={n} t={t} {FIELD_NAMES[args.field]} {N} secrets/GPU"
                        + (" (BASELINE configs[1])" if headline else "")}


# ------------------------------------------------- the line and the detail file -------------------------------------------------
def digest(detail):
    """one number per leg for the compact line: fractions of the ceiling each leg is quoted against (HBM unless named)"""
    out = {}
    cfgs = detail.get("configs") or {}
    for key, c in cfgs.items():
        if "error" in c:
            continue
        d = {"share_frac": c["share_frac"], "recover_frac": c["recover_frac"]}
        if "share_roofline" in c:
            d["share_frac_int8"] = c["share_roofline"]["frac"]
            d["share_frac_int8_executed"] = c["share_roofline"]["executed_frac"]
        if "recover_roofline_compute" in c:
            d["recover_frac_lds"] = c["recover_roofline_compute"]["frac"]
            d["share_frac_valu"] = c["share_roofline_compute"]["frac"]
        out[key] = d
    ew = detail.get("ew") or {}
    for fname, fld in (ew.get("fields") or {}).items():
        d = {op: fld[op]["frac"] for op in ("add", "mul", "inv") if op in fld}
        if "inv" in fld and "roofline_compute" in fld["inv"]:
            d["inv_frac_valu"] = fld["inv"]["roofline_compute"]["frac"]
        out["ew " + fname] = d
    mm = detail.get("matmul") or {}
    for shape, v in (mm.get("shapes") or {}).items():
        out["matmul " + shape] = {"T_mac_per_s": v["T_multiply_adds_per_s"], "frac_int8_executed": v["frac_of_int8_peak"]}
    prg = detail.get("prg_mode") or {}
    if "share_ms" in prg:
        out["prg_mode"] = {"share_ms": prg["share_ms"], "aes_blocks_per_s": prg["k_prg_blocks"]["blocks_per_s"],
                           "frac_lds": prg["k_prg_blocks"]["roofline_compute"]["frac"]}
    op = detail.get("open") or {}
    if "c4_all_gather" in op and "error" not in op["c4_all_gather"]:
        c4 = op["c4_all_gather"]
        out["open c4"] = {"opened_per_s": c4["opened_secrets_per_s"], "rccl_busbw_GBps": c4["rccl_busbw_GBps"],
                          "reconstruct_hbm_frac": c4["reconstruct_hbm_frac"]}
    return out


def finish_line(line, detail):
    """`verified` = AND over the headline and every leg of the detail object that ran; a leg that failed (an {"error": ..}
    object anywhere) is listed in `errors`.  Then the line is cut to size: floats to six figures and, should it still pass
    LINE_LIMIT (it does not at today's leg count), the digest and then the per-leg map give way to their summaries."""
    legs, errors = side_legs(detail)
    headline_ok = bool(line.get("verified_headline", line.get("verified", True)))
    line["verified_headline"] = headline_ok
    line["verified"] = headline_ok and all(legs.values())
    line["verified_legs"] = legs
    if errors:
        line["errors"] = [e[:240] for e in errors[:12]]
    line["legs"] = digest(detail)
    out = sig(line)
    if len(json.dumps(out)) > LINE_LIMIT:
        out.pop("legs", None)
    if len(json.dumps(out)) > LINE_LIMIT:
        out["verified_legs"] = {"count": len(legs), "failed": [k for k, v in legs.items() if not v][:20]}
    assert len(json.dumps(out)) <= LINE_LIMIT
    return out, legs, errors


def write_detail(path, line, detail):
    if not path:
        return None
    body = dict(detail, line=line)
    for target in (path, os.path.join("/tmp", os.path.basename(path))):
        try:
            tmp = target + ".tmp%d" % os.getpid()
            with open(tmp, "w") as fh:
                json.dump(body, fh, indent=1)
                fh.write("\n")
            os.replace(tmp, target)
            return target
        except OSError:
            continue
    return None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)          # never returns

    # The contract is ONE line on stdout.  Libraries write there too (RCCL prints a version banner from ncclCommInitRank
    # when NCCL_DEBUG asks for it): from here on file descriptor 1 goes to stderr, and the result line alone is written to
    # the descriptor stdout had.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line, detail=None):
        """the detail file first, then the line (so that whoever reads the line finds the file it names)"""
        if detail is not None and args.detail:
            def shown(path):
                return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
            line["detail"] = shown(args.detail)
            where = write_detail(args.detail, line, detail)
            if where and shown(where) != line["detail"]:       # (the directory was not writable: the copy under /tmp)
                line["detail"] = shown(where)
                write_detail(where, line, detail)
            if where is None:
                line["detail"] = None
            print(f"bench.py: detail in {where}", file=sys.stderr)
        text = json.dumps(line)
        assert len(text) <= LINE_LIMIT + 200, len(text)
        os.write(result_fd, (text + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    ctx = Ctx(args, world, rank, local_rank)
    ctx.init_torch()
    torch, dist = ctx.torch, ctx.dist
    pl = plan(args, world, rank)

    if ctx.dry:
        elapsed, rank_elapsed = ctx.timed_region(lambda k: None, args.steps, args.warmup)
        rccl = ctx.rccl_report()
        mine = torch.tensor([pl["mine"]], dtype=torch.int64)
        if world > 1 and pl["key"] != "c4":
            dist.all_reduce(mine)          # the shards add up to the total (c4: every rank opens every secret)
        if rank == 0:
            emit(sig({"metric": "shamir_reconstructions_per_sec", "value": 0.0, "unit": "reconstructions/s",
                      "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": 1e3 * elapsed / max(1, args.steps), "higher_is_better": True,
                      "ms_per_step_by_rank": [1e3 * x / max(1, args.steps) for x in rank_elapsed], "rccl": rccl,
                      "scaling": pl["scaling"], "vs_baseline": None, "dtype": pl["dtype"] or "u64",
                      "data": "none (dry run)",
                      "config": {"workload": "dry run of the launcher: " + pl["workload"], "n": pl["n"], "t": pl["t"],
                                 "field": FIELD_NAMES[pl["field"]], "total_secrets": pl["total"],
                                 "secrets_over_ranks": int(mine.item()), "parallelism": pl["parallelism"]}}, 9))
        ctx.finish()
        return

    ctx.init_scl()
    scl = ctx.scl
    from bench_legs import configs as legs_configs
    from bench_legs import ew as legs_ew
    from bench_legs import open_step as legs_open

    def base_line(value, ms_per_step, dtype, config, roofline, steps=None, warmup=None, metric="shamir_reconstructions_per_sec",
                  scaling=None):
        return {"metric": metric, "value": value, "unit": "reconstructions/s", "n_gpus": world,
                "steps": args.steps if steps is None else steps, "warmup": args.warmup if warmup is None else warmup,
                "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling or pl["scaling"], "vs_baseline": None,
                "dtype": dtype, "data": "synthetic", "config": config, "roofline": roofline}

    if pl["key"] == "c4":
        # BASELINE configs[3]: the exchange step is the timed step.  Every rank holds ceil(40 / G) parties' share vectors of
        # ALL the secrets, one all-gather per chunk brings the 40 rows together, every rank reconstructs every secret.
        c4 = legs_open.open_step(ctx, "gf2_128", 40, 13, pl["total"], args.open_chunk, args.steps, args.warmup, b"scl-bench-c4")
        if rank == 0:
            line = base_line(c4["opened_secrets_per_s"], c4["pipeline_ms"], "u128",
                             {"workload": pl["workload"], "field": FIELD_NAMES["gf2_128"], "n": 40, "t": 13,
                              "total_secrets": pl["total"], "chunk": c4["chunk"], "parallelism": pl["parallelism"]},
                             {"bound": "hbm", "kernel": "shamir_recover", "achieved": c4["reconstruct_GBps"],
                              "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": c4["reconstruct_hbm_frac"], "traffic": None},
                             scaling="strong")
            line["rccl_busbw_GBps"] = c4["rccl_busbw_GBps"]
            line["verified_headline"] = c4["verified"]
            detail = {"open": {"c4_all_gather": c4}}
            if world == 1 and args.cpu_sample > 0:
                detail["cpu_baseline"] = cpu_baseline("gf2_128", 40, 13, min(args.cpu_sample, 20_000))
                line["cpu_baseline"] = {k: detail["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample", "cpu_model")}
            out, _, _ = finish_line(line, detail)
            emit(out, detail)
        ctx.finish()
        return

    if args.mode == "open":
        rep = legs_open.open_report(ctx)
        if rank == 0:
            c4 = rep["c4_all_gather"]
            line = base_line(c4["opened_secrets_per_s"], c4["pipeline_ms"], "u128",
                             {"workload": f"open (all-gather + reconstruct) n=40 t=13 GF(2^128) {c4['secrets']} secrets, "
                                          f"{c4['parties_per_rank']} parties per rank (BASELINE configs[3] exchange step)",
                              "parallelism": f"parties{world}"},
                             {"bound": "hbm", "kernel": "shamir_recover", "achieved": c4["reconstruct_GBps"],
                              "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": c4["reconstruct_hbm_frac"], "traffic": None},
                             steps=3, warmup=1, metric="shamir_open_reconstructions_per_sec", scaling="strong")
            line["verified_headline"] = c4["verified"]
            detail = {"open": rep}
            out, _, _ = finish_line(line, detail)
            emit(out, detail)
        ctx.finish()
        return

    # =============================================== headline ===============================================
    if pl["key"] == "c5":
        args.field, args.n, args.t, args.secrets = pl["field"], pl["n"], pl["t"], pl["mine"]
        args.configs = args.open = 0
        if args.steps == 20 and args.warmup == 3:   # the defaults are the headline's: a (128,42) step is ~80 ms per 1.25e8
            args.steps, args.warmup = 5, 1
        args.cpu_sample = min(args.cpu_sample, 20_000)
    f, L = ctx.tag_limbs(args.field)
    E = 8 * L
    n, t, N = args.n, args.t, args.secrets
    seed = f"scl-bench-{args.field}-{rank}".encode()
    lam = scl.lagrange_basis(f, n)
    tc = t if args.share_mode == "coeffs" else 0
    # The steps rotate over `nsets` independently allocated operand sets, all alive at once: where a buffer lands moves these
    # kernels by up to 10 % (Placement), so ONE 12 GB set says as much about the allocation as about the
    # kernel.  Every step does the same work whichever set it runs on; the detail file carries the per-allocation means.
    nsets = args.allocations or (3 if (pl["key"] == "c2" and N * n * E <= 8_000_000_000) else 1)
    sets = []
    for a in range(nsets):
        secrets = scl.empty(f, N)
        coeffs = scl.empty(f, tc, N) if tc else None
        ctx.fill_random(secrets, f, seed + (b"-secrets" if a == 0 else b"-secrets%d" % a))
        if tc:
            ctx.fill_random(coeffs, f, seed + (b"-coeffs" if a == 0 else b"-coeffs%d" % a))
        sets.append((secrets, coeffs, scl.empty(f, n, N), scl.empty(f, N)))
    timers = [(scl.Timer(), scl.Timer()) for _ in range(args.steps)]
    turn = [0]

    def step(k):
        tm = timers[k] if k is not None else None
        secrets, coeffs, shares, out = sets[(k if k is not None else turn[0]) % nsets]
        turn[0] += 1
        if tm:
            tm[0].start()
        if args.share_mode == "coeffs":
            scl.shamir_share(f, secrets, coeffs, n, out=shares)
        else:
            scl.shamir_share_prg(f, secrets, t, n, seed, out=shares)
        if tm:
            tm[0].stop()
            tm[1].start()
        scl.shamir_recover(f, shares, lam, out=out)
        if tm:
            tm[1].stop()

    elapsed, rank_elapsed = ctx.timed_region(step, args.steps, args.warmup)

    # ---- per-kernel durations from the HIP events recorded inside the timed region -------------------
    share_ms = sum(tm[0].elapsed_ms() for tm in timers) / max(1, args.steps)
    rec_ms = sum(tm[1].elapsed_ms() for tm in timers) / max(1, args.steps)
    by_alloc = []
    for a in range(nsets):
        mine_ = [tm for k, tm in enumerate(timers) if k % nsets == a]
        if mine_:
            by_alloc.append({"steps": len(mine_), "share_ms": sum(tm[0].elapsed_ms() for tm in mine_) / len(mine_),
                             "recover_ms": sum(tm[1].elapsed_ms() for tm in mine_) / len(mine_)})
    ran = max(min(nsets, args.steps), min(nsets, args.warmup))      # the sets a step has run on (rotation from set 0)
    verified = all(bool(scl.equals(f, st[3], st[0])) for st in sets[:ran])
    if world > 1:      # the headline's verdict covers every rank's shard, not rank 0's alone
        flag = torch.tensor([1 if verified else 0], dtype=torch.int64, device="cpu" if args.backend == "gloo" else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verified = bool(flag.item())
    secrets, coeffs, shares, out = sets[0]
    del sets[1:]
    ctx.free()

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
    del secrets, coeffs, shares, out, src, dst, sets
    ctx.free()

    def headline_line(rccl):
        rec_bytes = (n + 1) * E            # n shares in, 1 secret out            
        share_bytes = (1 + t) * E + n * E if args.share_mode == "coeffs" else E + n * E
        kernels = {}
        for name, ms, nb in (("shamir_share", share_ms, share_bytes), ("shamir_recover", rec_ms, rec_bytes)):
            gbps = nb * N / (ms * 1e-3) / 1e9
            kernels[name] = {"ms": ms, "bytes_per_secret": nb, "GBps": gbps, "frac": gbps / HBM_PEAK_GBPS}
        dom = "shamir_share" if share_ms >= rec_ms else "shamir_recover"
        ach = kernels[dom]["GBps"]
        traffic = pmc_traffic(dom, args)
        algo = kernels[dom]["bytes_per_secret"] * N
        roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "algorithmic_bytes": algo,
                    "traffic_over_algorithmic": (traffic / algo) if traffic else None,
                    "traffic_source": "stamped: profiles/pmc_traffic.json" if traffic else None,
                    "measured_copy_GBps": copy_gbps, "frac_of_measured_copy": ach / copy_gbps}
        if dom == "shamir_share" and args.share_mode == "coeffs" and on_matrix_cores(args.field, n, t):
            # --config c5: the dominant kernel runs on the matrix cores; its HBM-equivalent figures stay beside the matrix roofline
            hbm = roofline
            roofline = mfma_share_roofline(n, t, N, share_ms)
            roofline["traffic"] = hbm["traffic"]
            if roofline["traffic"] is None and (n, t, N) == (128, 42, 125_000_000):
                # the stamped PMC run measured this very launch as its side configuration C5 (same kernel, same shard size)
                side = pmc_config_traffic("C5_shard_mersenne61_128_42")
                roofline["traffic"] = side["share"] if side else None
            roofline["hbm_equivalent"] = {k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes")}
        total = pl["total"] * args.steps       # c2: N per GPU x ranks; c5: BASELINE's total, split over the ranks
        line = base_line(total / elapsed, 1e3 * elapsed / args.steps, DTYPES[L],
                         {"workload": pl["workload"], "field": FIELD_NAMES[args.field], "n": n, "t": t, "secrets_per_gpu": N,
                          "total_secrets": pl["total"], "share_mode": args.share_mode, "layout": "SoA [party][secret]",
                          "allocations": nsets, "parallelism": pl["parallelism"]}, roofline)
        line.update({"kernels": kernels, "ms_per_step_by_rank": [1e3 * x / args.steps for x in rank_elapsed], "rccl": rccl,
                     "verified_headline": verified,
                     "reconstruct_only_per_s": N * world / (rec_ms * 1e-3), "share_only_per_s": N * world / (share_ms * 1e-3)})
        return line, roofline, dom

    # The multi-rank legs below run RCCL through torch and through the C ABI -- code no one-GPU box can rehearse with real RCCL.
    # If one of them never returns, the headline measured above must not be lost with it: after --side-timeout seconds rank 0
    # writes the line with what it has (the missing legs as errors) and every rank leaves.
    watchdog = None
    if world > 1 and args.side_timeout > 0:
        import threading

        def give_up():
            if rank == 0:
                why = f"no result after {args.side_timeout:g} s: the process was ended by bench.py's own watchdog"
                early, _, _ = headline_line({"error": why})
                detail = {"by_allocation": by_alloc, "open": {"error": why}}
                out, _, _ = finish_line(early, detail)
                emit(out, detail)
                print("bench.py: " + why, file=sys.stderr)
            # (exit code: see the end of main -- on a multi-rank run the line is the report; every rank leaves the same way, so the
            # launcher does not turn a measured headline into a failed job)
            os._exit(0 if verified else 3)
        # (the other ranks leave a little later: a launcher that sees a worker die ends the rest, rank 0's line must be out by then)
        watchdog = threading.Timer(args.side_timeout + (0 if rank == 0 else 5), give_up)
        watchdog.daemon = True
        watchdog.start()
    rccl = ctx.rccl_report()                             # (collective: every rank)
    if args.inject_error == "hang" and rank == world - 1:
        time.sleep(10 ** 6)                              # (test: a rank that never reaches the collectives)
    open_rep = legs_open.open_report(ctx) if args.open else None      # every rank takes part in the collectives
    if watchdog is not None:
        watchdog.cancel()

    if rank != 0:
        ctx.finish()
        return
    line, roofline, dom = headline_line(rccl)
    detail = {"by_allocation": by_alloc}
    if open_rep is not None:
        detail["open"] = open_rep

    def leg(key, fn):
        try:
            detail[key] = fn(ctx)
        except Exception as e:   # a failed leg is reported, not hidden, and never costs the headline
            detail[key] = {"error": f"{type(e).__name__}: {e}", "verified": False}
            ctx.free()
    if world == 1 and pl["key"] == "c2":
        if args.configs:
            leg("prg_mode", legs_configs.prg_mode_report)
            leg("c1_additive", legs_configs.c1_additive_report)
        if args.ew:
            leg("ew", legs_ew.ew_report)
            leg("layout", legs_ew.layout_report)
            leg("matmul", legs_ew.matmul_report)
    if world == 1 and args.configs:
        # the other BASELINE configurations at the size ONE GPU holds of them (C4, C5: an eighth), after the timed region
        detail["configs"] = legs_configs.configs_report(ctx)
    if world == 1 and args.cpu_sample > 0:
        detail["cpu_baseline"] = cpu_baseline(args.field, n, t, args.cpu_sample, all_cores=bool(args.cpu_all_cores))
        line["cpu_baseline"] = {k: detail["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample", "cpu_model")}
    if (world == 1 and args.pmc_live and pl["key"] == "c2" and roofline.get("bound") == "hbm"
            and (args.field, args.n, args.t, args.share_mode) == ("m61", 10, 3, "coeffs") and args.secrets >= 10_000_000):
        ctx.free()
        live, detail["pmc_live"] = live_pmc_traffic(args, os.path.abspath(__file__))
        if live is not None:
            cal = live.get("calibration_k_copy16", {})
            roofline["traffic_stamped"] = roofline["traffic"]
            roofline["traffic"] = live[dom]["bytes"]
            roofline["traffic_over_algorithmic"] = live[dom]["bytes"] / roofline["algorithmic_bytes"]
            roofline["traffic_source"] = "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run"
            detail["traffic_live"] = {"fetch_doubled": True, "fetch_correction_measured_on_k_copy16": cal.get("fetch_correction"),
                                      "shamir_share": live["shamir_share"], "shamir_recover": live["shamir_recover"],
                                      "how": "two child runs of this command's GPU legs (headline 3 steps, after the timed regions), mean "
                                             "per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md and checked on k_copy16 in the same pass"}
            for key, c in live.get("configs", {}).items():   # the side configurations' kernels, from the same two passes
                if key in detail.get("configs", {}) and "error" not in detail["configs"][key]:
                    detail["configs"][key]["traffic_stamped"] = detail["configs"][key].get("traffic")
                    detail["configs"][key]["traffic"] = {"share": c["share"]["bytes"], "recover": c["recover"]["bytes"],
                                                         "share_kernel": c["share"]["kernel"], "recover_kernel": c["recover"]["kernel"],
                                                         "share_over_algorithmic": c["share"]["traffic_over_algorithmic"],
                                                         "recover_over_algorithmic": c["recover"]["traffic_over_algorithmic"],
                                                         "source": "observed in this run"}
    # the line tells the truth about its side legs: `verified` is the AND over the headline and every leg that ran, and a
    # leg that failed makes the process exit non-zero AFTER the line is out
    out, legs, errors = finish_line(line, detail)
    emit(out, detail)
    ctx.finish()
    if errors or not out["verified"]:
        print("bench.py: " + ("; ".join(errors) if errors else "a leg did not verify: " +
                               ", ".join(k for k, v in legs.items() if not v)), file=sys.stderr)
        # One GPU: everything in this run has been rehearsed on such a box, so a failed leg fails the process (after the line is
        # out).  Several GPUs: the legs after the headline run RCCL code no one-GPU box could rehearse; their failure is IN the
        # line (`verified`: false, `errors`, `verified_legs`) and on stderr, but a verified headline keeps exit code 0 -- a scaling
        # record must not be lost to a side leg.  A headline that does not verify fails the process at any size.
        if world == 1 or not out["verified_headline"]:
            sys.exit(1)


if __name__ == "__main__":
    main()
