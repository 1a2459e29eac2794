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
the headline region and reported under "open" with RCCL's bus bandwidth beside the reconstruct kernel's HBM rate.

Prints ONE JSON line on rank 0 (fields described in DESIGN.md section "Measurement").
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "secure-computation-library_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

FIELD_TAGS = {"m61": 0, "m127": 1, "mont128": 2, "gf2_128": 3, "secp256k1": 4, "secp256k1_field": 5}
FIELD_NAMES = {"m61": "Mersenne61", "m127": "Mersenne127", "mont128": "Mont128", "gf2_128": "GF(2^128)",
               "secp256k1": "secp256k1_order", "secp256k1_field": "secp256k1_field"}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable)
I8_PEAK_TOPS = 5000.0   # MI355X_MICROARCH.md, matrix cores: I8 runs at 2x the BF16 rate per clock; BF16 dense ~2.5 PFLOP/s


def mfma_share_roofline(n, t, N, ms):
    """The matrix-core share kernel (k_share_mfma_m61_*, Mersenne61 at n > 96 / t >= 32) is bound by the matrix pipe, not by
    HBM.  Algorithmic work of its formulation: V (n x (t+1)) times C ((t+1) x N) in 8 signed base-256
    digits each = 64 digit-pair products, 2 int8 operations per multiply-accumulate.  `executed` counts what the instructions
    do: K padded to 64 coefficient slots and the party tile to 16 rows."""
    ops = 2.0 * 64 * (t + 1) * n * N
    executed = 2.0 * 64 * 64 * ((n + 15) // 16 * 16) * N
    ach = ops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "shamir_share", "achieved": ach, "peak": I8_PEAK_TOPS, "unit": "TOP/s (int8)",
            "frac": ach / I8_PEAK_TOPS, "algorithmic_ops": ops, "executed_TOPs": executed / (ms * 1e-3) / 1e12,
            "executed_frac": executed / (ms * 1e-3) / 1e12 / I8_PEAK_TOPS}


def on_matrix_cores(fkey, n, t):
    """the shapes scl_hip_shamir_share sends to k_share_mfma_m61_p16 (capi.hip: Mersenne61, n > 96, 32 <= t <= 63)"""
    return fkey == "m61" and n > 96 and 32 <= t <= 63
KERNEL_SOURCES = ("secure-computation-library_amd/csrc/kernels.hpp", "secure-computation-library_amd/csrc/capi.hip",
                  "secure-computation-library_amd/csrc/share_mfma.hpp", "secure-computation-library_amd/csrc/gemm_mfma.hpp", "secure-computation-library_amd/csrc/gemm_unit.hip",
                  "include/scl_hip/detail/field.hpp")


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
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="secrets timed on the CPU baseline (0 = skip)")
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
                    help="1: after the headline, the element-wise add / mul / inverse path (Mersenne61 10^8, Mersenne127 and GF(2^128) 10^7)")
    ap.add_argument("--ew-elements", type=int, default=0,
                    help="tests only: run the element-wise and layout legs on this many elements per field instead of 10^8 / 10^7")
    ap.add_argument("--pmc-live", type=int, default=1,
                    help="1: roofline.traffic observed in THIS run -- two child runs of the headline under rocprofv3 --pmc FETCH_SIZE / "
                         "WRITE_SIZE after everything else (one GPU, BASELINE configs[1] only; falls back to profiles/pmc_traffic.json)")
    ap.add_argument("--side-timeout", type=float, default=300.0,
                    help="multi-rank runs: seconds the legs after the headline (RCCL through torch and through the C ABI) may take "
                         "before rank 0 writes the line with what it has and every rank exits (0 = no limit)")
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


C5_PER_GPU_CAP = 125_000_000   # (128,42) Mersenne61: 128 GB of shares + 42 GB of coefficients + 2 GB per GPU (of 288 GB)


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


CPU_BASELINE_SECONDS = 8.0   # per leg of cpu_baseline (single thread, all cores, hoisted basis): the default run stays in minutes


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
    # A pilot of a few secrets bounds the sample to about CPU_BASELINE_SECONDS of work per leg whatever the shape costs (the
    # per-secret Lagrange basis is n(n-1) field inversions: 40 parties over GF(2^128) in the oracle port take milliseconds per
    # secret, where (10,3) over Mersenne61 takes two microseconds).
    pilot_n = min(sample, 16)
    try:
        pilot = lib.time_shamir(f, pilot_n, t, n)
    except O.OracleError:
        if kind != "reference":
            raise
        # a field the reference library does not have (GF(2^128)): the oracle port is the CPU baseline for it
        lib, kind = O.Port(), "port"
        pilot = lib.time_shamir(f, pilot_n, t, n)
    while pilot["share_s"] + pilot["recover_s"] < 0.25 and pilot_n < sample:   # (the first calls also pay for cold caches)
        pilot_n = min(sample, pilot_n * 8)
        pilot = lib.time_shamir(f, pilot_n, t, n)
    per_secret = max((pilot["share_s"] + pilot["recover_s"]) / pilot_n, 1e-9)
    sample = max(16, min(sample, int(CPU_BASELINE_SECONDS / per_secret)))
    r = lib.time_shamir(f, sample, t, n)
    if r["mismatches"]:
        raise RuntimeError("CPU baseline failed its own round trip")
    total = r["share_s"] + r["recover_s"]
    out = {
        "value": sample / total, "unit": "reconstructions/s", "cores": 1, "kind": kind, "cpu_model": cpu_model(),
        "host_cores_available": len(os.sched_getaffinity(0)),
        "sample": f"{sample} secrets, per-secret shamirSecretShare + shamirRecoverP (n={n}, t={t}), "
                  f"share {r['share_s']:.2f}s + recover {r['recover_s']:.2f}s",
        "recover_only_per_s": sample / r["recover_s"], "share_only_per_s": sample / r["share_s"],
    }
    # SURVEY.md section 8d / BASELINE.md section 2 ask for more figures beside the faithful single-thread run (SCL itself is
    # single-threaded; the threads below are this harness's, one PRG and one slab of secrets each):
    #   all_cores      the same per-secret path on every PHYSICAL core this process may run on, count stated (a container's
    #                  cgroup quota, when there is one, is stated beside it: it caps what those threads get)
    #   cores_per_gpu  the same on the 16 host threads that go with one GPU on the bench boxes
    #   hoisted_basis  one thread, Lagrange basis computed once instead of per secret (reference library only)
    from concurrent.futures import ThreadPoolExecutor
    phys = physical_cores()
    out["physical_cores"], out["cgroup_cpu_limit"] = phys, cgroup_cpu_limit()
    for key, cores in (("all_cores", phys), ("cores_per_gpu", min(len(os.sched_getaffinity(0)), 16))):
        try:
            per = max(1, sample // 4)          # a quarter of the sample per thread keeps the leg to a few seconds
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:   # ctypes releases the GIL for the duration of each call
                rs = list(ex.map(lambda i: lib.time_shamir(f, per, t, n, b"scl-bench-%d" % i), range(cores)))
            wall = time.perf_counter() - t0
            if not any(x["mismatches"] for x in rs):
                out[key] = {"value": per * cores / wall, "cores": cores,
                            "sample": f"{per} secrets on each of {cores} threads, wall {wall:.2f}s"}
        except Exception as e:  # the extra legs never fail the bench line
            out[key] = {"error": str(e)}
    if kind == "reference":
        try:
            hr = lib.time_shamir_hoisted(f, sample, t, n)
            if not hr["mismatches"]:
                out["hoisted_basis"] = {"value": sample / (hr["share_s"] + hr["recover_s"]), "cores": 1,
                                        "recover_only_per_s": sample / hr["recover_s"]}
        except Exception as e:
            out["hoisted_basis"] = {"error": str(e)}
    return out


def physical_cores():
    """physical cores among the logical CPUs this process may run on (distinct (socket, core) pairs of /proc/cpuinfo)"""
    allowed = os.sched_getaffinity(0)
    cores, cpu, phys = set(), None, 0
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                k, _, v = ln.partition(":")
                k = k.strip()
                if k == "processor":
                    cpu = int(v)
                elif k == "physical id":
                    phys = int(v)
                elif k == "core id" and cpu in allowed:
                    cores.add((phys, int(v)))
    except (OSError, ValueError):
        pass
    return len(cores) or len(allowed)


def cgroup_cpu_limit():
    """CPUs the container may use at once when a cgroup quota says so (cpu.max "quota period"), else None"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def kernel_source_hash():
    """sha256 over the kernel sources: what profiles/pmc_traffic.json is stamped with (a rebuilt .so of the same
    sources need not be byte-identical, the sources are)"""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(dom, args):
    """HBM bytes per launch of the dominant kernel from the committed PMC run -- only while that run still describes
    the code: the file carries the configuration, the kernel symbols and a hash of the kernel sources it was taken
    with; any mismatch gives null (bench.py cannot read PMCs itself)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        c = pmc["config"]
        if (c["field"], c["n"], c["t"], c["secrets_per_gpu"], c["share_mode"]) != (
                args.field, args.n, args.t, args.secrets, args.share_mode):
            return None
        if pmc.get("kernel_source_sha256_16") != kernel_source_hash():
            return None
        return pmc[dom]["bytes"]
    except Exception:
        return None


def pmc_config_traffic(key):
    """{"share": bytes, "recover": bytes} per launch for a side configuration, from the same stamped PMC file (null when the
    kernel sources have changed since, or the configuration was not in the PMC run)"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        if pmc.get("kernel_source_sha256_16") != kernel_source_hash():
            return None
        c = pmc["configs"][key]
        return {"share": c["share"]["bytes"], "recover": c["recover"]["bytes"],
                "share_kernel": c["share"]["kernel"], "recover_kernel": c["recover"]["kernel"]}
    except Exception:
        return None


PMC_NEEDLES = {"shamir_share": "k_share_small_t<sclhip::M61", "shamir_recover": "k_recover_fixed<sclhip::M61"}
# the side configurations by the kernels they launch, with their algorithmic bytes per launch (share, reconstruct)
PMC_CONFIGS = {
    "C3_mersenne127_10_3": ("k_share_small_t<sclhip::M127", "k_recover_fixed<sclhip::M127", (224 * 10**7, 176 * 10**7)),
    "C3_mont128_10_3": ("k_share_small_t<sclhip::Mont128", "k_recover_small<sclhip::Mont128", (224 * 10**7, 176 * 10**7)),
    "F3_secp256k1_scalar_10_3": ("k_share_small_pair<sclhip::Mont256<sclhip::SecpOrderParams>", "k_recover_small<sclhip::Mont256<sclhip::SecpOrderParams>",
                                 (448 * 10**7, 352 * 10**7)),
    "C4_shard_gf2_128_40_13": ("k_share_gf_tiles<13>", "k_recover_gf128_pos<512", (864 * 125 * 10**5, 656 * 125 * 10**5)),
    "C5_shard_mersenne61_128_42": ("k_share_mfma_m61", "k_recover_table<sclhip::M61", (1368 * 125 * 10**6, 1032 * 125 * 10**6)),
}


def pmc_means(d, counter):
    """{kernel name: (mean counter value over the launches of the largest size, their number)} from a rocprofv3 --pmc output
    directory.  One kernel may run at several sizes in a bench run (the first-use self-check of the GF(2^128) reconstruct kernel
    is a 4096-secret launch of the kernel C4 then runs at 1.25e7): launches within a factor of two of the maximum count."""
    import csv
    import glob
    from collections import defaultdict
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        big = [x for x in v if x >= 0.5 * max(v)]
        out[k] = (sum(big) / len(big), len(big))
    return out


def pmc_report(fetch, write, copy_bytes):
    """HBM bytes per launch of the headline's two kernels and of every side configuration's whose kernels are in the passes,
    from the two tables of pmc_means.  Counters are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: it reports
    half the bytes of 16-byte-per-lane streaming reads) and the factor is checked on k_copy16, whose byte count is known."""
    def pick(table, needle):
        hits = [(k, v) for k, v in table.items() if needle in k]
        return hits[0] if len(hits) == 1 else None

    def entry(needle):
        f, w = pick(fetch, needle), pick(write, needle)
        if f is None or w is None:
            return None
        name, (f_kib, nl) = f
        w_kib = w[1][0]
        return {"kernel": name.split("(")[0].replace("void ", ""), "launches": nl, "fetch_kib_reported": f_kib, "write_kib": w_kib,
                "bytes": int(round(2 * f_kib * 1024 + w_kib * 1024))}
    out = {}
    cf, cw = pick(fetch, "k_copy16"), pick(write, "k_copy16")
    if cf and cw:
        out["calibration_k_copy16"] = {"bytes_read_per_launch": copy_bytes, "fetch_kib_reported": cf[1][0], "write_kib": cw[1][0],
                                       "fetch_correction": copy_bytes / (cf[1][0] * 1024.0), "launches": cf[1][1]}
    for key, needle in PMC_NEEDLES.items():
        out[key] = entry(needle)
    out["configs"] = {}
    for cfg, (share_needle, rec_needle, algo) in PMC_CONFIGS.items():
        sh, rc = entry(share_needle), entry(rec_needle)
        if sh is None or rc is None:
            continue
        sh["algorithmic_bytes"], rc["algorithmic_bytes"] = algo
        sh["traffic_over_algorithmic"], rc["traffic_over_algorithmic"] = sh["bytes"] / algo[0], rc["bytes"] / algo[1]
        out["configs"][cfg] = {"share": sh, "recover": rc}
    return out


def live_pmc_traffic(args, timeout_s=240):
    """HBM bytes per launch OBSERVED in this run: two child processes run this command's GPU legs once more (headline 3 steps,
    the side configurations when --configs is on; no open step, no CPU baseline) under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (the two cannot share a pass on gfx950; no trace domain is combined with --pmc), and pmc_report reads the
    kernels' counter means from their CSVs.  The program itself follows `--`.  Children of this process, started after every
    timed region.  Returns (report or None, info): info = {"ran", "seconds", "fallback_reason"} goes into the line as
    `pmc_live`, so a fallback to the stamped figures says why."""
    import shutil
    import tempfile
    t_begin = time.perf_counter()

    def done(rep, reason=None):
        return rep, {"ran": rep is not None, "seconds": round(time.perf_counter() - t_begin, 1), "fallback_reason": reason}
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return done(None, "rocprofv3 not found")
    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ):   # this process is itself being profiled: no nesting
        return done(None, "this process is itself running under a profiler")
    child = [sys.executable, os.path.abspath(__file__), "--configs", str(args.configs), "--open", "0", "--cpu-sample", "0",
             "--pmc-live", "0", "--ew", "0", "--allocations", "1", "--steps", "3", "--warmup", "1", "--field", args.field, "--n", str(args.n), "--t", str(args.t),
             "--secrets", str(args.secrets), "--share-mode", args.share_mode]
    env = dict(os.environ, TMPDIR="/tmp")
    work = tempfile.mkdtemp(prefix="scl_pmc_", dir="/tmp")
    try:
        tables = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            p = subprocess.Popen([exe, "--pmc", counter, "-d", d, "--output-format", "csv", "--"] + child, cwd="/tmp", env=env,
                                 stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)      # exactly the process group this call started
                p.wait()
                return done(None, f"the {counter} pass did not finish in {timeout_s} s")
            if rc != 0:
                return done(None, f"the {counter} pass exited with {rc}")
            tables[counter] = pmc_means(d, counter)
        copy_bytes = float(min(4 << 30, args.n * args.secrets * 8 * (1 if args.field == "m61" else 2) // 2) & ~15)
        rep = pmc_report(tables["FETCH_SIZE"], tables["WRITE_SIZE"], copy_bytes)
        if rep.get("shamir_share") is None or rep.get("shamir_recover") is None:
            return done(None, "the headline's kernels are not in the counter tables")
        return done(rep)
    except Exception as e:
        return done(None, f"{type(e).__name__}: {e}")
    finally:
        shutil.rmtree(work, ignore_errors=True)


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

    def emit(line):
        os.write(result_fd, (json.dumps(line) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    dry = args.dry_run
    if dry:
        if world > 1:
            dist.init_process_group("gloo")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        # SCL_BENCH_ONE_DEVICE=1: a rehearsal of the multi-rank logic on a one-GPU box (every rank on device 0, gloo
        # collectives); never a measurement
        one_device = os.environ.get("SCL_BENCH_ONE_DEVICE") == "1" and args.backend == "gloo"
        dev_index = 0 if one_device else local_rank
        torch.cuda.set_device(dev_index)
        if world > 1:
            dist.init_process_group(args.backend, device_id=torch.device("cuda", dev_index) if args.backend == "nccl" else None)

    def sync():
        if not dry:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if not dry:
                torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cpu" if dry or args.backend == "gloo" else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def all_ranks(x):
        """every rank's value, in rank order (what the driver needs to see that N ranks really ran)"""
        if world == 1:
            return [x]
        xs = [None] * world
        dist.all_gather_object(xs, float(x))
        return xs

    def rccl_report():
        """what the process group itself says about the job: a SCALE record can check that the collective library saw N ranks"""
        if world == 1:
            return {"ranks": 1, "backend": None, "devices": [0] if dry else [torch.cuda.current_device()]}
        devs = [None] * world
        dist.all_gather_object(devs, -1 if dry else int(torch.cuda.current_device()))
        rep = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "devices": devs}
        if not dry and args.backend == "nccl":
            # one all-reduce of ones through the communicator the timed collectives use: RCCL itself counts the ranks
            ones = torch.ones(1, dtype=torch.int64, device="cuda")
            dist.all_reduce(ones)
            rep["allreduce_of_ones"] = int(ones.item())
        return rep

    def timed_region(step, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks"""
        for _ in range(warmup):
            step(None)
        sync()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        sync()
        mine_s = time.perf_counter() - t0
        return max_over_ranks(mine_s), all_ranks(mine_s)

    pl = plan(args, world, rank)
    if dry:
        elapsed, rank_elapsed = timed_region(lambda k: None, args.steps, args.warmup)
        rccl = rccl_report()
        mine = torch.tensor([pl["mine"]], dtype=torch.int64)
        if world > 1 and pl["key"] != "c4":
            dist.all_reduce(mine)          # the shards add up to the total (c4: every rank opens every secret)
        if rank == 0:
            emit({"metric": "shamir_reconstructions_per_sec", "value": 0.0, "unit": "reconstructions/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": 1e3 * elapsed / max(1, args.steps), "higher_is_better": True,
                              "ms_per_step_by_rank": [1e3 * x / max(1, args.steps) for x in rank_elapsed], "rccl": rccl,
                              "scaling": pl["scaling"], "vs_baseline": None, "dtype": pl["dtype"] or "u64",
                              "data": "none (dry run)",
                              "config": {"workload": "dry run of the launcher: " + pl["workload"], "n": pl["n"], "t": pl["t"],
                                         "field": FIELD_NAMES[pl["field"]], "total_secrets": pl["total"],
                                         "secrets_over_ranks": int(mine.item()), "parallelism": pl["parallelism"]}})
        if world > 1:
            dist.destroy_process_group()
        return

    import scl_amd as scl
    from scl_amd import dist as sd

    def tag_limbs(fkey):
        f_ = FIELD_TAGS[fkey]
        return f_, scl.limbs(f_)

    def fill_random(dst, f_, seed, counter0=0):
        """uniform field elements from the device AES-CTR PRG straight into dst ([rows][N][L] or [N][L])"""
        rows = dst if dst.dim() == 3 else dst.unsqueeze(0)
        N_ = rows.shape[1]
        per_row = (N_ * 8 * rows.shape[2] + 15) // 16
        for k in range(rows.shape[0]):
            scl.vector_random(f_, N_, seed, counter0=counter0 + k * per_row, out=rows[k])

    def share_recover_config(fkey, n, t, N, steps, seed, warmup=1, allocations=1):
        """one configuration end to end on this GPU: plain allocations, share + reconstruct timed with HIP events on the
        launch stream, round trip verified.  Returns the per-kernel figures.  `warmup` untimed launches first: the first
        25-30 ms of load after an idle gap (allocation, fill) run under a clock ramp -- from a cold start the 0.38 ms Mont128
        share kernel reads 0.51, 0.50, 0.49 .. and reaches 0.38 only after about sixty launches, the headline's 1.7 ms kernel after
        five (profiles/r4_probe_c3_seq.txt, r4_probe_headline_seq.txt) -- so the sub-millisecond configurations take 100
        warm-up launches (40 ms) and 50 timed ones per kernel.  `allocations` > 1: the whole measurement on that many
        independently allocated operand sets, all alive at once (so they are different memory); the figures are the MEAN over
        all of them -- where the operands land moves these kernels by up to 10 % (Placement), and one
        1.6 GB set says more about the allocation than about the kernel -- with the per-allocation means beside it."""
        f_, L = tag_limbs(fkey)
        E = 8 * L
        lam = scl.lagrange_basis(f_, n)
        sets = []
        for a in range(allocations):
            secrets = scl.empty(f_, N)
            coeffs = scl.empty(f_, t, N)
            fill_random(secrets, f_, seed + b"-secrets%d" % a)
            fill_random(coeffs, f_, seed + b"-coeffs%d" % a)
            sets.append((secrets, coeffs, scl.empty(f_, n, N), scl.empty(f_, N)))
        share_all, rec_all, share_by, rec_by, ok = [], [], [], [], True
        for secrets, coeffs, shares, out in sets:
            tms = [(scl.Timer(), scl.Timer()) for _ in range(steps)]
            # each kernel on its own: `steps` launches of the share kernel back to back, then `steps` of the reconstruct kernel,
            # HIP events around every launch.  (Alternating the two, as the headline's step does by contract, leaves every
            # launch behind the other kernel's tail: the same kernels then spread 6-10 % from launch to launch instead of 3-5 %,
            # profiles/r4_probe_c3_bench.txt; these entries are per-kernel figures: each kernel in its own steady state.)
            for k in range(-warmup, steps):
                if k >= 0:
                    tms[k][0].start()
                scl.shamir_share(f_, secrets, coeffs, n, out=shares)
                if k >= 0:
                    tms[k][0].stop()
            for k in range(-warmup, steps):
                if k >= 0:
                    tms[k][1].start()
                scl.shamir_recover(f_, shares, lam, out=out)
                if k >= 0:
                    tms[k][1].stop()
            torch.cuda.synchronize()
            sh, rc = [tm[0].elapsed_ms() for tm in tms], [tm[1].elapsed_ms() for tm in tms]
            share_all += sh
            rec_all += rc
            share_by.append(sum(sh) / steps)
            rec_by.append(sum(rc) / steps)
            ok = ok and bool(scl.equals(f_, out, secrets))
        sm, rm = sum(share_all) / len(share_all), sum(rec_all) / len(rec_all)
        sb, rb = (1 + t + n) * E, (n + 1) * E
        res = {"field": FIELD_NAMES[fkey], "n": n, "t": t, "secrets": N, "dtype": {1: "u64", 2: "u128", 4: "u256"}[L],
               "share_ms": sm, "recover_ms": rm, "bytes_per_secret": {"share": sb, "recover": rb},
               "share_GBps": sb * N / sm / 1e6, "recover_GBps": rb * N / rm / 1e6,
               "share_frac": sb * N / sm / 1e6 / HBM_PEAK_GBPS, "recover_frac": rb * N / rm / 1e6 / HBM_PEAK_GBPS,
               "round_trips_per_s": N / ((sm + rm) * 1e-3), "reconstructions_per_s": N / (rm * 1e-3), "verified": ok,
               "steps": steps, "warmup": warmup, "allocations": allocations,
               "share_ms_min_max": [min(share_all), max(share_all)], "recover_ms_min_max": [min(rec_all), max(rec_all)]}
        if allocations > 1:
            res["share_ms_by_allocation"], res["recover_ms_by_allocation"] = share_by, rec_by
        if on_matrix_cores(fkey, n, t):
            res["share_roofline"] = mfma_share_roofline(n, t, N, sm)   # share_frac above is its HBM-equivalent rate only
        del sets, secrets, coeffs, shares, out
        torch.cuda.empty_cache()
        return res

    def prg_mode_report():
        """BASELINE configs[1] in the reference's OWN mode: scl::ss::shamirSecretShare(secret, t, n, prg) draws the
        coefficients from the PRG (shamir.h:51-68, prg.cc:124-146); here that is scl_hip_shamir_share_prg, bit-identical to
        the per-secret calls on one PRG.  After and outside the headline's timed region; AES rate of k_prg_blocks in the same
        run beside it (the share draws 2 blocks per secret: Vector::random(4) of 8-byte elements)."""
        f_, n_, t_, N_ = scl.M61, 10, 3, 100_000_000
        sd_ = b"scl-bench-prg-mode"
        secrets_ = scl.empty(f_, N_)
        fill_random(secrets_, f_, sd_ + b"-secrets")
        shares_ = scl.empty(f_, n_, N_)
        out_ = scl.empty(f_, N_)
        lam_ = scl.lagrange_basis(f_, n_)
        reps = 5
        ts, tr = [scl.Timer() for _ in range(reps)], [scl.Timer() for _ in range(reps)]
        scl.shamir_share_prg(f_, secrets_, t_, n_, sd_, out=shares_)
        for k in range(reps):
            ts[k].start()
            scl.shamir_share_prg(f_, secrets_, t_, n_, sd_, out=shares_)
            ts[k].stop()
            tr[k].start()
            scl.shamir_recover(f_, shares_, lam_, out=out_)
            tr[k].stop()
        torch.cuda.synchronize()
        sm = sum(x.elapsed_ms() for x in ts) / reps
        rm = sum(x.elapsed_ms() for x in tr) / reps
        ok = bool(scl.equals(f_, out_, secrets_))
        del shares_, out_
        torch.cuda.empty_cache()
        bps = scl.blocks_per_secret(f_, t_)
        nb = bps * N_
        blocks = scl.prg_blocks(nb, sd_)
        tb = scl.Timer()
        tb.start()
        for _ in range(3):
            scl.prg_blocks(nb, sd_, out=blocks)     # (into the same buffer: no 3 GB allocation inside the timed loop)
        tb.stop()
        bm = tb.elapsed_ms() / 3
        del blocks, secrets_
        torch.cuda.empty_cache()
        sb = (1 + n_) * 8
        return {"workload": f"scl_hip_shamir_share_prg n={n_} t={t_} Mersenne61 {N_} secrets (coefficients drawn from the "
                            "AES-128-CTR PRG inside the call, the reference's mode) + reconstruct",
                "share_ms": sm, "share_secrets_per_s": N_ / (sm * 1e-3), "aes_blocks_per_secret": bps,
                "share_aes_blocks_per_s": nb / (sm * 1e-3), "share_GBps": sb * N_ / sm / 1e6,
                "share_frac": sb * N_ / sm / 1e6 / HBM_PEAK_GBPS, "recover_ms": rm,
                "round_trips_per_s": N_ / ((sm + rm) * 1e-3),
                "k_prg_blocks": {"blocks": nb, "ms": bm, "blocks_per_s": nb / (bm * 1e-3)}, "verified": ok}

    def c1_additive_report():
        """BASELINE configs[0]: additive sharing over Mersenne61, n = 3, 10^6 secrets -- the reference's per-secret
        additiveShare + Vector::sum on one host core (oracle/_ref), beside the GPU kernels at the SAME size (10^6 secrets is
        24 MB: the kernels are launch-bound there) and at 10^8."""
        import oracle_lib as O
        f_, n_ = scl.M61, 3
        rep = {"workload": "additive sharing Mersenne61 n=3 (BASELINE configs[0])"}
        for N_, key in ((1_000_000, "gpu_1e6"), (100_000_000, "gpu_1e8")):
            sd_ = b"scl-bench-c1"
            secrets_ = scl.empty(f_, N_)
            fill_random(secrets_, f_, sd_ + b"-secrets")
            shares_ = scl.empty(f_, n_, N_)
            out_ = scl.empty(f_, N_)
            reps = 20 if N_ <= 1_000_000 else 5
            scl.additive_share_prg(f_, secrets_, n_, sd_, out=shares_)
            scl.additive_recover(f_, shares_, out=out_)
            ts, tr = scl.Timer(), scl.Timer()
            ts.start()
            for _ in range(reps):
                scl.additive_share_prg(f_, secrets_, n_, sd_, out=shares_)
            ts.stop()
            tr.start()
            for _ in range(reps):
                scl.additive_recover(f_, shares_, out=out_)
            tr.stop()
            sm, rm = ts.elapsed_ms() / reps, tr.elapsed_ms() / reps
            rep[key] = {"secrets": N_, "share_ms": sm, "recover_ms": rm, "share_secrets_per_s": N_ / (sm * 1e-3),
                        "reconstructions_per_s": N_ / (rm * 1e-3), "round_trips_per_s": N_ / ((sm + rm) * 1e-3),
                        "recover_GBps": (n_ + 1) * 8 * N_ / rm / 1e6, "verified": bool(scl.equals(f_, out_, secrets_))}
            del secrets_, shares_, out_
            torch.cuda.empty_cache()
        try:
            lib, kind = O.Ref(), "reference"
        except Exception:
            lib, kind = O.Port(), "port"
        r = lib.time_additive(O.M61, 1_000_000, n_)
        if r["mismatches"]:
            raise RuntimeError("CPU additive baseline failed its own round trip")
        rep["cpu"] = {"kind": kind, "cores": 1, "secrets": 1_000_000, "share_s": r["share_s"], "recover_s": r["recover_s"],
                      "share_secrets_per_s": 1e6 / r["share_s"], "reconstructions_per_s": 1e6 / r["recover_s"],
                      "round_trips_per_s": 1e6 / (r["share_s"] + r["recover_s"])}
        rep["verified"] = rep["gpu_1e6"]["verified"] and rep["gpu_1e8"]["verified"]
        return rep

    def ew_report():
        """The element-wise path north_star names first: Vector::add / multiplyEntryWise (vector.h:199-245) and FF::inverse
        (ff.h:203-246, small_ff.h:61-92) over Mersenne61 (10^8 elements), Mersenne127, the 128-bit Montgomery prime field (north_star's
        "Fp": BASELINE configs[2]) and GF(2^128) (10^7 each) through scl_hip_ew.
        HIP events around every launch, each kernel in its own steady state (100 warm-up launches for the sub-millisecond
        sizes, see share_recover_config); algorithmic bytes 3E for a binary op, 2E for a unary one.  The
        inverse's figure includes its 4-byte zero-flag read-back (the call returns the reference's error for a zero).  Checked
        against the CPU oracle on a window at each end of the batch and through x * x^-1 = 1, (a + b) - b = a over all of it."""
        import numpy as np
        import oracle_lib as O
        port = O.Port()
        rep = {"workload": "scl_hip_ew: add, multiplyEntryWise, inverse over whole vectors resident in HBM",
               "bytes_per_element": "3E for add / mul, 2E for inv (E = element bytes)", "fields": {}}
        win = min(2048, args.ew_elements) if args.ew_elements else 2048
        for fkey, N_ in (("m61", 100_000_000), ("m127", 10_000_000), ("mont128", 10_000_000), ("gf2_128", 10_000_000)):
            N_ = args.ew_elements or N_
            f_, L_ = tag_limbs(fkey)
            E_ = 8 * L_
            a, b, out = scl.empty(f_, N_), scl.empty(f_, N_), scl.empty(f_, N_)
            fill_random(a, f_, b"scl-bench-ew-a-" + fkey.encode())
            fill_random(b, f_, b"scl-bench-ew-b-" + fkey.encode())
            ha = np.concatenate([scl.to_host(a[:win]), scl.to_host(a[-win:])])
            hb = np.concatenate([scl.to_host(b[:win]), scl.to_host(b[-win:])])
            warm, reps = (100, 50) if N_ * E_ < 400_000_000 else (5, 10)
            legs, ok_all = {}, True
            for name, op, nb, two in (("add", scl.ADD, 3 * E_, True), ("mul", scl.MUL, 3 * E_, True), ("inv", scl.INV, 2 * E_, False)):
                tms = [scl.Timer() for _ in range(reps)]
                for k in range(-warm, reps):
                    if k >= 0:
                        tms[k].start()
                    scl.ew(f_, op, a, b if two else None, out=out)
                    if k >= 0:
                        tms[k].stop()
                torch.cuda.synchronize()
                ms = [tm.elapsed_ms() for tm in tms]
                mean = sum(ms) / reps
                got = np.concatenate([scl.to_host(out[:win]), scl.to_host(out[-win:])])
                ok = bool(np.array_equal(got, port.ew(f_, {"add": O.ADD, "mul": O.MUL, "inv": O.INV}[name], ha, hb if two else None)))
                if name == "add":
                    ok = ok and bool(scl.equals(f_, scl.ew(f_, scl.SUB, out, b), a))
                if name == "inv":
                    prod = scl.ew(f_, scl.MUL, out, a)
                    one = scl.to_device(np.ascontiguousarray(np.broadcast_to(port.from_int(f_, 1), (N_, L_))))   # FF::one(): R mod p in a Montgomery field
                    ok = ok and bool(scl.equals(f_, prod, one))
                    del prod, one
                # what limits the kernel: the streaming ops are HBM-bound; inverses are vector-ALU work (3 + I / L
                # modular products per element by simultaneous inversion); GF(2^128) products run on per-lane window tables in LDS
                bound = ("hbm" if name == "add" or (name == "mul" and fkey != "gf2_128") else
                         "lds tables + vector ALU" if fkey == "gf2_128" else "vector ALU (near HBM)" if fkey == "m61" else "vector ALU")
                legs[name] = {"ms": mean, "ms_min_max": [min(ms), max(ms)], "elements_per_s": N_ / (mean * 1e-3),
                              "bytes_per_element": nb, "GBps": nb * N_ / mean / 1e6, "frac": nb * N_ / mean / 1e6 / HBM_PEAK_GBPS,
                              "bound": bound, "verified": ok}
                ok_all = ok_all and ok
            # the reference's own element-wise path on one host core beside it (oracle/_ref: Vector::add / multiplyEntryWise and
            # FF::inverse element by element; the oracle port for GF(2^128), which the reference does not have), 10^6 elements
            cpu = None
            try:
                n_cpu = min(N_, 1_000_000 if fkey in ("m61", "m127") else 20_000)   # (the port's Fermat / bit-serial inverses are slow)
                ca, cb = scl.to_host(a[:n_cpu]), scl.to_host(b[:n_cpu])
                lib_, kind_ = port, "port"
                if fkey in ("m61", "m127"):       # (the reference has neither a 128-bit Montgomery field nor GF(2^128): the port)
                    try:
                        lib_, kind_ = O.Ref(), "reference"
                    except Exception:
                        pass
                cpu = {"kind": kind_, "cores": 1, "elements": n_cpu}
                for name, op, two in (("add", O.ADD, True), ("mul", O.MUL, True), ("inv", O.INV, False)):
                    t0_ = time.perf_counter()
                    lib_.ew(f_, op, ca, cb if two else None)
                    cpu[name + "_ns_per_element"] = (time.perf_counter() - t0_) * 1e9 / n_cpu
            except Exception as e:
                cpu = {"error": str(e)}
            rep["fields"][FIELD_NAMES[fkey]] = {"elements": N_, "dtype": {1: "u64", 2: "u128"}[L_], "warmup": warm, "launches": reps,
                                                **legs, "cpu_reference": cpu, "verified": ok_all}
            del a, b, out
            torch.cuda.empty_cache()
        rep["verified"] = all(v["verified"] for v in rep["fields"].values())
        return rep

    def layout_report():
        """The bridge every reference-layout caller crosses: AoS [secret][party] (the Vector per secret shamirSecretShare returns,
        shamir.h:52-68) <-> SoA [party][secret] (what the kernels stream), scl_hip_aos_to_soa / scl_hip_soa_to_aos at n = 10.
        Algorithmic bytes: every element read once and written once, 2 n E per secret."""
        rep = {"workload": "scl_hip_aos_to_soa / scl_hip_soa_to_aos, n = 10 parties", "bytes_per_secret": "2 n E", "fields": {}}
        for fkey, N_ in (("m61", 100_000_000), ("m127", 10_000_000), ("secp256k1", 10_000_000)):
            N_ = args.ew_elements or N_
            f_, L_ = tag_limbs(fkey)
            n_, E_ = 10, 8 * L_
            soa_ = scl.empty(f_, n_, N_)
            fill_random(soa_, f_, b"scl-bench-layout-" + fkey.encode())
            warm, reps = (100, 50) if N_ * n_ * E_ < 2_000_000_000 else (5, 10)
            legs = {}
            aos_ = scl.soa_to_aos(f_, soa_)
            back = scl.aos_to_soa(f_, aos_)
            ok = bool(scl.equals(f_, back.view(-1, L_), soa_.view(-1, L_)))
            # (AoS order checked against the definition on a window: aos[s][i] = soa[i][s])
            w_ = min(4096, N_)
            ok = ok and bool(torch.equal(aos_[:w_].transpose(0, 1), soa_[:, :w_])) and bool(torch.equal(aos_[-w_:].transpose(0, 1), soa_[:, -w_:]))
            del back
            for name in ("soa_to_aos", "aos_to_soa"):
                tms = [scl.Timer() for _ in range(reps)]
                for k in range(-warm, reps):
                    if k >= 0:
                        tms[k].start()
                    if name == "soa_to_aos":
                        scl.lib.scl_hip_soa_to_aos(f_, scl._dev(aos_), scl._dev(soa_), N_, N_, n_, scl._stream())
                    else:
                        scl.lib.scl_hip_aos_to_soa(f_, scl._dev(soa_), N_, scl._dev(aos_), N_, n_, scl._stream())
                    if k >= 0:
                        tms[k].stop()
                torch.cuda.synchronize()
                ms = [tm.elapsed_ms() for tm in tms]
                mean = sum(ms) / reps
                nb = 2 * n_ * E_
                legs[name] = {"ms": mean, "ms_min_max": [min(ms), max(ms)], "GBps": nb * N_ / mean / 1e6,
                              "frac": nb * N_ / mean / 1e6 / HBM_PEAK_GBPS}
            rep["fields"][FIELD_NAMES[fkey]] = {"secrets": N_, "n": n_, **legs, "verified": ok}
            del soa_, aos_
            torch.cuda.empty_cache()
        rep["verified"] = all(v["verified"] for v in rep["fields"].values())
        return rep

    def matmul_report():
        """Matrix::multiply (matrix.h:477-495) beyond the sharing shapes: square Mersenne61 products on the general matrix-core kernel
        (csrc/gemm_mfma.hpp: 8 signed base-256 digits per value, 64 digit-pair int8 products per 61-bit multiply-add, all fifteen
        digit diagonals accumulated in int32 over 8192 inner columns at a time), and the reference-shaped (200 x 7000)(7000 x 300).
        Roofline: the int8 matrix peak on executed operations, 2 x 64 per multiply-add.  Checked against the CPU oracle's i-k-j loop
        on a window of rows and columns that takes in the tile edges."""
        import numpy as np
        import oracle_lib as O
        port = O.Port()
        f_ = scl.M61
        rep = {"workload": "scl_hip_matmul over Mersenne61", "shapes": {}}
        for (M_, K_, N_) in ((4096, 4096, 4096), (1024, 1024, 1024), (200, 7000, 300)):
            n_ew = args.ew_elements
            if n_ew:      # tests: a small cube that still takes the matrix cores' general kernel
                M_, K_, N_ = (160, 8300, 96) if (M_, K_, N_) == (4096, 4096, 4096) else (M_ // 8 + 1, K_ // 8 + 1, N_ // 8 + 1)
            A = scl.vector_random(f_, M_ * K_, b"scl-bench-mm-A").reshape(M_, K_, 1)
            B = scl.vector_random(f_, K_ * N_, b"scl-bench-mm-B").reshape(K_, N_, 1)
            out = scl.empty(f_, M_, N_)
            warm, reps = 5, 10
            tms = [scl.Timer() for _ in range(reps)]
            for k in range(-warm, reps):
                if k >= 0:
                    tms[k].start()
                scl.matmul(f_, A, B, out=out)
                if k >= 0:
                    tms[k].stop()
            torch.cuda.synchronize()
            ms = sum(tm.elapsed_ms() for tm in tms) / reps
            rows = sorted({0, 31, 32, M_ // 2, M_ - 1} & set(range(M_)))
            cols = sorted({0, 31, 32, 63, 64, N_ // 2, N_ - 1} & set(range(N_)))
            hA, hB = scl.to_host(A[rows]), scl.to_host(B[:, cols])
            want = port.matmul(f_, np.ascontiguousarray(hA), np.ascontiguousarray(hB))
            got = scl.to_host(out)[np.ix_(rows, cols)]
            macs = M_ * K_ * N_
            on_cores = K_ > 64 and M_ >= 33 and N_ >= 33 and macs >= (1 << 25)
            rep["shapes"][f"{M_}x{K_}x{N_}"] = {
                "ms": ms, "T_multiply_adds_per_s": macs / ms / 1e9, "path": "matrix cores, general kernel" if on_cores else "vector ALU (tiled / split-K)",
                "int8_TOPs_executed": (2 * 64 * macs / ms / 1e9) if on_cores else None,
                "frac_of_int8_peak": (2 * 64 * macs / ms / 1e9 / I8_PEAK_TOPS) if on_cores else None, "verified": bool(np.array_equal(got, want))}
            del A, B, out
            torch.cuda.empty_cache()
        rep["verified"] = all(v["verified"] for v in rep["shapes"].values())
        return rep

    def open_step(fkey, n, t, N, chunk, steps, warmup, seed):
        """The MPC open of N secrets: every rank holds ceil(n/G) parties' share vectors, one all-gather per chunk
        brings all n rows to every rank, every rank reconstructs (as every MPC party does).  Timed three ways:
        the collective alone, the reconstruct kernel alone (on gathered chunks), and the double-buffered pipeline."""
        f_, L = tag_limbs(fkey)
        E = 8 * L
        per = sd.parties_per_rank(n, world)
        first, cnt = sd.party_slab(n, rank, world)
        # this rank's slab of a real sharing: all n rows are produced chunk-wise from the same seeds on every rank
        # (identical bits everywhere) and only the rank's own rows are kept
        secrets = scl.empty(f_, N)
        fill_random(secrets, f_, seed + b"-open-secrets")
        local = torch.zeros((per, N, L), dtype=torch.int64, device="cuda")
        gen = min(N, chunk)
        for s0 in range(0, N, gen):
            c = min(gen, N - s0)
            full = scl.shamir_share_prg(f_, secrets[s0:s0 + c], t, n, seed + b"-open", first_secret=s0)
            if cnt:
                local[:cnt, s0:s0 + c].copy_(full[first:first + cnt])
            del full
        lam = scl.lagrange_basis(f_, n)
        c0 = min(chunk, N)
        gathered = torch.empty((world * per, c0, L), dtype=torch.int64, device="cuda")
        piece = local[:, :c0].contiguous()

        def gather_only(k):
            if world > 1:
                dist.all_gather_into_tensor(gathered, piece)
            else:
                gathered.copy_(piece)
        t_gather = timed_region(gather_only, steps, warmup)[0] / steps
        out_c = scl.empty(f_, c0)
        tm = scl.Timer()
        scl.shamir_recover(f_, gathered[:n], lam, out=out_c)
        tm.start()
        for _ in range(steps):
            scl.shamir_recover(f_, gathered[:n], lam, out=out_c)
        tm.stop()
        rec_ms = tm.elapsed_ms() / steps
        result = {}

        def pipeline(k):
            result["out"] = sd.open_and_reconstruct(f_, local, n, lam, chunk=chunk) if world > 1 else \
                sd.open_and_reconstruct_local(f_, local, n, lam, chunk=chunk)
        t_pipe = timed_region(pipeline, steps, warmup)[0] / steps
        ok = bool(scl.equals(f_, result["out"], secrets))
        # the same open by partial sums (any field): each rank reduces its own parties, the ranks all-gather one element
        # per secret and rank, every rank adds them -- 1/parties_per_rank of the volume, every rank still learns every secret
        mine_rows = local[:cnt].contiguous() if cnt != per else local

        def partial_pipeline(k):
            result["pg"] = sd.open_by_partial_gather(f_, mine_rows, lam[first:first + cnt], chunk=chunk) if world > 1 else \
                sd.open_and_reconstruct_local(f_, local, n, lam, chunk=chunk)
        t_pg = timed_region(partial_pipeline, steps, warmup)[0] / steps
        ok_pg = bool(scl.equals(f_, result["pg"], secrets))
        # the same two opens behind the C ABI: RCCL called by the library itself (scl_hip_open_all_gather /
        # scl_hip_open_partial_gather; per-row grouped all-gathers, no packing copy), what a C++ caller of include/scl_hip/ gets
        c_abi = None
        try:
            if one_device and world > 1:   # a leg that did not run is not a leg that failed
                raise InterruptedError("skipped in the one-device rehearsal: RCCL does not take two ranks on one GPU")
            if args.inject_error == "c_abi":
                raise RuntimeError("injected by --inject-error c_abi")
            comm = sd.Communicator()
            try:
                def c_pipeline(k):
                    result["c"] = sd.open_all_gather_c(comm, f_, local, n, lam, chunk=chunk)
                t_c = timed_region(c_pipeline, steps, warmup)[0] / steps

                def c_partial(k):
                    result["cp"] = sd.open_partial_gather_c(comm, f_, mine_rows, lam[first:first + cnt], chunk=chunk)
                t_cp = timed_region(c_partial, steps, warmup)[0] / steps
                c_abi = {"pipeline_ms": 1e3 * t_c, "opened_secrets_per_s": N / t_c,
                         "partial_gather_pipeline_ms": 1e3 * t_cp, "partial_gather_opened_secrets_per_s": N / t_cp,
                         "verified": bool(scl.equals(f_, result["c"], secrets)) and bool(scl.equals(f_, result["cp"], secrets))}
            finally:
                comm.close()
        except InterruptedError as e:
            c_abi = {"skipped": str(e)}
        except Exception as e:   # reported in the line (and in its `errors`, with a non-zero exit code after the line is out)
            c_abi = {"error": str(e), "verified": False}
        gathered_bytes = world * per * c0 * E
        res = {"field": FIELD_NAMES[fkey], "n": n, "t": t, "secrets": N, "chunk": c0, "parties_per_rank": per, "c_abi": c_abi,
               "collective": "all_gather_into_tensor" if world > 1 else "none (1 rank: local copy)",
               "gather_ms_per_chunk": 1e3 * t_gather, "gathered_bytes_per_chunk": gathered_bytes,
               "rccl_algbw_GBps": gathered_bytes / t_gather / 1e9,
               "rccl_busbw_GBps": gathered_bytes / t_gather / 1e9 * (world - 1) / world,
               "reconstruct_ms_per_chunk": rec_ms, "reconstruct_GBps": (n + 1) * E * c0 / rec_ms / 1e6,
               "reconstruct_hbm_frac": (n + 1) * E * c0 / rec_ms / 1e6 / HBM_PEAK_GBPS,
               "pipeline_ms": 1e3 * t_pipe, "opened_secrets_per_s": N / t_pipe, "verified": ok,
               "partial_gather": {"collective": "all_gather_into_tensor of one partial sum per secret and rank" if world > 1
                                  else "none (1 rank: the chunked reconstruct)",
                                  "gathered_bytes_per_secret": world * E, "all_gather_bytes_per_secret": world * per * E,
                                  "pipeline_ms": 1e3 * t_pg, "opened_secrets_per_s": N / t_pg, "verified": ok_pg}}
        del secrets, local, gathered, piece, out_c, result
        torch.cuda.empty_cache()
        return res

    def open_partial_sums(n, t, N, steps, warmup, seed):
        """Mersenne61 alternative: canonical partial sums + ONE reduce-scatter(SUM)"""
        f_ = scl.M61
        first, cnt = sd.party_slab(n, rank, world)
        secrets = scl.empty(f_, N)
        fill_random(secrets, f_, seed + b"-ps-secrets")
        full = scl.shamir_share_prg(f_, secrets, t, n, seed + b"-ps")
        local = full[first:first + cnt].contiguous()
        del full
        lam = scl.lagrange_basis(f_, n)
        result = {}

        def run(k):
            result["mine"] = sd.open_by_partial_sums(local, lam[first:first + cnt]) if world > 1 else \
                scl.shamir_recover(f_, local, lam)
        t_ps = timed_region(run, steps, warmup)[0] / steps
        lo = rank * (N // world)
        ok = bool(scl.equals(f_, result["mine"].reshape(-1, 1), secrets[lo:lo + N // world]))
        moved = N * 8
        res = {"field": "Mersenne61", "n": n, "t": t, "secrets": N,
               "collective": "reduce_scatter_tensor(SUM, int64)" if world > 1 else "none (1 rank)",
               "ms": 1e3 * t_ps, "opened_secrets_per_s": N / t_ps, "reduce_scatter_input_bytes": moved,
               "rccl_busbw_GBps": moved / t_ps / 1e9 * (world - 1) / world, "verified": ok}
        del secrets, local, result
        torch.cuda.empty_cache()
        return res

    def c4_rank_shape(N, chunk):
        """What ONE rank of BASELINE configs[3] on 8 GPUs does per open by scl_hip_open_partial_gather: its 5 of the 40 parties'
        share vectors of all N = 10^8 GF(2^128) secrets (an 8 GB slab) go through the reconstruct kernel at m = 5 (one partial
        sum per secret), and each gathered chunk of 8 partial rows through k_additive_recover (Vector::sum per secret).  The
        kernels of the 8-GPU configuration that fit one GPU, at their real size; the all-gather between them is xGMI time
        (DESIGN.md section 5)."""
        f_, L = tag_limbs("gf2_128")
        E = 8 * L
        world8, per = 8, 5
        lam = scl.lagrange_basis(f_, 40)
        local = scl.empty(f_, per, N)
        fill_random(local, f_, b"scl-bench-c4-rank")
        partial = scl.empty(f_, N)
        c = min(chunk, N)
        gathered = scl.empty(f_, world8, c)
        fill_random(gathered, f_, b"scl-bench-c4-rank-g")
        outc = scl.empty(f_, c)
        reps = 3
        t1, t2 = scl.Timer(), scl.Timer()
        scl.shamir_recover(f_, local, lam[:per], out=partial)
        t1.start()
        for _ in range(reps):
            scl.shamir_recover(f_, local, lam[:per], out=partial)
        t1.stop()
        scl.additive_recover(f_, gathered, out=outc)
        t2.start()
        for _ in range(reps * 4):
            scl.additive_recover(f_, gathered, out=outc)
        t2.stop()
        p_ms, s_ms = t1.elapsed_ms() / reps, t2.elapsed_ms() / (reps * 4)
        # linearity as the check: the partial over 5 rows + the partial over the same rows with lambda' = the sum over five
        # other coefficients ... kept simple: partial(lam) + partial(lam2) == partial(lam + lam2), all three by the kernel
        lam2 = lam[per:2 * per]
        lam3 = scl.to_host(scl.ew(f_, scl.ADD, scl.to_device(lam[:per]), scl.to_device(lam2)))
        w = min(N, 1 << 20)
        pa = scl.shamir_recover(f_, local[:, :w], lam[:per])
        pb = scl.shamir_recover(f_, local[:, :w], lam2)
        pc = scl.shamir_recover(f_, local[:, :w], lam3)
        ok = bool(scl.equals(f_, scl.ew(f_, scl.ADD, pa, pb), pc))
        pbytes, sbytes = (per + 1) * E, (world8 + 1) * E
        sums_ms_total = s_ms * (N / c)
        res = {"workload": f"one rank's kernels of BASELINE configs[3] on 8 GPUs by the partial-sum open: {per} parties x {N} "
                           f"GF(2^128) secrets -> partial sums (reconstruct kernel, m = {per}), then Vector::sum over {world8} "
                           f"gathered partial rows per chunk of {c}",
               "partial_ms": p_ms, "partial_bytes_per_secret": pbytes, "partial_GBps": pbytes * N / p_ms / 1e6,
               "partial_frac": pbytes * N / p_ms / 1e6 / HBM_PEAK_GBPS,
               "sum_ms_per_chunk": s_ms, "sum_bytes_per_secret": sbytes, "sum_GBps": sbytes * c / s_ms / 1e6,
               "sum_frac": sbytes * c / s_ms / 1e6 / HBM_PEAK_GBPS,
               "kernels_ms_per_open": p_ms + sums_ms_total, "opened_secrets_per_s_kernels_only": N / ((p_ms + sums_ms_total) * 1e-3),
               "xgmi_bytes_received_per_rank": (world8 - 1) * E * N, "verified": ok}
        del local, partial, gathered, outc
        torch.cuda.empty_cache()
        return res

    def open_report():
        N_open = args.open_secrets or 12_500_000 * world
        N_ps = (args.secrets // world) * world
        rep = {"c4_all_gather": open_step("gf2_128", 40, 13, N_open, args.open_chunk, 3, 1, b"scl-bench-open"),
               "m61_partial_sums": open_partial_sums(10, 3, N_ps, 3, 1, b"scl-bench-open")}
        if world == 1 and args.c4_rank_secrets:
            try:
                rep["c4_rank_shape"] = c4_rank_shape(args.c4_rank_secrets, args.open_chunk)
            except Exception as e:
                rep["c4_rank_shape"] = {"error": str(e), "verified": False}
                torch.cuda.empty_cache()
        return rep

    if pl["key"] == "c4":
        # BASELINE configs[3]: the exchange step is the timed step.  Every rank holds ceil(40 / G) parties' share vectors of
        # ALL the secrets, one all-gather per chunk brings the 40 rows together, every rank reconstructs every secret.
        c4 = open_step("gf2_128", 40, 13, pl["total"], args.open_chunk, args.steps, args.warmup, b"scl-bench-c4")
        if rank == 0:
            line = {"metric": "shamir_reconstructions_per_sec", "value": c4["opened_secrets_per_s"],
                    "unit": "reconstructions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": c4["pipeline_ms"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                    "dtype": "u128", "data": "synthetic",
                    "config": {"workload": pl["workload"], "field": FIELD_NAMES["gf2_128"], "n": 40, "t": 13,
                               "total_secrets": pl["total"], "chunk": c4["chunk"], "parallelism": pl["parallelism"]},
                    "roofline": {"bound": "hbm", "kernel": "shamir_recover", "achieved": c4["reconstruct_GBps"],
                                 "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": c4["reconstruct_hbm_frac"], "traffic": None},
                    "rccl_busbw_GBps": c4["rccl_busbw_GBps"], "verified": c4["verified"] and c4["partial_gather"]["verified"],
                    "open": {"c4_all_gather": c4}}
            if world == 1 and args.cpu_sample > 0:
                line["cpu_baseline"] = cpu_baseline("gf2_128", 40, 13, min(args.cpu_sample, 20_000))
            emit(line)
        if world > 1:
            dist.destroy_process_group()
        return

    if args.mode == "open":
        rep = open_report()
        if rank == 0:
            c4 = rep["c4_all_gather"]
            line = {"metric": "shamir_open_reconstructions_per_sec", "value": c4["opened_secrets_per_s"],
                    "unit": "reconstructions/s", "n_gpus": world, "steps": 3, "warmup": 1,
                    "ms_per_step": c4["pipeline_ms"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                    "dtype": "u128", "data": "synthetic",
                    "config": {"workload": f"open (all-gather + reconstruct) n=40 t=13 GF(2^128) {c4['secrets']} secrets, "
                                           f"{c4['parties_per_rank']} parties per rank (BASELINE configs[3] exchange step)",
                               "parallelism": f"parties{world}"},
                    "roofline": {"bound": "hbm", "kernel": "shamir_recover", "achieved": c4["reconstruct_GBps"],
                                 "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": c4["reconstruct_hbm_frac"], "traffic": None},
                    "open": rep}
            emit(line)
        if world > 1:
            dist.destroy_process_group()
        return

    # =============================================== headline ===============================================
    if pl["key"] == "c5":
        args.field, args.n, args.t, args.secrets = pl["field"], pl["n"], pl["t"], pl["mine"]
        args.configs = args.open = 0
        if args.steps == 20 and args.warmup == 3:   # the defaults are the headline's: a (128,42) step is ~80 ms per 1.25e8
            args.steps, args.warmup = 5, 1
        args.cpu_sample = min(args.cpu_sample, 20_000)
    f, L = tag_limbs(args.field)
    E = 8 * L
    n, t, N = args.n, args.t, args.secrets
    seed = f"scl-bench-{args.field}-{rank}".encode()
    lam = scl.lagrange_basis(f, n)
    tc = t if args.share_mode == "coeffs" else 0
    # The steps rotate over `nsets` independently allocated operand sets, all alive at once: where a buffer lands moves these
    # kernels by up to 10 % (Placement), so ONE 12 GB set says as much about the allocation as about the
    # kernel.  Every step does the same work whichever set it runs on; the line carries the mean and the per-allocation means.
    nsets = args.allocations or (3 if (pl["key"] == "c2" and N * n * E <= 8_000_000_000) else 1)
    sets = []
    for a in range(nsets):
        secrets = scl.empty(f, N)
        coeffs = scl.empty(f, tc, N) if tc else None
        fill_random(secrets, f, seed + (b"-secrets" if a == 0 else b"-secrets%d" % a))
        if tc:
            fill_random(coeffs, f, seed + (b"-coeffs" if a == 0 else b"-coeffs%d" % a))
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

    elapsed, rank_elapsed = timed_region(step, args.steps, args.warmup)

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
    secrets, coeffs, shares, out = sets[0]
    del sets[1:]
    torch.cuda.empty_cache()

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
    torch.cuda.empty_cache()

    def headline_line(rccl):
        rec_bytes = (n + 1) * E            # n shares in, 1 secret out            
        share_bytes = (1 + t) * E + n * E if args.share_mode == "coeffs" else E + n * E
        kernels = {
            "shamir_recover": {"ms": rec_ms, "bytes_per_secret": rec_bytes, "GBps": rec_bytes * N / (rec_ms * 1e-3) / 1e9},
            "shamir_share": {"ms": share_ms, "bytes_per_secret": share_bytes,
                             "GBps": share_bytes * N / (share_ms * 1e-3) / 1e9},
        }
        dom = "shamir_share" if share_ms >= rec_ms else "shamir_recover"
        ach = kernels[dom]["GBps"]
        roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBPS, "traffic": pmc_traffic(dom, args),
                    "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes run by the builder over this command, stamped "
                                      "with a hash of the kernel sources; null once they differ) -- not observed by this run",
                    "algorithmic_bytes": kernels[dom]["bytes_per_secret"] * N,
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
        line = {
            "metric": "shamir_reconstructions_per_sec", "value": total / elapsed, "unit": "reconstructions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_by_rank": [1e3 * x / args.steps for x in rank_elapsed], "rccl": rccl,
            "higher_is_better": True, "scaling": pl["scaling"], "vs_baseline": None,
            "dtype": {1: "u64", 2: "u128", 4: "u256"}[L], "data": "synthetic",
            "config": {"workload": pl["workload"],
                       "field": FIELD_NAMES[args.field], "n": n, "t": t, "secrets_per_gpu": N, "total_secrets": pl["total"],
                       "share_mode": args.share_mode, "layout": "SoA [party][secret]",
                       "allocation": f"plain; the steps rotate over {nsets} independently allocated operand set(s)",
                       "parallelism": pl["parallelism"]},
            "roofline": roofline, "kernels": kernels, "by_allocation": by_alloc, "verified": verified,
            "reconstruct_only_per_s": N * world / (rec_ms * 1e-3), "share_only_per_s": N * world / (share_ms * 1e-3),
        }
        return line, roofline, kernels, dom

    # The multi-rank legs below run RCCL through torch and through the C ABI -- code no one-GPU box can rehearse with real RCCL.
    # If one of them never returns, the headline measured above must not be lost with it: after --side-timeout seconds rank 0
    # writes the line with what it has (the missing legs as errors) and every rank leaves.
    watchdog = None
    if world > 1 and args.side_timeout > 0:
        import threading

        def give_up():
            if rank == 0:
                why = f"no result after {args.side_timeout:g} s: the process was ended by bench.py's own watchdog"
                early, _, _, _ = headline_line({"error": why})
                early["open"] = {"error": why}
                legs, errors = side_legs(early)
                early["verified_headline"], early["verified"], early["verified_legs"], early["errors"] = verified, False, legs, errors
                emit(early)
                print("bench.py: " + why, file=sys.stderr)
            os._exit(3)
        # (the other ranks leave a little later: a launcher that sees a worker die ends the rest, rank 0's line must be out by then)
        watchdog = threading.Timer(args.side_timeout + (0 if rank == 0 else 5), give_up)
        watchdog.daemon = True
        watchdog.start()
    rccl = rccl_report()                                 # (collective: every rank)
    if args.inject_error == "hang" and rank == world - 1:
        time.sleep(10 ** 6)                              # (test: a rank that never reaches the collectives)
    open_rep = open_report() if args.open else None      # every rank takes part in the collectives
    if watchdog is not None:
        watchdog.cancel()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    line, roofline, kernels, dom = headline_line(rccl)
    if open_rep is not None:
        line["open"] = open_rep
    if world == 1 and args.configs and pl["key"] == "c2":
        try:
            line["prg_mode"] = prg_mode_report()
        except Exception as e:
            line["prg_mode"] = {"error": str(e), "verified": False}
            torch.cuda.empty_cache()
        try:
            line["c1_additive"] = c1_additive_report()
        except Exception as e:
            line["c1_additive"] = {"error": str(e), "verified": False}
            torch.cuda.empty_cache()
    if world == 1 and args.ew and pl["key"] == "c2":
        try:
            line["ew"] = ew_report()
        except Exception as e:
            line["ew"] = {"error": str(e), "verified": False}
            torch.cuda.empty_cache()
    if world == 1 and args.ew and pl["key"] == "c2":
        try:
            line["layout"] = layout_report()
        except Exception as e:
            line["layout"] = {"error": str(e), "verified": False}
            torch.cuda.empty_cache()
    if world == 1 and args.ew and pl["key"] == "c2":
        try:
            line["matmul"] = matmul_report()
        except Exception as e:
            line["matmul"] = {"error": str(e), "verified": False}
            torch.cuda.empty_cache()
    if world == 1 and args.configs:
        # the other BASELINE configurations at the size ONE GPU holds of them (C4, C5: an eighth), after the timed region
        cfgs = {}
        for key, (fk, n_, t_, N_, st, wu, na) in {
            "C3_mersenne127_10_3": ("m127", 10, 3, 10_000_000, 50, 100, 3),
            "C3_mont128_10_3": ("mont128", 10, 3, 10_000_000, 50, 100, 3),
            "F3_secp256k1_scalar_10_3": ("secp256k1", 10, 3, 10_000_000, 50, 100, 1),   # Feldman / Pedersen's field
            "C4_shard_gf2_128_40_13": ("gf2_128", 40, 13, 12_500_000, 5, 2, 1),
            "C5_shard_mersenne61_128_42": ("m61", 128, 42, 125_000_000, 2, 1, 1),
        }.items():
            try:
                cfgs[key] = share_recover_config(fk, n_, t_, N_, st, b"scl-bench-" + key.encode(), warmup=wu, allocations=na)
                cfgs[key]["traffic"] = pmc_config_traffic(key)
            except Exception as e:  # a failed side configuration is reported, not hidden, and never fails the headline
                cfgs[key] = {"error": str(e), "verified": False}
                torch.cuda.empty_cache()
        line["configs"] = cfgs
    if world == 1 and args.cpu_sample > 0:
        line["cpu_baseline"] = cpu_baseline(args.field, n, t, args.cpu_sample)
    if (world == 1 and args.pmc_live and pl["key"] == "c2" and roofline.get("bound") == "hbm"
            and (args.field, args.n, args.t, args.share_mode) == ("m61", 10, 3, "coeffs") and args.secrets >= 10_000_000):
        torch.cuda.empty_cache()
        live, line["pmc_live"] = live_pmc_traffic(args)
        if live is not None:
            cal = live.get("calibration_k_copy16", {})
            roofline["traffic_stamped"] = roofline["traffic"]
            roofline["traffic"] = live[dom]["bytes"]
            roofline["traffic_over_algorithmic"] = live[dom]["bytes"] / roofline["algorithmic_bytes"]
            roofline["traffic_live"] = {"fetch_doubled": True, "fetch_correction_measured_on_k_copy16": cal.get("fetch_correction"),
                                        "shamir_share": live["shamir_share"], "shamir_recover": live["shamir_recover"]}
            roofline["traffic_source"] = ("observed in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over two "
                                          "child runs of this command's GPU legs (headline 3 steps, after the timed regions), mean per "
                                          "launch; FETCH_SIZE doubled per MI355X_MICROARCH.md and checked on k_copy16 in the same pass; "
                                          "traffic_stamped = the builder's figure from profiles/pmc_traffic.json")
            for key, c in live.get("configs", {}).items():   # the side configurations' kernels, from the same two passes
                if key in line.get("configs", {}) and "error" not in line["configs"][key]:
                    line["configs"][key]["traffic_stamped"] = line["configs"][key].get("traffic")
                    line["configs"][key]["traffic"] = {"share": c["share"]["bytes"], "recover": c["recover"]["bytes"],
                                                       "share_kernel": c["share"]["kernel"], "recover_kernel": c["recover"]["kernel"],
                                                       "share_over_algorithmic": c["share"]["traffic_over_algorithmic"],
                                                       "recover_over_algorithmic": c["recover"]["traffic_over_algorithmic"],
                                                       "source": "observed in this run (roofline.traffic_source)"}
    # the line tells the truth about its side legs: `verified` is the AND over the headline and every leg that ran, and a
    # leg that failed (an {"error": ..} object anywhere in the line) makes the process exit non-zero AFTER the line is out
    legs, errors = side_legs(line)
    line["verified_headline"] = verified
    line["verified"] = verified and all(legs.values())
    line["verified_legs"] = legs
    if errors:
        line["errors"] = errors
    emit(line)
    if world > 1:
        dist.destroy_process_group()
    if errors or not line["verified"]:
        print("bench.py: " + ("; ".join(errors) if errors else "a leg did not verify: " +
                               ", ".join(k for k, v in legs.items() if not v)), file=sys.stderr)
        sys.exit(1)


def side_legs(line):
    """({leg path: verified}, [error strings]) over every object of the result line that carries a `verified` or an `error`
    key below the top level (configs.*, prg_mode, c1_additive, open.* and open.*.c_abi / partial_gather ..)"""
    legs, errors = {}, []

    def walk(obj, path):
        if isinstance(obj, dict):
            if path:
                if "error" in obj:
                    errors.append(f"{path}: {obj['error']}")
                    legs[path] = False
                elif "verified" in obj:
                    legs[path] = bool(obj["verified"])
            for k, v in obj.items():
                if k not in ("cpu_baseline", "cpu_reference", "verified_legs"):
                    walk(v, f"{path}.{k}" if path else k)
    walk(line, "")
    return legs, errors


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
