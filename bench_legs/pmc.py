"""HBM traffic from the PMC counters: the stamped figures of profiles/pmc_traffic.json (valid while the kernel sources hash
the same), the parsing of rocprofv3's counter CSVs, and the opt-in live passes (two child runs of bench.py under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`)."""
import hashlib
import json
import os
import subprocess
import sys
import time

from .common import ROOT

KERNEL_SOURCES = ("secure-computation-library_amd/csrc/kernels.hpp", "secure-computation-library_amd/csrc/capi.hip",
                  "secure-computation-library_amd/csrc/share_mfma.hpp", "secure-computation-library_amd/csrc/gemm_mfma.hpp",
                  "secure-computation-library_amd/csrc/gemm_unit.hip", "include/scl_hip/detail/field.hpp")
PMC_NEEDLES = {"shamir_share": "k_share_small_t<sclhip::M61", "shamir_recover": "k_recover_fixed<sclhip::M61"}
# the side configurations by the kernels they launch, with their algorithmic bytes per launch (share, reconstruct)
PMC_CONFIGS = {
    "C3_mersenne127_10_3": ("k_share_small_t<sclhip::M127", "k_recover_fixed<sclhip::M127", (224 * 10**7, 176 * 10**7)),
    "C3_mont128_10_3": ("k_share_small_t<sclhip::Mont128", "k_recover_small<sclhip::Mont128", (224 * 10**7, 176 * 10**7)),
    "F3_secp256k1_scalar_10_3": ("k_share_small_pair<sclhip::Mont256<sclhip::SecpOrderParams>",
                                 "k_recover_small<sclhip::Mont256<sclhip::SecpOrderParams>", (448 * 10**7, 352 * 10**7)),
    "C4_shard_gf2_128_40_13": ("k_share_gf_tiles<13>", "k_recover_gf128_pos<512", (864 * 125 * 10**5, 656 * 125 * 10**5)),
    "C5_shard_mersenne61_128_42": ("k_share_mfma_m61", "k_recover_table<sclhip::M61", (1368 * 125 * 10**6, 1032 * 125 * 10**6)),
}


def kernel_source_hash():
    """sha256 over the kernel sources: what profiles/pmc_traffic.json is stamped with (a rebuilt .so of the same
    sources need not be byte-identical, the sources are)"""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stamped():
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
        pmc = json.load(fh)
    return pmc if pmc.get("kernel_source_sha256_16") == kernel_source_hash() else None


def pmc_traffic(dom, args):
    """HBM bytes per launch of the dominant kernel from the committed PMC run -- only while that run still describes
    the code: the file carries the configuration, the kernel symbols and a hash of the kernel sources it was taken
    with; any mismatch gives null (bench.py cannot read PMCs itself)."""
    try:
        pmc = _stamped()
        c = pmc["config"]
        if (c["field"], c["n"], c["t"], c["secrets_per_gpu"], c["share_mode"]) != (
                args.field, args.n, args.t, args.secrets, args.share_mode):
            return None
        return pmc[dom]["bytes"]
    except Exception:
        return None


def pmc_config_traffic(key):
    """{"share": bytes, "recover": bytes} per launch for a side configuration, from the same stamped PMC file (null when the
    kernel sources have changed since, or the configuration was not in the PMC run)"""
    try:
        c = _stamped()["configs"][key]
        return {"share": c["share"]["bytes"], "recover": c["recover"]["bytes"],
                "share_kernel": c["share"]["kernel"], "recover_kernel": c["recover"]["kernel"]}
    except Exception:
        return None


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


def live_pmc_traffic(args, bench_py, timeout_s=240):
    """HBM bytes per launch OBSERVED in this run: two child processes run this command's GPU legs once more (headline 3 steps,
    the side configurations when --configs is on; no open step, no CPU baseline) under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (the two cannot share a pass on gfx950; no trace domain is combined with --pmc), and pmc_report reads the
    kernels' counter means from their CSVs.  The program itself follows `--`.  Children of this process, started after every
    timed region.  Returns (report or None, info): info = {"ran", "seconds", "fallback_reason"} goes into the detail file as
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
    work = tempfile.mkdtemp(prefix="scl_pmc_", dir="/tmp")
    child = [sys.executable, bench_py, "--configs", str(args.configs), "--open", "0", "--cpu-sample", "0",
             "--pmc-live", "0", "--ew", "0", "--allocations", "1", "--steps", "3", "--warmup", "1", "--field", args.field,
             "--n", str(args.n), "--t", str(args.t), "--secrets", str(args.secrets), "--share-mode", args.share_mode,
             "--detail", os.path.join(work, "detail.json")]
    env = dict(os.environ, TMPDIR="/tmp")
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
