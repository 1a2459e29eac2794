"""cpu_baseline: the reference's own CPU path (oracle/_ref, else the oracle port) timed on this box's host cores."""
import os
import time

from .common import FIELD_TAGS

CPU_BASELINE_SECONDS = 10.0   # of single-thread work in the default run (the contract asks for a bounded sample of 10-30 s)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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


def cpu_baseline(field_key, n, t, sample, all_cores=False, seconds=CPU_BASELINE_SECONDS):
    """The reference CPU path on this box's host cores (single thread, like SCL itself): per secret
    shamirSecretShare + shamirRecoverP(shares).  oracle/_ref (the real reference, prebuilt) when it
    is there, else the oracle port.  `all_cores`: also the harness-threaded and hoisted-basis variants SURVEY.md section 8d
    asks for (three more bounded legs; opt-in with --cpu-all-cores 1, they go to the detail file)."""
    import oracle_lib as O
    kind = "reference"
    try:
        lib = O.Ref()
    except Exception:
        lib, kind = O.Port(), "port"
    f = FIELD_TAGS[field_key]
    # A pilot of a few secrets bounds the sample to about `seconds` of work whatever the shape costs (the per-secret
    # Lagrange basis is n(n-1) field inversions: 40 parties over GF(2^128) in the oracle port take milliseconds per
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
    sample = max(16, min(sample, int(seconds / per_secret)))
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
    if not all_cores:
        return out
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
