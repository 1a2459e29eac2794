"""The other BASELINE configurations at the size one GPU holds of them, and BASELINE configs[1] in the reference's own
(PRG-driven) mode."""
from .common import DTYPES, FIELD_NAMES, HBM_PEAK_GBPS, mfma_share_roofline, on_matrix_cores
from .compute_roofline import lds_roofline, valu_roofline
from .pmc import pmc_config_traffic

# key: (field, n, t, secrets, timed launches, warm-up launches, independently allocated operand sets)
SIDE_CONFIGS = {
    "C3_mersenne127_10_3": ("m127", 10, 3, 10_000_000, 50, 100, 3),
    "C3_mont128_10_3": ("mont128", 10, 3, 10_000_000, 50, 100, 3),
    "F3_secp256k1_scalar_10_3": ("secp256k1", 10, 3, 10_000_000, 50, 100, 1),   # Feldman / Pedersen's field
    "C4_shard_gf2_128_40_13": ("gf2_128", 40, 13, 12_500_000, 5, 2, 1),
    "C5_shard_mersenne61_128_42": ("m61", 128, 42, 125_000_000, 2, 1, 1),
}


def share_recover_config(ctx, fkey, n, t, N, steps, seed, warmup=1, allocations=1):
    """one configuration end to end on this GPU: plain allocations, share + reconstruct timed with HIP events on the
    launch stream, round trip verified.  Returns the per-kernel figures.  `warmup` untimed launches first: the first
    25-30 ms of load after an idle gap (allocation, fill) run under a clock ramp -- from a cold start the 0.38 ms Mont128
    share kernel reads 0.51, 0.50, 0.49 .. and reaches 0.38 only after about sixty launches, the headline's 1.7 ms kernel after
    five (profiles/r4_probe_c3_seq.txt, r4_probe_headline_seq.txt) -- so the sub-millisecond configurations take 100
    warm-up launches (40 ms) and 50 timed ones per kernel.  `allocations` > 1: the whole measurement on that many
    independently allocated operand sets, all alive at once (so they are different memory); the figures are the MEAN over
    all of them -- where the operands land moves these kernels by up to 10 % (Placement), and one
    1.6 GB set says more about the allocation than about the kernel -- with the per-allocation means beside it.
    Each kernel is timed on its own, `steps` launches back to back (alternating the two, as the headline's step does by
    contract, leaves every launch behind the other kernel's tail: 6-10 % spread instead of 3-5 %, profiles/r4_probe_c3_bench.txt)."""
    scl = ctx.scl
    f_, L = ctx.tag_limbs(fkey)
    E = 8 * L
    lam = scl.lagrange_basis(f_, n)
    sets = []
    for a in range(allocations):
        secrets = scl.empty(f_, N)
        coeffs = scl.empty(f_, t, N)
        ctx.fill_random(secrets, f_, seed + b"-secrets%d" % a)
        ctx.fill_random(coeffs, f_, seed + b"-coeffs%d" % a)
        sets.append((secrets, coeffs, scl.empty(f_, n, N), scl.empty(f_, N)))
    share_all, rec_all, share_by, rec_by, ok = [], [], [], [], True
    for secrets, coeffs, shares, out in sets:
        sh = ctx.timed_launches(lambda: scl.shamir_share(f_, secrets, coeffs, n, out=shares), steps, warmup)
        rc = ctx.timed_launches(lambda: scl.shamir_recover(f_, shares, lam, out=out), steps, warmup)
        share_all += sh
        rec_all += rc
        share_by.append(sum(sh) / steps)
        rec_by.append(sum(rc) / steps)
        ok = ok and bool(scl.equals(f_, out, secrets))
    sm, rm = sum(share_all) / len(share_all), sum(rec_all) / len(rec_all)
    sb, rb = (1 + t + n) * E, (n + 1) * E
    res = {"field": FIELD_NAMES[fkey], "n": n, "t": t, "secrets": N, "dtype": DTYPES[L],
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
    if fkey == "gf2_128" and (n, t) == (40, 13):
        # neither kernel of C4 is HBM-bound: the ceilings they do run against (bench_legs/compute_roofline.py)
        res["share_roofline_compute"] = valu_roofline("c4_share", N / (sm * 1e-3))
        res["recover_roofline_compute"] = lds_roofline("c4_recover", N / (rm * 1e-3))
        res["recover_roofline_valu"] = valu_roofline("c4_recover", N / (rm * 1e-3))     # (its second limit: 3.7 vector instructions per lookup)
    del sets, secrets, coeffs, shares, out
    ctx.free()
    return res


def configs_report(ctx):
    cfgs = {}
    for key, (fk, n_, t_, N_, st, wu, na) in SIDE_CONFIGS.items():
        try:
            cfgs[key] = share_recover_config(ctx, fk, n_, t_, N_, st, b"scl-bench-" + key.encode(), warmup=wu, allocations=na)
            cfgs[key]["traffic"] = pmc_config_traffic(key)
        except Exception as e:  # a failed side configuration is reported, not hidden, and never fails the headline
            cfgs[key] = {"error": str(e), "verified": False}
            ctx.free()
    return cfgs


def prg_mode_report(ctx):
    """BASELINE configs[1] in the reference's OWN mode: scl::ss::shamirSecretShare(secret, t, n, prg) draws the
    coefficients from the PRG (shamir.h:51-68, prg.cc:124-146); here that is scl_hip_shamir_share_prg, bit-identical to
    the per-secret calls on one PRG.  After and outside the headline's timed region; AES rate of k_prg_blocks in the same
    run beside it (the share draws 2 blocks per secret: Vector::random(4) of 8-byte elements)."""
    scl, torch = ctx.scl, ctx.torch
    f_, n_, t_, N_ = scl.M61, 10, 3, 100_000_000
    sd_ = b"scl-bench-prg-mode"
    secrets_ = scl.empty(f_, N_)
    ctx.fill_random(secrets_, f_, sd_ + b"-secrets")
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
    ctx.free()
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
    ctx.free()
    sb = (1 + n_) * 8
    return {"workload": f"scl_hip_shamir_share_prg n={n_} t={t_} Mersenne61 {N_} secrets (coefficients drawn from the "
                        "AES-128-CTR PRG inside the call, the reference's mode) + reconstruct",
            "share_ms": sm, "share_secrets_per_s": N_ / (sm * 1e-3), "aes_blocks_per_secret": bps,
            "share_aes_blocks_per_s": nb / (sm * 1e-3), "share_GBps": sb * N_ / sm / 1e6,
            "share_frac": sb * N_ / sm / 1e6 / HBM_PEAK_GBPS, "recover_ms": rm,
            "round_trips_per_s": N_ / ((sm + rm) * 1e-3),
            "k_prg_blocks": {"blocks": nb, "ms": bm, "blocks_per_s": nb / (bm * 1e-3),
                             "roofline_compute": lds_roofline("prg_blocks", nb / (bm * 1e-3))},
            "share_roofline_compute": lds_roofline("prg_blocks", nb / (sm * 1e-3)), "verified": ok}


def c1_additive_report(ctx):
    """BASELINE configs[0]: additive sharing over Mersenne61, n = 3, 10^6 secrets -- the reference's per-secret
    additiveShare + Vector::sum on one host core (oracle/_ref), beside the GPU kernels at the SAME size (10^6 secrets is
    24 MB: the kernels are launch-bound there) and at 10^8."""
    import oracle_lib as O
    scl = ctx.scl
    f_, n_ = scl.M61, 3
    rep = {"workload": "additive sharing Mersenne61 n=3 (BASELINE configs[0])"}
    for N_, key in ((1_000_000, "gpu_1e6"), (100_000_000, "gpu_1e8")):
        sd_ = b"scl-bench-c1"
        secrets_ = scl.empty(f_, N_)
        ctx.fill_random(secrets_, f_, sd_ + b"-secrets")
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
        ctx.free()
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
