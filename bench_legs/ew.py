"""The element-wise path north_star names first, the layout bridge and Matrix::multiply."""
import time

from .common import DTYPES, FIELD_NAMES, HBM_PEAK_GBPS, I8_PEAK_TOPS
from .compute_roofline import SQ, lds_roofline, valu_roofline

EW_FIELDS = (("m61", 100_000_000), ("m127", 10_000_000), ("mont128", 10_000_000), ("gf2_128", 10_000_000))
# what the rolled inversion moves through HBM per element once its chains outgrow L2 + MALL: x read,
# the prefix product written, prefix and x read again on the walk back, the result written = 5E against 2E algorithmic
ROLLED_INVERSE_TRAFFIC_OVER_ALGORITHMIC = 2.5


def ew_report(ctx):
    """Vector::add / multiplyEntryWise (vector.h:199-245) and FF::inverse (ff.h:203-246, small_ff.h:61-92) over Mersenne61
    (10^8 elements), Mersenne127, the 128-bit Montgomery prime field (north_star's "Fp": BASELINE configs[2]) and GF(2^128)
    (10^7 each) through scl_hip_ew.  HIP events around every launch, each kernel in its own steady state (100 warm-up launches
    for the sub-millisecond sizes, see configs.share_recover_config); algorithmic bytes 3E for a binary op, 2E for a unary one
    (SURVEY.md section 8d).  The inverse is timed twice: `inv` is the CALL the reference's semantics ask for (it returns the
    reference's error for a zero, so it ends in a stream synchronisation) and `inv_async` is scl_hip_ew_status, which leaves
    the zero flag on the device.  Legs HBM does not bound carry `roofline_compute` (bench_legs/compute_roofline.py).
    Checked against the CPU oracle on a window at each end of the batch and through x * x^-1 = 1, (a + b) - b = a over all of it."""
    import numpy as np
    import oracle_lib as O
    scl, args = ctx.scl, ctx.args
    port = O.Port()
    rep = {"workload": "scl_hip_ew: add, multiplyEntryWise, inverse over whole vectors resident in HBM",
           "bytes_per_element": "3E for add / mul, 2E for inv (E = element bytes)", "fields": {}}
    win = min(2048, args.ew_elements) if args.ew_elements else 2048
    for fkey, N_ in EW_FIELDS:
        N_ = args.ew_elements or N_
        f_, L_ = ctx.tag_limbs(fkey)
        E_ = 8 * L_
        a, b, out = scl.empty(f_, N_), scl.empty(f_, N_), scl.empty(f_, N_)
        ctx.fill_random(a, f_, b"scl-bench-ew-a-" + fkey.encode())
        ctx.fill_random(b, f_, b"scl-bench-ew-b-" + fkey.encode())
        ha = np.concatenate([scl.to_host(a[:win]), scl.to_host(a[-win:])])
        hb = np.concatenate([scl.to_host(b[:win]), scl.to_host(b[-win:])])
        # (every leg follows a host-side check of the previous one -- an idle gap: 60 launches of the 0.3 ms kernels are the
        # 20 ms of load the clocks need to come back, profiles/r4_probe_headline_seq.txt)
        warm, reps = (100, 50) if N_ * E_ < 400_000_000 else (60, 20)
        status = scl.ew_status_buffer()
        legs, ok_all = {}, True
        for name, op, nb, two in (("add", scl.ADD, 3 * E_, True), ("mul", scl.MUL, 3 * E_, True), ("inv", scl.INV, 2 * E_, False),
                                  ("inv_async", scl.INV, 2 * E_, False)):
            if name == "inv_async":
                ms = ctx.timed_launches(lambda: scl.ew_status(f_, op, a, None, status, out=out), reps, warm)
            else:
                ms = ctx.timed_launches(lambda: scl.ew(f_, op, a, b if two else None, out=out), reps, warm)
            mean = sum(ms) / reps
            got = np.concatenate([scl.to_host(out[:win]), scl.to_host(out[-win:])])
            want_op = {"add": O.ADD, "mul": O.MUL, "inv": O.INV, "inv_async": O.INV}[name]
            ok = bool(np.array_equal(got, port.ew(f_, want_op, ha, hb if two else None)))
            if name == "add":
                ok = ok and bool(scl.equals(f_, scl.ew(f_, scl.SUB, out, b), a))
            if name.startswith("inv"):
                prod = scl.ew(f_, scl.MUL, out, a)
                one = scl.to_device(np.ascontiguousarray(np.broadcast_to(port.from_int(f_, 1), (N_, L_))))   # FF::one(): R mod p in a Montgomery field
                ok = ok and bool(scl.equals(f_, prod, one))
                del prod, one
            if name == "inv_async":
                ok = ok and int(status.item()) == 0           # no zero among the operands: the device flag stayed clear
            rate = N_ / (mean * 1e-3)
            # what limits the kernel: the streaming ops are HBM-bound; inverses are vector-ALU work (3 + I / L
            # modular products per element by simultaneous inversion); GF(2^128) products run on per-lane window tables in LDS
            inv = name.startswith("inv")
            bound = ("hbm" if name == "add" or (name == "mul" and fkey != "gf2_128") else
                     "lds tables + vector ALU" if fkey == "gf2_128" else "vector ALU (near HBM)" if fkey == "m61" else "vector ALU")
            legs[name] = {"ms": mean, "ms_min_max": [min(ms), max(ms)], "elements_per_s": rate,
                          "bytes_per_element": nb, "GBps": nb * N_ / mean / 1e6, "frac": nb * N_ / mean / 1e6 / HBM_PEAK_GBPS,
                          "bound": bound, "verified": ok}
            if inv:
                legs[name]["roofline_compute"] = valu_roofline(fkey + "_inv", rate)
                if fkey != "m61":
                    two_level = fkey == "m127" and N_ >= 6_000_000      # (capi.hip inv_two_level_default: Mersenne127 from chains of 64)
                    legs[name]["traffic_over_algorithmic"] = {
                        "expected": 1.75 if two_level else ROLLED_INVERSE_TRAFFIC_OVER_ALGORITHMIC,
                        "measured": (SQ.get(fkey + "_inv") or {}).get("traffic_over_algorithmic"),     # profiles/sq_counters.json (PMC passes)
                        "why": "two levels: x read twice, a checkpoint written and read per block of 4, the result written = 3.5E of 2E"
                               if two_level else
                               "rolled chain: x read, prefix written, prefix + x read again, result written = 5E of 2E once the "
                               "chains outgrow L2"}
            elif name == "mul" and fkey == "gf2_128":
                legs[name]["roofline_compute"] = valu_roofline("gf2_128_mul", rate)
                legs[name]["roofline_lds"] = lds_roofline("gf2_128_mul", rate)
            ok_all = ok_all and ok
        # small batches: the synchronous call against the asynchronous form (what item "host-synchronous INV/DIV" costs)
        small = {}
        for n_small in (10_000, 100_000, 1_000_000):
            if n_small > N_:
                continue
            sa, so = a[:n_small], out[:n_small]
            ms_sync = ctx.timed_launches(lambda: scl.ew(f_, scl.INV, sa, None, out=so), 50, 20)
            ms_async = ctx.timed_launches(lambda: scl.ew_status(f_, scl.INV, sa, None, status, out=so), 50, 20)
            t0 = time.perf_counter()
            for _ in range(50):
                scl.ew(f_, scl.INV, sa, None, out=so)
            wall_sync = (time.perf_counter() - t0) / 50
            t0 = time.perf_counter()
            for _ in range(50):
                scl.ew_status(f_, scl.INV, sa, None, status, out=so)
            ctx.torch.cuda.synchronize()
            wall_async = (time.perf_counter() - t0) / 50
            small[str(n_small)] = {"sync_call_ms": sum(ms_sync) / 50, "async_call_ms": sum(ms_async) / 50,
                                   "sync_wall_us_per_call": 1e6 * wall_sync, "async_wall_us_per_call": 1e6 * wall_async}
        # the reference's own element-wise path on one host core beside it (oracle/_ref: Vector::add / multiplyEntryWise and
        # FF::inverse element by element; the oracle port for GF(2^128), which the reference does not have), 10^6 elements
        cpu = None
        try:
            n_cpu = min(N_, 1_000_000 if fkey in ("m61", "m127") else 20_000)   # (the port's Fermat / bit-serial inverses are slow)
            ca, cb = scl.to_host(a[:n_cpu]), scl.to_host(b[:n_cpu])
            lib_, kind_ = port, "port"
            if fkey in ("m61", "m127", "mont128"):       # (the reference has no GF(2^128): the port)
                try:
                    lib_, kind_ = O.Ref(), "reference"
                    lib_.ew(f_, O.ADD, ca[:4], cb[:4])
                except Exception:
                    lib_, kind_ = port, "port"
            cpu = {"kind": kind_, "cores": 1, "elements": n_cpu}
            for name, op, two in (("add", O.ADD, True), ("mul", O.MUL, True), ("inv", O.INV, False)):
                t0_ = time.perf_counter()
                lib_.ew(f_, op, ca, cb if two else None)
                cpu[name + "_ns_per_element"] = (time.perf_counter() - t0_) * 1e9 / n_cpu
        except Exception as e:
            cpu = {"error": str(e)}
        rep["fields"][FIELD_NAMES[fkey]] = {"elements": N_, "dtype": DTYPES[L_], "warmup": warm, "launches": reps,
                                            **legs, "inverse_small_batches": small, "cpu_reference": cpu, "verified": ok_all}
        del a, b, out
        ctx.free()
    rep["verified"] = all(v["verified"] for v in rep["fields"].values())
    return rep


def layout_report(ctx):
    """The bridge every reference-layout caller crosses: AoS [secret][party] (the Vector per secret shamirSecretShare returns,
    shamir.h:52-68) <-> SoA [party][secret] (what the kernels stream), scl_hip_aos_to_soa / scl_hip_soa_to_aos at n = 10.
    Algorithmic bytes: every element read once and written once, 2 n E per secret."""
    scl, torch, args = ctx.scl, ctx.torch, ctx.args
    rep = {"workload": "scl_hip_aos_to_soa / scl_hip_soa_to_aos, n = 10 parties", "bytes_per_secret": "2 n E", "fields": {}}
    for fkey, N_ in (("m61", 100_000_000), ("m127", 10_000_000), ("secp256k1", 10_000_000)):
        N_ = args.ew_elements or N_
        f_, L_ = ctx.tag_limbs(fkey)
        n_, E_ = 10, 8 * L_
        soa_ = scl.empty(f_, n_, N_)
        ctx.fill_random(soa_, f_, b"scl-bench-layout-" + fkey.encode())
        warm, reps = (100, 50) if N_ * n_ * E_ < 2_000_000_000 else (5, 10)
        legs = {}
        aos_ = scl.soa_to_aos(f_, soa_)
        back = scl.aos_to_soa(f_, aos_)
        ok = bool(scl.equals(f_, back.view(-1, L_), soa_.view(-1, L_)))
        # (AoS order checked against the definition on a window: aos[s][i] = soa[i][s])
        w_ = min(4096, N_)
        ok = ok and bool(torch.equal(aos_[:w_].transpose(0, 1), soa_[:, :w_])) and bool(torch.equal(aos_[-w_:].transpose(0, 1), soa_[:, -w_:]))
        del back
        calls = {"soa_to_aos": lambda: scl.lib.scl_hip_soa_to_aos(f_, scl._dev(aos_), scl._dev(soa_), N_, N_, n_, scl._stream()),
                 "aos_to_soa": lambda: scl.lib.scl_hip_aos_to_soa(f_, scl._dev(soa_), N_, scl._dev(aos_), N_, n_, scl._stream())}
        for name, fn in calls.items():
            ms = ctx.timed_launches(fn, reps, warm)
            mean = sum(ms) / reps
            nb = 2 * n_ * E_
            legs[name] = {"ms": mean, "ms_min_max": [min(ms), max(ms)], "GBps": nb * N_ / mean / 1e6,
                          "frac": nb * N_ / mean / 1e6 / HBM_PEAK_GBPS}
        rep["fields"][FIELD_NAMES[fkey]] = {"secrets": N_, "n": n_, **legs, "verified": ok}
        del soa_, aos_
        ctx.free()
    rep["verified"] = all(v["verified"] for v in rep["fields"].values())
    return rep


def matmul_report(ctx):
    """Matrix::multiply (matrix.h:477-495) beyond the sharing shapes: square Mersenne61 products on the general matrix-core kernel
    (csrc/gemm_mfma.hpp: 8 signed base-256 digits per value, 64 digit-pair int8 products per 61-bit multiply-add, all fifteen
    digit diagonals accumulated in int32 over 8192 inner columns at a time), and the reference-shaped (200 x 7000)(7000 x 300).
    Roofline: the int8 matrix peak on executed operations, 2 x 64 per multiply-add.  Checked against the CPU oracle's i-k-j loop
    on a window of rows and columns that takes in the tile edges."""
    import numpy as np
    import oracle_lib as O
    scl, args = ctx.scl, ctx.args
    port = O.Port()
    f_ = scl.M61
    rep = {"workload": "scl_hip_matmul over Mersenne61", "shapes": {}}
    for (M_, K_, N_) in ((4096, 4096, 4096), (1024, 1024, 1024), (200, 7000, 300)):
        if args.ew_elements:      # tests: a small cube that still takes the matrix cores' general kernel
            M_, K_, N_ = (160, 8300, 96) if (M_, K_, N_) == (4096, 4096, 4096) else (M_ // 8 + 1, K_ // 8 + 1, N_ // 8 + 1)
        A = scl.vector_random(f_, M_ * K_, b"scl-bench-mm-A").reshape(M_, K_, 1)
        B = scl.vector_random(f_, K_ * N_, b"scl-bench-mm-B").reshape(K_, N_, 1)
        out = scl.empty(f_, M_, N_)
        ms = sum(ctx.timed_launches(lambda: scl.matmul(f_, A, B, out=out), 10, 5)) / 10
        rows = sorted({0, 31, 32, M_ // 2, M_ - 1} & set(range(M_)))
        cols = sorted({0, 31, 32, 63, 64, N_ // 2, N_ - 1} & set(range(N_)))
        hA, hB = scl.to_host(A[rows]), scl.to_host(B[:, cols])
        want = port.matmul(f_, np.ascontiguousarray(hA), np.ascontiguousarray(hB))
        got = scl.to_host(out)[np.ix_(rows, cols)]
        macs = M_ * K_ * N_
        on_cores = K_ > 64 and M_ >= 33 and N_ >= 33 and macs >= (1 << 25)
        rep["shapes"][f"{M_}x{K_}x{N_}"] = {
            "ms": ms, "T_multiply_adds_per_s": macs / ms / 1e9,
            "path": "matrix cores, general kernel" if on_cores else "vector ALU (tiled / split-K)",
            "int8_TOPs_executed": (2 * 64 * macs / ms / 1e9) if on_cores else None,
            "frac_of_int8_peak": (2 * 64 * macs / ms / 1e9 / I8_PEAK_TOPS) if on_cores else None,
            "verified": bool(np.array_equal(got, want))}
        del A, B, out
        ctx.free()
    rep["verified"] = all(v["verified"] for v in rep["shapes"].values())
    return rep
