"""The one exchange step of the path: the MPC open (reference: Network::send + Network::recv, include/scl/net/network.h:148-152,
178-185) as RCCL all-gather / reduce-scatter + reconstruct, through torch.distributed and through the C ABI."""
from .common import FIELD_NAMES, HBM_PEAK_GBPS


def open_step(ctx, fkey, n, t, N, chunk, steps, warmup, seed):
    """The MPC open of N secrets: every rank holds ceil(n/G) parties' share vectors, one all-gather per chunk
    brings all n rows to every rank, every rank reconstructs (as every MPC party does).  Timed three ways:
    the collective alone, the reconstruct kernel alone (on gathered chunks), and the double-buffered pipeline."""
    scl, sd, torch, dist, args = ctx.scl, ctx.sd, ctx.torch, ctx.dist, ctx.args
    world, rank = ctx.world, ctx.rank
    f_, L = ctx.tag_limbs(fkey)
    E = 8 * L
    per = sd.parties_per_rank(n, world)
    first, cnt = sd.party_slab(n, rank, world)
    # this rank's slab of a real sharing: all n rows are produced chunk-wise from the same seeds on every rank
    # (identical bits everywhere) and only the rank's own rows are kept
    secrets = scl.empty(f_, N)
    ctx.fill_random(secrets, f_, seed + b"-open-secrets")
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
    t_gather = ctx.timed_region(gather_only, steps, warmup)[0] / steps
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
    t_pipe = ctx.timed_region(pipeline, steps, warmup)[0] / steps
    ok = bool(scl.equals(f_, result["out"], secrets))
    # the same open by partial sums (any field): each rank reduces its own parties, the ranks all-gather one element
    # per secret and rank, every rank adds them -- 1/parties_per_rank of the volume, every rank still learns every secret
    mine_rows = local[:cnt].contiguous() if cnt != per else local

    def partial_pipeline(k):
        result["pg"] = sd.open_by_partial_gather(f_, mine_rows, lam[first:first + cnt], chunk=chunk) if world > 1 else \
            sd.open_and_reconstruct_local(f_, local, n, lam, chunk=chunk)
    t_pg = ctx.timed_region(partial_pipeline, steps, warmup)[0] / steps
    ok_pg = bool(scl.equals(f_, result["pg"], secrets))
    # the same two opens behind the C ABI: RCCL called by the library itself (scl_hip_open_all_gather /
    # scl_hip_open_partial_gather; per-row grouped all-gathers, no packing copy), what a C++ caller of include/scl_hip/ gets
    c_abi = None
    try:
        if ctx.one_device and world > 1:   # a leg that did not run is not a leg that failed
            raise InterruptedError("skipped in the one-device rehearsal: RCCL does not take two ranks on one GPU")
        if args.inject_error == "c_abi":
            raise RuntimeError("injected by --inject-error c_abi")
        comm = sd.Communicator()
        try:
            def c_pipeline(k):
                result["c"] = sd.open_all_gather_c(comm, f_, local, n, lam, chunk=chunk)
            t_c = ctx.timed_region(c_pipeline, steps, warmup)[0] / steps

            def c_partial(k):
                result["cp"] = sd.open_partial_gather_c(comm, f_, mine_rows, lam[first:first + cnt], chunk=chunk)
            t_cp = ctx.timed_region(c_partial, steps, warmup)[0] / steps
            c_abi = {"pipeline_ms": 1e3 * t_c, "opened_secrets_per_s": N / t_c,
                     "partial_gather_pipeline_ms": 1e3 * t_cp, "partial_gather_opened_secrets_per_s": N / t_cp,
                     "verified": bool(scl.equals(f_, result["c"], secrets)) and bool(scl.equals(f_, result["cp"], secrets))}
        finally:
            comm.close()
    except InterruptedError as e:
        c_abi = {"skipped": str(e)}
    except Exception as e:   # reported in the result (and in its `errors`, with a non-zero exit code after the line is out)
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
    ctx.free()
    return res


def open_partial_sums(ctx, n, t, N, steps, warmup, seed):
    """Mersenne61 alternative: canonical partial sums + ONE reduce-scatter(SUM)"""
    scl, sd, world, rank = ctx.scl, ctx.sd, ctx.world, ctx.rank
    f_ = scl.M61
    first, cnt = sd.party_slab(n, rank, world)
    secrets = scl.empty(f_, N)
    ctx.fill_random(secrets, f_, seed + b"-ps-secrets")
    full = scl.shamir_share_prg(f_, secrets, t, n, seed + b"-ps")
    local = full[first:first + cnt].contiguous()
    del full
    lam = scl.lagrange_basis(f_, n)
    result = {}

    def run(k):
        result["mine"] = sd.open_by_partial_sums(local, lam[first:first + cnt]) if world > 1 else \
            scl.shamir_recover(f_, local, lam)
    t_ps = ctx.timed_region(run, steps, warmup)[0] / steps
    lo = rank * (N // world)
    ok = bool(scl.equals(f_, result["mine"].reshape(-1, 1), secrets[lo:lo + N // world]))
    moved = N * 8
    res = {"field": "Mersenne61", "n": n, "t": t, "secrets": N,
           "collective": "reduce_scatter_tensor(SUM, int64)" if world > 1 else "none (1 rank)",
           "ms": 1e3 * t_ps, "opened_secrets_per_s": N / t_ps, "reduce_scatter_input_bytes": moved,
           "rccl_busbw_GBps": moved / t_ps / 1e9 * (world - 1) / world, "verified": ok}
    del secrets, local, result
    ctx.free()
    return res


def c4_rank_shape(ctx, N, chunk):
    """What ONE rank of BASELINE configs[3] on 8 GPUs does per open by scl_hip_open_partial_gather: its 5 of the 40 parties'
    share vectors of all N = 10^8 GF(2^128) secrets (an 8 GB slab) go through the reconstruct kernel at m = 5 (one partial
    sum per secret), and each gathered chunk of 8 partial rows through k_additive_recover (Vector::sum per secret).  The
    kernels of the 8-GPU configuration that fit one GPU, at their real size; the all-gather between them is xGMI time
    (DESIGN.md section 5)."""
    scl = ctx.scl
    f_, L = ctx.tag_limbs("gf2_128")
    E = 8 * L
    world8, per = 8, 5
    lam = scl.lagrange_basis(f_, 40)
    local = scl.empty(f_, per, N)
    ctx.fill_random(local, f_, b"scl-bench-c4-rank")
    partial = scl.empty(f_, N)
    c = min(chunk, N)
    gathered = scl.empty(f_, world8, c)
    ctx.fill_random(gathered, f_, b"scl-bench-c4-rank-g")
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
    # linearity as the check: partial(lam) + partial(lam2) == partial(lam + lam2), all three by the kernel
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
    ctx.free()
    return res


def open_report(ctx):
    args, world = ctx.args, ctx.world
    N_open = args.open_secrets or 12_500_000 * world
    N_ps = (args.secrets // world) * world
    rep = {"c4_all_gather": open_step(ctx, "gf2_128", 40, 13, N_open, args.open_chunk, 3, 1, b"scl-bench-open"),
           "m61_partial_sums": open_partial_sums(ctx, 10, 3, N_ps, 3, 1, b"scl-bench-open")}
    if world == 1 and args.c4_rank_secrets:
        try:
            rep["c4_rank_shape"] = c4_rank_shape(ctx, args.c4_rank_secrets, args.open_chunk)
        except Exception as e:
            rep["c4_rank_shape"] = {"error": str(e), "verified": False}
            ctx.free()
    return rep
