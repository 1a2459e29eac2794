"""What every leg shares: the hardware peaks the fractions are quoted against, the run context, the matrix-pipe roofline of
the (128,42) share kernel, and the walk over a result object that decides `verified`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "secure-computation-library_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FIELD_TAGS = {"m61": 0, "m127": 1, "mont128": 2, "gf2_128": 3, "secp256k1": 4, "secp256k1_field": 5}
FIELD_NAMES = {"m61": "Mersenne61", "m127": "Mersenne127", "mont128": "Mont128", "gf2_128": "GF(2^128)",
               "secp256k1": "secp256k1_order", "secp256k1_field": "secp256k1_field"}
DTYPES = {1: "u64", 2: "u128", 4: "u256"}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable)
I8_PEAK_TOPS = 5000.0   # MI355X_MICROARCH.md, matrix cores: I8 runs at 2x the BF16 rate per clock; BF16 dense ~2.5 PFLOP/s
C5_PER_GPU_CAP = 125_000_000   # (128,42) Mersenne61: 128 GB of shares + 42 GB of coefficients + 2 GB per GPU (of 288 GB)


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


def side_legs(line):
    """({leg path: verified}, [error strings]) over every object of a result that carries a `verified` or an `error`
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


def sig(x, digits=6):
    """floats of the compact line to `digits` significant figures (the detail file keeps them whole)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sig(v, digits) for v in x]
    return x


class Ctx:
    """One rank's view of the run: arguments, world, the torch / scl modules once they are imported, and the collective
    helpers of the contract (barrier + synchronize on both sides of a timed region, max over ranks)."""

    def __init__(self, args, world, rank, local_rank):
        self.args, self.world, self.rank, self.local_rank = args, world, rank, local_rank
        self.dry = args.dry_run
        self.one_device = False
        self.torch = self.dist = self.scl = self.sd = None

    # ---- set-up ---------------------------------------------------------------------------------------------------
    def init_torch(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        args, world = self.args, self.world
        if self.dry:
            if world > 1:
                dist.init_process_group("gloo")
            return
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        # SCL_BENCH_ONE_DEVICE=1: a rehearsal of the multi-rank logic on a one-GPU box (every rank on device 0, gloo
        # collectives); never a measurement
        self.one_device = os.environ.get("SCL_BENCH_ONE_DEVICE") == "1" and args.backend == "gloo"
        dev_index = 0 if self.one_device else self.local_rank
        torch.cuda.set_device(dev_index)
        if world > 1:
            dist.init_process_group(args.backend, device_id=torch.device("cuda", dev_index) if args.backend == "nccl" else None)

    def init_scl(self):
        import scl_amd as scl
        from scl_amd import dist as sd
        self.scl, self.sd = scl, sd

    def finish(self):
        if self.world > 1:
            self.dist.destroy_process_group()

    # ---- the contract's timing skeleton -----------------------------------------------------------------------------
    def sync(self):
        if not self.dry:
            self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            if not self.dry:
                self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        tt = self.torch.tensor([x], dtype=self.torch.float64,
                               device="cpu" if self.dry or self.args.backend == "gloo" else "cuda")
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def all_ranks(self, x):
        """every rank's value, in rank order (what the driver needs to see that N ranks really ran)"""
        if self.world == 1:
            return [x]
        xs = [None] * self.world
        self.dist.all_gather_object(xs, float(x))
        return xs

    def timed_region(self, step, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks"""
        for _ in range(warmup):
            step(None)
        self.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        self.sync()
        mine_s = time.perf_counter() - t0
        return self.max_over_ranks(mine_s), self.all_ranks(mine_s)

    def rccl_report(self):
        """what the process group itself says about the job: a SCALE record can check that the collective library saw N ranks"""
        torch, dist = self.torch, self.dist
        if self.world == 1:
            return {"ranks": 1, "backend": None, "devices": [0] if self.dry else [torch.cuda.current_device()]}
        devs = [None] * self.world
        dist.all_gather_object(devs, -1 if self.dry else int(torch.cuda.current_device()))
        rep = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "devices": devs}
        if not self.dry and self.args.backend == "nccl":
            # one all-reduce of ones through the communicator the timed collectives use: RCCL itself counts the ranks
            ones = torch.ones(1, dtype=torch.int64, device="cuda")
            dist.all_reduce(ones)
            rep["allreduce_of_ones"] = int(ones.item())
        return rep

    # ---- device helpers ---------------------------------------------------------------------------------------------
    def tag_limbs(self, fkey):
        f_ = FIELD_TAGS[fkey]
        return f_, self.scl.limbs(f_)

    def fill_random(self, dst, f_, seed, counter0=0):
        """uniform field elements from the device AES-CTR PRG straight into dst ([rows][N][L] or [N][L])"""
        rows = dst if dst.dim() == 3 else dst.unsqueeze(0)
        N_ = rows.shape[1]
        per_row = (N_ * 8 * rows.shape[2] + 15) // 16
        for k in range(rows.shape[0]):
            self.scl.vector_random(f_, N_, seed, counter0=counter0 + k * per_row, out=rows[k])

    def timed_launches(self, fn, reps, warm):
        """`warm` untimed calls of fn, then `reps` calls with HIP events around each (on the launch stream): [ms]"""
        tms = [self.scl.Timer() for _ in range(reps)]
        for k in range(-warm, reps):
            if k >= 0:
                tms[k].start()
            fn()
            if k >= 0:
                tms[k].stop()
        self.torch.cuda.synchronize()
        return [tm.elapsed_ms() for tm in tms]

    def free(self):
        self.torch.cuda.empty_cache()
