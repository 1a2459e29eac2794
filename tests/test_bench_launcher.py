"""bench.py's multi-rank plumbing on CPU: `--gpus N` started plainly must start N ranks itself (a child
`python -m torch.distributed.run`, before torch or the GPU is touched) and print ONE line with n_gpus = N; a world
size that disagrees with --gpus must fail.  `--backend gloo --dry-run` keeps the rendezvous, the barriers and the
max-over-ranks timing and skips the GPU work (cf. the reference's exchange step, include/scl/net/network.h:148-185,
which the real run times over RCCL)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_plain_start_with_gpus_2_launches_two_ranks():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert len(lines[0]) < 8192
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["config"]["parallelism"] == "shard2"
    assert line["metric"] == "shamir_reconstructions_per_sec" and line["scaling"] == "weak"
    # what the process group itself reports, so that a scaling record can check that N ranks really ran: the collective
    # library's own world size and backend, each rank's device and each rank's time per step (the line's is their maximum)
    assert line["rccl"]["ranks"] == 2 and line["rccl"]["backend"] == "gloo" and len(line["rccl"]["devices"]) == 2
    assert len(line["ms_per_step_by_rank"]) == 2 and abs(max(line["ms_per_step_by_rank"]) - line["ms_per_step"]) < 1e-9


def test_single_rank_dry_run_prints_one_line():
    r = _run(["--dry-run", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("\n") == 1 and len(r.stdout) < 8192      # ONE line, short enough for the driver's record
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1
    assert line["rccl"] == {"ranks": 1, "backend": None, "devices": [0]} and len(line["ms_per_step_by_rank"]) == 1
    # the driver's own command, dry: every key of the contract is there whatever the flags
    r = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in line, k
    assert line["steps"] == 20 and line["warmup"] == 5 and "workload" in line["config"] and "model" not in line["config"]


def test_world_size_that_disagrees_with_gpus_fails():
    # what a mis-launched rank sees: WORLD_SIZE says 2, the command line says 4
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)
    r = _run(["--gpus", "1", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0


def test_pmc_traffic_is_stamped_and_only_quoted_for_its_own_kernels():
    """profiles/pmc_traffic.json carries the configuration, the kernel symbols and a hash of the kernel sources it was measured
    with; bench.py reports `roofline.traffic` only for that configuration and while the hash still matches -- a kernel edit
    makes the number null instead of silently stale.  The committed file must describe the committed sources."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
        pmc = json.load(fh)
    assert pmc["kernel_source_sha256_16"] == bench.kernel_source_hash(), "regenerate profiles/pmc_traffic.json (tools/make_pmc_traffic.py)"
    assert "k_share_small" in pmc["shamir_share"]["kernel"] and "k_recover_fixed" in pmc["shamir_recover"]["kernel"]
    args = bench.parse_args([])
    share, rec = bench.pmc_traffic("shamir_share", args), bench.pmc_traffic("shamir_recover", args)
    # algorithmic bytes of BASELINE configs[1]: 112 B and 88 B per secret; measured traffic within 0.5 % of them
    assert abs(share / (112 * args.secrets) - 1) < 5e-3 and abs(rec / (88 * args.secrets) - 1) < 5e-3
    assert bench.pmc_traffic("shamir_share", bench.parse_args(["--secrets", "1000"])) is None
    assert bench.pmc_traffic("shamir_share", bench.parse_args(["--t", "4"])) is None
    # the side configurations of the same PMC passes: every BASELINE configuration's two kernels, traffic = algorithmic bytes
    want = {"C3_mersenne127_10_3": ("k_share_small_t<sclhip::M127", "k_recover_fixed<sclhip::M127"),
            "C3_mont128_10_3": ("k_share_small_t<sclhip::Mont128", "k_recover_small<sclhip::Mont128"),
            "F3_secp256k1_scalar_10_3": ("k_share_small_pair<sclhip::Mont256", "k_recover_small<sclhip::Mont256"),
            "C4_shard_gf2_128_40_13": ("k_share_gf_tiles<13>", "k_recover_gf128_pos"),
            "C5_shard_mersenne61_128_42": ("k_share_mfma_m61_p16", "k_recover_table<sclhip::M61")}
    for key, (sk, rk) in want.items():
        tr = bench.pmc_config_traffic(key)
        assert tr is not None, key
        assert sk in tr["share_kernel"] and rk in tr["recover_kernel"], (key, tr)
        c = pmc["configs"][key]
        assert abs(c["share"]["traffic_over_algorithmic"] - 1) < 1e-2 and abs(c["recover"]["traffic_over_algorithmic"] - 1) < 1e-2, (key, c)
    import bench_legs.pmc as pmc_mod
    real = pmc_mod.kernel_source_hash
    try:
        pmc_mod.kernel_source_hash = lambda: "0" * 16      # what an edited kernel source looks like
        assert bench.pmc_traffic("shamir_share", args) is None
        assert bench.pmc_config_traffic("C4_shard_gf2_128_40_13") is None
    finally:
        pmc_mod.kernel_source_hash = real


def test_configs_quoted_on_eight_gpus_dry_run():
    """`--gpus 8 --config c4 / c5`: BASELINE configs[3] and [4] as the multi-rank headline -- strong scaling, the totals
    split over the ranks (c5: 8 x 1.25e8 = 10^9 shards adding up; c4: every rank opens all 10^8 secrets from its 5 parties'
    slabs).  gloo + --dry-run checks the launch, the plan and the strings of the result line; the arithmetic needs the GPUs."""
    r = _run(["--gpus", "8", "--config", "c4", "--backend", "gloo", "--dry-run", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["dtype"] == "u128"
    c = line["config"]
    assert c["total_secrets"] == 100_000_000 and c["n"] == 40 and c["t"] == 13 and c["parallelism"] == "parties8"
    assert "GF(2^128)" in c["workload"] and "5 parties per rank" in c["workload"] and "BASELINE configs[3]" in c["workload"]
    r = _run(["--gpus", "8", "--config", "c5", "--backend", "gloo", "--dry-run", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = line["config"]
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["dtype"] == "u64"
    assert c["total_secrets"] == 1_000_000_000 == c["secrets_over_ranks"] and c["n"] == 128 and c["t"] == 42
    assert "BASELINE configs[4]" in c["workload"] and c["parallelism"] == "shard8"


def test_c5_on_fewer_gpus_is_capacity_bound_and_says_so():
    sys.path.insert(0, ROOT)
    import bench
    pl = bench.plan(bench.parse_args(["--config", "c5", "--gpus", "2"]), 2, 1)
    assert pl["total"] == 250_000_000 and pl["mine"] == 125_000_000 and pl["scaling"] == "weak"
    assert "needs 8 GPUs" in pl["workload"]
    # ragged totals: the first total % world ranks take one more
    a = bench.parse_args(["--config", "c5", "--gpus", "8", "--total-secrets", "1000003"])
    assert sum(bench.plan(a, 8, r)["mine"] for r in range(8)) == 1000003
    assert bench.plan(bench.parse_args([]), 4, 0)["total"] == 400_000_000


def test_matrix_core_share_is_priced_against_the_matrix_pipe():
    """C5's share kernel runs on the matrix cores: bench.py quotes it against the int8 matrix peak (its HBM-equivalent rate
    stays beside it), only for the shapes capi.hip sends there, with the formulation's own operation count."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.on_matrix_cores("m61", 128, 42) and bench.on_matrix_cores("m61", 97, 32)
    assert not bench.on_matrix_cores("m61", 10, 3) and not bench.on_matrix_cores("m61", 96, 42)
    assert not bench.on_matrix_cores("m127", 128, 42) and not bench.on_matrix_cores("m61", 128, 64)
    r = bench.mfma_share_roofline(128, 42, 1_000_000, 1.0)
    assert r["bound"] == "mfma" and r["peak"] == bench.I8_PEAK_TOPS
    assert r["algorithmic_ops"] == 2 * 64 * 43 * 128 * 1_000_000           # 64 digit pairs, 43 coefficients, 128 parties
    assert abs(r["achieved"] - r["algorithmic_ops"] / 1e-3 / 1e12) < 1e-9 and abs(r["frac"] - r["achieved"] / 5000.0) < 1e-12
    assert abs(r["executed_frac"] / r["frac"] - 64 / 43) < 1e-9             # K padded from 43 to 64 slots


def test_side_legs_and_pmc_parsing_are_pure_functions():
    """bench.side_legs: `verified` of the line = AND over the headline and every leg, an {"error": ..} object anywhere is listed
    and counts as unverified, a {"skipped": ..} leg is neither; bench.pmc_means / pmc_report: the parsing of rocprofv3's
    counter CSVs that both the live passes and tools/make_pmc_traffic.py use (KiB counters, FETCH_SIZE doubled, the launches of
    the largest size only)."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.side_legs is __import__("bench_legs.common", fromlist=["side_legs"]).side_legs
    line = {"verified": True, "roofline": {"frac": 0.8}, "cpu_baseline": {"all_cores": {"error": "x"}},
            "configs": {"A": {"verified": True, "share_roofline": {"frac": 0.3}}, "B": {"error": "boom", "verified": False}},
            "open": {"c4": {"verified": True, "c_abi": {"skipped": "why"}, "partial_gather": {"verified": False}}}}
    legs, errors = bench.side_legs(line)
    assert legs == {"configs.A": True, "configs.B": False, "open.c4": True, "open.c4.partial_gather": False}
    assert errors == ["configs.B: boom"]
    assert bench.cpu_model() and isinstance(bench.cpu_model(), str)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for counter, rows in (("FETCH_SIZE", [("void sclhip::k_share_small_t<sclhip::M61, 2, 3, 64>(a)", 1562500.0),
                                              ("void sclhip::k_share_small_t<sclhip::M61, 2, 3, 64>(a)", 10.0),      # a tiny launch
                                              ("void sclhip::k_recover_fixed<sclhip::M61, 2, 10, true, 64>(b)", 3906250.0),
                                              ("void sclhip::k_copy16<0>(c)", 1953125.0)]),
                              ("WRITE_SIZE", [("void sclhip::k_share_small_t<sclhip::M61, 2, 3, 64>(a)", 7812500.0),
                                              ("void sclhip::k_recover_fixed<sclhip::M61, 2, 10, true, 64>(b)", 781250.0),
                                              ("void sclhip::k_copy16<0>(c)", 3906250.0)])):
            os.makedirs(os.path.join(d, counter, "host"))
            with open(os.path.join(d, counter, "host", "1_counter_collection.csv"), "w") as fh:
                fh.write("Kernel_Name,Counter_Name,Counter_Value\n")
                for k, v in rows:
                    fh.write(f'"{k}",{counter},{v}\n')
        f, w = bench.pmc_means(os.path.join(d, "FETCH_SIZE"), "FETCH_SIZE"), bench.pmc_means(os.path.join(d, "WRITE_SIZE"), "WRITE_SIZE")
        rep = bench.pmc_report(f, w, 4e9)
    assert rep["shamir_share"]["bytes"] == 112 * 10**8 and rep["shamir_share"]["launches"] == 1
    assert rep["shamir_recover"]["bytes"] == 88 * 10**8
    assert abs(rep["calibration_k_copy16"]["fetch_correction"] - 2.0) < 1e-9 and rep["configs"] == {}


def test_the_line_is_cut_to_size_and_the_detail_keeps_everything():
    """bench.finish_line: `verified` = AND over the headline and every leg of the detail object, errors listed, floats to six
    figures, and a line that would pass the limit sheds its digest and then its per-leg map -- it never grows past what the
    driver's record takes (BENCH_r05: a 20 KB line, parsed = null)."""
    sys.path.insert(0, ROOT)
    import bench
    line = {"metric": "shamir_reconstructions_per_sec", "value": 3.123456789e10, "verified_headline": True,
            "roofline": {"frac": 0.8123456789}}
    detail = {"configs": {"A": {"verified": True, "share_frac": 0.5, "recover_frac": 0.4},
                          "B": {"error": "boom " * 100, "verified": False}},
              "open": {"c4_all_gather": {"verified": True, "opened_secrets_per_s": 1.0, "rccl_busbw_GBps": 0.0,
                                         "reconstruct_hbm_frac": 0.4, "c_abi": {"skipped": "why"}}}}
    out, legs, errors = bench.finish_line(dict(line), detail)
    assert out["verified"] is False and out["verified_headline"] is True
    assert legs == {"configs.A": True, "configs.B": False, "open.c4_all_gather": True} and len(errors) == 1
    assert len(out["errors"][0]) <= 240 and out["value"] == 3.12346e10 and out["roofline"]["frac"] == 0.812346
    assert out["legs"]["A"] == {"share_frac": 0.5, "recover_frac": 0.4} and "B" not in out["legs"]
    assert len(json.dumps(out)) < bench.LINE_LIMIT
    many = {"configs": {f"leg{i:04d}": {"verified": True, "share_frac": 0.5, "recover_frac": 0.4} for i in range(400)}}
    out, legs, _ = bench.finish_line(dict(line), many)
    assert len(json.dumps(out)) < bench.LINE_LIMIT and "legs" not in out
    assert out["verified_legs"] == {"count": 400, "failed": []} and out["verified"] is True


def test_first_contact_kit_dry_run():
    """tools/first_contact_8gpu.sh -- the one script for the first 8-GPU node (the open step over real RCCL on 2 / 4 / 8 ranks,
    the headline at 1 / 2 / 4 / 8, --config c4 / c5 at 8) -- with FIRST_CONTACT_DRY_RUN=1: every bench command of it through
    gloo + --dry-run, its checks of each line (rccl.ranks == N, one time per rank, line < 8 KB) applied.  Its N = 1 line is the
    driver's BENCH command's workload."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, FIRST_CONTACT_DRY_RUN="1")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run(["bash", os.path.join(ROOT, "tools", "first_contact_8gpu.sh"), d], capture_output=True, text=True,
                           env=env, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        assert "first contact complete" in r.stdout
        one = json.loads(open(os.path.join(d, "scale_1.json")).read().strip())
        bench_cmd = json.loads(_run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--dry-run"]).stdout.strip())
        assert one["config"] == bench_cmd["config"] and one["metric"] == bench_cmd["metric"] and one["steps"] == 20
        for n in (2, 4, 8):
            ln = json.loads(open(os.path.join(d, f"scale_{n}.json")).read().strip())
            assert ln["n_gpus"] == n and ln["config"]["total_secrets"] == n * 100_000_000 and ln["scaling"] == "weak"
        c4 = json.loads(open(os.path.join(d, "c4_8.json")).read().strip())
        assert c4["config"]["parallelism"] == "parties8" and "BASELINE configs[3]" in c4["config"]["workload"]
