"""world_size-2 (and 3) gloo runs on CPU of the multi-GPU plumbing: secret-axis sharding with PRG
counter origins, and the all-gather 'open' step's layout.  The HIP kernels cannot run here, so the
reconstruct step is injected: the CPU oracle acts as the checker of what the collective assembled."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(fn, world, *args):
    port = _free_port()
    mp.spawn(_entry, args=(fn, world, port) + args, nprocs=world, join=True)


def _entry(rank, fn, world, port, *args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


def _soa(aos):
    return np.ascontiguousarray(np.transpose(aos, (1, 0, 2)))


def _open_worker(rank, world, field, n, t, N, chunk):
    from scl_amd import dist as sd
    port = O.Port()
    L = O.LIMBS[field]
    secrets = port.vector_random(field, b"open-secrets", N)
    full = _soa(port.shamir_share(field, b"open-seed", secrets, t, n))  # [n][N][L], same on every rank
    per = sd.parties_per_rank(n, world)
    first, cnt = sd.party_slab(n, rank, world)
    local = np.zeros((per, N, L), dtype=np.uint64)
    local[:cnt] = full[first:first + cnt]
    lt = torch.from_numpy(local.view(np.int64))
    got = sd.open_shares(lt, n).numpy().view(np.uint64)
    assert np.array_equal(got, full)
    nodes = np.stack([port.from_int(field, i + 1) for i in range(n)])
    lam = port.lagrange_basis(field, nodes, port.from_int(field, 0))
    calls = []

    def checker(f, shares, lam_, out):  # stands in for the HIP kernel: oracle on the gathered chunk
        sh = shares.numpy().view(np.uint64)
        assert sh.shape[0] == n
        rec = port.shamir_recover_lambda(f, np.ascontiguousarray(np.transpose(sh, (1, 0, 2))), lam_)
        out.copy_(torch.from_numpy(rec.view(np.int64)))
        calls.append(sh.shape[1])

    out = sd.open_and_reconstruct(field, lt, n, lam, chunk=chunk, recover=checker)
    assert np.array_equal(out.numpy().view(np.uint64), secrets)
    assert sum(calls) == N and len(calls) == -(-N // chunk)


@pytest.mark.parametrize("world,field,n,t,N,chunk", [(2, O.M61, 10, 3, 1000, 256), (2, O.M127, 5, 2, 77, 1 << 20),
                                                     (3, O.M61, 40, 13, 300, 128)])
def test_open_all_gather_layout(world, field, n, t, N, chunk):
    _run(_open_worker, world, field, n, t, N, chunk)


def _shard_worker(rank, world, field, n, t, N):
    """each rank shares its own slice with first_secret = its origin; concatenated == the one-PRG run"""
    from scl_amd import dist as sd
    port = O.Port()
    L = O.LIMBS[field]
    secrets = port.vector_random(field, b"shard-secrets", N)
    first, cnt = sd.shard_bounds(N, rank, world)
    # the rank's slice under the reference PRG discipline: coefficients from blocks [s*B, (s+1)*B)
    B = (t + 2) // 2 if L == 1 else t + 1
    elems = port.from_bytes(field, port.prg_blocks(b"shard-seed", first * B, cnt * B)).reshape(cnt, -1, L)
    mine = port.shamir_share_coeffs(field, secrets[first:first + cnt], np.ascontiguousarray(elems[:, 1:t + 1]), n)
    pieces = [None] * world
    dist.all_gather_object(pieces, (first, cnt, mine))
    assert sorted(p[0] for p in pieces) == [sd.shard_bounds(N, r, world)[0] for r in range(world)]
    assert sum(p[1] for p in pieces) == N
    whole = np.concatenate([p[2] for p in sorted(pieces, key=lambda p: p[0])])
    assert np.array_equal(whole, port.shamir_share(field, b"shard-seed", secrets, t, n))


@pytest.mark.parametrize("world,N", [(2, 101), (3, 10)])
def test_sharded_prg_origins_reproduce_the_single_prg_run(world, N):
    _run(_shard_worker, world, O.M61, 10, 3, N)


def test_shard_bounds_cover_everything():
    from scl_amd import dist as sd
    for N in (0, 1, 7, 8, 100_000_000):
        for world in (1, 2, 4, 8):
            spans = [sd.shard_bounds(N, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == N
            for (a, c), (b, _) in zip(spans, spans[1:]):
                assert a + c == b
    assert [sd.party_slab(40, r, 8) for r in range(8)] == [(5 * r, 5) for r in range(8)]
    assert [sd.party_slab(10, r, 4) for r in range(4)] == [(0, 3), (3, 3), (6, 3), (9, 1)]


def _partial_sum_worker(rank, world, n, t, N):
    """open_by_partial_sums over gloo: the oracle computes each rank's partial sums and the final fold, the
    collective is the real reduce-scatter; every rank's slice must equal the secrets"""
    from scl_amd import dist as sd
    port = O.Port()
    f = O.M61
    secrets = port.vector_random(f, b"ps-secrets", N)
    secrets[0] = port.from_int(f, -1)                       # p - 1: the partial sums at their largest
    full = _soa(port.shamir_share(f, b"ps-seed", secrets, t, n))
    nodes = np.stack([port.from_int(f, i + 1) for i in range(n)])
    lam = port.lagrange_basis(f, nodes, port.from_int(f, 0))
    first, cnt = sd.party_slab(n, rank, world)
    local = torch.from_numpy(np.ascontiguousarray(full[first:first + cnt]).view(np.int64))

    def partial(field, shares, lam_):
        sh = shares.numpy().view(np.uint64)
        if sh.shape[0] == 0:
            return torch.zeros((sh.shape[1], 1), dtype=torch.int64)
        rec = port.shamir_recover_lambda(field, np.ascontiguousarray(np.transpose(sh, (1, 0, 2))), lam_)
        assert int(rec.max()) < (1 << 61) - 1                # canonical, so that 8 of them cannot wrap 64 bits
        return torch.from_numpy(rec.view(np.int64))

    def fold(field, words):
        return torch.from_numpy(port.from_bytes(field, words.numpy().tobytes()).view(np.int64))

    mine = sd.open_by_partial_sums(local, lam[first:first + cnt], partial=partial, fold=fold)
    lo = rank * (N // world)
    assert np.array_equal(mine.numpy().view(np.uint64), secrets[lo:lo + N // world])


@pytest.mark.parametrize("world,n,t,N", [(2, 10, 3, 64), (3, 7, 2, 33), (2, 3, 1, 8), (3, 2, 1, 9)])  # last: a rank with no party
def test_open_by_partial_sums(world, n, t, N):
    _run(_partial_sum_worker, world, n, t, N)


def test_open_on_one_rank_is_the_chunked_reconstruct():
    """world size 1 (what bench.py times on a single GPU): no collective, same chunking and output"""
    from scl_amd import dist as sd
    port = O.Port()
    f, n, t, N, chunk = O.M61, 5, 2, 77, 20
    secrets = port.vector_random(f, b"one-rank", N)
    full = _soa(port.shamir_share(f, b"one-rank-seed", secrets, t, n))
    nodes = np.stack([port.from_int(f, i + 1) for i in range(n)])
    lam = port.lagrange_basis(f, nodes, port.from_int(f, 0))
    calls = []

    def checker(field, shares, lam_, out):
        sh = shares.numpy().view(np.uint64)
        rec = port.shamir_recover_lambda(field, np.ascontiguousarray(np.transpose(sh, (1, 0, 2))), lam_)
        out.copy_(torch.from_numpy(rec.view(np.int64)))
        calls.append(sh.shape[1])

    out = sd.open_and_reconstruct_local(f, torch.from_numpy(full.view(np.int64)), n, lam, chunk=chunk, recover=checker)
    assert np.array_equal(out.numpy().view(np.uint64), secrets) and calls == [20, 20, 20, 17]


def _partial_gather_worker(rank, world, field, n, t, N, chunk):
    """open_by_partial_gather over gloo, any field: the oracle computes each rank's partial sums and adds the gathered
    partials, the collective is the real all-gather; EVERY rank must end with every secret, and with exactly what
    the all-gather open gives"""
    from scl_amd import dist as sd
    port = O.Port()
    L = O.LIMBS[field]
    secrets = port.vector_random(field, b"pg-secrets", N)
    secrets[0] = port.from_int(field, -1)
    if field == O.GF2_128:      # the oracle's sharing walks x++ (meaningless in characteristic 2): explicit nodes 1..n as bit patterns
        nodes = O.from_ints(list(range(1, n + 1)), L)
        coeffs = port.vector_random(field, b"pg-coeffs", max(t, 1) * N).reshape(N, max(t, 1), L)[:, :t]
        full = _soa(np.stack([port.poly_eval(field, np.concatenate([secrets[s:s + 1], coeffs[s]]), nodes) for s in range(N)]))
    else:
        nodes = np.stack([port.from_int(field, i + 1) for i in range(n)])
        full = _soa(port.shamir_share(field, b"pg-seed", secrets, t, n))
    lam = port.lagrange_basis(field, nodes, port.from_int(field, 0))
    first, cnt = sd.party_slab(n, rank, world)
    local = torch.from_numpy(np.ascontiguousarray(full[first:first + cnt]).view(np.int64))
    calls = []

    def partial(f, shares, lam_, out):
        sh = np.ascontiguousarray(shares.numpy()).view(np.uint64)
        rec = port.shamir_recover_lambda(f, np.ascontiguousarray(np.transpose(sh, (1, 0, 2))), lam_)
        out.copy_(torch.from_numpy(rec.view(np.int64)))
        calls.append(sh.shape[1])

    def total(f, rows, out):
        r = rows.numpy().view(np.uint64)
        assert r.shape[0] == world
        acc = r[0].copy()
        for k in range(1, world):
            acc = port.ew(f, O.ADD, acc, r[k])
        out.copy_(torch.from_numpy(acc.view(np.int64)))

    out = sd.open_by_partial_gather(field, local, lam[first:first + cnt], chunk=chunk, partial=partial, total=total)
    assert np.array_equal(out.numpy().view(np.uint64), secrets)
    assert (sum(calls) == N and len(calls) == -(-N // chunk)) if cnt else not calls


@pytest.mark.parametrize("world,field,n,t,N,chunk", [(2, O.GF2_128, 40, 13, 50, 16), (3, O.M127, 7, 2, 33, 1 << 20),
                                                     (2, O.M61, 10, 3, 200, 64), (3, O.MONT128, 2, 1, 9, 4),   # a rank with no party
                                                     (2, O.SECP256K1_SCALAR, 5, 2, 12, 5)])
def test_open_by_partial_gather(world, field, n, t, N, chunk):
    _run(_partial_gather_worker, world, field, n, t, N, chunk)


@pytest.mark.parametrize("world,n", [(8, 40), (4, 10), (2, 7), (1, 10), (8, 3)])
def test_c_abi_open_row_order_matches_the_grouped_gather_layout(world, n):
    """scl_hip_open_all_gather moves a chunk by one all-gather PER PARTY ROW of the slab (no packing copy), so row
    j * world + r of the gathered buffer holds party r * per + j, and the reconstruct kernel is handed lambda in that order
    (zero for padding rows).  Host-only entry point: the layout is rebuilt here in numpy and reconstructed by the oracle."""
    import scl_amd  # noqa: F401  (loads the library; no GPU call below)
    from scl_amd import dist as sd
    port = O.Port()
    f, t, N = O.M61, min(3, n - 1), 50
    per = sd.parties_per_rank(n, world)
    order = sd.open_row_order(n, world)
    assert len(order) == per * world and sorted(p for p in order if p >= 0) == list(range(n))
    secrets = port.vector_random(f, b"ro-secrets", N)
    full = _soa(port.shamir_share(f, b"ro-seed", secrets, t, n))            # [n][N][1]
    slabs = []
    for r in range(world):
        first, cnt = sd.party_slab(n, r, world)
        slab = np.full((per, N, 1), 0xDEADBEEF, dtype=np.uint64)            # padding rows hold junk
        slab[:cnt] = full[first:first + cnt]
        slabs.append(slab)
    # what `per` grouped all-gathers leave in the buffer: for each row j, the ranks' rows j one after the other
    buf = np.stack([slabs[r][j] for j in range(per) for r in range(world)])
    nodes = np.stack([port.from_int(f, i + 1) for i in range(n)])
    lam = port.lagrange_basis(f, nodes, port.from_int(f, 0))
    lam_rows = np.stack([lam[p] if p >= 0 else np.zeros_like(lam[0]) for p in order])
    buf = np.where(np.array(order)[:, None, None] >= 0, buf, 0)             # lambda = 0 kills the padding rows
    rec = port.shamir_recover_lambda(f, np.ascontiguousarray(np.transpose(buf, (1, 0, 2))), lam_rows)
    assert np.array_equal(rec, secrets)
