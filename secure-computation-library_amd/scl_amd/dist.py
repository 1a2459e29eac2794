"""Multi-GPU plumbing: one process per GPU over torch.distributed (backend "nccl" = RCCL on xGMI).

Two things only:

* sharding -- secrets are independent, so a batch splits along the secret axis with no data-path
  collective; `shard_bounds` also gives the PRG counter origin (`first_secret`) that keeps every
  rank bit-identical to the single-PRG reference run (SURVEY.md section 8a note P).
* the "open" step -- every party sends its share vector to every party, then reconstructs
  (reference: Network::send + Network::recv, include/scl/net/network.h:148-185, exercised by
  test/scl/protocol/beaver.h:43-55).  Parties map onto ranks; one all-gather of each rank's
  `[parties_per_rank][chunk]` slab per chunk, overlapped with the reconstruct kernel of the
  previous chunk on the compute stream.

The collective/layout code is device-agnostic (it runs on CPU tensors over gloo in the tests);
the arithmetic is always the HIP kernels.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(N: int, rank: int, world: int) -> tuple[int, int]:
    """contiguous split of [0, N): (first_secret, count); the first N % world ranks get one extra"""
    base, rem = divmod(N, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def party_slab(n: int, rank: int, world: int) -> tuple[int, int]:
    """parties owned by `rank` when n parties are dealt in contiguous blocks of ceil(n/world)"""
    per = -(-n // world)
    first = min(rank * per, n)
    return first, max(0, min(per, n - first))


def parties_per_rank(n: int, world: int) -> int:
    return -(-n // world)


def open_shares(local: torch.Tensor, n: int, group=None, out: torch.Tensor | None = None) -> torch.Tensor:
    """All-gather of party slabs.  local: [parties_per_rank(n)][N][L] (rows past this rank's party
    count are padding).  Returns [n][N][L] with party p = rank*per + j at row p."""
    world = dist.get_world_size(group)
    per = parties_per_rank(n, world)
    if local.shape[0] != per:
        raise ValueError(f"local slab must have {per} party rows (pad the last rank)")
    gathered = out if out is not None else torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype,
                                                        device=local.device)
    dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)
    return gathered[:n]


def open_and_reconstruct(field: int, local: torch.Tensor, n: int, lam, chunk: int = 1 << 24, group=None,
                         recover=None) -> torch.Tensor:
    """The MPC 'open': gather all n parties' shares of N secrets and reconstruct on every rank.
    Chunked over the secret axis; the all-gather of chunk k+1 (RCCL, its own stream) overlaps the
    reconstruct kernel of chunk k.  `recover(field, shares[n][c][L], lam, out)` defaults to the HIP
    kernel; tests inject a checker to exercise the collective on CPU."""
    if recover is None:
        from . import shamir_recover as recover  # the HIP path; raises without a GPU
    world = dist.get_world_size(group)
    per = parties_per_rank(n, world)
    N, L = local.shape[1], local.shape[2]
    out = torch.empty((N, L), dtype=local.dtype, device=local.device)
    bufs = [torch.empty((world * per, min(chunk, N), L), dtype=local.dtype, device=local.device) for _ in range(2)]
    pending = None
    starts = list(range(0, N, chunk))
    for k, s0 in enumerate(starts):
        c = min(chunk, N - s0)
        buf = bufs[k % 2][:, :c] if c == bufs[k % 2].shape[1] else torch.empty((world * per, c, L), dtype=local.dtype,
                                                                                device=local.device)
        work = dist.all_gather_into_tensor(buf, local[:, s0:s0 + c].contiguous(), group=group, async_op=True)
        if pending is not None:
            pwork, pbuf, ps0, pc = pending
            pwork.wait()
            recover(field, pbuf[:n], lam, out[ps0:ps0 + pc])
        pending = (work, buf, s0, c)
    if pending is not None:
        pwork, pbuf, ps0, pc = pending
        pwork.wait()
        recover(field, pbuf[:n], lam, out[ps0:ps0 + pc])
    return out


def open_and_reconstruct_local(field: int, local: torch.Tensor, n: int, lam, chunk: int = 1 << 24, recover=None) -> torch.Tensor:
    """World size 1: the one rank holds all n parties, so the open step is the chunked reconstruct alone (same chunking
    and output as open_and_reconstruct, no collective)."""
    if recover is None:
        from . import shamir_recover as recover
    N, L = local.shape[1], local.shape[2]
    out = torch.empty((N, L), dtype=local.dtype, device=local.device)
    for s0 in range(0, N, chunk):
        c = min(chunk, N - s0)
        recover(field, local[:n, s0:s0 + c], lam, out[s0:s0 + c])
    return out


def open_by_partial_sums(local: torch.Tensor, lam_local, group=None, partial=None, fold=None) -> torch.Tensor:
    """Mersenne61 only (the alternative to the all-gather): every rank reduces ITS parties
    to the canonical partial sum sum_j lambda_j * share_j (< p = 2^61 - 1), one reduce-scatter(SUM) over <= 8
    ranks adds the partials as plain 64-bit integers -- 8 (2^61 - 2) < 2^64 cannot wrap -- and each rank folds its
    1/G slice of the secrets modulo p once.  Moves 1/n of the all-gather's volume.  Returns this rank's slice
    [N / world][1]; N must divide by the world size.

    local: [parties of this rank][N][1]; lam_local: their Lagrange coefficients.  `partial(field, shares, lam)` and
    `fold(field, words)` default to the HIP kernels (shamir_recover, from_bytes = FF::read's "% p"); the gloo tests
    inject CPU checkers."""
    from . import M61
    world = dist.get_world_size(group)
    if world > 8:
        raise ValueError("the unreduced 64-bit sum is only safe for at most 8 ranks")
    N = local.shape[1]
    if N % world:
        raise ValueError("N must be a multiple of the world size")
    if partial is None:
        from . import shamir_recover as partial
    if fold is None:
        from . import from_bytes

        def fold(field, words):
            return from_bytes(field, words.view(torch.uint8).reshape(-1))
    if local.shape[0] == 0:   # more ranks than parties: this rank holds none and contributes zeros
        part = torch.zeros(N, dtype=local.dtype, device=local.device)
    else:
        part = partial(M61, local, lam_local).reshape(N)          # canonical, < 2^61
    mine = torch.empty(N // world, dtype=part.dtype, device=part.device)
    dist.reduce_scatter_tensor(mine, part.contiguous(), op=dist.ReduceOp.SUM, group=group)
    return fold(M61, mine).reshape(N // world, 1)


def open_by_partial_gather(field: int, local: torch.Tensor, lam_local, chunk: int = 1 << 24, group=None, partial=None,
                           total=None) -> torch.Tensor:
    """The open step for ANY field with 1/parties_per_rank of the all-gather's volume: reconstruction at a point is linear in
    the shares, so every rank first reduces ITS parties to the partial sum  sum_j lambda_j * share_j  (one field element per
    secret, a reconstruct kernel over its own rows only), the ranks all-gather those partials -- world x N elements instead
    of n x N -- and every rank adds the world partials with the field's addition (Vector::sum per secret).  Every rank ends
    with every secret, bit-identical to open_and_reconstruct (field arithmetic is exact); per rank the reconstruct work is
    its own parties' share of it.  For C4 (40 parties of 16 bytes on 8 ranks) a rank receives 7 x 16 bytes per secret instead
    of 35 x 16.  (What it gives up: no rank sees the other parties' individual shares, so nothing can re-check them --
    the semi-honest open.  Mersenne61 has the cheaper reduce-scatter form above when each rank needs only its slice.)

    local: [parties of this rank][N][L] (may have zero rows: a rank without parties contributes zeros); lam_local: their
    Lagrange coefficients.  Chunked over the secret axis, the all-gather of chunk k overlapping the partial sums of chunk
    k + 1.  `partial(field, shares, lam, out)` and `total(field, rows, out)` default to the HIP kernels (shamir_recover,
    additive_recover); the gloo tests inject CPU checkers."""
    if partial is None:
        from . import shamir_recover as partial
    if total is None:
        from . import additive_recover

        def total(field_, rows, out):
            return additive_recover(field_, rows, out=out)
    world = dist.get_world_size(group)
    N, L = local.shape[1], local.shape[2]
    out = torch.empty((N, L), dtype=local.dtype, device=local.device)
    pending = None
    for s0 in range(0, N, chunk):
        c = min(chunk, N - s0)
        mine = torch.zeros((c, L), dtype=local.dtype, device=local.device)
        if local.shape[0]:
            partial(field, local[:, s0:s0 + c], lam_local, mine)
        buf = torch.empty((world, c, L), dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(buf.view(world * c, L), mine, group=group, async_op=True)  # (concatenation form)
        if pending is not None:
            pwork, pbuf, ps0, pc, _ = pending
            pwork.wait()
            total(field, pbuf, out[ps0:ps0 + pc])
        pending = (work, buf, s0, c, mine)
    if pending is not None:
        pwork, pbuf, ps0, pc, _ = pending
        pwork.wait()
        total(field, pbuf, out[ps0:ps0 + pc])
    return out


# ---- the same open step behind the C ABI (scl_hip_comm_*, scl_hip_open_*: RCCL called from the library itself) -------------
class Communicator:
    """The C ABI's communicator handle: an RCCL communicator made by the library (ncclCommInitRank) plus the stream, events and
    gather buffers of the open step.  With torch.distributed initialised the 128-byte unique id travels from rank 0 by a
    broadcast on that process group (any backend); a single process needs no process group at all."""

    def __init__(self, group=None):
        import ctypes as C

        from . import _chk, lib
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            _chk(lib.scl_hip_comm_unique_id(ident))
        if self.world > 1:
            dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
            t = torch.tensor(list(ident), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = (C.c_ubyte * 128)(*t.cpu().tolist())
        self._h = C.c_void_p()
        _chk(lib.scl_hip_comm_init_rank(C.byref(self._h), self.world, self.rank, ident))

    def close(self):
        from . import lib
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.scl_hip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def open_row_order(n: int, world: int) -> list[int]:
    """party held by each row of a gathered chunk (scl_hip_open_row_order; -1 = padding): host only"""
    import ctypes as C

    from . import _chk, lib
    per = parties_per_rank(n, world)
    order = (C.c_long * (per * world))()
    _chk(lib.scl_hip_open_row_order(n, world, order))
    return list(order)


def open_all_gather_c(comm: Communicator, field: int, local: torch.Tensor, n: int, lam, chunk: int = 1 << 24,
                      out: torch.Tensor | None = None) -> torch.Tensor:
    """scl_hip_open_all_gather: `local` is this rank's [parties_per_rank(n)][N][L] slab; every rank returns all N secrets"""
    from . import _chk, _dev, _host, _hp, _stream, lib
    per, N, L = local.shape
    if per != parties_per_rank(n, comm.world):
        raise ValueError(f"local slab must have {parties_per_rank(n, comm.world)} party rows (pad the last rank)")
    if out is None:
        out = torch.empty((N, L), dtype=local.dtype, device=local.device)
    lam_h = _host(lam)
    _chk(lib.scl_hip_open_all_gather(comm._h, field, _dev(out), _dev(local), N, n, _hp(lam_h), N, chunk, _stream()))
    return out


def open_partial_gather_c(comm: Communicator, field: int, mine: torch.Tensor, lam_local, chunk: int = 1 << 24,
                          out: torch.Tensor | None = None) -> torch.Tensor:
    """scl_hip_open_partial_gather: `mine` is the rank's OWN [parties][N][L] rows (may have zero rows), lam_local their
    Lagrange coefficients"""
    from . import _chk, _dev, _host, _hp, _stream, lib
    cnt, N, L = mine.shape
    if out is None:
        out = torch.empty((N, L), dtype=mine.dtype, device=mine.device)
    lam_h = _host(lam_local) if cnt else None
    _chk(lib.scl_hip_open_partial_gather(comm._h, field, _dev(out), _dev(mine) if cnt else None, N, cnt,
                                         _hp(lam_h) if cnt else None, N, chunk, _stream()))
    return out


def open_reduce_scatter_c(comm: Communicator, field: int, mine: torch.Tensor, lam_local, chunk: int = 1 << 24,
                          out: torch.Tensor | None = None, all_ranks: bool = True) -> torch.Tensor:
    """scl_hip_open_reduce_scatter (Mersenne61, at most 8 ranks): the ranks' canonical partial sums meet in an
    ncclReduceScatter of plain 64-bit sums, each rank folds its slice mod p; all_ranks: an all-gather then hands every rank every
    secret, else out[s] is written only on the rank that owns s (slice_owner)"""
    from . import _chk, _dev, _host, _hp, _stream, lib
    cnt, N, L = mine.shape
    if out is None:
        out = torch.zeros((N, L), dtype=mine.dtype, device=mine.device)
    lam_h = _host(lam_local) if cnt else None
    _chk(lib.scl_hip_open_reduce_scatter(comm._h, field, _dev(out), _dev(mine) if cnt else None, N, cnt,
                                         _hp(lam_h) if cnt else None, N, chunk, 1 if all_ranks else 0, _stream()))
    return out


def slice_owner(s: int, N: int, world: int, chunk: int = 1 << 24) -> int:
    """the rank scl_hip_open_reduce_scatter(all_ranks = 0) leaves secret s on (scl_hip.h)"""
    chunk = (min(chunk or (1 << 24), N) + 1) & ~1
    s0 = s // chunk * chunk
    cnt = min(chunk, N - s0)
    return (s - s0) // -(-cnt // world)
