"""One image split into row slabs over the GPUs of a node.

The reference transforms a whole image per call (apply_periodic_conv_2d.py:3-28: fft2, multiply,
ifft2); an image too large for one GPU, or one that should be solved faster than one GPU can, is
split here by rows.  Rank r owns H / P consecutive image rows of every iterate array.  Everything
the solver loops do per pixel — the row transforms, D and its adjoint (a horizontal difference,
multiplying_matrix.py:55-59), the prox maps, the relaxation and dual updates, the loss sums — is
local to a row and runs in the fused row kernel on the slab.  Only the column stage (column
transform, resolvent algebra, inverse column transform) needs whole columns: before it the
half-spectra are redistributed so that every rank holds all H rows of its share of the spectral
tiles, and afterwards back.  Both redistributions are one all-to-all each (RCCL over xGMI with the
"nccl" backend; CPU tensors with "gloo" in tests).

Layouts (C = columns per spectral tile, complex elements):
  row side     [n_tiles][2][rows][C]                    the tiles that go to one rank are contiguous
  column side  [source rank][local tiles][2][rows][C]   what the all-to-all delivers
The column kernels address the per-source row blocks of the column side directly (rows per rank a
power of two), so neither side repacks anything; for other row counts one strided device copy on the
receiving side builds [2][local tiles][H][C] instead.

Tile groups: with `exchange_groups` = G > 1 (the default picks G so that a message between two ranks stays below
256 MiB, and at least 4 when there is an exchange at all) the tiles of every rank are cut into G groups and the column
stage becomes a pipeline: inbound exchange of group g + 1, column kernels of group g and outbound exchange of group
g - 1 are in flight together (the collectives on RCCL's stream, ordered against the plan's stream by events).

Loss records add up over ranks: every rank finalises the sums of its own rows, the sums of |d| and
d^2 are all-reduced, and l2 / mse / psnr are recomputed from the totals.  Without a loss_function that
happens once per solve.  With loss_function = l1_loss / l2_loss / mse / psnr of this package (and tol) the
reference's early stop (admm:152-154, pdr:159-161, pddr:119-121) is reproduced: after every iteration the
four local sums are all-reduced (32 bytes, enqueued on the stream like the exchanges: no host round trip),
every rank evaluates the same stop test on the device and a set flag freezes the iterates of all ranks at the
reference's iteration; the host looks at the flag every `check_every` iterations to leave its launch loop.
ssim and any callable that is not one of those four do not add up over slabs: for them the rows of all ranks are
gathered every iteration and the function sees the whole image (WholeImageLoss), the reference's own host loop.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .distributed import shard_range

SOLVER_ADMM, SOLVER_PDR, SOLVER_PDDR = 0, 1, 2


class SlabLayout:
    """Who owns what: `rows` image rows per rank and a contiguous span of spectral tiles per rank."""

    def __init__(self, H: int, W: int, tile_cols: int, n_tiles: int, rank: int, world: int):
        if H % world:
            raise ValueError(f"image height {H} is not a multiple of the number of ranks {world}")
        self.H, self.W, self.C, self.n_tiles, self.rank, self.world = int(H), int(W), int(tile_cols), int(n_tiles), int(rank), int(world)
        self.rows = H // world
        self.row_lo = rank * self.rows
        self.tile_spans: List[Tuple[int, int]] = [shard_range(n_tiles, r, world) for r in range(world)]
        self.tile_lo, self.tile_hi = self.tile_spans[rank]
        self.local_tiles = self.tile_hi - self.tile_lo


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class SlabExchange:
    """The spectral buffers of a rank and the all-to-all transposes between them.

    blocked (one rank, or a power-of-two number of rows per rank): spec_cols is
    [source rank][local tiles][spectra][rows][C], exactly what the all-to-all delivers; the column kernels
    address the per-source row blocks themselves (n_row_blocks = world), so there is no repacking — with one
    rank spec_cols IS spec_rows.  Otherwise spec_cols is [spectra][local tiles][H][C] and one strided device
    copy on the receiving side puts every source's rows at their global row index (n_row_blocks = 0)."""

    def __init__(self, lay: SlabLayout, complex_dtype: torch.dtype, device, group=None, n_spectra: int = 2, force_exchange: bool = False,
                 exchange_groups: int = 0):
        """force_exchange: with one rank the all-to-all is an identity and is skipped; True makes it happen anyway (the
        collective then runs with a one-rank group: the call shape of the multi-rank exchange, on one device)."""
        self.lay, self.group, self.ns = lay, group, int(n_spectra)
        ns = self.ns
        self.exchange = lay.world > 1 or (bool(force_exchange) and dist.is_available() and dist.is_initialized())
        self.blocked = lay.world == 1 or (lay.rows & (lay.rows - 1)) == 0
        self.n_row_blocks = lay.world if self.blocked else 0
        self.spec_rows = torch.zeros((lay.n_tiles, ns, lay.rows, lay.C), dtype=complex_dtype, device=device)
        stage = None
        if self.exchange:
            stage = torch.zeros((lay.world, lay.local_tiles, ns, lay.rows, lay.C), dtype=complex_dtype, device=device)
        if self.blocked:
            self.spec_cols = stage if stage is not None else self.spec_rows.unsqueeze(0)
            self._stage = None
        else:
            self.spec_cols = torch.zeros((ns, lay.local_tiles, lay.H, lay.C), dtype=complex_dtype, device=device)
            self._stage = stage
        per_tile = ns * lay.rows * lay.C * 2  # real numbers per tile of the row-side layout
        self._row_splits = [(hi - lo) * per_tile for lo, hi in lay.tile_spans]
        self._col_splits = [lay.local_tiles * per_tile] * lay.world
        # tile groups of the pipelined column stage (blocked layout only: the other one repacks whole buffers)
        self.groups = 1
        if self.exchange and self.blocked:
            esz = 16 if complex_dtype == torch.complex128 else 8
            biggest = max(hi - lo for lo, hi in lay.tile_spans) * ns * lay.rows * lay.C * esz   # bytes between two ranks per exchange
            want = int(exchange_groups) if exchange_groups else max(4, -(-biggest // (256 << 20)))
            self.groups = max(1, min(want, min(hi - lo for lo, hi in lay.tile_spans) or 1))
        self._comm = None

    def group_range(self, n_tiles: int, g: int) -> Tuple[int, int]:
        """[a, b) of group g among n_tiles tiles: the same rule on the sending and the receiving side."""
        c = -(-n_tiles // self.groups)
        return min(g * c, n_tiles), min((g + 1) * c, n_tiles)

    def _a2a_group(self, g: int, to_cols: bool):
        """Exchange of tile group g: rows -> columns (to_cols) or back.  Returns a work handle to wait on (or None)."""
        lay = self.lay
        a, b = self.group_range(lay.local_tiles, g)
        col_side = [self.spec_cols[src, a:b] for src in range(lay.world)]
        row_side = []
        for lo, hi in lay.tile_spans:
            ra, rb = self.group_range(hi - lo, g)
            row_side.append(self.spec_rows[lo + ra:lo + rb])
        out_list, in_list = (col_side, row_side) if to_cols else (row_side, col_side)
        out_list = [torch.view_as_real(t) for t in out_list]
        in_list = [torch.view_as_real(t) for t in in_list]
        if dist.get_backend(self.group) == "nccl":
            return dist.all_to_all(out_list, in_list, group=self.group, async_op=True)
        # gloo (tests): one host-staged all_to_all_single with the pieces packed back to back
        inp = torch.cat([t.reshape(-1) for t in in_list]).cpu()
        out = torch.empty(sum(t.numel() for t in out_list), dtype=inp.dtype)
        dist.all_to_all_single(out, inp, [t.numel() for t in out_list], [t.numel() for t in in_list], group=self.group)
        o = 0
        for t in out_list:
            t.copy_(out[o:o + t.numel()].view(t.shape))
            o += t.numel()
        return None

    def pipelined(self, col_fn):
        """The whole column stage in tile groups: col_fn(first, count) launches the column kernels of the local tiles
        [first, first + count) on the current stream.  Inbound exchanges run two groups ahead of the kernels."""
        dev = self.spec_rows.device
        main = torch.cuda.current_stream(dev)
        nccl = dist.get_backend(self.group) == "nccl"
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=dev)
        comm = self._comm

        def issue(g, to_cols):
            if not nccl:
                return self._a2a_group(g, to_cols)
            ev = main.record_event()
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                return self._a2a_group(g, to_cols)

        G = self.groups
        inbound = {g: issue(g, True) for g in range(min(2, G))}
        outbound = []
        for g in range(G):
            w = inbound.pop(g)
            if w is not None:
                w.wait()                      # the current stream waits for group g to have arrived
            a, b = self.group_range(self.lay.local_tiles, g)
            if b > a:
                col_fn(a, b - a)
            outbound.append(issue(g, False))
            if g + 2 < G:
                inbound[g + 2] = issue(g + 2, True)
        for w in outbound:
            if w is not None:
                w.wait()

    def _a2a(self, out: torch.Tensor, inp: torch.Tensor, out_splits: Sequence[int], in_splits: Sequence[int]):
        o = torch.view_as_real(out).reshape(-1)
        i = torch.view_as_real(inp).reshape(-1)
        if dist.get_backend(self.group) == "gloo" and o.is_cuda:
            # gloo has no device all-to-all: stage through host memory (tests on a single GPU only)
            oc, ic = torch.empty(o.shape, dtype=o.dtype), i.cpu()
            dist.all_to_all_single(oc, ic, list(out_splits), list(in_splits), group=self.group)
            o.copy_(oc)
            return
        dist.all_to_all_single(o, i, list(out_splits), list(in_splits), group=self.group)

    def _cols_view(self) -> torch.Tensor:
        l = self.lay  # [spectra][tiles][H][C] seen as [spectra][tiles][source rank][rows][C]
        return self.spec_cols.view(self.ns, l.local_tiles, l.world, l.rows, l.C)

    def rows_to_cols(self):
        """Every rank's rows of my tiles -> spec_cols (all H rows of the local tiles)."""
        if not self.exchange:
            return  # spec_cols is spec_rows
        if self.blocked:
            self._a2a(self.spec_cols, self.spec_rows, self._col_splits, self._row_splits)
            return
        self._a2a(self._stage, self.spec_rows, self._col_splits, self._row_splits)
        self._cols_view().copy_(self._stage.permute(2, 1, 0, 3, 4))

    def cols_to_rows(self):
        """spec_cols -> my rows of every tile in spec_rows."""
        if not self.exchange:
            return
        if self.blocked:
            self._a2a(self.spec_rows, self.spec_cols, self._row_splits, self._col_splits)
            return
        self._stage.copy_(self._cols_view().permute(2, 1, 0, 3, 4))
        self._a2a(self.spec_rows, self._stage, self._row_splits, self._col_splits)


def combine_loss_records(local: torch.Tensor, npix: float, group=None) -> torch.Tensor:
    """local: [iters][8] per-slab records (l1, l2, mse, psnr of two pairs).  Returns the records of the
    whole image: l1 and l2^2 are sums over the slabs; mse and psnr follow from the totals."""
    sums = torch.stack((local[:, 0], local[:, 1] ** 2, local[:, 4], local[:, 5] ** 2), dim=1).contiguous()
    _, world = _world(group)
    if world > 1:
        if dist.get_backend(group) == "gloo" and sums.is_cuda:
            h = sums.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            sums = h.to(local.device)
        else:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    out = torch.empty_like(local)
    for pair in range(2):
        s1, s2 = sums[:, 2 * pair], sums[:, 2 * pair + 1]
        mse = s2 / npix
        out[:, 4 * pair + 0] = s1
        out[:, 4 * pair + 1] = torch.sqrt(s2)
        out[:, 4 * pair + 2] = mse
        out[:, 4 * pair + 3] = torch.where(mse == 0, torch.full_like(mse, math.inf), 10.0 * torch.log10(1.0 / mse))
    return out


class SlabSolver:
    """Plan, buffers and the two local stages of one slab-decomposed solve (fp32 or fp64)."""

    def __init__(self, kernel, image_shape: Tuple[int, int], t_resolvent: float, s_resolvent: float, dtype=torch.float32, device=None, group=None,
                 n_spectra: int = 2, eig_arrays=None, local_multipliers: bool = True, force_exchange: bool = False, exchange_groups: int = 0):
        from . import _lib as L
        from .plan import create_plan, compute_device, real_dtype
        self.L = L
        self.group = group
        self.rank, self.world = _world(group)
        H, W = int(image_shape[0]), int(image_shape[1])
        self.dt = real_dtype(dtype)
        self.dev = compute_device(device)
        self.injected = eig_arrays is not None
        # a rank keeps the multipliers of its own spectral tiles only; injected eigen arrays (the fp64 parity route)
        # and local_multipliers=False take a plan with the whole plane, as a single-device solve has
        self.local = bool(local_multipliers) and not self.injected
        self.plan = create_plan(H, W, 1, self.dt, self.dev, slab=self.local)
        c, n = C.c_int(0), C.c_int(0)
        L.check(L.lib().ddn_slab_layout(self.plan._h, C.byref(c), C.byref(n)))
        self.lay = SlabLayout(H, W, c.value, n.value, self.rank, self.world)
        if self.local:
            self.plan.restrict_tiles(self.lay.tile_lo, self.lay.local_tiles)
        # eig_vals_mat = 1 + t^2 |eK|^2 + (t^2 + t s) |eD|^2   (multiplying_matrix.py:61-72 with D1 == D2)
        cK, cD = float(t_resolvent) ** 2, float(t_resolvent) ** 2 + float(t_resolvent) * float(s_resolvent)
        if self.injected:
            # the reference's own eigen arrays of the WHOLE image, consumed bit for bit (fp64 parity)
            self.plan.set_eig(eig_arrays["K"], eig_arrays.get("D"))
            self.plan.set_resolvent(cK, cD, eig_arrays["M"])
        else:
            self.plan.set_kernel(kernel)
            self.plan.set_resolvent(cK, cD)
        self.xch = SlabExchange(self.lay, torch.complex128 if self.dt == torch.float64 else torch.complex64, self.dev, group, n_spectra, force_exchange, exchange_groups)
        self._rec = torch.zeros((8,), dtype=torch.float64, device=self.dev)
        self._sums = torch.zeros((4,), dtype=torch.float64, device=self.dev)
        self._done = torch.full((1,), -1, dtype=torch.int32, device=self.dev)
        L.check(L.lib().ddn_slab_begin(self.plan._h))

    def close(self):
        self.plan.close()

    def rows(self, solver: int, state: Sequence[Optional[torch.Tensor]], b: torch.Tensor, t: float, rho: float, gamma: float, objective: int,
             consume: bool, produce: bool, sol: Optional[torch.Tensor] = None, sol_only: bool = False):
        from .plan import _ptr
        s = list(state) + [None] * (4 - len(state))
        self.L.check(self.L.lib().ddn_slab_rows(self.plan._h, solver, self.lay.rows, _ptr(self.xch.spec_rows), _ptr(s[0]), _ptr(s[1]), _ptr(s[2]), _ptr(s[3]),
                                                _ptr(b), float(t), float(rho), float(gamma), int(objective), int(consume), int(produce), _ptr(sol), int(sol_only)))

    def column_stage(self):
        """rows -> columns all-to-all, local column stage, columns -> rows all-to-all."""
        from .plan import _ptr
        if self.xch.groups > 1:
            self.xch.pipelined(lambda first, count: self.L.check(self.L.lib().ddn_slab_cols_part(
                self.plan._h, _ptr(self.xch.spec_cols), self.lay.tile_lo, self.lay.local_tiles, self.xch.n_row_blocks, 0, first, count)))
            return
        self.xch.rows_to_cols()
        self.L.check(self.L.lib().ddn_slab_cols(self.plan._h, _ptr(self.xch.spec_cols), self.lay.tile_lo, self.lay.local_tiles, self.xch.n_row_blocks))
        self.xch.cols_to_rows()

    def rows_cp(self, state, b: torch.Tensor, t: float, s: float, gamma: float, objective: int, step: int, consume: bool, produce: bool):
        from .plan import _ptr
        if self.injected and getattr(self, "_aux", None) is None:
            self._aux = torch.zeros_like(self.xch.spec_rows)
        aux = self._aux if self.injected else None
        self.L.check(self.L.lib().ddn_slab_rows_cp(self.plan._h, self.lay.rows, _ptr(self.xch.spec_rows), _ptr(aux), _ptr(state[0]), _ptr(state[1]), _ptr(state[2]), _ptr(b),
                                                   float(t), float(s), float(gamma), int(objective), int(step), int(consume), int(produce)))

    def column_multiply(self, conj: bool):
        """rows -> columns all-to-all, spectrum times eK (or its conjugate) on the local tiles, all-to-all back."""
        from .plan import _ptr
        if self.xch.groups > 1:
            self.xch.pipelined(lambda first, count: self.L.check(self.L.lib().ddn_slab_cols_part(
                self.plan._h, _ptr(self.xch.spec_cols), self.lay.tile_lo, self.lay.local_tiles, self.xch.n_row_blocks, 2 if conj else 1, first, count)))
            return
        self.xch.rows_to_cols()
        self.L.check(self.L.lib().ddn_slab_cols_multiply(self.plan._h, _ptr(self.xch.spec_cols), self.lay.tile_lo, self.lay.local_tiles, int(conj),
                                                         self.xch.n_row_blocks))
        self.xch.cols_to_rows()

    def stop_test(self, kind: int, tol: float, it: int, first_it: int, record: Optional[torch.Tensor]):
        """Whole-image loss record of the last consuming row launch into `record` ([8], may be None) and the stop test
        `loss < tol` on the device: local sums -> all-reduce (SUM) of four doubles -> every rank decides alike."""
        from .plan import _ptr
        self.L.check(self.L.lib().ddn_slab_loss_sums(self.plan._h, self.lay.rows, _ptr(self._sums)))
        if self.world > 1 or self.xch.exchange:
            if dist.get_backend(self.group) == "gloo" and self._sums.is_cuda:
                h = self._sums.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                self._sums.copy_(h)
            else:
                dist.all_reduce(self._sums, op=dist.ReduceOp.SUM, group=self.group)
        self.L.check(self.L.lib().ddn_slab_stop_test(self.plan._h, _ptr(self._sums), int(kind), float(tol), int(it), int(first_it), _ptr(record)))

    def stopped_at(self) -> int:
        """-1 while running, else the iteration at which the device froze the solve (reads one int: a host sync)."""
        from .plan import _ptr
        self.L.check(self.L.lib().ddn_slab_iters_done(self.plan._h, _ptr(self._done)))
        return int(self._done.item())

    def loss_record(self, out: torch.Tensor):
        """out[8] <- the loss sums of the local rows seen by the last consuming row launch."""
        from .plan import _ptr
        self.L.check(self.L.lib().ddn_slab_loss(self.plan._h, self.lay.rows, _ptr(out)))

    def gather_rows(self, rows: torch.Tensor) -> torch.Tensor:
        """(H / P, W) rows of every rank -> the whole (H, W) image on every rank (an all-gather in rank order, which
        is row order)."""
        rows = rows.contiguous()
        if self.world == 1:
            return rows
        staged = dist.get_backend(self.group) == "gloo" and rows.is_cuda
        src = rows.cpu() if staged else rows
        full = torch.empty((self.world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather(list(full.chunk(self.world)), src, group=self.group)
        return full.to(rows.device) if staged else full

    def agree(self, v: float) -> float:
        """The value of the group's first rank on every rank: all ranks then take the same branch on it."""
        if self.world == 1:
            return v
        cpu = dist.get_backend(self.group) == "gloo"
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if cpu else self.dev)
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        return float(t.item())


class WholeImageLoss:
    """loss_function(iterate, other) for the losses whose value is not a sum over row slabs: ssim (its windows reach
    across slab borders) and any callable that is not one of this package's own.  The rows of every rank are gathered
    and the function sees the whole image, as in the reference's loop (admm:139-154, pdr:143-161, pddr:103-121): one
    all-gather and one host decision per iteration tested.  This package's ssim runs on the compute device; any other
    callable gets its tensors on the device the caller's rows live on.  `values` keeps what was evaluated."""

    def __init__(self, sv: SlabSolver, fn, tol: float, b_rows: torch.Tensor, caller_device):
        from .core.loss import ssim
        self.sv, self.fn, self.tol = sv, fn, float(tol)
        self.where = None if fn is ssim else caller_device
        self.b = self._place(sv.gather_rows(b_rows))
        self.values = []
        self.stopped = -1

    def _place(self, t: torch.Tensor) -> torch.Tensor:
        return t if self.where is None else t.to(self.where)

    def test(self, it: int, a_rows: torch.Tensor, other_rows: Optional[torch.Tensor] = None) -> bool:
        """Evaluate at iteration `it`; True when the solve stops here (loss < tol, the same answer on every rank)."""
        a = self._place(self.sv.gather_rows(a_rows))
        o = self.b if other_rows is None else self._place(self.sv.gather_rows(other_rows))
        v = self.fn(a, o)
        v = float(v.reshape(-1)[0]) if isinstance(v, torch.Tensor) else float(v)
        v = self.sv.agree(v)
        self.values.append(v)
        if v < self.tol:
            self.stopped = it
            return True
        return False


WHOLE_IMAGE = -1      # the `kind` of a loss that needs the gathered image (WholeImageLoss)


def _stop_args(kwargs):
    """(kind, tol, check_every) of the optional early stop.  This package's l1_loss, l2_loss, mse and psnr add up over
    slabs: kinds 1-4, tested on the device from all-reduced sums.  ssim and any other callable need the whole image:
    kind WHOLE_IMAGE, evaluated every iteration on the gathered rows (check_every does not apply)."""
    fn = kwargs.get("loss_function")
    if fn is None:
        return 0, 0.0, 0
    if not callable(fn):
        raise ValueError("slab solvers: loss_function must be callable")
    kind = int(getattr(fn, "_ddn_kind", 0))
    if not 1 <= kind <= 4:
        kind = WHOLE_IMAGE
    return kind, float(kwargs.get("tol", 1e-6)), max(1, int(kwargs.get("check_every", 10)))


def _finish_records(sv, ring, n, kind, H, W, group, whole=None):
    """(records of the iterations applied, iterations applied).  With a whole-image loss the records are the values
    the function returned, one per iteration tested."""
    if whole is not None:
        return torch.tensor(whole.values, dtype=torch.float64), (n if whole.stopped < 0 else whole.stopped)
    if kind:
        st = sv.stopped_at()
        nd = n if st < 0 else st
        return ring[:nd].cpu(), nd
    return combine_loss_records(ring[:n], float(H) * float(W), group).cpu(), n


def _local_rows(full: torch.Tensor, lay: SlabLayout) -> torch.Tensor:
    return full[..., lay.row_lo:lay.row_lo + lay.rows, :]


def admm_solver_slab(b_rows: torch.Tensor, image_shape: Tuple[int, int], objective_function: str = "l1", t: float = 18, rho: float = 0.001,
                     gamma: float = 0.5, kernel=None, niters: int = 500, **kwargs):
    """admm_solver (admm/algorithm.py:20-201) for one image whose rows are spread over the ranks of
    `group`.  b_rows: this rank's (H / P, W) rows of the blurred image.  Returns this rank's rows of the
    solution of the final solve (admm:160-173).  Runs up to niters - 1 iterations; with loss_function, tol and
    check_every it stops where the reference does (admm:139-154): l1_loss, l2_loss, mse or psnr of this package are tested
    on the device from all-reduced sums, ssim or any other callable on the gathered image (return_losses then gives the
    values of the function, one per iteration tested).

    Keyword extensions: init_state = (u, w, y, z) local slabs ((rows, W) and (3, rows, W)); without it
    the reference's draws (admm:49-54) are made for the WHOLE image on every rank from torch's CPU
    generator and the local rows kept, so that all ranks must share the generator state;
    return_state, return_losses (records of the whole image, [iterations][8]: l1, l2, mse, psnr of the
    primary pair, then of the secondary pair), eig_arrays = {"K", "D", "M"} (the reference's eigen arrays
    of the whole image, for fp64 parity), dtype, device, group."""
    if objective_function not in ("l1", "l2"):
        raise ValueError("Invalid objective function. Choose 'l1' or 'l2'.")
    from . import _lib as L
    from .plan import stage
    H, W = int(image_shape[0]), int(image_shape[1])
    sv = SlabSolver(kernel, (H, W), 1, 1, dtype=kwargs.get("dtype", b_rows.dtype), device=kwargs.get("device", b_rows.device), group=kwargs.get("group"),
                    eig_arrays=kwargs.get("eig_arrays"), local_multipliers=kwargs.get("local_multipliers", True), force_exchange=kwargs.get("force_exchange", False), exchange_groups=kwargs.get("exchange_groups", 0))
    try:
        lay, dev, dt = sv.lay, sv.dev, sv.dt
        if tuple(b_rows.shape) != (lay.rows, W):
            raise ValueError(f"b_rows must hold this rank's {lay.rows} x {W} rows, got {tuple(b_rows.shape)}")
        init = kwargs.get("init_state")
        if init is None:
            _x = torch.clamp(torch.randn((H, W), dtype=torch.float32), min=0, max=1)
            full = [torch.clamp(torch.randn(shape, dtype=torch.float32), min=0, max=1) for shape in ((H, W), (H, W), (3, H, W), (3, H, W))]
            init = [_local_rows(v, lay) for v in full]
        u, w, y, z = [stage(torch.as_tensor(v), dev, dt) for v in init]
        bd = stage(b_rows, dev, dt)
        sol = torch.zeros_like(bd)
        n = max(int(niters) - 1, 0)
        obj = L.OBJ[objective_function]
        ring = torch.zeros((max(n, 1), 8), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            sv.plan.sync_stream()
            st = (u, w, y, z)
            kind, tol, every = _stop_args(kwargs)
            whole = WholeImageLoss(sv, kwargs["loss_function"], tol, bd, b_rows.device) if kind == WHOLE_IMAGE else None
            sv.rows(SOLVER_ADMM, st, bd, t, rho, gamma, obj, False, True)
            stopped = False
            for c in range(1, n + 1):
                sv.column_stage()
                # with a stop test every consuming launch stores the look-ahead solution, so that the one of the stopping
                # iteration is what `sol` holds when the flag freezes the rows (admm:118-154)
                sv.rows(SOLVER_ADMM, st, bd, t, rho, gamma, obj, True, True, sol=sol if kind else None)
                if c >= 2:
                    if whole is not None:
                        if c - 1 >= 2 and whole.test(c - 1, sol):   # loss(sol, b) for i > 1 (admm:118)
                            stopped = True
                            break
                    elif kind:
                        sv.stop_test(kind, tol, c - 1, 2, ring[c - 2])
                        if c % every == 0 and sv.stopped_at() >= 0:
                            stopped = True
                            break
                    else:
                        sv.loss_record(ring[c - 2])   # the look-ahead solution of iteration c - 1 (admm:118-139)
            if not stopped:
                sv.column_stage()
                sv.rows(SOLVER_ADMM, st, bd, t, rho, gamma, obj, True, False, sol=sol, sol_only=True)
                if n >= 1:
                    if whole is not None:
                        if n >= 2:
                            whole.test(n, sol)
                    elif kind:
                        sv.stop_test(kind, tol, n, 2, ring[n - 1])
                    else:
                        sv.loss_record(ring[n - 1])
            torch.cuda.synchronize(dev)
        ret = [sol.to(b_rows.device)]
        if kwargs.get("return_state"):
            ret.append(tuple(v.to(b_rows.device) for v in st))
        if kwargs.get("return_losses"):
            ret.append(_finish_records(sv, ring, n, kind, H, W, kwargs.get("group"), whole)[0])
        return ret[0] if len(ret) == 1 else tuple(ret)
    finally:
        sv.close()


def _dr_slab(solver: int, n: int, sv: SlabSolver, st, bd, t, rho, gamma, obj, ring, kind=0, tol=0.0, every=0, first_it=1, whole=None):
    from .core.proximal_operators import prox_box
    sv.rows(solver, st, bd, t, rho, gamma, obj, False, True)
    prev = torch.empty_like(st[0]) if (whole is not None and solver == SOLVER_PDR) else None
    for c in range(1, n + 1):
        sv.column_stage()
        if prev is not None:
            prev.copy_(st[0])
        sv.rows(solver, st, bd, t, rho, gamma, obj, True, c < n)
        if whole is not None:
            # primal DR: loss(z1, z1prev) (pdr:143-161); primal-dual DR: loss(clip(p), b) (pddr:103-121)
            if c >= first_it and (whole.test(c, st[0], prev) if solver == SOLVER_PDR else whole.test(c, prox_box(st[0], 1))):
                break
        elif kind:
            sv.stop_test(kind, tol, c, first_it, ring[c - 1])
            if c % every == 0 and sv.stopped_at() >= 0:
                break
        else:
            sv.loss_record(ring[c - 1])


def primal_dr_splitting_slab(problem: str, kernel, b_rows: torch.Tensor, image_shape: Tuple[int, int], niters: int = 500, gamma: float = 0.049,
                             t: float = 2.0, rho: float = 0.1, **kwargs):
    """primal_dr_splitting (primal_dr/algorithm.py:21-187) on row slabs: niters iterations from
    z1 = b, z2 = (b, b, b) (pdr:88-89) unless init_state = (z1, z2) local slabs is given; problem 'l1'
    selects the l1 data term, anything else the squared-l2 one (pdr:101-110).  Returns this rank's rows
    of clip(z1) (pdr:182-187).  Loss records: primary pair (z1, z1prev), secondary (z1, b)."""
    from .plan import stage
    from .core.proximal_operators import prox_box
    H, W = int(image_shape[0]), int(image_shape[1])
    sv = SlabSolver(kernel, (H, W), t, t, dtype=kwargs.get("dtype", b_rows.dtype), device=kwargs.get("device", b_rows.device), group=kwargs.get("group"),
                    eig_arrays=kwargs.get("eig_arrays"), local_multipliers=kwargs.get("local_multipliers", True), force_exchange=kwargs.get("force_exchange", False), exchange_groups=kwargs.get("exchange_groups", 0))
    try:
        dev, dt = sv.dev, sv.dt
        bd = stage(b_rows, dev, dt)
        init = kwargs.get("init_state")
        if init is None:
            z1, z2 = bd.clone(), torch.stack((bd, bd, bd)).contiguous()
        else:
            z1, z2 = [stage(torch.as_tensor(v), dev, dt) for v in init]
        n = max(int(niters), 0)
        ring = torch.zeros((max(n, 1), 8), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            sv.plan.sync_stream()
            kind, tol, every = _stop_args(kwargs)
            whole = WholeImageLoss(sv, kwargs["loss_function"], tol, bd, b_rows.device) if kind == WHOLE_IMAGE else None
            if n > 0:
                _dr_slab(SOLVER_PDR, n, sv, (z1, z2), bd, t, rho, gamma, 0 if problem == "l1" else 1, ring, kind, tol, every, 3, whole)   # j > 1 (pdr:143)
            torch.cuda.synchronize(dev)
        ret = [prox_box(z1, 1).to(b_rows.device)]                                                     # pdr:184
        if kwargs.get("return_state"):
            ret.append((z1.to(b_rows.device), z2.to(b_rows.device)))
        if kwargs.get("return_losses"):
            ret.append(_finish_records(sv, ring, n, kind, H, W, kwargs.get("group"), whole)[0])
        return ret[0] if len(ret) == 1 else tuple(ret)
    finally:
        sv.close()


def primal_dual_dr_splitting_slab(b_rows: torch.Tensor, kernel, image_shape: Tuple[int, int], niters: int = 1000, t: float = 3, rho: float = 0.4,
                                  gamma: float = 0.03, **kwargs):
    """primal_dual_dr_splitting (primal_dual_dr/algorithm.py:20-139) on row slabs: niters - 1
    iterations.  init_state = (p, q) local slabs; without it the reference's draws (pddr:49-50) are made
    for the whole image on every rank and the local rows kept.  Loss pair: (clip(p), b)."""
    from .plan import stage
    from .core.proximal_operators import prox_box
    H, W = int(image_shape[0]), int(image_shape[1])
    sv = SlabSolver(kernel, (H, W), t, t, dtype=kwargs.get("dtype", b_rows.dtype), device=kwargs.get("device", b_rows.device), group=kwargs.get("group"),
                    eig_arrays=kwargs.get("eig_arrays"), local_multipliers=kwargs.get("local_multipliers", True), force_exchange=kwargs.get("force_exchange", False), exchange_groups=kwargs.get("exchange_groups", 0))
    try:
        lay, dev, dt = sv.lay, sv.dev, sv.dt
        bd = stage(b_rows, dev, dt)
        init = kwargs.get("init_state")
        if init is None:
            init = [_local_rows(torch.rand((H, W), dtype=dt), lay), _local_rows(torch.rand((3, H, W), dtype=dt), lay)]
        p, q = [stage(torch.as_tensor(v), dev, dt) for v in init]
        n = max(int(niters) - 1, 0)
        ring = torch.zeros((max(n, 1), 8), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            sv.plan.sync_stream()
            kind, tol, every = _stop_args(kwargs)
            whole = WholeImageLoss(sv, kwargs["loss_function"], tol, bd, b_rows.device) if kind == WHOLE_IMAGE else None
            if n > 0:
                _dr_slab(SOLVER_PDDR, n, sv, (p, q), bd, t, rho, gamma, 0, ring, kind, tol, every, 2, whole)   # k > 1 (pddr:103)
            torch.cuda.synchronize(dev)
        ret = [prox_box(p, 1).to(b_rows.device)]                                                      # pddr:136
        if kwargs.get("return_state"):
            ret.append((p.to(b_rows.device), q.to(b_rows.device)))
        if kwargs.get("return_losses"):
            ret.append(_finish_records(sv, ring, n, kind, H, W, kwargs.get("group"), whole)[0])
        return ret[0] if len(ret) == 1 else tuple(ret)
    finally:
        sv.close()


def chambolle_pock_slab(b_rows: torch.Tensor, kernel, image_shape: Tuple[int, int], objective_function: str = "l1", t: float = 0.4, s: float = 0.7,
                        gamma: float = 0.01, max_iter: int = 500, **kwargs):
    """chambolle_pock (chambolle_pock/algorithm.py:21-143) on row slabs: max_iter - 1 iterations, two column
    stages (K, then K^T) and therefore four all-to-all exchanges per iteration.  b_rows: (H / P, W); returns
    this rank's rows of x with the reference's leading dimension, (1, H / P, W).  init_state = (x, z, y) local
    slabs; without it the reference's draws (cp:50-52) are made for the whole image on every rank and the
    local rows kept.  Loss pair: (x_next, b) (cp:96)."""
    if objective_function not in ("l1", "l2"):
        raise ValueError("Invalid objective function. Choose 'l1' or 'l2'.")
    from . import _lib as L
    from .plan import stage
    H, W = int(image_shape[0]), int(image_shape[1])
    sv = SlabSolver(kernel, (H, W), 1, 1, dtype=kwargs.get("dtype", b_rows.dtype), device=kwargs.get("device", b_rows.device), group=kwargs.get("group"),
                    n_spectra=1, eig_arrays=kwargs.get("eig_arrays"), local_multipliers=kwargs.get("local_multipliers", True), force_exchange=kwargs.get("force_exchange", False), exchange_groups=kwargs.get("exchange_groups", 0))
    try:
        lay, dev, dt = sv.lay, sv.dev, sv.dt
        b2 = b_rows.reshape(lay.rows, W)
        init = kwargs.get("init_state")
        if init is None:
            init = [_local_rows(torch.rand(size=shape, dtype=dt), lay) for shape in ((H, W), (H, W), (3, H, W))]
        x, z, y = [stage(torch.as_tensor(v), dev, dt) for v in init]
        bd = stage(b2, dev, dt)
        n = max(int(max_iter) - 1, 0)
        obj = L.OBJ[objective_function]
        ring = torch.zeros((max(n, 1), 8), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            sv.plan.sync_stream()
            st = (x, z, y)
            if n > 0:
                sv.rows_cp(st, bd, t, s, gamma, obj, 1, False, True)
                for c in range(1, n + 1):
                    sv.column_multiply(False)
                    sv.rows_cp(st, bd, t, s, gamma, obj, 0, True, True)
                    sv.column_multiply(True)
                    sv.rows_cp(st, bd, t, s, gamma, obj, 1, True, c < n)
                    sv.loss_record(ring[c - 1])
            torch.cuda.synchronize(dev)
        ret = [x.to(b_rows.device).unsqueeze(0)]
        if kwargs.get("return_state"):
            ret.append(tuple(v.to(b_rows.device) for v in st))
        if kwargs.get("return_losses"):
            ret.append(combine_loss_records(ring[:n], float(H) * float(W), kwargs.get("group")).cpu())
        return ret[0] if len(ret) == 1 else tuple(ret)
    finally:
        sv.close()
