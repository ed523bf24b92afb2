"""One-process-per-GPU sharding of a scan over the trial-frequency / trial-period grid.

Every trial frequency is independent given the (small, replicated) sample set, so the grid is
cut into ``world`` contiguous equal slabs, rank r computes slab r, and the only exchange is one
all-gather of the result array (RCCL over xGMI when the process group's backend is ``nccl``;
``gloo`` on CPU for the tests).  No sample is ever sharded, so no reduction is needed
(SURVEY.md §8e).  The reference's own parallelism — ``multiprocessing.Pool.map`` over periods,
``/root/reference/src/periodicity/phase.py:69-70,185-186`` — has the same shape with pickling
instead of a collective.

NOT part of the product package: this is launcher-side plumbing for jobs that are started as one process
per GPU (``torch.distributed.run``, as the bench driver does).  torch provides rendezvous and the
collective, never compute, and nothing under ``periodicity_amd/`` imports torch
(``tests/test_host_api.py`` imports the package with torch blocked).  The product's multi-GPU path - one
process driving N devices with RCCL directly - is the persistent plan ``pdc_gls_plan_*`` /
``pdc_gls_scan_multi`` in ``csrc/multi.hip`` (``_cabi.GlsPlan``, ``GLS(devices=...)``).
A process that uses both must initialise torch's GPU context (``torch.cuda.set_device``) BEFORE the
first scan call: the torch wheel carries its own HIP runtime, and the one loaded first serves both.
"""
import numpy as np


def slab_bounds(n_grid, world, rank):
    """``(begin, end, per)``: rank's half-open slab of ``[0, n_grid)`` and the padded slab length
    ``per = ceil(n_grid / world)`` every rank contributes to the all-gather."""
    if world < 1 or not 0 <= rank < world or n_grid < 0:
        raise ValueError("bad world/rank/grid size")
    per = -(-n_grid // world) if n_grid else 0
    begin = min(rank * per, n_grid)
    end = min(begin + per, n_grid)
    return begin, end, per


def gather_slabs(local, n_grid, group=None):
    """All-gather the per-rank slabs into the full array on every rank.

    ``local`` is this rank's slab as a 1-D float64 ``torch.Tensor`` (on the GPU for ``nccl``, on
    the CPU for ``gloo``) of length ``end - begin``; returns a tensor of length ``n_grid``.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    begin, end, per = slab_bounds(n_grid, world, rank)
    if local.numel() != end - begin:
        raise ValueError(f"rank {rank}: slab has {local.numel()} entries, expected {end - begin}")
    if n_grid == 0:
        return local.new_empty(0)
    send = local
    if end - begin < per:  # last rank(s): pad so that all contributions are equal-sized
        send = local.new_zeros(per)
        send[:end - begin] = local
    out = torch.empty(per * world, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    return out[:n_grid]


def _place(array, device, group=None):
    """A float64 numpy array as a tensor where the group's backend wants it: on the rank's GPU for
    ``nccl`` (RCCL), on the host for ``gloo`` (the CPU tests)."""
    import torch
    import torch.distributed as dist
    tensor = torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64))
    return tensor if dist.get_backend(group) == "gloo" else tensor.to(f"cuda:{device}")


def sharded_scan(compute_slab, n_grid, group=None):
    """Run ``compute_slab(begin, count) -> torch.Tensor`` for this rank's slab and all-gather."""
    import torch.distributed as dist
    begin, end, _ = slab_bounds(n_grid, dist.get_world_size(group), dist.get_rank(group))
    return gather_slabs(compute_slab(begin, end - begin), n_grid, group)


def sharded_gls(t, y, dy, f0, delta, nf, fit_mean=True, psd=False, device=None, group=None):
    """Generalized Lomb-Scargle power on ``f0 + j*delta``, j < nf, with the grid sharded over the
    ranks of ``group`` (one GPU per rank, ``cuda:LOCAL_RANK`` unless ``device`` is given).

    With an ``nccl`` (RCCL) group the slab never leaves the GPU before the exchange: the samples go
    up once, ``pdc_gls_scan_dev`` writes the slab into a device tensor on torch's current stream, the
    all-gather runs on device tensors, and only the gathered array comes back.  With ``gloo`` (CPU
    groups of the tests, or a host-side exchange) the slab is computed through the host entry point."""
    import torch
    import torch.distributed as dist

    from periodicity_amd import _cabi
    on_host = dist.get_backend(group) == "gloo"
    if device is None:
        device = _cabi.default_device() if on_host else torch.cuda.current_device()
    if on_host:
        def compute(begin, count):
            part = _cabi.gls_scan(t, y, dy, f0, delta, count, fit_mean, psd, j_begin=begin, device=device)
            return torch.from_numpy(part)
        return sharded_scan(compute, nf, group).numpy()

    where = f"cuda:{device}"
    d_t, d_y = (torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(where) for a in (t, y))
    d_dy = None if dy is None else torch.as_tensor(np.ascontiguousarray(dy, dtype=np.float64)).to(where)
    if d_y.numel() != d_t.numel() or (d_dy is not None and d_dy.numel() != d_t.numel()):
        raise ValueError("Input arrays have incompatible lengths.")

    def compute(begin, count):
        slab = torch.empty(count, dtype=torch.float64, device=where)
        if count:
            lib = _cabi.lib()
            wb = lib.pdc_gls_work_bytes(d_t.numel(), 1, count)
            work = torch.empty(wb, dtype=torch.uint8, device=where)
            _cabi.check(lib.pdc_gls_scan_dev(
                device, torch.cuda.current_stream(device).cuda_stream, d_t.data_ptr(), d_y.data_ptr(),
                None if d_dy is None else d_dy.data_ptr(), None, d_t.numel(), 1, 0, f0, delta, begin,
                count, int(bool(fit_mean)), int(bool(psd)), slab.data_ptr(), None, None,
                work.data_ptr(), wb))
        return slab

    return sharded_scan(compute, nf, group).cpu().numpy()


def sharded_periods(scan, periods, device=None, group=None):
    """Shard a trial-period sweep: ``scan(periods_slab, device) -> ndarray`` is run on this rank's
    contiguous slab of ``periods`` and the slabs are all-gathered.  Used for PDM / StringLength,
    whose per-period work is independent exactly like the frequency grid of the periodogram
    (the reference fans these out with ``Pool.map``, ``phase.py:69-70,185-186``)."""
    import torch
    import torch.distributed as dist
    periods = np.ascontiguousarray(periods, dtype=np.float64)
    if device is None and dist.get_backend(group) != "gloo":
        device = torch.cuda.current_device()

    def compute(begin, count):
        return _place(scan(periods[begin:begin + count], device), device, group)

    return sharded_scan(compute, periods.size, group).cpu().numpy()


def sharded_pdm(t, x, periods, nb=5, nc=2, sigma=None, device=None, group=None):
    """theta for every trial period, period grid sharded over the ranks of ``group``."""
    from periodicity_amd import _cabi
    sigma = np.var(x, ddof=1) if sigma is None else sigma
    return sharded_periods(lambda p, dev: _cabi.pdm_scan(t, x, p, nb, nc, sigma, device=dev),
                           periods, device, group)


def sharded_stringlength(t, m, periods, device=None, group=None):
    """String length for every trial period, period grid sharded over the ranks of ``group``."""
    from periodicity_amd import _cabi
    return sharded_periods(lambda p, dev: _cabi.stringlength_scan(t, m, p, device=dev),
                           periods, device, group)
