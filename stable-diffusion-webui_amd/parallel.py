"""Multi-GPU data parallelism for the hot path: one process per GPU, images sharded, no per-step communication.

The reference is single-device; images of a batch are independent units — own generator
(modules/rng.py:108), own cond row, per-image CFG combine (modules/sd_samplers_cfg_denoiser.py:78-80) and decode
(modules/processing.py:631) — so the job [0, batch_size*n_iter) is split contiguously across ranks and each rank's images
are bit-identical to the single-GPU result (seeds are ``seed + global_index``).  Collectives (RCCL over xGMI via
``torch.distributed`` backend "nccl"; "gloo" on CPU for the tests):
  * once per checkpoint: weights packed into one blob on rank 0 and sent scatter + all-gather, so all 7 xGMI links of every
    GPU carry 1/W of the blob instead of one link carrying all of it (a ring/chain broadcast is bound by one 153 GB/s link);
  * once per job: gather of the uint8 images to rank 0.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Join the process group the launcher (torch.distributed.run) described in the environment."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:                                   # SDMI_DIST_BACKEND=gloo: several ranks on ONE GPU (the single-GPU test of the N > 1 bench path)
            backend = os.environ.get("SDMI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of [0, n_items); the first (n_items % world) ranks take one extra item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


COLLECTIVES = {"weights": 0, "job": 0}                       # calls issued by this process (bench.py prints them per job)


def blob_algorithm(backend: str, nbytes: int, requested: str = "scatter_allgather") -> str:
    """Which collective carries a weight blob — decided UP FRONT from the backend and the size, identically on every rank (never by
    catching an exception on some ranks, which would leave the others inside a different collective): scatter + all-gather where the
    backend has ``scatter`` (nccl = RCCL, gloo), a plain broadcast elsewhere, for small blobs, or on request."""
    if requested == "broadcast" or nbytes < (1 << 20) or backend not in ("nccl", "gloo"):
        return "broadcast"
    return "scatter_allgather"


def broadcast_blob(blob: torch.Tensor, src: int = 0, algo: str = "scatter_allgather") -> torch.Tensor:
    """Broadcast a flat uint8 tensor (same length on every rank; contents valid on ``src``)."""
    world = dist.get_world_size()
    if world == 1:
        return blob
    n = blob.numel()
    if blob_algorithm(dist.get_backend(), n, algo) == "broadcast":
        dist.broadcast(blob, src=src)
        COLLECTIVES["weights"] += 1
        return blob
    assert n % world == 0, "pad the blob to a multiple of the world size"
    shard = n // world
    views = list(blob.view(world, shard).unbind(0))
    mine = torch.empty(shard, dtype=blob.dtype, device=blob.device)
    dist.scatter(mine, scatter_list=[v.contiguous() for v in views] if dist.get_rank() == src else None, src=src)
    dist.all_gather(views, mine)          # each rank pulls the other W-1 shards from their owners (all links busy)
    COLLECTIVES["weights"] += 2
    return blob


def broadcast_state_dict(sd: Optional[dict], src: int = 0, device="cpu", algo: str = "scatter_allgather") -> dict:
    """Rank ``src`` passes the checkpoint dict; every rank returns an identical dict whose tensors are views into one
    device blob (fp16 tensors stay fp16: ~2.1 GB for SD1.5)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return {k: v.to(device) for k, v in sd.items()}
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]
    dist.broadcast_object_list(meta, src=src)
    entries = meta[0]
    offs, total = [], 0
    for _, shape, dt in entries:
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=getattr(torch, dt)).element_size()
        total = _pad_to(total, 256)
        offs.append((total, nbytes))
        total += nbytes
    total = _pad_to(total, 256 * world)
    blob = torch.empty(total, dtype=torch.uint8, device=device)
    if rank == src:
        for (k, shape, dt), (o, nb) in zip(entries, offs):
            blob[o:o + nb].copy_(sd[k].contiguous().view(-1).view(torch.uint8).to(device))
    broadcast_blob(blob, src=src, algo=algo)
    out = {}
    for (k, shape, dt), (o, nb) in zip(entries, offs):
        out[k] = blob[o:o + nb].view(getattr(torch, dt)).view(shape)
    return out


def gather_to_rank0(t: torch.Tensor, counts: List[int]) -> Optional[torch.Tensor]:
    """Concatenate per-rank tensors (rank r contributes counts[r] leading rows) on rank 0.  The collective is chosen UP FRONT from the
    backend, identically on every rank (gather on nccl = RCCL and gloo; all-gather-and-drop elsewhere) — never by catching an exception
    on some ranks, which would leave the others in a different collective."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return t
    rank = dist.get_rank()
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    if dist.get_backend() in ("nccl", "gloo"):
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
        dist.gather(pad, gather_list=bufs, dst=0)
    else:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
    COLLECTIVES["job"] += 1
    if rank != 0:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)])


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------------------------------------
# sharded job entry: split all_seeds = [0, batch_size * n_iter) across ranks, gather the uint8 images
# ------------------------------------------------------------------------------------------------------------------------------
_PER_IMAGE_FIELDS = ("c", "uc", "y", "uy", "hr_c", "hr_uc", "refiner_c", "refiner_uc", "refiner_y", "refiner_uy", "init_images")


def _slice_rows(v, lo, hi, n_total):
    """Rows [lo, hi) of a per-image field: tensors / lists with a leading dimension of n_total, or the prompt_parser containers."""
    if v is None:
        return None
    from . import prompt_parser
    if prompt_parser.is_multicond(v):
        return prompt_parser.MulticondLearnedConditioning((hi - lo,), v.batch[lo:hi])
    if isinstance(v, (list, tuple)):
        return list(v[lo:hi]) if len(v) == n_total else v
    if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n_total:
        return v[lo:hi]
    return v


def shard_job(p, world: int, rank: int):
    """The sub-job of rank ``rank``: a shallow copy of ``p`` holding images [lo, hi) of the global job with their global seeds
    (``seed + global_index``, modules/processing.py:901-909) — so a rank's images are the single-process images (own generator per
    image, modules/rng.py:108; per-image CFG combine, modules/sd_samplers_cfg_denoiser.py:78-80).  The per-call batch size is kept
    at ``p.batch_size`` (bit-identity with a single process running the same batch size holds call by call); a rank with fewer
    than batch_size images left runs one smaller last batch, exactly like the reference's last iteration would."""
    import copy
    n_total = p.batch_size * p.n_iter
    lo, hi = shard_range(n_total, world, rank)
    seed = 1000 if p.seed is None or isinstance(p.seed, (list, tuple)) or p.seed == -1 else int(p.seed)
    all_seeds = list(p.seed) if isinstance(p.seed, (list, tuple)) else [seed + (i if p.subseed_strength == 0 else 0) for i in range(n_total)]
    subseed = 2000 if p.subseed is None or isinstance(p.subseed, (list, tuple)) or p.subseed == -1 else int(p.subseed)
    all_subseeds = list(p.subseed) if isinstance(p.subseed, (list, tuple)) else [subseed + i for i in range(n_total)]
    q = copy.copy(p)
    count = hi - lo
    for f in _PER_IMAGE_FIELDS:
        if hasattr(q, f):
            setattr(q, f, _slice_rows(getattr(p, f), lo, hi, n_total))
    for f in ("latent_mask",):
        if hasattr(q, f) and torch.is_tensor(getattr(p, f)) and getattr(p, f).shape[0] == n_total and n_total > 1:
            setattr(q, f, getattr(p, f)[lo:hi])
    q.seed, q.subseed = all_seeds[lo:hi], all_subseeds[lo:hi]
    q.batch_size = max(1, min(p.batch_size, count))
    q.n_iter = (count + q.batch_size - 1) // q.batch_size if count else 0
    q.extra_generation_params = dict(p.extra_generation_params)
    return q, lo, hi, all_seeds


def process_images_sharded(p, runner=None, world: Optional[int] = None, rank: Optional[int] = None, gather: bool = True):
    """``process_images`` for a job sharded by independent images over the ranks of the default process group (one process per
    GPU).  No per-step communication: each rank runs its contiguous slice of the job, then ONE gather of the uint8 images to
    rank 0 (RCCL gather of [count, H, W, 3] bytes).  Returns the ``Processed`` of the whole job on rank 0 and of the rank's own
    slice elsewhere.  ``runner`` defaults to processing.process_images; ``world`` / ``rank`` override the process group (used to
    replay one rank's slice in a single process; implies no gather)."""
    from . import processing
    runner = runner or processing.process_images_one_device
    explicit = world is not None
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
    q, lo, hi, all_seeds = shard_job(p, world, rank)
    n_total = p.batch_size * p.n_iter
    if not explicit and world > 1 and gather and rank != 0 and dist.get_backend() == "nccl":
        q.images_to_host = False                              # these images leave the GPU through the gather only
    if hi > lo:
        if q.n_iter * q.batch_size != hi - lo:                # ragged tail: full batches first, then the remainder as its own job
            full = (hi - lo) // q.batch_size * q.batch_size
            parts = []
            for a, b in ((0, full), (full, hi - lo)):
                if b > a:
                    parts.append(runner(_with_rows(q, a, b, hi - lo)))
            res = parts[0]
            for extra in parts[1:]:
                res.images = list(res.images) + list(extra.images)
                if res.latents is not None and extra.latents is not None:
                    res.latents = torch.cat([res.latents, extra.latents])
                both = res.images_device is not None and extra.images_device is not None
                res.images_device = torch.cat([res.images_device, extra.images_device]) if both else None
                for f in ("all_seeds", "all_subseeds"):
                    if isinstance(getattr(res, f, None), list) and isinstance(getattr(extra, f, None), list):
                        setattr(res, f, getattr(res, f) + getattr(extra, f))
        else:
            res = runner(q)
    else:
        res = processing.Processed(p, [], all_seeds[0] if all_seeds else -1, [], None)
    res.shard = (lo, hi)
    if explicit or world == 1 or not gather:
        return res
    counts = [shard_range(n_total, world, r)[1] - shard_range(n_total, world, r)[0] for r in range(world)]
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    dev_images = getattr(res, "images_device", None)          # what sdmi_image_to_u8 wrote, still on the GPU (process_images keeps it)
    if on_gpu and dev_images is not None and dev_images.shape[0] == hi - lo:
        mine = dev_images                                     # RCCL gathers straight from that buffer: no host bounce on the way out
    elif res.images:
        mine = torch.from_numpy(np.stack(res.images)).to(dev)
    else:
        # a rank WITHOUT images (fewer images than ranks) does not know the size of the job's final images (a hires job's are not
        # p.height x p.width): rank 0, which owns at least as many images as any other rank, tells it.  ``counts`` is the same list on
        # every rank, so all of them take this branch or none does; a job with an image on every rank needs no such exchange.
        mine = None
    if min(counts) == 0:
        hw = torch.tensor(list(res.images[0].shape[:2]) if res.images else [0, 0], dtype=torch.int64, device=dev)
        dist.broadcast(hw, src=0)
        COLLECTIVES["job"] += 1
        if mine is None:
            mine = torch.zeros((0, int(hw[0]), int(hw[1]), 3), dtype=torch.uint8, device=dev)
    elif mine is None:
        #  this rank owns images but has none to send — an interrupted / skipped job left its device buffer short while
        # images_to_host was off.  It must still enter the gather with a tensor of the job's image size, or the other ranks wait in
        # dist.gather forever; the size is the job's own (hires target if there is one), known locally: no extra collective.
        h = int(getattr(q, "hr_upscale_to_y", 0) or getattr(q, "height", 0))
        w = int(getattr(q, "hr_upscale_to_x", 0) or getattr(q, "width", 0))
        mine = torch.zeros((0, h, w, 3), dtype=torch.uint8, device=dev)
    allv = gather_to_rank0(mine, counts)
    if dist.get_rank() == 0:
        res.images = list(allv.cpu().numpy())
        res.all_seeds = all_seeds
        res.shard = (0, n_total)
    return res


def _with_rows(q, a, b, n):
    """Copy of sub-job ``q`` (n images) restricted to its rows [a, b), as a batch_size x n_iter job of its own."""
    import copy
    r = copy.copy(q)
    for f in _PER_IMAGE_FIELDS:
        if hasattr(r, f):
            setattr(r, f, _slice_rows(getattr(q, f), a, b, n))
    r.seed, r.subseed = list(q.seed[a:b]), list(q.subseed[a:b])
    r.batch_size = min(q.batch_size, b - a)
    r.n_iter = (b - a) // r.batch_size
    return r


# ------------------------------------------------------------------------------------------------------------------------------
# ONE process, N devices ("one process driving 8 devices with one host thread per device ... the webui front-end
# stays a single process behind queue_lock", modules/call_queue.py:8-13; the reference itself selects ONE device, modules/cmd_args.py:106,
# modules/devices.py:35-44).  The torchrun path above needs one process per GPU, which a webui is not: here a pool keeps one engine (a
# replica of the checkpoint) and one worker thread per device inside the webui's own process, and a job is cut exactly as shard_job cuts
# it for ranks — contiguous image ranges, seeds seed + global index, per-call batch size kept — so the result is the single-device job's,
# image for image and bit for bit (same per-call batch size).  No collective: a replica is packed from the checkpoint dict the first
# model holds (SdModel._checkpoint; device-to-device copies when it already lives on a GPU), the images come back as host arrays.
# ------------------------------------------------------------------------------------------------------------------------------
_MODEL_FIELDS = ("sd_model", "hr_sd_model", "refiner_sd_model")


class DevicePool:
    """Engines of one checkpoint on several devices of this process, and the job splitter over them.

        pool = DevicePool(model, devices=[0, 1, 2, 3])        # replicas packed once per checkpoint
        res = pool.process_images(p)                          # p.sd_model is ``model``; Processed of the whole job

    ``devices`` may name a device twice (two engines on one GPU: how the single-GPU test box exercises the threads).  ``serial=True`` runs
    the workers one after the other in the calling thread (the host-emulated CPU tier, whose "device" is not re-entrant)."""

    def __init__(self, model, devices, serial: bool = False):
        assert len(devices) >= 1
        self.devices = [int(d) for d in devices]
        self.serial = bool(serial)
        self._replicas = {}                                   # (id(source model), slot) -> replica
        self._sources = {}
        self.primary = model
        for slot in range(len(self.devices)):
            self.replica(model, slot)

    def replica(self, model, slot: int):
        """The engine-side twin of ``model`` for worker ``slot``: the model itself where its device matches (slot 0 keeps the primary),
        otherwise a second SdModel packed from the same checkpoint dict on that worker's device, with the first model's per-model
        settings (accuracy mode, range-extended VAE, external VAE) carried over."""
        key = (id(model), slot)
        if key in self._replicas:
            return self._replicas[key]
        dev = self.devices[slot]
        first_slot_of_dev = self.devices.index(dev)
        if model.engine.device == dev and first_slot_of_dev == slot:
            rep = model
        else:
            from . import sd_models, shared
            keep = shared.sd_model
            try:
                rep = sd_models.SdModel(model._checkpoint, model.unet_cfg, model.vae_cfg, device=dev, load_vae=model.has_vae,
                                        vae_decoder_only=model._vae_decoder_only, parameterization=model.parameterization,
                                        cond_stage_key=model.cond_stage_key, conditioning_key=model.model.conditioning_key,
                                        embedder=model.embedder, noise_augmentor=model.noise_augmentor, depth_model=model.depth_model)
            finally:
                shared.sd_model = keep                        # the constructor publishes itself as the reference's global: the primary stays
            rep.alphas_cumprod = model.alphas_cumprod.clone()
            if getattr(model, "accuracy_mode", False):
                rep.set_accuracy_mode(True)
            if model.vae_range_extended:
                rep.set_vae_range_extended(True)
        self._replicas[key] = rep
        self._sources[id(model)] = model
        return rep

    def for_each_model(self, fn, model=None):
        """``fn(replica)`` for every worker's twin of ``model`` (default: the pool's first model), one after the other in the calling
        thread — how per-model state that lives outside the checkpoint reaches the replicas: ``pool.for_each_model(lambda m:
        networks.load_networks(m, names, state_dicts, ...))`` merges a LoRA into every device's engine (the loader keeps module-level
        state, so this is not something to do from the workers)."""
        model = model or self.primary
        done = []
        for slot in range(len(self.devices)):
            rep = self.replica(model, slot)
            if not any(rep is d for d in done):
                fn(rep)
                done.append(rep)
        return done

    def close(self):
        for (mid, slot), rep in list(self._replicas.items()):
            if rep is not self._sources.get(mid):
                rep.engine.close()
        self._replicas.clear()

    # ------------------------------------------------------------------------------------------------------
    def _job_for(self, p, slot: int):
        """``p`` as worker ``slot`` sees it: its models replaced by that worker's replicas, its per-image tensors on that device."""
        import copy
        q = copy.copy(p)
        dev = torch.device("cuda", self.devices[slot])
        for f in _MODEL_FIELDS:
            m = getattr(p, f, None)
            if m is not None:
                setattr(q, f, self.replica(m, slot))
        for f in _PER_IMAGE_FIELDS + ("latent_mask", "image_mask", "init_latent", "firstpass_image"):
            v = getattr(p, f, None)
            if torch.is_tensor(v):
                setattr(q, f, v.to(dev))
        q.images_to_host = True
        return q

    def process_images(self, p, runner=None):
        """The whole job ``p`` over the pool's devices; returns the ``Processed`` a single-device ``process_images(p)`` would return
        (images in job order as host arrays, ``all_seeds`` of the whole job, latents concatenated on the primary's device when kept)."""
        import threading
        n = len(self.devices)
        n_total = p.batch_size * p.n_iter
        results, errors = [None] * n, [None] * n

        # the workers' jobs — and with them any replica a hires / refiner checkpoint still needs — are built here, one after the other in
        # the calling thread, before a worker starts
        jobs = []
        for slot in range(n):
            if torch.cuda.is_available():
                torch.cuda.set_device(self.devices[slot])
            jobs.append(self._job_for(p, slot))
        if torch.cuda.is_available():
            torch.cuda.set_device(self.primary.engine.device)

        def work(slot):
            try:
                dev = self.devices[slot]
                if torch.cuda.is_available():
                    torch.cuda.set_device(dev)                # torch's current device is per thread
                q = jobs[slot]
                results[slot] = process_images_sharded(q, runner=runner, world=n, rank=slot)
                if torch.cuda.is_available():
                    torch.cuda.synchronize(dev)
            except BaseException as ex:                       # re-raised in the caller's thread
                errors[slot] = ex

        if self.serial or n == 1:
            for slot in range(n):
                work(slot)
        else:
            threads = [threading.Thread(target=work, args=(slot,), name=f"sdmi-device-{self.devices[slot]}-{slot}") for slot in range(n)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        for ex in errors:
            if ex is not None:
                raise ex
        if torch.cuda.is_available():
            torch.cuda.set_device(self.primary.engine.device)
        res = results[0]
        lat_dev = self.primary.device
        for r in results[1:]:
            res.images = list(res.images) + list(r.images)
            if res.latents is not None and r.latents is not None:
                res.latents = torch.cat([res.latents.to(lat_dev), r.latents.to(lat_dev)])
            elif r.latents is not None and not res.images:
                res.latents = r.latents.to(lat_dev)
            for f in ("infotexts",):
                if isinstance(getattr(res, f, None), list) and isinstance(getattr(r, f, None), list):
                    setattr(res, f, getattr(res, f) + getattr(r, f))
        res.images_device = None
        whole, _, _, all_seeds = shard_job(p, 1, 0)
        res.all_seeds = all_seeds
        res.all_subseeds = list(whole.subseed)
        res.batch_size = p.batch_size
        res.shard = (0, n_total)
        res.devices = list(self.devices)
        return res


_POOLS = {}


def process_images_devices(p, devices, runner=None, serial: bool = False):
    """``process_images`` over several devices of THIS process (one module-level pool per device list: the replicas are packed once
    per checkpoint and released when a job arrives with another one).  The extension calls this when ``opts.mi355x_devices`` names more than
    one device (extension/scripts/mi355x_engine.py); ``bench.py`` keeps the one-process-per-GPU path above."""
    key = (tuple(int(d) for d in devices), bool(serial))
    pool = _POOLS.get(key)
    if pool is not None and pool.primary is not p.sd_model:
        # another checkpoint: the old pool's replicas (one packed model per extra device) are released before the new ones are packed —
        # a pool holds its first model, so a cached pool per checkpoint would keep every checkpoint ever loaded resident on every device
        pool.close()
        pool = None
    if pool is None:
        pool = DevicePool(p.sd_model, devices, serial=serial)
        _POOLS[key] = pool
    return pool.process_images(p, runner=runner)
