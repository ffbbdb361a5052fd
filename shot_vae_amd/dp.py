"""Data parallelism, MI355X-native: one process per GPU, each with a full replica and its shard of the
labelled + unlabelled minibatch; ONE RCCL all-reduce (sum) of the flat fp32 gradient buffer per step
over xGMI, the 1/world scaling folded into the SGD kernel.

Replaces the 11 nn.DataParallel wrappers of the reference (wideresnet.py:78-93, vae.py:108-132,
decoder.py:63-64): no per-forward parameter broadcast, no activation scatter/gather.  Same semantics
where it matters: per-replica BatchNorm statistics, global-batch gradient (equal shards: mean of
per-rank gradients whose losses use the local batch size).  Deviation: mixup / label-smoothing pairs
are drawn inside the shard."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun-style env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard(t, rank, world):
    """Rows [rank*B/world, (rank+1)*B/world) of a batch (SURVEY.md §8e partitioning)."""
    B = t.shape[0]
    per = B // world
    return t[rank * per: (rank + 1) * per]


def broadcast_parameters(model, src=0):
    """Rank `src`'s parameters and BatchNorm buffers to every rank (start-up / after loading a checkpoint on one rank).
    Takes the MODEL: the in-place c10d write does not bump any version counter, so the engine is told explicitly that
    its packed weight shadows are stale."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        eng = model._engine
        dist.broadcast(eng.param, src)
        dist.broadcast(eng.bufs, src)
        eng.mark_dirty()


def all_reduce_gradients(flat_grad):
    """The single collective of a step.  Returns the scale the optimizer must apply (1/world)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return 1.0 / dist.get_world_size()
    return 1.0


def all_reduce_module_gradients(module, average=True):
    """Data parallelism for modules whose parameters are ordinary torch tensors (the smooth-ELBO SmoothVAE, config 5):
    the .grad tensors are flattened into ONE bucket, all-reduced once (RCCL over xGMI) and scattered back, averaged over
    the ranks -- the same single-collective scheme as the flat-buffer path.  Returns the number of elements reduced."""
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return 0
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return sum(g.numel() for g in grads)
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return off
