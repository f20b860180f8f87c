"""Stream sharding across GPUs: input streams are independent, so rank r = GPU r owns
whole streams and there is no collective on the data path. The only communication is the
bookkeeping below: which streams a GPU gets, and the whole-job throughput (bytes of all ranks over
the slowest rank's time), exchanged with one all_reduce each on whatever backend the process group
uses (RCCL on the GPUs, gloo in the CPU tests)."""


def assign_streams(sizes, n_gpus):
    """Longest-first greedy assignment of streams (files) to GPUs; returns a list of index lists.
    Config 4 (Silesia, 12 files over 8 GPUs) uses this; config 5 (one shard per GPU) is the identity."""
    loads = [0] * n_gpus
    out = [[] for _ in range(n_gpus)]
    for i in sorted(range(len(sizes)), key=lambda i: (-sizes[i], i)):
        g = min(range(n_gpus), key=lambda g: (loads[g], g))
        out[g].append(i)
        loads[g] += sizes[i]
    return out


def shard_seed(rank, stream=0, streams_per_gpu=1):
    """Seed of the synthetic S-enwik8 shard a (rank, local stream) pair processes."""
    return 1000 + rank * streams_per_gpu + stream


def aggregate_throughput(local_bytes, local_seconds, device=None):
    """(total bytes of all ranks, max seconds over ranks, bytes/s). Works without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_bytes, local_seconds, local_bytes / local_seconds
    t = torch.tensor([float(local_seconds)], dtype=torch.float64, device=device)
    b = torch.tensor([float(local_bytes)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(b, op=dist.ReduceOp.SUM)
    return b.item(), t.item(), b.item() / t.item()
