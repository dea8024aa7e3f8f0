"""One process per GPU; clips are the data-parallel unit.

The reference has no distributed code at all (single process, CUDA_VISIBLE_DEVICES="0",
city_sender.py:39); its outer loop over videos (city_sender.py:495) is embarrassingly parallel: nothing is
exchanged inside a clip.  So the only collective on the path is ONE broadcast of the weight arena from rank 0
at start-up (RCCL over xGMI: backend "nccl" on ROCm) and an optional gather of per-clip results.
"""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def needs_self_launch(n_ranks):
    """True when a script was asked for ``n_ranks`` > 1 but runs outside a torchrun environment."""
    return n_ranks > 1 and "WORLD_SIZE" not in os.environ


def self_launch(script, argv, n_ranks, timeout=None):
    """Start ``n_ranks`` fresh ranks of ``script argv`` (``python -m torch.distributed.run``, one process per GPU,
    rendezvous on 127.0.0.1 and a free port) as CHILD processes and return the launcher's exit code.

    Must be called before the calling process makes any HIP call: the parent never touches the GPU (on this pool a
    process that has initialised the GPU must not exec or be replaced), it only relays the children's output --
    rank 0 prints the result line -- and their exit status (non-zero if any rank failed)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: required by RCCL on this driver
    env["EVC_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


def init(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); returns
    (rank, world_size, device).  Single-process runs need no environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = os.environ.get("EVC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    elif torch.cuda.is_available() and os.environ.get("EVC_DIST_SHARE_GPU") == "1":
        # rehearsal on a 1-GPU box: several ranks share the visible GPUs, collectives go over gloo
        n = torch.cuda.device_count()
        torch.cuda.set_device(local % n)
        device = torch.device("cuda", local % n)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {"device_id": device} if backend == "nccl" else {}
        # A rank that dies (or never arrives) must not leave the others waiting in a collective for ever: every collective
        # gets a bounded wait (EVC_DIST_TIMEOUT_S, default 15 min -- longer than any timed region of the benchmark), after
        # which the survivors raise; torchrun then tears the job down and exits non-zero.
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(os.environ.get("EVC_DIST_TIMEOUT_S", "900")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def shard_range(n_items, rank, world):
    """Contiguous block partition, remainder to the first ranks (46 clips over 8 -> 6,6,6,6,6,6,5,5)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_state_dict(sd, src=0, device=None, world=None):
    """Rank ``src`` passes a state dict, the others pass None; every rank returns the full dict.
    Metadata goes as one object broadcast; all float32 payload as ONE flat buffer (a single large
    collective suits point-to-point xGMI links better than 446 small ones), other dtypes per tensor."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return sd
    rank = dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    meta = [None]
    if rank == src:
        meta[0] = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    f32 = [(k, s) for k, s, dt in meta if dt == "float32"]
    total = sum(int(torch.Size(s).numel()) for _, s in f32)
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if rank == src:
        off = 0
        for k, s in f32:
            n = int(torch.Size(s).numel())
            flat[off:off + n] = sd[k].reshape(-1).to(device)
            off += n
    dist.broadcast(flat, src=src)
    out, off = {}, 0
    for k, s, dt in meta:
        if dt == "float32":
            n = int(torch.Size(s).numel())
            out[k] = flat[off:off + n].view(s)
            off += n
        else:
            t = sd[k].to(device) if rank == src else torch.empty(s, dtype=getattr(torch, dt), device=device)
            dist.broadcast(t, src=src)
            out[k] = t
    return out


def backend_name():
    """"nccl" is RCCL on ROCm (collectives over xGMI); "gloo" in CPU rehearsals; "none" for a single process."""
    if not dist.is_initialized():
        return "none"
    b = dist.get_backend()
    return "nccl (RCCL)" if b == "nccl" else str(b)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device):
    """Every rank's value, in rank order, on every rank (one small all_gather)."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def sum_over_ranks(value, device):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
