"""world_size-2 CPU (gloo) tests of the multi-GPU plumbing: shard plan and the one-shot weight broadcast."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import evc_amd  # noqa: F401
    from evc_amd import dist as D
    r, w, dev = D.init(backend="gloo")
    g = torch.Generator().manual_seed(5)
    sd = None
    if r == 0:
        sd = {"a.weight": torch.randn(7, 3, generator=g), "b.bias": torch.randn(5, generator=g),
              "tab._cdf": torch.arange(12, dtype=torch.int32).reshape(3, 4), "c.weight": torch.randn(2, 2, 2, generator=g)}
    out = D.broadcast_state_dict(sd, src=0, device=dev)
    lo, hi = D.shard_range(46, r, w)
    tot = D.sum_over_ranks(hi - lo, dev)
    mx = D.max_over_ranks(float(r + 1), dev)
    D.barrier()
    q.put((r, {k: (str(v.dtype), v.cpu().numpy()) for k, v in out.items()}, (lo, hi), tot, mx))


def test_broadcast_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(5)
    ref = {"a.weight": torch.randn(7, 3, generator=g), "b.bias": torch.randn(5, generator=g),
           "tab._cdf": torch.arange(12, dtype=torch.int32).reshape(3, 4), "c.weight": torch.randn(2, 2, 2, generator=g)}
    for r, sd, rng, tot, mx in res:
        assert list(sd) == list(ref)
        for k in ref:
            assert sd[k][0] == str(ref[k].dtype) and (sd[k][1] == ref[k].numpy()).all(), (r, k)
        assert tot == 46 and mx == 2.0
    assert res[0][2] == (0, 23) and res[1][2] == (23, 46)


def test_shard_plan_46_clips_over_8():
    sys.path.insert(0, REPO)
    import evc_amd  # noqa: F401
    from evc_amd.dist import shard_range
    sizes = [shard_range(46, r, 8) for r in range(8)]
    assert [b - a for a, b in sizes] == [6, 6, 6, 6, 6, 6, 5, 5]
    assert sizes[0][0] == 0 and sizes[-1][1] == 46 and all(sizes[i][1] == sizes[i + 1][0] for i in range(7))
    assert shard_range(3, 5, 8) == (3, 3)   # more ranks than items: empty shard


def _run_bench(args, env_extra, timeout=240):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` outside torchrun starts 2 ranks (child processes, gloo here) and relays rank 0's line:
    n_gpus and ranks_seen are 2, never a silent 1-rank measurement."""
    rc, line, err = _run_bench(["--gpus", "2", "--plumbing-only"], {"EVC_DIST_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["self_launched"] is True
    assert line["backend"] == "gloo" and line["broadcast_ok"] is True


def test_bench_names_the_stated_multi_gpu_configs_and_their_shards():
    """BASELINE configs[2] (46 clips block-sharded over the ranks) and configs[4] (256 clips, F-PNDM-50) as bench
    workloads: 2 gloo ranks print the shard sizes they would decode and the workload name; the weak-scaling default
    keeps 9 clips on every rank."""
    rc, line, err = _run_bench(["--gpus", "2", "--plumbing-only", "--total-clips", "46"], {"EVC_DIST_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    assert line["per_rank_clips"] == [23, 23] and line["scaling"] == "strong"
    assert line["config"]["workload"].startswith("configs[2]: 46 clips over 2 rank(s) (23,23)")
    rc, line, err = _run_bench(["--gpus", "2", "--plumbing-only", "--total-clips", "255", "--sampler", "FPNDM",
                                "--subsample", "50"], {"EVC_DIST_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    assert line["per_rank_clips"] == [128, 127] and "59 forwards (FPNDM-50)" in line["config"]["workload"]
    rc, line, err = _run_bench(["--gpus", "2", "--plumbing-only"], {"EVC_DIST_BACKEND": "gloo"})
    assert rc == 0 and line["per_rank_clips"] == [9, 9] and line["scaling"] == "weak"
    assert line["config"]["workload"].startswith("configs[1]: 9 clips/GPU")
    # the plan for the stated 8-rank shapes, without starting 8 processes
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import argparse
    a = argparse.Namespace(total_clips=46, clips=9, sampler="DDPM", subsample=100)
    sizes = [bench.rank_clips(a, r, 8)[0] for r in range(8)]
    assert sizes == [6, 6, 6, 6, 6, 6, 5, 5]
    assert bench.rank_clips(a, 0, 8)[1].startswith("configs[2]: 46 clips over 8 rank(s) (6,6,6,6,6,6,5,5)")
    a = argparse.Namespace(total_clips=256, clips=9, sampler="FPNDM", subsample=50)
    assert [bench.rank_clips(a, r, 8)[0] for r in range(8)] == [32] * 8
    assert bench.rank_clips(a, 0, 8)[1].startswith("configs[4]: 256 clips over 8 rank(s)")


def test_a_failed_rank_ends_the_job_non_zero_in_bounded_time():
    """One rank dies before the first collective: the launcher must come back non-zero (torchrun tears the job down; the
    surviving rank's collective has a bounded wait, EVC_DIST_TIMEOUT_S) -- never a hang, never a result line."""
    import time
    t0 = time.time()
    rc, line, err = _run_bench(["--gpus", "2", "--plumbing-only", "--fail-rank", "1"],
                               {"EVC_DIST_BACKEND": "gloo", "EVC_DIST_TIMEOUT_S": "30"}, timeout=200)
    assert rc != 0 and line is None, (rc, line)
    assert time.time() - t0 < 150
    assert "simulated failure" in err


def test_bench_refuses_world_size_mismatch():
    rc, line, err = _run_bench(["--gpus", "2", "--plumbing-only"], {"EVC_DIST_BACKEND": "gloo", "WORLD_SIZE": "1", "RANK": "0"})
    assert rc == 2 and line is None and "WORLD_SIZE=1" in err


def test_bench_parent_launches_before_any_gpu_call(monkeypatch):
    """The self-launching parent must hand over to its children before it touches the GPU (a process that has
    initialised HIP must not spawn-and-wait on this pool): with every torch.cuda entry point booby-trapped, main()
    still reaches self_launch."""
    import importlib.util
    sys.path.insert(0, REPO)
    import evc_amd  # noqa: F401
    from evc_amd import dist as D
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []

    def boom(*a, **k):
        raise AssertionError("GPU touched before the ranks were launched")
    for name in ("is_available", "set_device", "current_stream", "synchronize", "init", "device_count"):
        monkeypatch.setattr(torch.cuda, name, boom)
    monkeypatch.setattr(D, "self_launch", lambda script, argv, n, timeout=None: calls.append((script, list(argv), n)) or 0)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "1"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert calls and calls[0][2] == 4 and calls[0][1] == ["--gpus", "4", "--steps", "1"]
