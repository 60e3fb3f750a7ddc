"""World-size-2 CPU (gloo) test of the per-bucket gradient all-reduce used by the N > 1 step.
Checks what stock DDP gets wrong for DGL: every bucket -- including the fusion head --
is summed across ranks exactly once, and the folded 1/world scale yields the mean."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-gdl_amd"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gdl.ddp import BucketReducer

    n_f, n_a, n_v = 6150, 20000, 30000  # fusion / audio / visual bucket sizes (scaled down)
    offs = [0, n_f, n_f + n_a, n_f + n_a + n_v]
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(offs[-1], generator=g)
    mine = flat.clone()
    red = BucketReducer(flat, {"fusion": (offs[0], offs[1]), "audio": (offs[1], offs[2]), "visual": (offs[2], offs[3])})
    ok = True
    # order of the real step: fusion first, then the two encoders
    red.launch("fusion")
    red.launch("audio")
    try:
        red.launch("audio")
        ok = False  # a bucket must not be reduced twice
    except RuntimeError:
        pass
    try:
        red.wait_all()
        ok = False  # a step that forgot a bucket must fail loudly
    except RuntimeError:
        pass
    red.launch("visual")
    red.wait_all()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    expect = sum(gathered)
    ok = ok and torch.allclose(flat, expect, rtol=1e-6, atol=1e-6)
    ok = ok and abs(red.grad_scale - 1.0 / world) < 1e-12
    mean = flat * red.grad_scale
    ok = ok and torch.allclose(mean, expect / world, rtol=1e-6, atol=1e-6)
    # buffers follow rank 0
    buf = torch.full((5,), float(rank))
    red.broadcast_buffers([buf])
    ok = ok and bool((buf == 0).all())
    q.put((rank, ok, float(flat.double().sum())))
    dist.destroy_process_group()


def test_bucket_allreduce_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    sums = [s for _, _, s in res]
    np.testing.assert_allclose(sums[0], sums[1], rtol=1e-9)  # both ranks hold identical reduced grads


def _worker5(rank, world, port, q):
    """The five buckets of DGLTrainer in the issue orders of its data-parallel step (trainer.py: audio_l4, visual_l4,
    audio_rest, fusion, visual_rest with the early backward; fusion, audio_l4, visual_l4, audio_rest, visual_rest without),
    ranks seeded differently, replica state synchronised from rank 0 first."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gdl.ddp import BucketReducer

    sizes = {"fusion": 615, "audio_rest": 2780, "audio_l4": 8390, "visual_rest": 2790, "visual_l4": 8390}  # real sizes / 1000
    layout = ["fusion", "audio_rest", "audio_l4", "visual_rest", "visual_l4"]  # arena order: head | audio (rest, l4) | visual
    offs, o = {}, 0
    for n in layout:
        offs[n] = (o, o + sizes[n])
        o += sizes[n]
    g = torch.Generator().manual_seed(7 + 13 * rank)  # every rank its own "initialisation" and gradients
    params, momentum, bn = torch.randn(o, generator=g), torch.randn(o, generator=g), torch.randn(40, generator=g)
    flat = torch.randn(o, generator=g)
    mine = flat.clone()
    red = BucketReducer(flat, offs)
    ok = True
    red.sync_state([params, momentum, bn])
    ref = [torch.empty_like(params) for _ in range(world)]
    dist.all_gather(ref, params)
    ok = ok and all(torch.equal(r, ref[0]) for r in ref)  # every replica starts from rank 0's parameters
    g0 = torch.Generator().manual_seed(7)
    ok = ok and torch.equal(params, torch.randn(o, generator=g0))
    for step in range(2):
        order = ("audio_l4", "visual_l4", "audio_rest", "fusion", "visual_rest") if step == 0 else \
            ("fusion", "audio_l4", "visual_l4", "audio_rest", "visual_rest")
        for n in order:
            red.launch(n)
        red.wait_all()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        expect = sum(gathered)
        ok = ok and torch.allclose(flat, expect, rtol=1e-5, atol=1e-5)
        mine = flat.clone()  # next step reduces the reduced values again: sums grow by `world`
    # timing helper runs (gloo): one number per bucket
    t = red.time_buckets(lambda: None, reps=1)
    ok = ok and set(t) == set(layout) and all(v >= 0 for v in t.values())
    # a disabled reducer keeps the bookkeeping and moves nothing
    red.enabled = False
    before = flat.clone()
    for n in layout:
        red.launch(n)
    red.wait_all()
    ok = ok and torch.equal(before, flat)
    q.put((rank, ok, float(flat.double().sum())))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_five_buckets_world(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker5, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    sums = [s for _, _, s in res]
    np.testing.assert_allclose(sums, sums[0], rtol=1e-9)


def test_bench_plain_multi_gpu_invocation_self_launches():
    """`python bench.py --gpus 2` typed without a launcher must start torch.distributed.run itself, as a
    child process and before any GPU call -- not exit with a usage error.  On this GPU-less container the two ranks it starts
    each stop at "no MI355X visible" (the product has no CPU path): that message, twice, and the parent's non-zero exit code
    show that the launch happened with the right rank environment; on a GPU box tests/test_ddp_gpu.py runs the same form to
    its JSON line."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        pytest.skip("GPU box: tests/test_ddp_gpu.py::test_bench_two_rank_control_flow[plain] covers the full run")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and "launch with" not in r.stderr, r.stderr[-1500:]
    assert r.stderr.count("no MI355X visible") >= 2, r.stderr[-1500:]
