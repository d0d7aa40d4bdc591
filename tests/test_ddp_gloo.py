"""CPU-only, world sizes 2 / 4 / 8 over gloo: the N>1 exchange step (flat gradient bucket all-reduce, parameter/buffer
broadcast, flat-parameter optimizer, sum + ``grad_scale`` folding, the two-bucket overlap with even and uneven splits) behaves
like single-process training on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import spcl_amd  # noqa: F401
    from spcl_amd import ddp
    torch.manual_seed(100 + rank)  # different init per rank -> broadcast must equalise
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 2))
    ddp.broadcast_state(model)
    w0 = torch.cat([p.detach().flatten() for p in model.parameters()])
    flat = ddp.FlatParams(model.parameters())
    assert flat.nbytes == 4 * sum(p.numel() for p in model.parameters())
    # parameters are now views of one tensor; values unchanged
    assert torch.equal(torch.cat([p.detach().flatten() for p in model.parameters()]), w0)
    assert flat.param.data_ptr() == next(model.parameters()).data_ptr()
    opt = torch.optim.SGD([flat.param], lr=0.1)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(4 * world, 6, generator=g), torch.randn(4 * world, 2, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    model.eval()  # BN in eval so that per-rank statistics do not enter the comparison
    flat.zero_grad()
    loss = ((model(xs) - ys) ** 2).mean()
    loss.backward()
    red = flat.reduce().clone()
    opt.step()
    # fold_mean (what the epochers switch on under FusedRAdam): the exchange leaves the ranks' SUM in the bucket and the
    # factor 1 / world in grad_scale, for the optimizer kernel to apply (spcl_radam_step_scaled); sum * scale == the mean
    flat.zero_grad()
    ((model(xs) - ys) ** 2).mean().backward()
    flat.fold_mean = True
    # (valid BEFORE the collective: a captured update graph bakes the factor in without ever running the exchange)
    assert flat.grad_scale == 1.0 / world
    summed = flat.reduce().clone()
    assert flat.grad_scale == 1.0 / world
    flat.zero_grad()
    ((model(xs) - ys) ** 2).mean().backward()
    flat.fold_mean = False
    mean = flat.reduce().clone()
    assert flat.grad_scale == 1.0 and torch.equal(summed * (1.0 / world), mean)  # (world sizes are powers of two: exact)
    for p_ in model.parameters():
        p_.grad = None
    # plain GradBucket gives the same averaged gradients
    m2 = torch.nn.Linear(3, 2)
    ddp.broadcast_state(m2)
    b = ddp.GradBucket(m2.parameters())
    (m2(torch.full((1, 3), float(rank + 1))).sum()).backward()
    b.allreduce()
    q.put((rank, w0.tolist(), red.tolist(), torch.cat([p.detach().flatten() for p in model.parameters()]).tolist(),
           m2.weight.grad.tolist(), ddp.on_master()))  # plain lists: no shared-memory handles to outlive the worker
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_flat_bucket_allreduce_matches_single_process(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res = [(r, torch.tensor(a), torch.tensor(b), torch.tensor(c), torch.tensor(d), m) for r, a, b, c, d, m in res]
    r0, w0a, g0, p0, mg0, master0 = res[0]
    assert master0
    for r1, w0b, g1, p1, mg1, master1 in res[1:]:
        assert torch.equal(w0a, w0b)            # broadcast equalised the initial weights
        assert torch.allclose(g0, g1)           # every rank holds the same averaged gradient
        assert torch.allclose(p0, p1)           # and the same updated parameters
        assert not master1 and torch.allclose(mg0, mg1)
    assert torch.allclose(mg0, torch.full((2, 3), (world + 1) / 2.0))  # mean of 1 .. world
    # single-process reference on the concatenated batch: mean over 4 x world samples == mean of the rank means
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 2)).eval()
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(4 * world, 6, generator=g), torch.randn(4 * world, 2, generator=g)
    ((model(X) - Y) ** 2).mean().backward()
    ref = torch.cat([p.grad.flatten() for p in model.parameters()])
    assert torch.allclose(g0, ref, atol=1e-6)


def test_single_process_paths_are_noops():
    import spcl_amd  # noqa: F401
    from spcl_amd import ddp
    assert not ddp.is_distributed() and ddp.on_master()
    m = torch.nn.Linear(4, 3)
    ddp.broadcast_state(m)
    b = ddp.GradBucket(m.parameters())
    m(torch.ones(2, 4)).sum().backward()
    g = m.weight.grad.clone()
    flat = b.allreduce()
    assert flat.numel() == 15 and torch.equal(m.weight.grad, g) and m.weight.grad.data_ptr() == b.views[0].data_ptr()


def _overlap_worker(rank, world, port, q, early_layer=2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy

    import spcl_amd  # noqa: F401
    from spcl_amd import ddp
    torch.manual_seed(5)
    base = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4), torch.nn.Tanh(),
                               torch.nn.Linear(4, 2))
    g = torch.Generator().manual_seed(11 + rank)
    xs, ys = torch.randn(4, 6, generator=g), torch.randn(4, 2, generator=g)
    out = []
    fired = []
    for overlap in (False, True, "cut"):
        model = copy.deepcopy(base)
        flat = ddp.FlatParams(model.parameters())
        deferred = []
        if overlap == "cut":
            # a step replayed from hipGraphs (stepgraph.StepGraph.cut): during the capture the hook only GATHERS, the start of
            # the collective is handed to the cutter, which runs it between the two compute graphs of every replay
            flat.cutter = deferred.append
        if overlap:
            # members are in module order: the early bucket is the TAIL (what backward finishes first); it starts when the
            # gradient w.r.t. the first layer's output exists, i.e. once everything after that layer is differentiated
            # (early_layer 4: an UNEVEN split -- only the last layer's 10 values go early, the other 59 wait for the end)
            flat.overlap_from(model[early_layer].weight)
            model[early_layer - 2].register_full_backward_pre_hook(
                lambda m, go, flat=flat: (fired.append(flat._early is None), flat.reduce_early())[0] and None)
        for step in range(2):  # two steps: the per-step state is reset
            flat.zero_grad()
            ((model(xs * (step + 1)) - ys) ** 2).mean().backward()
            if overlap == "cut":
                assert flat._early is True and len(deferred) == 1  # gathered, nothing sent yet
                deferred.pop()()
            if overlap:
                assert flat._early is not None  # the hook started the early bucket during backward
                assert model[0].weight.grad is not None and flat._early_off == sum(
                    p.numel() for m in list(model)[:early_layer] for p in m.parameters())
            red = flat.reduce().clone()
            assert flat._early is None and flat.param.grad is flat.flat
            out.append(red.tolist())
    q.put((rank, out, fired))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,early_layer", [(2, 2), (2, 4), (4, 2), (8, 4)])
def test_two_bucket_overlap_equals_the_flat_allreduce(world, early_layer):
    """ddp.FlatParams.overlap_from / reduce_early (projector + Conv5..Conv3 early, Conv2..Conv1 late): the
    tail of the bucket is all-reduced asynchronously from a backward hook, the head after backward; same averaged
    gradients as the one-bucket reduce, bit for bit at world size 2, on every rank and in every step."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q, early_layer)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out, fired in res:
        plain, bucketed, cut = out[:2], out[2:4], out[4:]
        assert bucketed == cut              # the collective started by the cutter instead of the hook: the same two all-reduces
        if world == 2:
            assert plain == bucketed        # step by step, element by element (two addends: one order)
        else:                               # (gloo's ring adds a slice's addends in an order that depends on its length)
            assert torch.allclose(torch.tensor(plain), torch.tensor(bucketed), rtol=1e-6, atol=1e-7)
        assert fired == [True] * 4          # the hook ran once per step and found the early bucket not yet started
    for other in res[1:]:
        assert res[0][1] == other[1]        # every rank holds the same averaged gradients
    assert any(v != 0.0 for v in res[0][1][0])
