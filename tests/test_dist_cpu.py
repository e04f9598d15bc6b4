"""N > 1 path on CPU: two gloo ranks.  The product model cannot run without a GPU, so the data-parallel plumbing
(batch sharding, DDP wrapper, bf16 gradient-compression hook, metric reduction) is exercised with a small torch module;
the property checked is the one SURVEY.md section 8e names: gradients after the all-reduce equal the single-process gradients
of the full global batch."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(24, 48), torch.nn.GELU(), torch.nn.LayerNorm(48), torch.nn.Linear(48, 8))


def _worker(rank, world, port, bf16, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from lemevit_amd import dist as D
    r, l, w = D.init_distributed("gloo")
    assert (r, w) == (rank, world)
    net = D.wrap_ddp(_net(), None, bf16_grads=bf16, bucket_cap_mb=1)
    torch.manual_seed(1)
    x = torch.randn(16, 24); y = torch.randint(0, 8, (16,))
    idx = list(D.shard_batch(16, rank, world))
    loss = torch.nn.functional.cross_entropy(net(x[idx]), y[idx])
    loss.backward()
    m = D.all_reduce_mean(loss.detach())
    if rank == 0:
        torch.save(dict(grads=[p.grad.clone() for p in net.module.parameters()], loss=m), out)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("bf16", [False, True])
def test_ddp_gradients_match_single_process(tmp_path, bf16):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), bf16, out), nprocs=2, join=True)
    got = torch.load(out)
    net = _net()
    torch.manual_seed(1)
    x = torch.randn(16, 24); y = torch.randint(0, 8, (16,))
    loss = torch.nn.functional.cross_entropy(net(x), y)
    loss.backward()
    tol = 2e-2 if bf16 else 1e-6       # bf16 buckets: 8 mantissa bits
    for g, p in zip(got["grads"], net.parameters()):
        assert torch.allclose(g, p.grad, rtol=tol, atol=tol * float(p.grad.abs().max())), (g - p.grad).abs().max()
    assert abs(float(got["loss"]) - float(loss)) < 1e-6


def _sync_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from lemevit_amd import dist as D
    D.init_distributed("gloo")
    torch.manual_seed(100 + rank)
    flat = torch.randn(1000)                                       # this rank's flat block gradients
    rest = [torch.nn.Parameter(torch.zeros(7, 3)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 2))]
    for p in rest[:2]:
        p.grad = torch.randn_like(p)                               # the third parameter has no gradient this step
    mine = dict(flat=flat.clone(), rest=[None if p.grad is None else p.grad.clone() for p in rest])
    sync = D.FlatGradSync(flat, [(0, 300), (300, 640), (640, 1000)], rest)
    sync.chunk_ready(2)                                            # the backward pass finishes the LAST chunk first
    sync.chunk_ready(1)
    sync.finish()                                                  # sends chunk 0 and the remaining parameters, waits, averages
    torch.save(dict(mine=mine, flat=flat, rest=[None if p.grad is None else p.grad for p in rest]), out + f".{rank}")
    torch.distributed.destroy_process_group()


def test_flat_grad_sync_two_ranks(tmp_path):
    """FlatGradSync (the DDP-free exchange used with FlatAdamW): after finish() every rank holds the MEAN of the ranks' flat
    gradients and of the remaining parameters' gradients, whatever order the chunks were released in."""
    out = str(tmp_path / "s.pt")
    mp.spawn(_sync_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    want = (r0["mine"]["flat"] + r1["mine"]["flat"]) / 2
    assert torch.allclose(r0["flat"], want, atol=1e-6) and torch.allclose(r1["flat"], want, atol=1e-6)
    for i in range(2):
        w = (r0["mine"]["rest"][i] + r1["mine"]["rest"][i]) / 2
        assert torch.allclose(r0["rest"][i], w, atol=1e-6) and torch.allclose(r1["rest"][i], w, atol=1e-6)
    assert r0["rest"][2] is None and r1["rest"][2] is None


def _sync2_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from lemevit_amd import dist as D
    D.init_distributed("gloo")
    torch.manual_seed(200 + rank)
    micro = [torch.randn(512), torch.randn(512)]                   # two micro-batches of local gradients
    flat = torch.zeros(512)
    sync = D.FlatGradSync(flat, [(0, 200), (200, 512)], [], compress="bf16")
    with sync.no_sync():                                           # gradient accumulation: nothing may be sent yet
        flat += micro[0]
        sync.chunk_ready(1); sync.chunk_ready(0)
        assert not sync._work
        with pytest.raises(RuntimeError):
            sync.finish()
    flat += micro[1]
    sync.chunk_ready(1)
    sync.finish()
    torch.save(dict(mine=micro[0] + micro[1], flat=flat), out + f".{rank}")
    torch.distributed.destroy_process_group()


def test_flat_grad_sync_no_sync_and_bf16_wire(tmp_path):
    """Gradient accumulation under no_sync() sends nothing; the exchange after it carries the ACCUMULATED gradients, as bf16
    on the wire (compress='bf16'): the result is the mean of the ranks' sums within bf16 rounding of each addend."""
    out = str(tmp_path / "s2.pt")
    mp.spawn(_sync2_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    want = (r0["mine"].bfloat16().float() + r1["mine"].bfloat16().float()) / 2
    assert torch.equal(r0["flat"], r1["flat"])
    assert torch.allclose(r0["flat"], want, atol=2e-2, rtol=1e-2)
    assert (r0["flat"] - (r0["mine"] + r1["mine"]) / 2).abs().max() < 4e-2


def test_shard_batch():
    from lemevit_amd.dist import shard_batch
    assert [list(shard_batch(8, r, 4)) for r in range(4)] == [[0, 1], [2, 3], [4, 5], [6, 7]]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


def _bn_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from lemevit_amd import dist as D
    D.init_distributed("gloo")
    torch.manual_seed(7)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 6, 3, padding=1), torch.nn.BatchNorm2d(6), torch.nn.GELU(), torch.nn.Conv2d(6, 4, 3), torch.nn.BatchNorm2d(4)).train()
    sync = D.FlatGradSync(torch.zeros(8), [(0, 8)], [])
    sync.attach_buffer_broadcast(net, 0)
    seen = []
    net.register_forward_pre_hook(lambda m, a: seen.append({k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}))       # (runs behind the broadcast)
    torch.manual_seed(300 + rank)                                  # every rank normalises ITS shard: the running statistics drift apart between broadcasts
    for _ in range(3):
        net(torch.randn(4, 3, 8, 8) * (1.0 + rank) + rank)
    with torch.no_grad():
        net(torch.randn(4, 3, 8, 8))                               # a no-grad pass is not a training pass: no broadcast
    final = {k: v.clone() for k, v in net.state_dict().items() if "running" in k}
    torch.save(dict(seen=seen, final=final), out + f".{rank}")
    torch.distributed.destroy_process_group()


def test_buffer_broadcast_every_training_forward(tmp_path):
    """The reference's DistributedDataParallel runs with broadcast_buffers=True (main.py:333): rank 0's BatchNorm running statistics and batch counters reach every rank in front
    of EVERY training forward pass.  FlatGradSync.attach_buffer_broadcast does the same: at the start of each of three training passes both ranks hold the same buffers (rank 0's),
    although each rank has updated them from different data in between; and the statistics do drift between the broadcasts (the final buffers differ), i.e. the test sees
    what it claims to."""
    out = str(tmp_path / "bn.pt")
    mp.spawn(_bn_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert len(r0["seen"]) == 4 and len(r1["seen"]) == 4
    for step in range(3):
        for k in r0["seen"][step]:
            assert torch.equal(r0["seen"][step][k], r1["seen"][step][k]), (step, k)
    assert int(r1["seen"][2]["1.num_batches_tracked"]) == 2
    assert any(not torch.equal(r0["final"][k], r1["final"][k]) for k in r0["final"])
    assert any(not torch.equal(r0["seen"][3][k], r1["seen"][3][k]) for k in r0["final"])          # the no-grad pass saw the drifted statistics: no broadcast outside training
