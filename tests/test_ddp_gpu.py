"""Two ranks sharing the one GPU of the test box (gloo transport, CUDA tensors): the engine's
backward-overlapped, bucketed gradient all-reduce + fused Adam must equal the hand-computed
data-parallel step  theta' = Adam(theta, (g_rank0 + g_rank1) / 2)  with per-rank BatchNorm
statistics (the reference does not synchronise BatchNorm, monai_unet.py:529-538)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CH, ST, K, SIZE, B = (16, 32, 64), (2, 2), 16, 32, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_net():
    from oracle.unet_ref import RefUNet, deterministic_fill_
    from segmantic_amd.seg.monai_unet import Net
    ref = deterministic_fill_(RefUNet(3, 1, K, CH, ST), 0)
    net = Net(num_classes=K, channels=CH, strides=ST)
    net.load_state_dict({"_model." + k: v.clone() for k, v in ref.state_dict().items()})
    return net.to("cuda:0").train()


def _batch(rank):
    from oracle.unet_ref import synthetic_batch
    img, lab = synthetic_batch(B, SIZE, K, seed=10 + rank)
    return {"image": img.to("cuda:0"), "label": lab.to("cuda:0")}


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    from segmantic_amd.seg.distributed import init_distributed
    init_distributed(backend="gloo")
    net = _make_net()
    gs = net.enable_grad_sync(bucket_bytes=64 << 10)      # many buckets on this small net
    assert gs.world == 2
    net.training_step(_batch(rank))
    net.training_step(_batch(rank))                        # second step: moments / versions
    torch.cuda.synchronize()
    torch.save(net._engine.flat.detach().cpu(), os.path.join(out_dir, f"flat_{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_step_equals_manual_gradient_average(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    f0 = torch.load(tmp_path / "flat_0.pt")
    f1 = torch.load(tmp_path / "flat_1.pt")
    assert torch.equal(f0, f1)                             # replicas stay bit-identical
    # manual emulation in one process
    from segmantic_amd import ops
    nets = [_make_net(), _make_net()]
    master = nets[0]
    opt = master.optimizers()
    for step in range(2):
        grads = []
        for r, net in enumerate(nets):
            eng = net._engine_for()
            if net is not master:                          # same weights on both replicas
                eng.flat.copy_(master._engine.flat)
                eng.bump()
            b = _batch(r)
            logits = eng.forward(b["image"], train=True)
            from segmantic_amd.seg.losses import dice_backward, dice_forward
            st = net.loss_function._state
            dice_forward(st, logits, b["label"], 1e-5, 1e-5)
            eng.backward(dice_backward(st, logits))
            grads.append(eng.flat_grad.clone())
        master._engine.flat_grad.copy_(grads[0] + grads[1])
        opt.step(0.5)
        master._engine.bump()
    torch.cuda.synchronize()
    ref = master._engine.flat.detach().cpu()
    # identical arithmetic (sum of two f32 gradients, scale 1/2 inside Adam) -> bit exact
    assert torch.equal(f0, ref), float((f0 - ref).abs().max())
