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


def _fit_worker(rank, world, port, datalist, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0", SEGMI_DIST_BACKEND="gloo")
    import warnings

    from segmantic_amd.seg.monai_unet import train
    torch.manual_seed(7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = train(datalist=datalist, output_dir=out_dir, spatial_size=[16, 16, 16], channels=CH, strides=ST,
                    max_epochs=2, mixed_precision=True, num_samples=2, gpu_ids=[0, 0])
    torch.cuda.synchronize()
    torch.save({"flat": net._engine.flat.detach().cpu(), "best": net.best_val_dice},
               os.path.join(str(out_dir), f"final_rank{rank}.pt"))
    if dist.is_initialized():
        dist.destroy_process_group()


def test_two_rank_fit_runs_the_reference_train_entry_point_end_to_end(tmp_path):
    """`train()` under a 2-rank launch (both ranks on the one GPU, gloo): rank 0's split is shared, every
    rank runs the same number of steps with 5 training volumes (DistributedSampler padding), validation
    numbers come from rank 0, only rank 0 writes checkpoints / Dataset.json / the CSV log, and the ranks
    end with identical weights."""
    import json

    import numpy as np
    from oracle.unet_ref import synthetic_batch
    from segmantic_amd.data.nifti import write_nifti
    root = tmp_path / "data"
    (root / "image").mkdir(parents=True)
    (root / "label").mkdir()
    A = np.diag([1.0, 1.0, 1.0, 1.0])
    n = 7
    for i in range(n):
        img, lab = synthetic_batch(1, 24, 3, seed=40 + i)
        write_nifti(root / "image" / f"c{i}.nii.gz", (img[0, 0].numpy() * 100 + 300).astype(np.float32).transpose(2, 1, 0), A)
        write_nifti(root / "label" / f"c{i}.nii.gz", lab[0, 0].numpy().astype(np.uint8).transpose(2, 1, 0), A)
    dl = {"labels": {"1": "a", "2": "b"},
          "training": [{"image": f"image/c{i}.nii.gz", "label": f"label/c{i}.nii.gz"} for i in range(5)],
          "validation": [{"image": f"image/c{i}.nii.gz", "label": f"label/c{i}.nii.gz"} for i in (5, 6)],
          "test": []}
    (root / "dataset.json").write_text(json.dumps(dl))
    out = tmp_path / "results"
    port = _free_port()
    mp.spawn(_fit_worker, args=(2, port, root / "dataset.json", out), nprocs=2, join=True)
    ckpts = sorted(out.glob("epoch=*-val_loss=*-val_dice=*.ckpt"))
    assert 1 <= len(ckpts) <= 2                      # written once (rank 0), at most one per epoch
    rows = (out / "logs" / "metrics.csv").read_text().strip().splitlines()
    assert len(rows) == 3 and all(np.isfinite(float(r.split(",")[1])) for r in rows[1:])
    assert (out / "Dataset.json").exists()
    r0, r1 = torch.load(out / "final_rank0.pt"), torch.load(out / "final_rank1.pt")
    assert torch.equal(r0["flat"], r1["flat"])       # replicas never diverge
    assert r0["best"] == r1["best"]                  # the validation number is rank 0's on every rank


def _predict_worker(rank, world, port, ckpt, images, labels, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0", SEGMI_DIST_BACKEND="gloo")
    from segmantic_amd.seg.monai_unet import predict
    predict(model_file=ckpt, test_images=images, test_labels=labels, output_dir=out_dir,
            tissue_dict={"bg": 0, "a": 1, "b": 2}, channels=CH, strides=ST, gpu_ids=[0, 0])
    if dist.is_initialized():
        dist.destroy_process_group()


def test_two_rank_predict_equals_the_single_process_run(tmp_path):
    """`predict()` under a 2-rank launch: volumes dealt round-robin, label files identical to the
    single-process run, the score file in `test_images` order."""
    import warnings

    import numpy as np
    from oracle.unet_ref import synthetic_batch
    from segmantic_amd.data.nifti import read_nifti, write_nifti
    from segmantic_amd.seg.monai_unet import Net, predict
    root = tmp_path / "data"
    root.mkdir()
    A = np.diag([1.0, 1.0, 1.0, 1.0])
    images, labels = [], []
    for i in range(3):
        img, lab = synthetic_batch(1, 24, 3, seed=60 + i)
        write_nifti(root / f"i{i}.nii.gz", (img[0, 0].numpy() * 100 + 300).astype(np.float32).transpose(2, 1, 0), A)
        write_nifti(root / f"l{i}.nii.gz", lab[0, 0].numpy().astype(np.uint8).transpose(2, 1, 0), A)
        images.append(root / f"i{i}.nii.gz")
        labels.append(root / f"l{i}.nii.gz")
    torch.manual_seed(3)
    net = Net(num_classes=3, channels=CH, strides=ST, spatial_size=[16, 16, 16]).to("cuda:0")
    ckpt = tmp_path / "m.ckpt"
    net.save_checkpoint(ckpt, epoch=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        predict(model_file=ckpt, test_images=images, test_labels=labels, output_dir=tmp_path / "one",
                tissue_dict={"bg": 0, "a": 1, "b": 2}, channels=CH, strides=ST, gpu_ids=[0])
    port = _free_port()
    mp.spawn(_predict_worker, args=(2, port, ckpt, images, labels, tmp_path / "two"), nprocs=2, join=True)
    for i in range(3):
        a, _ = read_nifti(tmp_path / "one" / f"i{i}.nii.gz")
        b, _ = read_nifti(tmp_path / "two" / f"i{i}.nii.gz")
        assert np.array_equal(a, b), i
    s1 = np.loadtxt(tmp_path / "one" / "mean_dice_m_generalized_score.txt", delimiter=",")
    s2 = np.loadtxt(tmp_path / "two" / "mean_dice_m_generalized_score.txt", delimiter=",")
    # (the running mean is taken on the device in one process and on gathered host rows on rank 0: 1 ulp)
    assert s1.shape == s2.shape == (3,) and np.allclose(s1, s2, rtol=1e-6, atol=0)


def _rccl_worker(rank, world, port, out_dir, force):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      SEGMI_GRADSYNC_FORCE="1" if force else "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)          # nccl = RCCL on ROCm
    net = _make_net()
    gs = net.enable_grad_sync(bucket_bytes=64 << 10)
    assert gs.world == 1 and gs._force == bool(force)
    for step in range(3):
        net.training_step(_batch(0))
    torch.cuda.synchronize()
    torch.save(net._engine.flat.detach().cpu(), os.path.join(out_dir, f"rccl_{int(force)}.pt"))
    dist.destroy_process_group()


def test_gradient_buckets_through_rccl_with_one_rank(tmp_path):
    """The box has one GPU, so RCCL cannot run a real exchange -- but it can run the whole path: process
    group `nccl`, every gradient bucket all-reduced (over one rank: the identity) on the side stream
    behind the engine's events, `Work.wait()` on the training stream.  Three steps must leave exactly
    the weights of the same steps without the collectives."""
    port = _free_port()
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path), True), nprocs=1, join=True)
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path), False), nprocs=1, join=True)
    assert torch.equal(torch.load(tmp_path / "rccl_1.pt"), torch.load(tmp_path / "rccl_0.pt"))


def _fresh_env():
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, PYTHONPATH=str(root), SEGMI_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return root, env, sys.executable


def test_bench_gpus_2_starts_its_own_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` un-wrapped: the parent -- a fresh process that has
    not touched the GPU -- starts two ranks (both on this box's one GPU, gloo transport), rank 0 prints
    the JSON line, the launcher's exit code comes back."""
    import json
    import subprocess
    root, env, py = _fresh_env()
    p = subprocess.run([py, str(root / "bench.py"), "--gpus", "2", "--workload", "train", "--steps", "2",
                        "--warmup", "1", "--size", "32", "--batch", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 4
    gx = out["config"]["gradient_exchange"]
    assert gx["backend"] == "gloo" and gx["bytes_per_step"] > 0 and gx["exposed_allreduce_ms_per_step_rank0"] >= 0


def test_cli_train_with_two_gpu_ids_starts_its_own_ranks(tmp_path):
    """`segmantic-unet train-config` with `gpu_ids: [0, 0]` from a plain process: train() hands the call
    to two ranks it starts itself (reference: pl.Trainer(devices=len(gpu_ids)), monai_unet.py:529-538)."""
    import json
    import subprocess

    import numpy as np
    from oracle.unet_ref import synthetic_batch
    from segmantic_amd.data.nifti import write_nifti
    root, env, py = _fresh_env()
    data = tmp_path / "data"
    (data / "image").mkdir(parents=True)
    (data / "label").mkdir()
    A = np.diag([1.0, 1.0, 1.0, 1.0])
    for i in range(4):
        img, lab = synthetic_batch(1, 24, 3, seed=80 + i)
        write_nifti(data / "image" / f"c{i}.nii.gz", (img[0, 0].numpy() * 100 + 300).astype(np.float32).transpose(2, 1, 0), A)
        write_nifti(data / "label" / f"c{i}.nii.gz", lab[0, 0].numpy().astype(np.uint8).transpose(2, 1, 0), A)
    dl = {"labels": {"1": "a", "2": "b"},
          "training": [{"image": f"image/c{i}.nii.gz", "label": f"label/c{i}.nii.gz"} for i in range(3)],
          "validation": [{"image": "image/c3.nii.gz", "label": "label/c3.nii.gz"}], "test": []}
    (data / "dataset.json").write_text(json.dumps(dl))
    out = tmp_path / "results"
    cfg = {"datalist": str(data / "dataset.json"), "output_dir": str(out), "spatial_size": [16, 16, 16],
           "channels": list(CH), "strides": list(ST), "max_epochs": 1, "num_samples": 2, "gpu_ids": [0, 0]}
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    p = subprocess.run([py, "-m", "segmantic_amd.commands.monai_unet_cli", "train-config", "-c",
                        str(tmp_path / "cfg.json")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "[rank 0/2]" in p.stdout and "[rank 1/2]" in p.stdout
    assert len(list(out.glob("epoch=0-val_loss=*-val_dice=*.ckpt"))) == 1      # rank 0 only
    assert not list(out.glob("ranks_*.json"))                                  # the hand-over file is gone


def test_bench_infer_gpus_2_shards_one_volume_by_z_slabs():
    """`bench.py --workload infer --gpus 2` (self-launched, gloo): next to the replica figure the line carries
    ONE volume cut into z-slabs, one per rank, with the label all-gather as its only exchange (north_star's
    inference split)."""
    import json
    import subprocess
    root, env, py = _fresh_env()
    p = subprocess.run([py, str(root / "bench.py"), "--gpus", "2", "--workload", "infer", "--steps", "1", "--warmup", "1",
                        "--volume", "192", "--size", "64", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["parallelism"] == "replicas2"
    sh = out["one_volume_sharded"]
    assert sh["value"] > 0 and sh["labels_shape"] == [192, 192, 192] and sh["label_allgather_ms"] >= 0
    assert out["lanes"]["labels_identical"] is True
