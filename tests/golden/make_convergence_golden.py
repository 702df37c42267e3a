"""Generates tests/golden/convergence_c1.json: the CPU oracle's training trajectory on BASELINE
config 1 (32^3 patches, 3 labels, batch 8 = 2 volumes x num_samples 4, fp32) -- the target of
north_star's last gate, "Dice within 1e-4 of the CPU reference on synthetic data".

Run in the build container:  python tests/golden/make_convergence_golden.py

Protocol (mirrored by tests/test_e2e_gpu.py::test_convergence_matches_cpu_reference):
  * network: the reference's default UNet (channels 16-32-64-128-256, strides 2, K = 3), weights from
    the build-owned counter-hash filler (seed 0);
  * 30 ``training_step``s (reference order, monai_unet.py:339-348) with Adam(lr) on the batches
    ``synthetic_batch(8, 32, 3, seed=100 + step)``;
  * after steps 10, 20, 30: the reference's validation (``validation_step`` :350-363 +
    ``on_validation_epoch_end`` :365-397) on two 32^3 volumes ``synthetic_batch(1, 32, 3, seed=900 + i)``:
    sliding window roi 160^3 (the volume is zero-padded to the roi), sw_batch 4, Dice loss, argmax,
    DiceMetric(include_background=False) mean.
Pure data: no reference source is involved (the oracle is this repository's own restatement).
"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle.metrics_ref import ref_argmax, ref_dice_metric          # noqa: E402
from oracle.sliding_ref import ref_sliding_window_inference          # noqa: E402
from oracle.unet_ref import (RefUNet, deterministic_fill_, ref_dice_loss, ref_train_step,   # noqa: E402
                             synthetic_batch)

K, SIZE, B, STEPS, EVERY, LR = 3, 32, 8, 30, 10, 1e-4


def validate(net):
    net.eval()
    dices, losses = [], []
    with torch.no_grad():
        for i in range(2):
            img, lab = synthetic_batch(1, SIZE, K, seed=900 + i)
            out, _, _ = ref_sliding_window_inference(img, (160,) * 3, 4, net, overlap=0.25)
            losses.append(float(ref_dice_loss(out, lab)))
            d, _ = ref_dice_metric(ref_argmax(out), lab.long(), K)
            dices.append(d)
    net.train()
    d = torch.cat(dices)                                   # [2, K-1]; mean over classes then volumes
    return float(torch.nanmean(torch.nanmean(d, 1))), sum(losses) / len(losses)


def trajectory(threads: int) -> dict:
    torch.set_num_threads(threads)
    net = deterministic_fill_(RefUNet(3, 1, K), 0).train()
    opt = torch.optim.Adam(net.parameters(), lr=LR)
    out = {"threads": threads, "train_loss": [], "val_dice": [], "val_loss": []}
    out["val_dice_initial"], out["val_loss_initial"] = validate(net)
    for step in range(STEPS):
        img, lab = synthetic_batch(B, SIZE, K, seed=100 + step)
        _, loss = ref_train_step(net, opt, img, lab)
        out["train_loss"].append(float(loss))
        if (step + 1) % EVERY == 0:
            d, l = validate(net)
            out["val_dice"].append(d)
            out["val_loss"].append(l)
            print(threads, step + 1, float(loss), d, l, flush=True)
    return out


def main():
    # Two runs of the SAME oracle with different oneDNN thread counts (= different f32 summation orders):
    # their spread is the CPU reference's own reproducibility on this protocol.  The validation Dice is a
    # discrete function of near-tied logits on 2 x 32^3 voxels, so the spread is not negligible against
    # north_star's 1e-4 (measured: 2e-6 / 2e-5 / 1.1e-4 at steps 10 / 20 / 30); the GPU test therefore
    # asserts |HIP - nearest oracle run| <= 1e-4.
    runs = [trajectory(8), trajectory(3)]
    out = {"config": {"labels": K, "patch": SIZE, "batch": B, "steps": STEPS, "validate_every": EVERY, "lr": LR,
                      "optimizer": "Adam", "val_volumes": 2, "val_roi": 160, "val_overlap": 0.25,
                      "torch": torch.__version__},
           "runs": runs,
           "oracle_self_spread": {
               "val_dice": [abs(a - b) for a, b in zip(runs[0]["val_dice"], runs[1]["val_dice"])],
               "val_loss": [abs(a - b) for a, b in zip(runs[0]["val_loss"], runs[1]["val_loss"])],
               "train_loss_max": max(abs(a - b) for a, b in zip(runs[0]["train_loss"], runs[1]["train_loss"]))}}
    (Path(__file__).parent / "convergence_c1.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
