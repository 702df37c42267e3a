"""CPU oracle for the segmantic 3D-UNet hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import anything from this package.  The product (``segmantic_amd``) never does: it fails
loudly when its HIP library is missing instead of falling back to this code.

What this is
------------
A plain torch-CPU / numpy fp32 restatement of the arithmetic that the reference
(dyollb/segmantic, ``/root/reference``) delegates to third-party libraries on its hot path:

* ``unet_ref``      MONAI ``UNet(num_res_units=2, norm=BATCH, act=PRELU)`` as built at
                    reference ``src/segmantic/seg/monai_unet.py:114-124``, MONAI ``DiceLoss(
                    to_onehot_y=True, softmax=True)`` (``:128``), the manual-optimisation step
                    order of ``training_step`` (``:339-348``) and Adam (``:299-304``).
* ``sliding_ref``   MONAI ``sliding_window_inference`` as called at ``:354-356`` / ``:637-639``.
* ``metrics_ref``   ``AsDiscrete(argmax)`` (``:129-134, :622, :673``), ``DiceMetric(
                    include_background=False)`` (``:136-138``), ``NormalizeIntensityd(
                    channel_wise=True)`` (``:164``).
* ``resample_ref``  ITK ``ResampleImageFilter`` semantics used by
                    ``src/segmantic/image/processing.py:49-120``.
* ``ensemble_ref``  ``SelectBestEnsemble`` of ``src/segmantic/seg/transforms.py:15-61``.

Third-party dependencies that hold the algorithm (all absent from ``/root/reference`` and from
this image, all *unpinned* in the reference's ``pyproject.toml:24-40``): ``monai`` (code needs
>= 1.2), ``pytorch-lightning`` (>= 2.0), ``SimpleITK``.  ``torch`` (present here, 2.10) is
where MONAI's UNet ultimately gets its numbers, so the oracle is composed of the very same
``torch.nn`` CPU operators.

PARITY UNPINNED
---------------
The reference's own tests hold **no** golden vector for logits, loss, gradients, sliding-window
output, Dice or resampled intensities, and the reference hot path cannot
be imported here (``ModuleNotFoundError: pytorch_lightning / monai / SimpleITK`` -- an ordinary
error, no permission was denied).  The value-level oracle is therefore **parity unpinned**.
What *is* pinned against the reference / its dependencies' published facts:

* UNet parameter count 4,808,917 for (in=1, out=2) -- the figure MONAI's spleen tutorial prints
  for exactly this configuration -- and the 148 MONAI state-dict key names / shapes
  (``tests/test_oracle.py``).
* ``resample`` output geometry asserted by the reference's ``tests/image/test_image.py:33-52``.
* ``Net`` hparams of ``tests/seg/test_unet.py:15-20``; config round trips of
  ``tests/utils/test_cli.py:19-80``.
* ``ensemble_ref.ref_select_best`` (``SelectBestEnsemble``, ``seg/transforms.py:39-61``): the one
  value-level vector the reference's tests hold for this path, ``tests/seg/test_transforms.py:9-43``
  (label and one-hot forms), committed as ``tests/golden/reference_select_best.json`` -- this
  function, and through it ``ops.ensemble_select``, is **pinned**.
"""
