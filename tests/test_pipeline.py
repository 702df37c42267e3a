"""Predict pre-/post-processing chain (reference ``src/segmantic/seg/monai_unet.py:151-176`` and
``:612-625``) against ``oracle/pipeline_ref.py`` -- host geometry on CPU, values on the GPU.

Which MONAI Spacing convention is restated, and why, is written in ``oracle/pipeline_ref.py``.
"""
import numpy as np
import pytest
import torch

from oracle import pipeline_ref as R
from segmantic_amd.seg import pipeline as P


def _rot(ax, deg):
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    m = np.eye(3)
    i, j = [(1, 2), (0, 2), (0, 1)][ax]
    m[i, i], m[i, j], m[j, i], m[j, j] = c, -s, s, c
    return m


def _affine(perm, signs, spacing, origin=(3.0, -7.5, 11.25), rot=None):
    """voxel axis v runs along world axis perm[v] with sign signs[v] and spacing[v] mm."""
    A = np.eye(4)
    M = np.zeros((3, 3))
    for v in range(3):
        M[perm[v], v] = signs[v] * spacing[v]
    if rot is not None:
        M = rot @ M
    A[:3, :3] = M
    A[:3, 3] = origin
    return A


AFFINES = [
    _affine((0, 1, 2), (1, 1, 1), (1.0, 1.0, 1.0)),
    _affine((0, 1, 2), (-1, -1, 1), (0.8, 0.8, 2.5)),                       # LPS, anisotropic
    _affine((2, 0, 1), (1, -1, 1), (1.5, 0.7, 1.1)),                        # permuted + flipped
    _affine((1, 2, 0), (-1, 1, -1), (0.9, 1.3, 2.0), rot=_rot(2, 12.0)),    # oblique
    _affine((0, 2, 1), (1, 1, -1), (2.0, 0.5, 1.0), rot=_rot(0, -20.0) @ _rot(1, 8.0)),
]


@pytest.mark.parametrize("A", AFFINES)
def test_orientation_matches_nibabel_rule(A):
    ornt = P.io_orientation(A)
    ref = R.ref_io_orientation(A)
    assert np.array_equal(ornt, ref.astype(np.int64))
    g = np.random.RandomState(0)
    vol = g.rand(2, 5, 6, 7).astype(np.float32)
    got, A_new, rec = P.to_ras(torch.from_numpy(vol), A)
    want, A_ref, o = R.ref_to_ras(vol, A)
    assert np.array_equal(got.numpy(), want)
    assert np.allclose(A_new, A_ref, atol=1e-12)
    # the re-oriented affine is RAS: dominant direction of voxel axis w is +world axis w
    assert np.array_equal(P.io_orientation(A_new), np.array([[0, 1], [1, 1], [2, 1]]))
    back = P.from_ras(got, rec)
    assert np.array_equal(back.numpy(), vol)
    assert np.array_equal(R.ref_from_ras(want, o), vol)


@pytest.mark.parametrize("A", AFFINES)
@pytest.mark.parametrize("pixdim", [(1.0, 1.0, 1.0), (2.0, 1.5, 0.6), (0.5, 0.5, 0.5)])
def test_spacing_geometry_matches_monai_rule(A, pixdim):
    vol = np.zeros((1, 9, 12, 7), np.float32)
    _, A_ras, _ = R.ref_to_ras(vol, A)
    shape = R.ref_to_ras(vol, A)[0].shape[1:]
    new, new_shape = P.spacing_geometry(A_ras, shape, pixdim)
    ref_new = R.ref_zoom_affine(A_ras, pixdim)
    ref_shape, ref_off = R.ref_compute_shape_offset(shape, A_ras, ref_new)
    ref_new[:3, 3] = ref_off
    assert list(new_shape) == [int(v) for v in ref_shape]
    assert np.allclose(new, ref_new, atol=1e-10)
    # axis-aligned input: the closed form round((n - 1) * in / out + 1)
    sp = np.sqrt((A_ras[:3, :3] ** 2).sum(0))
    if np.allclose(np.abs(A_ras[:3, :3]), np.diag(sp)):
        want = np.round((np.asarray(shape) - 1) * sp / np.asarray(pixdim) + 1).astype(int)
        assert list(new_shape) == list(want)


# --------------------------------------------------------------------------------------- GPU
def _blob_labels(shape, k, seed):
    g = np.random.RandomState(seed)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    lab = np.zeros(shape, np.uint8)
    for c in range(1, k):
        ctr = [g.uniform(0.3, 0.7) * s for s in shape]
        rad = [g.uniform(0.15, 0.3) * s for s in shape]
        m = ((zz - ctr[0]) / rad[0]) ** 2 + ((yy - ctr[1]) / rad[1]) ** 2 + ((xx - ctr[2]) / rad[2]) ** 2 < 1
        lab[m] = c
    return lab


def _smooth_logits(k, shape, seed):
    g = torch.Generator().manual_seed(seed)
    low = torch.randn((1, k) + tuple(max(2, s // 4) for s in shape), generator=g)
    return torch.nn.functional.interpolate(low, size=tuple(shape), mode="trilinear",
                                           align_corners=True)[0].contiguous()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [0, 1, 2])
def test_predict_chain_matches_oracle(tmp_path, case):
    from segmantic_amd.data.nifti import read_nifti, write_nifti
    A = [AFFINES[1], AFFINES[2], AFFINES[3]][case]
    spacing = [(1.0, 1.0, 1.0), (1.2, 0.9, 1.6), (1.0, 1.0, 1.0)][case]
    K = 5
    shape_zyx = (18, 22, 26)
    g = np.random.RandomState(10 + case)
    lab = _blob_labels(shape_zyx, K, 3 + case)
    lab[:3] = 0                      # background margins: CropForeground really crops
    lab[:, :2] = 0
    lab[:, :, -4:] = 0
    img = (g.rand(*shape_zyx).astype(np.float32) * 100 + 40 * lab).astype(np.float32)
    write_nifti(tmp_path / "img.nii.gz", img, A)
    write_nifti(tmp_path / "lab.nii.gz", lab, A)
    _, A_file = read_nifti(tmp_path / "img.nii.gz")          # f32-rounded affine, as stored

    pipe = P.PredictPipeline(device="cuda:0", spacing=spacing, with_label=True)
    item = pipe.load(tmp_path / "img.nii.gz", tmp_path / "lab.nii.gz")
    # oracle on the same voxels in MONAI's [C, i, j, k] order
    vol = np.ascontiguousarray(img.transpose(2, 1, 0))[None]
    labv = np.ascontiguousarray(lab.transpose(2, 1, 0))[None].astype(np.float32)
    rec = R.ref_preprocess(vol, A_file, spacing, labv, A_file)
    assert tuple(item["image"].shape) == rec["image"].shape
    assert item["crop"][0] == rec["crop"][0] and item["crop"][1] == rec["crop"][1]
    assert rec["image"].shape[1:] != tuple(item["crop"][2]), "the case must exercise the crop"
    assert np.allclose(item["affine"], rec["affine"], atol=1e-9)
    got = item["image"].cpu().numpy()
    # normalise: f32 kernel vs f64 oracle; resample: identical f64 tap arithmetic on those values
    assert np.abs(got - rec["image"]).max() < 2e-5
    assert np.abs(item["label"].cpu().numpy() - rec["label"]).max() < 1e-5

    # inverse chain on the K-channel logits: Spacing^-1 -> un-crop -> un-orient -> argmax
    logits = _smooth_logits(K, rec["image"].shape[1:], 77 + case)
    lab_gpu = pipe.invert_and_discretize(logits.to("cuda:0"), item).cpu().numpy()
    lab_ref = R.ref_invert_and_discretize(logits.numpy(), rec)
    assert lab_gpu.shape == lab_ref.shape == vol.shape[1:]
    assert np.array_equal(lab_gpu.astype(np.int64), lab_ref), \
        f"{int((lab_gpu != lab_ref).sum())} label voxels differ"
    # outside the crop box every logit is 0 -> class 0 (first max)
    assert (lab_ref == 0).any()

    # the saved file carries the source grid
    (tmp_path / "out").mkdir()
    out = pipe.save(torch.from_numpy(lab_gpu).to("cuda:0"), item, tmp_path / "out")
    arr, A_out = read_nifti(out)
    assert arr.shape == shape_zyx and np.allclose(A_out, A_file)
    assert np.array_equal(arr.transpose(2, 1, 0), lab_gpu)


@pytest.mark.gpu
def test_predict_chain_round_trip_reproduces_label(tmp_path):
    """load() then invert_and_discretize() of the one-hot label volume gives back the label
    (no spacing: flips, permutation, crop and zero padding are exact)."""
    from segmantic_amd.data.nifti import write_nifti
    A = AFFINES[2]
    K = 4
    lab = _blob_labels((16, 20, 24), K, 5)
    lab[:2] = 0
    lab[:, -3:] = 0
    img = (np.random.RandomState(1).rand(16, 20, 24) * 10 + lab).astype(np.float32)
    write_nifti(tmp_path / "img.nii.gz", img, A)
    write_nifti(tmp_path / "lab.nii.gz", lab, A)
    pipe = P.PredictPipeline(device="cuda:0", spacing=(), with_label=True)
    item = pipe.load(tmp_path / "img.nii.gz", tmp_path / "lab.nii.gz")
    pl = item["label"][0].long()
    onehot = torch.zeros((K,) + tuple(pl.shape), device="cuda:0").scatter_(0, pl[None], 1.0)
    got = pipe.invert_and_discretize(onehot, item).cpu().numpy()
    assert np.array_equal(got, lab.transpose(2, 1, 0))


@pytest.mark.gpu
def test_spacing_with_a_nearest_label_mode_keeps_integer_labels(tmp_path):
    """Spacingd(mode=[bilinear, nearest]) from a bundle config (utils/bundle.py plans
    ``spacing_label_nearest``): the label volume is resampled with MONAI's nearest rule (torch
    grid_sample: nearbyint, i.e. half to even -- a 2x spacing change hits x.5 on every other voxel),
    the image bilinearly as before."""
    from segmantic_amd.data.nifti import read_nifti, write_nifti
    A = AFFINES[1]
    K = 5
    lab = _blob_labels((18, 22, 26), K, 4)
    img = (np.random.RandomState(3).rand(18, 22, 26) * 50 + 10 * lab).astype(np.float32)
    write_nifti(tmp_path / "img.nii.gz", img, A)
    write_nifti(tmp_path / "lab.nii.gz", lab, A)
    _, A_file = read_nifti(tmp_path / "img.nii.gz")
    sp_in = np.sqrt((A_file[:3, :3] ** 2).sum(0))
    for factor in (2.0, 0.5, 1.3):
        spacing = tuple(float(v) * factor for v in sp_in)
        pipe = P.PredictPipeline(device="cuda:0", spacing=spacing, with_label=True, label_nearest=True)
        item = pipe.load(tmp_path / "img.nii.gz", tmp_path / "lab.nii.gz")
        got = item["label"].cpu().numpy()
        assert np.array_equal(got, np.round(got)) and set(np.unique(got)) <= set(range(K))
        # oracle: same chain, label through the nearest resampler
        vol = np.ascontiguousarray(img.transpose(2, 1, 0))[None]
        labv = np.ascontiguousarray(lab.transpose(2, 1, 0))[None].astype(np.float32)
        rec = R.ref_preprocess(vol, A_file, (), labv, A_file)          # up to the crop
        want, _ = R.ref_spacing(rec["label"], rec["affine_crop"], spacing, nearest=True)
        assert got.shape == want.shape and np.array_equal(got, want), factor
        # the bilinear default of the reference still interpolates the label
        soft = P.PredictPipeline(device="cuda:0", spacing=spacing, with_label=True).load(
            tmp_path / "img.nii.gz", tmp_path / "lab.nii.gz")["label"].cpu().numpy()
        if factor != 2.0:          # (a 2x coarser grid that starts on a voxel samples voxels exactly)
            assert not np.array_equal(soft, np.round(soft))
