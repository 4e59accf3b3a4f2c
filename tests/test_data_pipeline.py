"""Input pipeline: RandomCrop3D against vectors produced by the reference class
(tests/golden/make_golden_aug.py), the training-sample assembly on the GPU against a per-slice numpy restatement."""
import numpy as np
import pytest
import torch


def test_random_crop3d_matches_reference_golden(golden_dir):
    from mr3d_amd.datasets import RandomCrop3D
    g = np.load(f'{golden_dir}/random_crop3d.npz')
    for k in range(int(g['num_cases'])):
        H, W, D = (int(v) for v in g[f'hwd_{k}'])
        boxes, labels = g[f'boxes_{k}'], g[f'labels_{k}']
        aug = RandomCrop3D(rng=np.random.RandomState(int(g[f'seed_{k}'])))
        img = np.zeros((H, W, D), np.uint8)
        img[:] = (np.arange(W) % 251)[None, :, None]
        crop, nb, nl, nm = aug(img, boxes.copy(), labels.copy(), [img] * len(boxes))
        x0, y0, _ = g[f'origin_{k}']
        assert tuple(crop.shape) == tuple(g[f'crop_shape_{k}']) == (H // 4, W // 4, D), k
        assert np.array_equal(crop, img[y0:y0 + H // 4, x0:x0 + W // 4, :]), k
        np.testing.assert_array_equal(nb, g[f'new_boxes_{k}'])
        np.testing.assert_array_equal(nl, g[f'new_labels_{k}'])
        assert len(nm) == len(nb) and all(m.shape == crop.shape for m in nm)
        # every kept box lies strictly inside the patch
        assert (nb[:, [0, 1, 4]] >= 0).all() and (nb[:, 2] < W // 4).all() and (nb[:, 3] < H // 4).all() and (nb[:, 5] < D).all()


@pytest.mark.gpu
def test_prepare_train_sample_gpu(tmp_path):
    """.npy -> device -> RandomCrop3D -> both pathways + boxes + masks; pathway 1 bit-exact vs the per-slice restatement
    (oracle/input_ref.py), pathway 2 = the 1.5x resample of the SAME patch, masks cropped / padded / transposed"""
    from mr3d_amd.datasets import RandomCrop3D, prepare_train_sample, load_npy_volume, IMG_NORM_CFG
    from oracle import input_ref
    rs = np.random.RandomState(3)
    H, W, D = 200, 264, 12
    grey = rs.randint(0, 255, (H, W, D)).astype(np.uint8)
    path = tmp_path / 'scan.npy'
    np.save(path, grey)
    boxes = np.array([[100, 60, 120, 80, 2, 6], [150, 90, 170, 110, 1, 9], [30, 20, 45, 40, 3, 8]], np.float32)
    labels = np.array([1, 1, 1])
    masks = []
    for b in boxes.astype(int):
        m = np.zeros((H, W, D), np.uint8)
        m[b[1]:b[3] + 1, b[0]:b[2] + 1, b[4]:b[5] + 1] = 1
        masks.append(m)
    vol = load_npy_volume(str(path))
    assert vol.is_cuda and vol.dtype == torch.uint8 and tuple(vol.shape) == (H, W, D)
    aug = RandomCrop3D(rng=np.random.RandomState(11))
    data = prepare_train_sample(vol, boxes, labels, masks, upscale_factor=1.5, extra_aug=aug)
    # replay the same augmentation on the host
    crop, nb, nl, nm = RandomCrop3D(rng=np.random.RandomState(11))(grey, boxes, labels, masks)
    ph, pw = H // 4, W // 4
    assert tuple(crop.shape) == (ph, pw, D)
    exp = input_ref.prepare_volume_ref(crop, IMG_NORM_CFG['mean'], IMG_NORM_CFG['std'], 32)
    got = data['imgs'].cpu().numpy()[0]
    assert got.shape == exp.shape and data['img_meta']['img_shape'] == (ph, pw, 3, D)
    np.testing.assert_array_equal(got, exp)
    assert data['imgs_2'].shape[2:] == (int(round(D * 1.5)), -(-int(round(ph * 1.5)) // 32) * 32, -(-int(round(pw * 1.5)) // 32) * 32)
    np.testing.assert_allclose(data['gt_bboxes'].cpu().numpy(), nb)
    np.testing.assert_allclose(data['gt_bboxes_2'].cpu().numpy(), nb * 1.5)
    assert data['gt_labels'].tolist() == list(nl) and data['gt_masks_2'] is None
    gm = data['gt_masks'].cpu().numpy()
    hp, wp = data['img_meta']['pad_shape'][:2]
    assert gm.shape == (len(nb), D, hp, wp) and gm.dtype == np.uint8
    for k, m in enumerate(nm):
        np.testing.assert_array_equal(gm[k, :, :ph, :pw], m.transpose(2, 0, 1))
        assert gm[k, :, ph:, :].sum() == 0 and gm[k, :, :, pw:].sum() == 0
    # a scan without boxes is skipped like in the reference (coco_3d_2scales.py:206-207)
    assert prepare_train_sample(vol, np.zeros((0, 6), np.float32), np.zeros(0), None) is None
