"""CPU: the oracle's restatement of cv.resize(INTER_LINEAR, 8-bit) -- known answers and the bound against textbook fp64
bilinear.  (OpenCV itself is not in the image: parity with cv2 is unpinned, oracle/crop_resize.py says so.)"""
import numpy as np
import pytest

from oracle.crop_resize import crop, resize_linear_float, resize_linear_u8


def test_identity_and_exact_halving_known_answers():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (36, 52, 3), dtype=np.uint8)
    assert np.array_equal(resize_linear_u8(img, (52, 36)), img)
    e = img.astype(np.int64)
    avg = (e[0::2, 0::2] + e[0::2, 1::2] + e[1::2, 0::2] + e[1::2, 1::2] + 2) >> 2       # cv2: INTER_LINEAR at exactly 2x == area
    assert np.array_equal(resize_linear_u8(img, (26, 18)), avg.astype(np.uint8))
    flat = np.full((9, 11, 3), 137, np.uint8)
    assert (resize_linear_u8(flat, (224, 224)) == 137).all()
    one = rng.integers(0, 256, (1, 1, 3), dtype=np.uint8)
    assert (resize_linear_u8(one, (5, 7)) == one[0, 0]).all()


@pytest.mark.parametrize('shape,size', [((37, 53), (224, 224)), ((300, 200), (224, 224)), ((1080, 1920), (224, 224)),
                                        ((5, 7), (9, 3)), ((64, 64), (63, 65)), ((2, 2), (224, 224))])
def test_within_one_grey_level_of_fp64_bilinear(shape, size):
    rng = np.random.default_rng(shape[0] + size[0])
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    a, b = resize_linear_u8(img, size), resize_linear_float(img, size)
    assert a.shape == (size[1], size[0], 3)
    assert np.abs(a.astype(np.float64) - b).max() < 1.0


def test_crop_is_the_numpy_slice():
    frame = np.arange(20 * 30 * 3, dtype=np.uint8).reshape(20, 30, 3)
    assert np.array_equal(crop(frame, (3, 4, 10, 9)), frame[4:9, 3:10])
    assert crop(frame, (25, 15, 60, 40)).shape == (5, 5, 3)         # clamped at the frame like numpy
    assert crop(frame, (10, 10, 10, 12)).size == 0


def test_crop_box_from_keypoints_matches_the_oracle_restatement():
    from oracle.crop_resize import objectron_crop
    from torchdet3d.dataloaders import crop_cords_from_keypoints
    rng = np.random.default_rng(1)
    frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    for _ in range(50):
        kp = rng.integers(-40, 700, (9, 2)).astype(np.int64)            # some points outside the frame
        clipped, box = crop_cords_from_keypoints(kp, 640, 480)
        skp, crop, obox = objectron_crop(frame, kp)
        assert tuple(int(v) for v in box) == obox
        assert np.array_equal(clipped - np.asarray(obox[:2]), skp)
        assert crop.shape[0] == obox[3] - obox[1] and crop.shape[1] == obox[2] - obox[0]
