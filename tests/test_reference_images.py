"""The reference's extractor tests on its own two test images (test/stella_vslam/feature/orb_extractor.cc:79-330), with the images
derived by tests/golden/make_reference_images.py.  CPU: the oracle must satisfy the reference's assertions (keypoints found,
rows == size, nothing inside a mask).  GPU (-m gpu): the device path must reproduce the oracle bit for bit on this REAL imagery,
for every mask variant the reference exercises."""
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def _img(i):
    Image = pytest.importorskip("PIL.Image")
    return np.ascontiguousarray(np.asarray(Image.open(os.path.join(HERE, "golden", f"equirect_00{i}_gray.png"))), dtype=np.uint8)


def _rect_mask(rects, w, h):
    """create_rectangle_mask (orb_extractor.cc:138-151): [x_min, x_max, y_min, y_max] ratios -> std::round -> cv::rectangle filled
    with 0 on a 255 background, both corner points inclusive."""
    m = np.full((h, w), 255, np.uint8)
    for x0, x1, y0, y1 in rects:
        xa, xb, ya, yb = int(round(w * x0)), int(round(w * x1)), int(round(h * y0)), int(round(h * y1))
        m[max(ya, 0):yb + 1, max(xa, 0):xb + 1] = 0
    return m


def _cases():
    h, w = 960, 1920
    tb = np.ones((h, w), np.uint8)           # extract_with_image_mask_1 (:117-152): 25 % of top and bottom
    tb[0:h // 4] = 0
    tb[3 * h // 4:h - 1] = 0
    lr = np.ones((h, w), np.uint8)           # _2 (:154-189): 25 % of left and right
    lr[:, 0:w // 4] = 0
    lr[:, 3 * w // 4:w - 1] = 0
    yy, xx = np.mgrid[0:h, 0:w]
    disc = np.ones((h, w), np.uint8)         # _3 (:191-229): disc of radius 320 in the centre
    disc[(xx - w // 2) ** 2 + (yy - h // 2) ** 2 <= 320 ** 2] = 0
    r1 = [[0.0, 1.0, 0.0, 0.2], [0.0, 1.0, 0.8, 1.0]]                       # rectangle masks 1-3 (:231-330)
    r2 = [[0.0, 0.2, 0.0, 1.0], [0.8, 1.0, 0.0, 1.0]]
    return {
        "without_mask_1": (1, None, None, lambda k: True),
        "without_mask_2": (2, None, None, lambda k: True),
        "image_mask_1": (1, tb, None, lambda k: (k["y"] >= h // 4).all() and (k["y"] <= 3 * h // 4).all()),
        "image_mask_2": (2, lr, None, lambda k: (k["x"] >= w // 4).all() and (k["x"] <= 3 * w // 4).all()),
        # the lookup floors (y * scale, x * scale) (orb_extractor.cc:168-170), so a keypoint may sit up to one level-0 pixel
        # inside the hard-edged disc in the upper-left quadrant (the reference draws its disc with LINE_AA)
        "image_mask_3": (1, disc, None, lambda k: (np.hypot(k["x"] - w // 2, k["y"] - h // 2) >= 320 - 1.5).all()),
        "rectangle_mask_1": (1, None, r1, lambda k: (k["y"] >= h // 5).all() and (k["y"] <= 4 * h // 5).all()),
        "rectangle_mask_2": (2, None, r2, lambda k: (k["x"] >= w // 5).all() and (k["x"] <= 4 * w // 5).all()),
        "rectangle_mask_3": (2, None, r1 + r2, lambda k: (k["x"] >= w // 5).all() and (k["x"] <= 4 * w // 5).all()
                             and (k["y"] >= h // 5).all() and (k["y"] <= 4 * h // 5).all()),
    }


@pytest.mark.parametrize("name", list(_cases()))
def test_oracle_satisfies_the_reference_assertions(name):
    i, mask, rects, ok = _cases()[name]
    img = _img(i)
    m = mask if rects is None else _rect_mask(rects, img.shape[1], img.shape[0])
    k, d, _ = O.orb_extract(img, mask=m, min_area=1000)
    assert len(k) > 1000 and d.shape == (len(k), 32) and d.dtype == np.uint8
    assert ok(k), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_cases()))
def test_device_equals_oracle_on_the_reference_images(name):
    from stella_vslam_amd import feature as F
    i, mask, rects, ok = _cases()[name]
    img = _img(i)
    ext = F.orb_extractor(F.orb_params("ORB setting for test"), min_area=1000, mask_rects=rects or ())
    kg, dg = ext.extract(img, mask)
    m = mask if rects is None else _rect_mask(rects, img.shape[1], img.shape[0])
    assert rects is None or np.array_equal(m, ext._rectangle_mask(img.shape[1], img.shape[0]))
    ko, do, _ = O.orb_extract(img, mask=m, min_area=1000)
    assert len(kg) == len(ko) and np.array_equal(kg, ko) and np.array_equal(dg, do)
    assert ok(kg)


# ---- the reference's three toy samples at their own sizes (orb_extractor.cc:25-77 and 330-357): white image, black rectangle,
#      every keypoint within 2 x scale of the rectangle's free corner (cv::rectangle fills both corner points inclusively)
TOYS = {"toy_sample_1": ((600, 600), (300, 300, 600, 600), (300, 300)),          # (rows, cols), (x0, y0, x1, y1), corner (x, y)
        "toy_sample_2": ((2000, 2000), (0, 0, 1800, 1800), (1800, 1800)),
        "toy_sample_3": ((1200, 600), (300, 600, 600, 1200), (300, 600))}


def _toy(name):
    (rows, cols), (x0, y0, x1, y1), corner = TOYS[name]
    img = np.full((rows, cols), 255, np.uint8)
    img[y0:min(y1 + 1, rows), x0:min(x1 + 1, cols)] = 0
    return img, corner


def _near_corner(k, corner):
    sf = O.scale_tables(1.2, 8)[0]
    tol = 2.0 * sf[k["octave"]]
    return (np.abs(k["x"] - corner[0]) <= tol).all() and (np.abs(k["y"] - corner[1]) <= tol).all()


@pytest.mark.parametrize("name", list(TOYS))
def test_oracle_toy_samples(name):
    img, corner = _toy(name)
    k, d, _ = O.orb_extract(img, min_area=1000)
    assert len(k) > 0 and d.shape == (len(k), 32) and _near_corner(k, corner)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(TOYS))
def test_device_toy_samples_equal_oracle(name):
    from stella_vslam_amd import feature as F
    img, corner = _toy(name)
    kg, dg = F.orb_extractor(F.orb_params("ORB setting for test"), min_area=1000).extract(img)
    ko, do, _ = O.orb_extract(img, min_area=1000)
    assert len(kg) == len(ko) > 0 and np.array_equal(kg, ko) and np.array_equal(dg, do) and _near_corner(kg, corner)
