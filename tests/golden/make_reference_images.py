#!/usr/bin/env python3
"""Derives the two real-image fixtures from the REFERENCE's own test data (run in the build container, where /root/reference
exists and PIL is importable):  python tests/golden/make_reference_images.py

test/stella_vslam/feature/orb_extractor.cc:79-330 runs the extractor on test/data/equirectangular_image_00{1,2}.jpg with
cv::imread(.., IMREAD_GRAYSCALE).  OpenCV is not available; PIL decodes the same JPEGs and converts to 8-bit luma (ITU-R 601,
a decoder that may differ from OpenCV's by +-1 LSB).  The grey images are stored losslessly (PNG) so that the tests -- the
reference's invariants on the oracle, and bit-parity of the device path against the oracle on REAL imagery -- do not depend on
/root/reference or on a JPEG decoder's version at run time."""
import os
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/test/data"
for i in (1, 2):
    g = Image.open(f"{SRC}/equirectangular_image_00{i}.jpg").convert("L")
    g.save(os.path.join(HERE, f"equirect_00{i}_gray.png"), optimize=True)
    print(g.size, os.path.getsize(os.path.join(HERE, f"equirect_00{i}_gray.png")))
