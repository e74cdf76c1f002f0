"""Python mirror of the flat part of stella_vslam::data (frame_observation) over the C ABI.

`frame_observation` is built the way system::create_*_frame builds it (system.cc:384-395): undistort the extractor's
keypoints, convert them to bearings and bin them into the matcher grid -- one device call, no host arithmetic.
"""
from __future__ import annotations

import numpy as np

from . import camera as _camera


class frame_observation:
    """data/frame_observation.h:12-38.  `keypt_indices_in_cells_` is kept as CSR (cell = col * num_grid_rows_ + row):
    `cell_off_` (num_grid_cols_ * num_grid_rows_ + 1) and `cell_items_`; `indices_in_cell(col, row)` returns the
    reference's vector for one cell."""

    def __init__(self, camera: _camera.base, keypts, descriptors, num_grid_cols: int = 64, num_grid_rows: int = 48,
                 stereo_x_right=None, depths=None):
        self.num_grid_cols_, self.num_grid_rows_ = int(num_grid_cols), int(num_grid_rows)
        self.descriptors_ = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        und, brg, off, items = camera._observe(keypts, grid=(self.num_grid_cols_, self.num_grid_rows_))
        self.undist_keypts_, self.bearings_, self.cell_off_, self.cell_items_ = und, brg, off, items
        self.stereo_x_right_ = None if stereo_x_right is None else np.ascontiguousarray(stereo_x_right, np.float32)
        self.depths_ = None if depths is None else np.ascontiguousarray(depths, np.float32)

    def indices_in_cell(self, col: int, row: int) -> np.ndarray:
        c = col * self.num_grid_rows_ + row
        return self.cell_items_[self.cell_off_[c]:self.cell_off_[c + 1]]
