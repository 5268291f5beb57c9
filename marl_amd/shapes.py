"""Target-shape preprocessing: PNG image -> grid of cell centres.

Restates the image pipeline of the reference config module
(marl_llm/cfg/assembly_cfg.py:35-137) with PIL + numpy only (the reference uses
cv2 + matplotlib, neither of which this package depends on):

    grayscale -> Otsu binarisation -> crop to the black bounding box -> flip rows
    -> tile into 36-pixel cells -> keep cells that are 100 % black -> centre on the
    mean -> scale so the grid height is 2.2 -> (l_cell, grid_coords)

The result has the layout of the reference's ``fig/results.pkl`` entries
``l_cell`` and ``grid_coords`` (assembly_cfg.py:133-135).  ``binary_image`` and
``shape_bound_points`` are render-only in the reference and are not produced.

The seven grids derived from the reference's ``fig/*.png`` are committed as
``marl_amd/data/shapes.npz`` (data, produced by :func:`build_shapes_npz`), so
neither the PNGs nor PIL are needed at run time.
"""
from __future__ import annotations

import glob
import os
from typing import List, Sequence, Tuple

import numpy as np

GRID_SIZE_PX = 36          # assembly_cfg.py:60
TARGET_HEIGHT = 2.2        # assembly_cfg.py:99

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "shapes.npz")


def otsu_threshold(gray: np.ndarray) -> int:
    """Otsu threshold of a uint8 image: the level t maximising the between-class
    variance of {v <= t} vs {v > t} over the 256-bin histogram (first maximum).
    Stands in for cv2.threshold(..., THRESH_BINARY + THRESH_OTSU), assembly_cfg.py:48."""
    hist = np.bincount(gray.ravel(), minlength=256).astype(np.float64)
    total = hist.sum()
    levels = np.arange(256, dtype=np.float64)
    w0 = np.cumsum(hist)
    w1 = total - w0
    m0 = np.cumsum(hist * levels)
    mt = m0[-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        mu0 = m0 / w0
        mu1 = (mt - m0) / w1
        sigma_b = w0 * w1 * (mu0 - mu1) ** 2
    sigma_b[~np.isfinite(sigma_b)] = 0.0
    return int(np.argmax(sigma_b))


def grid_from_gray(gray: np.ndarray) -> Tuple[float, np.ndarray]:
    """(l_cell, grid_coords[n_g, 2] float64) from a uint8 grayscale image.
    Follows assembly_cfg.py:48-102 step by step."""
    t = otsu_threshold(gray)
    binary = np.where(gray > t, 255, 0).astype(np.uint8)            # cfg:48
    black = np.argwhere(binary == 0)                                  # cfg:51
    min_y, min_x = black.min(axis=0)
    max_y, max_x = black.max(axis=0)
    binary = binary[min_y:max_y + 1, min_x:max_x + 1]                 # cfg:54
    height, width = binary.shape
    binary = binary[::-1, :]                                          # cfg:57 (row flip)
    g = GRID_SIZE_PX
    coords: List[List[float]] = []
    for i in range(g, height - g, g):                                 # cfg:64-80
        for j in range(g, width - g, g):
            if np.all(binary[i:i + g, j:j + g] == 0):
                coords.append([j + g / 2, i + g / 2])
    xy = np.array(coords, dtype=np.float64)
    xy[:, 0] -= np.mean(xy[:, 0])                                     # cfg:87-90
    xy[:, 1] -= np.mean(xy[:, 1])
    real_h = np.max(xy[:, 1]) - np.min(xy[:, 1])                      # cfg:95-101
    h_scale = TARGET_HEIGHT / real_h
    return float(g * h_scale), h_scale * xy                           # cfg:102,134


def grid_from_png(path: str) -> Tuple[float, np.ndarray]:
    from PIL import Image
    gray = np.asarray(Image.open(path).convert("L"), dtype=np.uint8)  # cfg:47
    return grid_from_gray(gray)


def sorted_pngs(folder: str) -> List[str]:
    """*.png sorted by integer stem, as assembly_cfg.py:145-146."""
    return sorted(glob.glob(os.path.join(folder, "*.png")),
                  key=lambda x: int(os.path.basename(x).split(".")[0]))


def build_shapes_npz(folder: str, out_path: str = _DATA) -> None:
    """Run the pipeline over every PNG of ``folder`` and store the grids."""
    paths = sorted_pngs(folder)
    out = {"names": np.array([os.path.basename(p) for p in paths])}
    l_cells = []
    for k, p in enumerate(paths):
        l_cell, grid = grid_from_png(p)
        l_cells.append(l_cell)
        out[f"grid_{k}"] = grid
    out["l_cell"] = np.array(l_cells, dtype=np.float64)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    np.savez_compressed(out_path, **out)


def load_shapes(path: str = _DATA) -> Tuple[List[float], List[np.ndarray]]:
    """(l_cells, grid_coords) with grid_coords[k] of shape (n_g_k, 2), float64 —
    the two ``results.pkl`` entries the environment reads (assembly.py:114-115)."""
    z = np.load(path, allow_pickle=False)
    l_cells = [float(v) for v in z["l_cell"]]
    grids = [np.array(z[f"grid_{k}"], dtype=np.float64) for k in range(len(l_cells))]
    return l_cells, grids


def results_dict(path: str = _DATA) -> dict:
    """A dict with the keys of the reference's results.pkl (assembly_cfg.py:27-32).
    The two render-only entries are filled with placeholders of the right length."""
    l_cells, grids = load_shapes(path)
    return {
        "l_cell": l_cells,
        "grid_coords": grids,
        "binary_image": [np.zeros((1, 1), dtype=np.uint8) for _ in l_cells],
        "shape_bound_points": [np.array([g[:, 0].min(), g[:, 0].max(), g[:, 1].min(), g[:, 1].max()])
                               for g in grids],
    }


if __name__ == "__main__":  # python -m marl_amd.shapes <fig folder>
    import sys
    build_shapes_npz(sys.argv[1])
    lc, gr = load_shapes()
    print([g.shape[0] for g in gr], lc)
