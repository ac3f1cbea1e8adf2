"""Seeded synthetic inputs for the hot path.

The reference ships no fixtures for its parsers or its engine, so parity tests and
bench.py feed the GPU path and the CPU oracle with the SAME seeded synthetic tensors built here:

* ``paf_maps``  — conf ``[B,19,H,W]`` + paf ``[B,38,H,W]`` heat-maps of N posed 18-joint COCO skeletons,
  laid out exactly as ``hyperpose::parser::paf::process`` consumes them (reference src/paf.cpp:300-312;
  limb/channel tables src/coco.hpp:10-51).
* ``images_u8`` — uniform u8 HWC BGR frames, the input of ``dnn::tensorrt::inference``
  (reference src/tensorrt.cpp:436-461).

Only numpy is used; nothing here touches the GPU.
"""
from __future__ import annotations

import numpy as np

# reference src/coco.hpp:10-30 (PAF channel pairs) and :32-51 (part index pairs)
COCOPAIRS_NET = [(12, 13), (20, 21), (14, 15), (16, 17), (22, 23), (24, 25), (0, 1), (2, 3), (4, 5), (6, 7),
                 (8, 9), (10, 11), (28, 29), (30, 31), (34, 35), (32, 33), (36, 37), (18, 19), (26, 27)]
COCOPAIRS = [(1, 2), (1, 5), (2, 3), (3, 4), (5, 6), (6, 7), (1, 8), (8, 9), (9, 10), (1, 11), (11, 12),
             (12, 13), (1, 0), (0, 14), (14, 16), (0, 15), (15, 17), (2, 16), (5, 17)]

# 18-joint COCO template (x right, y down), body height ~ 1.
# nose neck Rsho Relb Rwri Lsho Lelb Lwri Rhip Rkne Rank Lhip Lkne Lank Reye Leye Rear Lear
_TEMPLATE = np.array([
    (0.00, -0.40), (0.00, -0.30), (-0.12, -0.30), (-0.17, -0.12), (-0.19, 0.04), (0.12, -0.30),
    (0.17, -0.12), (0.19, 0.04), (-0.07, 0.05), (-0.08, 0.28), (-0.08, 0.50), (0.07, 0.05),
    (0.08, 0.28), (0.08, 0.50), (-0.04, -0.44), (0.04, -0.44), (-0.08, -0.41), (0.08, -0.41),
], dtype=np.float64)


def rng_for(config_index: int, salt: int = 0) -> np.random.Generator:
    """numpy.random.Generator(PCG64(20240 + config index)); ``salt`` separates streams."""
    return np.random.Generator(np.random.PCG64(20240 + config_index + 1000 * salt))


def skeletons(rng: np.random.Generator, n_people: int, rows: int, cols: int,
              scale_range=(14.0, 34.0), jitter=0.03) -> np.ndarray:
    """``[n_people,18,2]`` continuous (x=col, y=row) joint positions in feature-map pixels."""
    out = np.zeros((n_people, 18, 2))
    for p in range(n_people):
        s = rng.uniform(*scale_range)
        th = rng.uniform(-0.5, 0.5)
        rot = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        pts = (_TEMPLATE + rng.normal(0, jitter, _TEMPLATE.shape)) @ rot.T * s
        lo, hi = pts.min(0), pts.max(0)
        # keep the whole skeleton inside the map with a 1.5 px margin when it fits
        tx = rng.uniform(1.5 - lo[0], max(1.5 - lo[0] + 1e-3, cols - 2.5 - hi[0]))
        ty = rng.uniform(1.5 - lo[1], max(1.5 - lo[1] + 1e-3, rows - 2.5 - hi[1]))
        out[p] = pts + (tx, ty)
    return out


def paf_maps(rng: np.random.Generator, batch: int, rows: int = 46, cols: int = 54, people=(1, 2, 4, 8),
             sigma: float = 1.0, band: float = 1.0, noise: float = 0.01, drop_joint_prob: float = 0.05,
             scale_range=(14.0, 34.0)):
    """Synthetic (conf ``[B,19,rows,cols]``, paf ``[B,38,rows,cols]``) float32 + the joint list per frame."""
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
    conf = np.zeros((batch, 19, rows, cols))
    paf = np.zeros((batch, 38, rows, cols))
    truth = []
    for b in range(batch):
        n = int(people[b % len(people)])
        sk = skeletons(rng, n, rows, cols, scale_range=scale_range)
        visible = rng.uniform(size=(n, 18)) >= drop_joint_prob
        truth.append((sk, visible))
        for p in range(n):
            for k in range(18):
                if visible[p, k]:
                    d2 = (xx - sk[p, k, 0]) ** 2 + (yy - sk[p, k, 1]) ** 2
                    conf[b, k] += np.exp(-d2 / (2 * sigma * sigma))
        conf[b, 18] = 1.0 - conf[b, :18].max(0)
        count = np.zeros((19, rows, cols))
        for p in range(n):
            for l, ((p1, p2), (cx, cy)) in enumerate(zip(COCOPAIRS, COCOPAIRS_NET)):
                if not (visible[p, p1] and visible[p, p2]):
                    continue
                a, c = sk[p, p1], sk[p, p2]
                v = c - a
                ln = np.hypot(*v)
                if ln < 1e-6:
                    continue
                u = v / ln
                rx, ry = xx - a[0], yy - a[1]
                along = rx * u[0] + ry * u[1]
                perp = np.abs(rx * u[1] - ry * u[0])
                m = (along >= -band) & (along <= ln + band) & (perp <= band)
                paf[b, cx][m] += u[0]
                paf[b, cy][m] += u[1]
                count[l][m] += 1
        for l, (cx, cy) in enumerate(COCOPAIRS_NET):
            m = count[l] > 1
            paf[b, cx][m] /= count[l][m]
            paf[b, cy][m] /= count[l][m]
    conf += rng.normal(0, noise, conf.shape)
    paf += rng.normal(0, noise, paf.shape)
    return conf.astype(np.float32), paf.astype(np.float32), truth


def images_u8(rng: np.random.Generator, batch: int, h: int, w: int) -> np.ndarray:
    """Uniform[0,255] u8 ``[B,h,w,3]`` HWC BGR frames (already network-sized: cv::resize is then a copy)."""
    return rng.integers(0, 256, size=(batch, h, w, 3), dtype=np.uint8)


# ---------------------------------------------------------------------------------------------------------------
# PoseProposal synthetic tensors (SURVEY.md 8d "PPN synthetic"): reference src/pose_proposal.cpp:12-41.
COCOPAIR_STD = [(1, 8), (8, 9), (9, 10), (1, 11), (11, 12), (12, 13), (1, 2), (2, 3), (3, 4), (1, 5), (5, 6), (6, 7),
                (1, 0), (0, 14), (0, 15), (14, 16), (15, 17)]


def ppn_maps(rng: np.random.Generator, batch: int, net: int = 384, grid: int = 12, people=(1, 2, 3, 4), nbr: int = 9,
             spurious: float = 0.01):
    """7 tensors per frame in parser order (conf_point, conf_iou, x, y, w, h, edge): 6 x [B,18,grid,grid] and
    edge [B,17,nbr,nbr,grid,grid]; x/y/w/h in input pixels as the network's restore_coor emits them
    (hyperpose/Model/pose_proposal/model.py:111-119).  True edges are U(0.8,1) (no exact ties), the rest U(0,0.04)."""
    cell = net / grid
    conf = rng.uniform(0.0, 0.08, (batch, 18, grid, grid))
    iou = rng.uniform(0.0, 1.0, (batch, 18, grid, grid))
    gy, gx = np.mgrid[0:grid, 0:grid]
    x = np.broadcast_to((gx + 0.5) * cell, (batch, 18, grid, grid)) + rng.normal(0, 4, (batch, 18, grid, grid))
    y = np.broadcast_to((gy + 0.5) * cell, (batch, 18, grid, grid)) + rng.normal(0, 4, (batch, 18, grid, grid))
    w = rng.uniform(20, 90, (batch, 18, grid, grid))
    h = rng.uniform(20, 90, (batch, 18, grid, grid))
    edge = rng.uniform(0.0, 0.04, (batch, 17, nbr, nbr, grid, grid))
    spur = rng.uniform(size=conf.shape) < spurious
    conf[spur] = rng.uniform(0.1, 0.5, int(spur.sum()))
    for b in range(batch):
        n = int(people[b % len(people)])
        sk = skeletons(rng, n, net, net, scale_range=(110.0, 260.0))
        for p in range(n):
            cells = np.clip(np.floor(sk[p] / cell).astype(int), 0, grid - 1)  # [18,(cx,cy)]
            for k in range(18):
                cx, cy = cells[k]
                conf[b, k, cy, cx] = rng.uniform(0.6, 1.0)
                x[b, k, cy, cx], y[b, k, cy, cx] = np.clip(sk[p, k], 0, net - 1)
                w[b, k, cy, cx], h[b, k, cy, cx] = rng.uniform(30, 80, 2)
            for l, (p1, p2) in enumerate(COCOPAIR_STD):
                dx, dy = cells[p2] - cells[p1]
                if abs(dx) <= nbr // 2 and abs(dy) <= nbr // 2:
                    edge[b, l, dy + nbr // 2, dx + nbr // 2, cells[p1][1], cells[p1][0]] = rng.uniform(0.8, 1.0)
    f = lambda a: np.ascontiguousarray(a, np.float32)
    return [f(conf), f(iou), f(x), f(y), f(w), f(h), f(edge)]


# ---------------------------------------------------------------------------------------------------------------
# PifPaf synthetic fields (SURVEY.md 8d "PifPaf synthetic"): OpenPifPaf 17-keypoint topology, bones as in
# reference src/pifpaf_decoder/openpifpaf_postprocessor.cpp:64-84 (1-based joint indices).
PIFPAF_BONES = [(16, 14), (14, 12), (17, 15), (15, 13), (12, 13), (6, 12), (7, 13), (6, 7), (6, 8), (7, 9), (8, 10),
                (9, 11), (2, 3), (1, 2), (1, 3), (2, 4), (3, 5), (4, 6), (5, 7)]
# 17 OpenPifPaf joints picked from the 18-joint template above (nose, l/r eye, l/r ear, l/r shoulder, ...)
_PIFPAF_FROM_18 = [0, 15, 14, 17, 16, 5, 2, 6, 3, 7, 4, 11, 8, 12, 9, 13, 10]


def pifpaf_maps(rng: np.random.Generator, batch: int, fh: int = 49, fw: int = 49, people=(1, 2, 3, 4), noise: float = 0.02):
    """(paf ``[B,19,9,fh,fw]``, pif ``[B,17,5,fh,fw]``) in the argument order of ``parser::pifpaf::process``
    (reference src/pifpaf.cpp:7).  pif = [conf, x, y, b, scale], paf = [conf, x1, y1, x2, y2, b1, b2, s1, s2], all
    positions/scales in field units (stride 8), as the decoder multiplies them by 8."""
    yy, xx = np.mgrid[0:fh, 0:fw].astype(np.float64)
    pif = np.zeros((batch, 17, 5, fh, fw))
    paf = np.zeros((batch, 19, 9, fh, fw))
    pif[:, :, 0] = rng.uniform(0, noise, (batch, 17, fh, fw))
    paf[:, :, 0] = rng.uniform(0, noise, (batch, 19, fh, fw))
    pif[:, :, 1], pif[:, :, 2] = xx, yy
    pif[:, :, 4] = 1.0
    for c in (1, 3):
        paf[:, :, c], paf[:, :, c + 1] = xx, yy
    paf[:, :, 7:9] = 1.0
    for b in range(batch):
        n = int(people[b % len(people)])
        sk18 = skeletons(rng, n, fh, fw, scale_range=(16.0, 34.0))
        for p in range(n):
            sk = sk18[p][_PIFPAF_FROM_18]
            size = float(np.ptp(sk[:, 1]))
            scale = max(0.6, size / 14.0)
            for k in range(17):
                d2 = (xx - sk[k, 0]) ** 2 + (yy - sk[k, 1]) ** 2
                m = d2 <= 2.6 ** 2  # ~21 voting cells: the decoder divides every vote by PIF_NN = 16
                c = 0.95 * np.exp(-d2 / (2 * 2.5 ** 2))
                upd = m & (c > pif[b, k, 0])
                pif[b, k, 0][upd] = c[upd]
                pif[b, k, 1][upd] = sk[k, 0] + rng.normal(0, 0.03, int(upd.sum()))
                pif[b, k, 2][upd] = sk[k, 1] + rng.normal(0, 0.03, int(upd.sum()))
                pif[b, k, 4][upd] = scale * rng.uniform(0.9, 1.1, int(upd.sum()))
            for l, (j1, j2) in enumerate(PIFPAF_BONES):
                a, c2 = sk[j1 - 1], sk[j2 - 1]
                v = c2 - a
                ln = np.hypot(*v)
                u = v / max(ln, 1e-6)
                rx, ry = xx - a[0], yy - a[1]
                along = rx * u[0] + ry * u[1]
                perp = np.abs(rx * u[1] - ry * u[0])
                m = (along >= -1.0) & (along <= ln + 1.0) & (perp <= 1.0)
                conf = 0.9 * np.exp(-perp ** 2 / 2.0) * rng.uniform(0.85, 1.0, perp.shape)
                upd = m & (conf > paf[b, l, 0])
                k = int(upd.sum())
                paf[b, l, 0][upd] = conf[upd]
                paf[b, l, 1][upd] = a[0] + rng.normal(0, 0.03, k)
                paf[b, l, 2][upd] = a[1] + rng.normal(0, 0.03, k)
                paf[b, l, 3][upd] = c2[0] + rng.normal(0, 0.03, k)
                paf[b, l, 4][upd] = c2[1] + rng.normal(0, 0.03, k)
                paf[b, l, 7][upd] = scale
                paf[b, l, 8][upd] = scale
    # keep every regressed position inside the field so that the reference's unchecked index arithmetic
    # (openpifpaf_postprocessor.cpp:693,745) stays in bounds
    pif[:, :, 1] = np.clip(pif[:, :, 1], 0, fw - 1.01)
    pif[:, :, 2] = np.clip(pif[:, :, 2], 0, fh - 1.01)
    for c in (1, 3):
        paf[:, :, c] = np.clip(paf[:, :, c], 0, fw - 1.01)
        paf[:, :, c + 1] = np.clip(paf[:, :, c + 1], 0, fh - 1.01)
    return np.ascontiguousarray(paf, np.float32), np.ascontiguousarray(pif, np.float32)
