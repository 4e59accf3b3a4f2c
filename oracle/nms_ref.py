"""numpy restatement of the reference's only CPU NMS3D, `nms_3d_python` of
mmdet/core/evaluation/coco_utils.py:245-282 (greedy by descending score, true 3D IoU with the inclusive +1
convention, keep `iou <= thr`).  TEST INFRASTRUCTURE: pinned by tests/golden/nms3d.npz (fixtures produced by the
imported reference function, tests/golden/make_golden.py) in tests/test_oracle_golden.py, and timed on the GPU
box's host cores as bench.py's `cpu_baseline` C2 leg.  Never imported by the product."""
import numpy as np


def nms3d_numpy(dets, iou_thr):
    """dets [n,7] = x1,y1,x2,y2,z1,z2,score -> indices of the kept rows in pick order (descending score).
    float64 arithmetic like the reference (it converts Python lists with np.array)."""
    d = np.asarray(dets, dtype=np.float64)
    if d.shape[0] == 0:
        return np.zeros((0,), np.int64)
    lo = d[:, [0, 1, 4]]
    hi = d[:, [2, 3, 5]]
    vol = np.prod(hi - lo + 1.0, axis=1)
    alive = np.argsort(-d[:, 6])
    picked = []
    while alive.size:
        head, rest = alive[0], alive[1:]
        picked.append(head)
        if not rest.size:
            break
        edge = np.minimum(hi[rest], hi[head]) - np.maximum(lo[rest], lo[head]) + 1.0
        inter = np.prod(np.maximum(edge, 0.0), axis=1)
        iou = inter / (vol[head] + vol[rest] - inter)
        alive = rest[iou <= iou_thr]
    return np.asarray(picked, np.int64)
