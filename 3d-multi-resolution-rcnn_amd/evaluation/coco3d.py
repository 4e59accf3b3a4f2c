"""3D COCO-style evaluation: the post-hoc path of the reference (mmdet/core/evaluation/coco_utils.py:12-100,233-282,
pycocotools_local/cocoeval.py with is3D=True, iouType='bbox').  Host-side numpy -- this is the per-epoch control
plane, not the hot path -- vectorised where the reference loops in Python (its IoU matrix is a double Python loop,
cocoeval.py:261-273; its matching a triple one).

Kept on purpose (all deviations of the reference from stock pycocotools):
  * boxes are [x, y, w, h, z, d] (coco_utils.py:233-242 xyxyzz2xywhzd); IoU = cocoeval.py:263-273;
  * 19 IoU thresholds 0.05:0.05:0.95 (cocoeval.py:870), maxDets [1, 10, 10000] (:93), area ranges of the 2D
    'area' field (:873), crowd gts are 'ignore', greedy matching in score order preferring non-ignored gts;
  * the 29-entry stats vector of summarize() (:790-820).
"""
import numpy as np


def xyxyzz2xywhzd(bbox):
    """coco_utils.py:233-242"""
    b = [float(v) for v in bbox]
    return [b[0], b[1], b[2] - b[0] + 1, b[3] - b[1] + 1, b[4], b[5] - b[4] + 1]


def det2json3D(results, img_ids, cat_ids):
    """coco_utils.py det2json3D: results[idx][label] = [k, 7] arrays (x1,y1,x2,y2,z1,z2,score) -> list of records"""
    out = []
    for idx, per_class in enumerate(results):
        for label, bboxes in enumerate(per_class):
            for b in np.asarray(bboxes):
                out.append(dict(image_id=img_ids[idx], bbox=xyxyzz2xywhzd(b), score=float(b[6]), category_id=cat_ids[label],
                                original_bbox=[float(v) for v in b[:7]]))
    return out


def nms_3d_records(records, boxes, iou_thr):
    """coco_utils.py:245-282: greedy NMS in descending score order over [n, 7] boxes; a box survives iff its IoU with
    every kept box is <= iou_thr.  Returns the kept records in keep order."""
    if len(boxes) == 0:
        return []
    b = np.asarray(boxes, dtype=np.float64)
    vol = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1) * (b[:, 5] - b[:, 4] + 1)
    idxs = np.argsort(-b[:, 6])
    keep = []
    while idxs.shape[0] > 0:
        i = idxs[0]
        keep.append(i)
        r = idxs[1:]
        inter = (np.maximum(0, np.minimum(b[r, 2], b[i, 2]) - np.maximum(b[r, 0], b[i, 0]) + 1) *
                 np.maximum(0, np.minimum(b[r, 3], b[i, 3]) - np.maximum(b[r, 1], b[i, 1]) + 1) *
                 np.maximum(0, np.minimum(b[r, 5], b[i, 5]) - np.maximum(b[r, 4], b[i, 4]) + 1))
        iou = inter / (vol[i] + vol[r] - inter)
        idxs = r[iou <= iou_thr]
    return [records[k] for k in keep]


def apply_nms_3d(json_results, iou_thr=0.1):
    """per-volume NMS over a result list (the reference's apply_nms step before COCOeval)"""
    by_img = {}
    for r in json_results:
        by_img.setdefault(r['image_id'], []).append(r)
    out = []
    for recs in by_img.values():
        out += nms_3d_records(recs, [r['original_bbox'] for r in recs], iou_thr)
    return out


def iou3d_xywhzd(d, g):
    """[D, G] IoU matrix, cocoeval.py:252-273"""
    d, g = np.asarray(d, np.float64).reshape(-1, 6), np.asarray(g, np.float64).reshape(-1, 6)
    dbx, dby, dbz = d[:, 0] + d[:, 2] - 1, d[:, 1] + d[:, 3] - 1, d[:, 4] + d[:, 5] - 1
    gbx, gby, gbz = g[:, 0] + g[:, 2] - 1, g[:, 1] + g[:, 3] - 1, g[:, 4] + g[:, 5] - 1
    ix = np.maximum(0, np.minimum(dbx[:, None], gbx[None]) - np.maximum(d[:, None, 0], g[None, :, 0]) + 1)
    iy = np.maximum(0, np.minimum(dby[:, None], gby[None]) - np.maximum(d[:, None, 1], g[None, :, 1]) + 1)
    iz = np.maximum(0, np.minimum(dbz[:, None], gbz[None]) - np.maximum(d[:, None, 4], g[None, :, 4]) + 1)
    inter = ix * iy * iz
    va, vb = g[:, 2] * g[:, 3] * g[:, 5], d[:, 2] * d[:, 3] * d[:, 5]
    return inter / (va[None] + vb[:, None] - inter)


class COCOeval3D:

    def __init__(self, gts, dts, iou_thrs=None, rec_thrs=None, max_dets=(1, 10, 10000), area_rng=None):
        """gts / dts: lists of records {id, image_id, category_id, bbox [x,y,w,h,z,d], area, iscrowd (gt), score (dt)}"""
        self.iou_thrs = np.linspace(.05, 0.95, 19) if iou_thrs is None else np.asarray(iou_thrs, np.float64)
        self.rec_thrs = np.linspace(.0, 1.00, 101) if rec_thrs is None else np.asarray(rec_thrs, np.float64)
        self.max_dets = sorted(max_dets)
        self.area_rng = [[0, 1e10], [0, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e10]] if area_rng is None else area_rng
        self.area_lbl = ['all', 'small', 'medium', 'large']
        self.img_ids = sorted({g['image_id'] for g in gts})
        self.cat_ids = sorted({g['category_id'] for g in gts})
        self._gts, self._dts = {}, {}
        for g in gts:
            self._gts.setdefault((g['image_id'], g['category_id']), []).append(dict(g, ignore=bool(g.get('iscrowd', 0))))
        for k, d in enumerate(dts):
            d = dict(d)
            d.setdefault('id', k + 1)
            d.setdefault('area', d['bbox'][2] * d['bbox'][3])
            self._dts.setdefault((d['image_id'], d['category_id']), []).append(d)
        self.eval_imgs, self.eval, self.stats = None, None, None

    def _evaluate_img(self, img, cat, rng, max_det):
        gt, dt = self._gts.get((img, cat), []), self._dts.get((img, cat), [])
        if not gt and not dt:
            return None
        g_ig = np.array([g['ignore'] or g['area'] < rng[0] or g['area'] > rng[1] for g in gt], dtype=bool)
        gtind = np.argsort(g_ig, kind='mergesort')
        gt, g_ig = [gt[i] for i in gtind], g_ig[gtind]
        dtind = np.argsort([-d['score'] for d in dt], kind='mergesort')[:max_det]
        dt = [dt[i] for i in dtind]
        crowd = [bool(g.get('iscrowd', 0)) for g in gt]
        T, G, D = len(self.iou_thrs), len(gt), len(dt)
        gtm, dtm, dt_ig = np.zeros((T, G)), np.zeros((T, D)), np.zeros((T, D), dtype=bool)
        if G and D:
            ious = iou3d_xywhzd([d['bbox'] for d in dt], [g['bbox'] for g in gt])
            for ti, t in enumerate(self.iou_thrs):
                for di in range(D):
                    best, m = min(t, 1 - 1e-10), -1
                    for gi in range(G):
                        if gtm[ti, gi] > 0 and not crowd[gi]:
                            continue
                        if m > -1 and not g_ig[m] and g_ig[gi]:
                            break
                        if ious[di, gi] < best:
                            continue
                        best, m = ious[di, gi], gi
                    if m == -1:
                        continue
                    dt_ig[ti, di] = g_ig[m]
                    dtm[ti, di] = gt[m]['id']
                    gtm[ti, m] = dt[di]['id']
        out_rng = np.array([d['area'] < rng[0] or d['area'] > rng[1] for d in dt], dtype=bool).reshape(1, D)
        dt_ig = dt_ig | ((dtm == 0) & np.repeat(out_rng, T, 0))
        return dict(dtm=dtm, dt_ig=dt_ig, g_ig=g_ig, scores=np.array([d['score'] for d in dt], np.float64))

    def evaluate(self):
        md = self.max_dets[-1]
        self.eval_imgs = {(c, a, i): self._evaluate_img(i, c, rng, md) for c in self.cat_ids
                          for a, rng in enumerate(self.area_rng) for i in self.img_ids}
        return self

    def accumulate(self):
        T, R, K, A, M = len(self.iou_thrs), len(self.rec_thrs), len(self.cat_ids), len(self.area_rng), len(self.max_dets)
        precision, recall, scores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
        for k, c in enumerate(self.cat_ids):
            for a in range(A):
                E = [e for e in (self.eval_imgs[(c, a, i)] for i in self.img_ids) if e is not None]
                if not E:
                    continue
                for m, md in enumerate(self.max_dets):
                    sc = np.concatenate([e['scores'][:md] for e in E])
                    inds = np.argsort(-sc, kind='mergesort')
                    sc = sc[inds]
                    dtm = np.concatenate([e['dtm'][:, :md] for e in E], axis=1)[:, inds]
                    dt_ig = np.concatenate([e['dt_ig'][:, :md] for e in E], axis=1)[:, inds]
                    npig = int(np.count_nonzero(~np.concatenate([e['g_ig'] for e in E])))
                    if npig == 0:
                        continue
                    tps = np.cumsum((dtm != 0) & ~dt_ig, axis=1).astype(np.float64)
                    fps = np.cumsum((dtm == 0) & ~dt_ig, axis=1).astype(np.float64)
                    for t in range(T):
                        tp, fp = tps[t], fps[t]
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = np.maximum.accumulate(pr[::-1])[::-1] if nd else pr       # precision envelope
                        idx = np.searchsorted(rc, self.rec_thrs, side='left')
                        q, ss = np.zeros(R), np.zeros(R)
                        ok = idx < nd
                        q[ok], ss[ok] = pr[idx[ok]], sc[idx[ok]]
                        precision[t, :, k, a, m], scores[t, :, k, a, m] = q, ss
        self.eval = dict(precision=precision, recall=recall, scores=scores)
        return self

    def _summarize(self, ap=1, iou_thr=None, area='all', max_dets=10000):
        a = [i for i, l in enumerate(self.area_lbl) if l == area]
        m = [i for i, v in enumerate(self.max_dets) if v == max_dets]
        s = self.eval['precision'] if ap else self.eval['recall']
        if iou_thr is not None:
            s = s[np.where(np.abs(iou_thr - self.iou_thrs) < 0.01)[0]]
        s = s[..., a, m]
        return float(np.mean(s[s > -1])) if (s > -1).any() else -1.0

    def summarize(self):
        """the reference's 29 numbers (cocoeval.py:790-820): AP over all thresholds, AP at each of the 19 thresholds,
        AP small/medium/large, AR at maxDets[0..2], AR small/medium/large"""
        md = self.max_dets
        st = [self._summarize(1)]
        st += [self._summarize(1, iou_thr=t, max_dets=md[2]) for t in np.arange(1, 20) * 0.05]
        st += [self._summarize(1, area=l, max_dets=md[2]) for l in ('small', 'medium', 'large')]
        st += [self._summarize(0, max_dets=v) for v in md[:3]]
        st += [self._summarize(0, area=l, max_dets=md[2]) for l in ('small', 'medium', 'large')]
        self.stats = np.array(st)
        return self.stats
