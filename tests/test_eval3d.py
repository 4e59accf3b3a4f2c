"""the 3D COCO-style evaluator and the result-list NMS against the reference's own
pycocotools_local/cocoeval.py (is3D=True) and coco_utils.nms_3d_python outputs (tests/golden/eval3d*.{npz,json},
written by tests/golden/make_golden.py)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_cocoeval3d_matches_reference():
    import mr3d_amd  # noqa: F401
    from mr3d_amd.evaluation import COCOeval3D
    g = np.load(os.path.join(HERE, 'golden', 'eval3d.npz'))
    inp = json.load(open(os.path.join(HERE, 'golden', 'eval3d_input.json')))
    E = COCOeval3D(inp['gts'], inp['dts']).evaluate().accumulate()
    assert np.array_equal(E.iou_thrs.round(6), g['iou_thrs'].round(6)) and E.max_dets == g['max_dets'].tolist()
    np.testing.assert_allclose(E.eval['precision'], g['precision'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(E.eval['recall'], g['recall'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(E.eval['scores'], g['scores'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(E.summarize(), g['stats'], rtol=0, atol=1e-12)


def test_result_list_nms_matches_reference_and_json_format():
    import mr3d_amd  # noqa: F401
    from mr3d_amd.evaluation import nms_3d_records, det2json3D, apply_nms_3d, xyxyzz2xywhzd
    g = np.load(os.path.join(HERE, 'golden', 'eval3d_nms.npz'))
    kept = nms_3d_records(list(range(len(g['boxes']))), g['boxes'], 0.1)
    assert kept == g['kept'].tolist()
    assert nms_3d_records([], np.zeros((0, 7)), 0.1) == []
    assert xyxyzz2xywhzd([10, 20, 19, 39, 5, 9]) == [10, 20, 10, 20, 5, 5]
    res = [[np.array([[10, 20, 19, 39, 5, 9, 0.9], [11, 21, 20, 40, 5, 9, 0.8]], np.float32)], [np.zeros((0, 7), np.float32)]]
    js = det2json3D(res, img_ids=[101, 102], cat_ids=[1])
    assert [r['image_id'] for r in js] == [101, 101] and js[0]['bbox'] == [10, 20, 10, 20, 5, 5] and js[0]['category_id'] == 1
    assert len(apply_nms_3d(js, 0.1)) == 1 and apply_nms_3d(js, 0.1)[0]['score'] == js[0]['score']
