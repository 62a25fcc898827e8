#!/usr/bin/env python
"""Container-only: record what the REFERENCE's own metric code returns for a seeded label pair
(tests/golden/metrics.npz).  TEST INFRASTRUCTURE.  Loads /root/reference/util/util.py by path (nothing
is copied) and uses sklearn's confusion_matrix exactly as test_segmentation.py:176 calls it.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_metrics.py
"""
import importlib.util
import os
import sys

import numpy as np
from sklearn.metrics import confusion_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import detweights as dw  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_util", "/root/reference/util/util.py")
ref_util = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_util)

shape = (2, 48, 64)
label = dw.det_labels("metrics_gt", shape, 9).numpy()
pred = dw.det_labels("metrics_pred", shape, 9).numpy()
pred = np.where(dw.det_labels("metrics_mix", shape, 3).numpy() == 0, pred, label)  # ~2/3 agreement
label[label == 7] = 6  # class 7 never occurs in the ground truth -> NaN recall
pred[pred == 5] = 4  # class 5 is never predicted -> NaN precision
pred[(pred == 8) | (label == 8)] = 0
label[label == 8] = 0  # class 8 absent from both -> NaN everywhere
conf = confusion_matrix(y_true=label.flatten(), y_pred=pred.flatten(), labels=[0, 1, 2, 3, 4, 5, 6, 7, 8])
prec, rec, iou = ref_util.compute_results(conf)
out = os.path.join(HERE, "..", "tests", "golden", "metrics.npz")
np.savez_compressed(out, label=label.astype(np.int64), pred=pred.astype(np.int32), conf=conf.astype(np.int64),
                    precision=prec, recall=rec, iou=iou)
print("wrote", os.path.normpath(out), conf.sum(), np.round(iou, 3))
