"""TEST INFRASTRUCTURE -- numpy restatement of the reference's detection post-processing.

``postprocess`` follows yolox/utils/boxes.py:33-77 line by line.  The NMS it calls lives in torchvision (0.16.1 pinned by
pip-requirements.txt:147), which is neither in the reference tree nor installed here: ``nms`` and ``batched_nms`` restate
torchvision's published algorithm (ops/boxes.py: batched_nms -> coordinate trick up to 20000 box coordinates on a CUDA
device / 4000 on the CPU, per-class loop above; csrc/ops/cuda/nms_kernel.cu: stable descending sort, devIoU in float32).
Parity unpinned: no torchvision to run.  Only tests/ may import this module.
"""
import numpy as np

F = np.float32


def nms(boxes, scores, thr):
    """indices kept, in descending score order (stable)."""
    boxes = np.asarray(boxes, F)
    order = np.argsort(-np.asarray(scores, F), kind='stable')
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    removed = np.zeros(len(boxes), bool)
    keep = []
    for pos, i in enumerate(order):
        if removed[i]:
            continue
        keep.append(i)
        rest = order[pos + 1:]
        left = np.maximum(boxes[i, 0], boxes[rest, 0]); right = np.minimum(boxes[i, 2], boxes[rest, 2])
        top = np.maximum(boxes[i, 1], boxes[rest, 1]); bottom = np.minimum(boxes[i, 3], boxes[rest, 3])
        inter = np.maximum(right - left, F(0)) * np.maximum(bottom - top, F(0))
        with np.errstate(divide='ignore', invalid='ignore'):
            over = (inter / (area[i] + area[rest] - inter)) > F(thr)
        removed[rest[over]] = True
    return np.array(keep, dtype=np.int64)


def batched_nms(boxes, scores, idxs, thr, trick_max_numel=20000):
    boxes = np.asarray(boxes, F)
    if boxes.size == 0:
        return np.zeros(0, np.int64)
    if boxes.size > trick_max_numel:                       # _batched_nms_vanilla
        keep_mask = np.zeros(len(boxes), bool)
        for c in np.unique(idxs):
            cur = np.where(idxs == c)[0]
            keep_mask[cur[nms(boxes[cur], scores[cur], thr)]] = True
        kept = np.where(keep_mask)[0]
        return kept[np.argsort(-np.asarray(scores, F)[kept], kind='stable')]
    max_coordinate = boxes.max()                           # _batched_nms_coordinate_trick
    offsets = np.asarray(idxs, F) * (max_coordinate + F(1))
    return nms(boxes + offsets[:, None], scores, thr)


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    prediction = np.array(prediction, F)
    box_corner = np.empty_like(prediction)
    box_corner[:, :, 0] = prediction[:, :, 0] - prediction[:, :, 2] / 2
    box_corner[:, :, 1] = prediction[:, :, 1] - prediction[:, :, 3] / 2
    box_corner[:, :, 2] = prediction[:, :, 0] + prediction[:, :, 2] / 2
    box_corner[:, :, 3] = prediction[:, :, 1] + prediction[:, :, 3] / 2
    prediction[:, :, :4] = box_corner[:, :, :4]
    output = [None for _ in range(len(prediction))]
    for i, image_pred in enumerate(prediction):
        if not image_pred.shape[0]:
            continue
        cls = image_pred[:, 5:5 + num_classes]
        class_pred = cls.argmax(1)                                        # first maximum
        class_conf = cls[np.arange(len(cls)), class_pred]
        conf_mask = image_pred[:, 4] * class_conf >= F(conf_thre)
        det = np.concatenate([image_pred[:, :5], class_conf[:, None], class_pred[:, None].astype(F)], 1)[conf_mask]
        if not det.shape[0]:
            continue
        if class_agnostic:
            keep = nms(det[:, :4], det[:, 4] * det[:, 5], nms_thre)
        else:
            keep = batched_nms(det[:, :4], det[:, 4] * det[:, 5], det[:, 6], nms_thre)
        output[i] = det[keep]
    return output
