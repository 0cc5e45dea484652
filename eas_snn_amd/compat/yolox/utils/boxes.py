"""Box utilities used by the head and the evaluators (reference: yolox/utils/boxes.py:33-77; pairwise IoU lives in the SimOTA kernel, csrc/simota.hip)."""
from eas_snn_amd import ops


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    """reference: yolox/utils/boxes.py:33-77 (torchvision NMS inside) -> one HIP call for the whole batch.  Unlike the reference
    the ``prediction`` tensor is left untouched (the reference rewrites its first four columns to corners in place)."""
    return ops.postprocess(prediction, num_classes, conf_thre, nms_thre, class_agnostic)
