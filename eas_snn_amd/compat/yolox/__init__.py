"""yolox-compatible surface of eas_snn_amd: the ``yolox.exp`` / ``yolox.core.launch`` / ``yolox.models`` /
``yolox.utils`` names that EAS-SNN's tools/train_event.py and tools/eval_event.py import, backed by the
MI355X HIP hot path.  Only the event-detection hot path is provided; COCO/VOC
datasets, evaluators, export tools and logging backends of upstream YOLOX are out of scope."""
__version__ = '0.3.0+eas_snn_amd'
