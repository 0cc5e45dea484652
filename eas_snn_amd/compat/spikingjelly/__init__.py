"""spikingjelly-compatible namespace backed by the MI355X HIP kernels of eas_snn_amd.

Only the part of ``spikingjelly==0.0.0.0.14`` that EAS-SNN calls is provided
(see eas_snn_amd/compat/spikingjelly/activation_based/*).  This is NOT the
upstream package; it keeps its import names so the reference's call sites
(yolox/utils/utils_snn.py:6, yolox/exp/event_yolox_base.py:12,
yolox/core/trainer.py:116, yolox/evaluators/event_evaluator.py:197) resolve unchanged.
"""
__version__ = '0.0.0.0.14+eas_snn_amd'
