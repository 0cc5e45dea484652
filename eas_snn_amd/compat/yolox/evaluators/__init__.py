"""Evaluators (reference: yolox/evaluators/__init__.py): the event-detection inference loop; COCO / VOC / Prophesee metric code is
outside the hot path."""
from .event_evaluator import EventEvaluator

__all__ = ['EventEvaluator']
