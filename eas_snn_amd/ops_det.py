"""Detections: batch-wide post-processing (confidence mask + class-aware NMS), SimOTA assignment, decode + loss terms + gradient
(reference: yolox/utils/boxes.py:33-77; yolox/models/yolo_head.py get_losses / get_assignments).

Part of the operator layer of ``eas_snn_amd.ops`` (split by kernel family; ``ops`` re-exports everything here, so ``ops.<name>`` keeps working)."""
import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr, stream
from .ops_core import _call, _dev, _f32c, _timer_add, _timer_mark


# ------------------------------------------------------------------------------------------------ detections
def postprocess_device(prediction, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    """(rows [B, A, 7], counts int32 [B]) on the device, no host synchronisation (eas_postprocess)."""
    _dev(prediction)
    pred = _f32c(prediction)
    B, A, row = pred.shape
    if row != 5 + num_classes:
        raise ValueError(f'prediction rows have {row} columns, expected 5 + {num_classes}')
    L = _lib.lib()
    out = torch.empty((B, A, 7), dtype=torch.float32, device=pred.device)
    cnt = torch.empty(B, dtype=torch.int32, device=pred.device)
    ws = torch.empty(L.eas_postprocess_workspace_bytes(B, A), dtype=torch.uint8, device=pred.device)
    check(L.eas_postprocess(ptr(pred), B, A, int(num_classes), float(conf_thre), float(nms_thre), int(bool(class_agnostic)), ptr(out),
                            ptr(cnt), ptr(ws), stream()), 'eas_postprocess')
    return out, cnt


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    """``yolox.utils.postprocess`` (boxes.py:33-77): list with one [n, 7] tensor per image (None where nothing is kept)."""
    out, cnt = postprocess_device(prediction, num_classes, conf_thre, nms_thre, class_agnostic)
    counts = cnt.tolist()                                   # the one host synchronisation: the result is a ragged python list
    return [out[i, :n] if n else None for i, n in enumerate(counts)]


def simota_supported(gt_valid, bbox_preds):
    return bbox_preds.is_cuda and bbox_preds.dtype == torch.float32 and gt_valid.shape[1] <= 255 and bbox_preds.shape[1] <= 12288


@torch.no_grad()
def simota_assign(grids, strides, gt_boxes, gt_cls, gt_valid, bbox_preds, obj_preds, cls_preds):
    """SimOTA assignment for the whole batch in one launch (eas_simota_assign): (fg bool [B,A], matched int64 [B,A],
    matched_iou float [B,A]) -- the outputs of YOLOXHead._assign."""
    _dev(grids, strides, gt_boxes, gt_cls, gt_valid, bbox_preds, obj_preds, cls_preds)
    B, A = bbox_preds.shape[:2]
    G, nc = gt_valid.shape[1], cls_preds.shape[-1]
    gr = _f32c(grids.reshape(-1, 2)[:A].float())
    st = _f32c(strides.reshape(-1)[:A].float())
    gb, gc = _f32c(gt_boxes.float()), _f32c(gt_cls.float())
    gv = gt_valid.to(torch.uint8).contiguous()
    bx, ob, cl = _f32c(bbox_preds.float()), _f32c(obj_preds.float().reshape(B, A)), _f32c(cls_preds.float())
    fg = torch.empty((B, A), dtype=torch.uint8, device=bx.device)
    matched = torch.empty((B, A), dtype=torch.int64, device=bx.device)
    miou = torch.empty((B, A), dtype=torch.float32, device=bx.device)
    check(_lib.lib().eas_simota_assign(ptr(gr), ptr(st), ptr(gb), ptr(gc), ptr(gv), ptr(bx), ptr(ob), ptr(cl), B, G, A, nc, ptr(fg),
                                       ptr(matched), ptr(miou), stream()), 'eas_simota_assign')
    return fg.bool(), matched, miou


_ANCHOR_CACHE = {}


def _anchor_tables(hws, strides, device):
    """grids [A,2] and strides [A] of the head levels (cached per geometry and device)"""
    key = (tuple(hws), tuple(float(s_) for s_ in strides), str(device))
    t = _ANCHOR_CACHE.get(key)
    if t is None:
        gs, ss = [], []
        for (h, w), s_ in zip(hws, strides):
            yv, xv = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
            gs.append(torch.stack((xv, yv), 2).reshape(-1, 2).float())
            ss.append(torch.full((h * w,), float(s_)))
        t = _ANCHOR_CACHE[key] = (torch.cat(gs).to(device), torch.cat(ss).to(device))
    return t


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _DetLossFn(torch.autograd.Function):
    """Decode + SimOTA assignment + loss terms + their gradient for the raw head maps of all levels: five launches forward
    (eas_det_decode, eas_simota_assign_rows, eas_det_loss x2) plus a few tiny label ops, one multiply backward."""

    @staticmethod
    def forward(ctx, labels, strides, nc, use_l1, *raw):
        ctx.set_materialize_grads(False)      # a result nobody differentiates arrives as None in backward, not as a zero tensor
        L = len(raw) // 3
        regs, objs, clss = [_f32c(t) for t in raw[0::3]], [_f32c(t) for t in raw[1::3]], [_f32c(t) for t in raw[2::3]]
        _dev(labels, *regs)
        lib = _lib.lib()
        dev = regs[0].device
        B = regs[0].shape[0]
        hws = [tuple(r.shape[-2:]) for r in regs]
        A = sum(h * w for h, w in hws)
        hw_arr = (C.c_int * (2 * L))(*[v for hw in hws for v in hw])
        st_arr = (C.c_float * L)(*[float(s_) for s_ in strides])
        dec = torch.empty((B, A, 5 + nc), dtype=torch.float32, device=dev)
        check(lib.eas_det_decode(L, _ptr_array(regs), _ptr_array(objs), _ptr_array(clss), hw_arr, st_arr, B, nc, ptr(dec), stream()),
              'eas_det_decode')
        labels = _f32c(labels.float())
        G = labels.shape[1]
        if B <= 1024 and labels.shape[2] == 5:
            # nlabel, the valid-row mask, class / box columns and the label count as ONE launch instead of ten tiny tensor operators
            gt_valid = torch.empty((B, G), dtype=torch.uint8, device=dev)
            gt_cls = torch.empty((B, G), dtype=torch.float32, device=dev)
            gt_boxes = torch.empty((B, G, 4), dtype=torch.float32, device=dev)
            num_gts = torch.empty((), dtype=torch.float32, device=dev)
            check(lib.eas_det_labels(ptr(labels), B, G, ptr(gt_valid), ptr(gt_cls), ptr(gt_boxes), ptr(num_gts), stream()), 'eas_det_labels')
        else:
            nlabel = (labels.sum(dim=2) > 0).sum(dim=1)
            gt_valid = (torch.arange(G, device=dev)[None] < nlabel[:, None]).to(torch.uint8)
            gt_cls, gt_boxes = labels[:, :, 0].contiguous(), labels[:, :, 1:5].contiguous()
            num_gts = nlabel.sum().float()
        grids, svec = _anchor_tables(hws, strides, dev)
        fg = torch.empty((B, A), dtype=torch.uint8, device=dev)
        matched = torch.empty((B, A), dtype=torch.int64, device=dev)
        miou = torch.empty((B, A), dtype=torch.float32, device=dev)
        check(lib.eas_simota_assign_rows(ptr(grids), ptr(svec), ptr(gt_boxes), ptr(gt_cls), ptr(gt_valid), ptr(dec), B, G, A, nc, ptr(fg),
                                         ptr(matched), ptr(miou), stream()), 'eas_simota_assign_rows')
        g_regs, g_objs, g_clss = [torch.empty_like(t) for t in regs], [torch.empty_like(t) for t in objs], [torch.empty_like(t) for t in clss]
        out = torch.empty(7, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.eas_det_loss_workspace_doubles(), dtype=torch.float64, device=dev)
        check(lib.eas_det_loss(L, _ptr_array(regs), _ptr_array(objs), _ptr_array(clss), _ptr_array(g_regs), _ptr_array(g_objs),
                               _ptr_array(g_clss), hw_arr, st_arr, B, nc, ptr(dec), ptr(gt_boxes), ptr(gt_cls), G, ptr(fg), ptr(matched),
                               ptr(miou), ptr(num_gts), int(bool(use_l1)), ptr(out), ptr(ws), stream()), 'eas_det_loss')
        ctx.grads = [g for trip in zip(g_regs, g_objs, g_clss) for g in trip]
        ctx.scale = out[6]
        outs = tuple(out[i] for i in range(6))
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g_total, *_unused):
        grads = ctx.grads
        if g_total is None:
            return (None,) * (4 + len(grads))
        torch._foreach_mul_(grads, g_total * ctx.scale)
        return (None, None, None, None) + tuple(grads)


def det_decode_eval_supported(raws):
    return (len(raws) <= 4 and all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 for trip in raws for t in trip)
            and sum(r.shape[-1] * r.shape[-2] for r, _, _ in raws) < (1 << 24))


@torch.no_grad()
def det_decode_eval(raws, strides, num_classes):
    """inference output [B, A, 5 + num_classes] of YOLOXHead from the raw (reg, obj, cls) maps of its levels: sigmoid on objectness and
    classes, levels along the anchors, boxes decoded -- one launch (eas_det_decode_eval)."""
    regs, objs, clss = [_f32c(r) for r, _, _ in raws], [_f32c(o) for _, o, _ in raws], [_f32c(c) for _, _, c in raws]
    _dev(*regs)
    L = len(regs)
    B = regs[0].shape[0]
    hws = [tuple(r.shape[-2:]) for r in regs]
    A = sum(h * w for h, w in hws)
    hw_arr = (C.c_int * (2 * L))(*[v for hw in hws for v in hw])
    st_arr = (C.c_float * L)(*[float(s_) for s_ in strides])
    dec = torch.empty((B, A, 5 + num_classes), dtype=torch.float32, device=regs[0].device)
    _call('eas_det_decode', 4 * 2 * dec.numel(), _lib.lib().eas_det_decode_eval, L, _ptr_array(regs), _ptr_array(objs), _ptr_array(clss), hw_arr,
          st_arr, B, int(num_classes), ptr(dec), stream())
    return dec


def det_loss_supported(raw_regs, labels, loss_type):
    A = sum(r.shape[-1] * r.shape[-2] for r in raw_regs)
    return (raw_regs[0].is_cuda and raw_regs[0].dtype == torch.float32 and len(raw_regs) <= 4 and A <= 12288 and labels.shape[1] <= 255
            and loss_type == 'iou' and all(r.dim() == 4 for r in raw_regs))


def det_loss(regs, objs, clss, labels, strides, num_classes, use_l1):
    """(total, 5*iou, obj, cls, l1, num_fg/num_gts) of YOLOXHead.get_losses from the raw head maps of every level."""
    raw = [t for trip in zip(regs, objs, clss) for t in trip]
    return _DetLossFn.apply(labels, tuple(float(s_) for s_ in strides), int(num_classes), bool(use_l1), *raw)
