"""IoU loss (reference: yolox/models/losses.py:10-53)."""
import torch
import torch.nn as nn


def _prod2(t):
    """Product over a last axis of length 2 (box width x height).  ``torch.prod``'s backward inspects the input for zeros
    on the host, which stalls the stream and cannot be captured in a HIP graph; the explicit product has neither problem."""
    return t[..., 0] * t[..., 1]



class IOUloss(nn.Module):
    def __init__(self, reduction='none', loss_type='iou'):
        super().__init__()
        self.reduction = reduction
        self.loss_type = loss_type

    def forward(self, pred, target):
        assert pred.shape[0] == target.shape[0]
        pred, target = pred.view(-1, 4), target.view(-1, 4)
        p_lo, p_hi = pred[:, :2] - pred[:, 2:] / 2, pred[:, :2] + pred[:, 2:] / 2
        t_lo, t_hi = target[:, :2] - target[:, 2:] / 2, target[:, :2] + target[:, 2:] / 2
        tl, br = torch.max(p_lo, t_lo), torch.min(p_hi, t_hi)
        area_p, area_g = _prod2(pred[:, 2:]), _prod2(target[:, 2:])
        en = _prod2((tl < br).type(tl.type()))
        area_i = _prod2(br - tl) * en
        area_u = area_p + area_g - area_i
        iou = area_i / (area_u + 1e-16)
        if self.loss_type == 'iou':
            loss = 1 - iou ** 2
        elif self.loss_type == 'giou':
            c_tl, c_br = torch.min(p_lo, t_lo), torch.max(p_hi, t_hi)
            area_c = _prod2(c_br - c_tl)
            giou = iou - (area_c - area_u) / area_c.clamp(1e-16)
            loss = 1 - giou.clamp(min=-1.0, max=1.0)
        else:
            raise NotImplementedError(self.loss_type)
        if self.reduction == 'mean':
            loss = loss.mean()
        elif self.reduction == 'sum':
            loss = loss.sum()
        return loss
