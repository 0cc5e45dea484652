"""TEST INFRASTRUCTURE -- torch-CPU restatement of the spiking YOLOX forward
(and loss) that surrounds the hot path.

Module / attribute names mirror the reference so that ``state_dict`` keys are
identical (e.g. ``backbone.backbone.dark2.0.conv.0.weight``,
``...dark2.0.act.w``, ``backbone.backbone.stem.0.conv.conv.weight``): golden
fixtures carry a reference ``state_dict`` and load here with strict=True.

Pinned by tests/golden/model_*.npz: logits / losses / gradients produced by the
UNMODIFIED reference classes (yolox/models/*.py, yolox/exp/event_yolox_base.py)
built on oracle.sj_ref (the one stand-in; see sj_ref header).

Reference locations restated:
  BaseConv / Bottleneck / SPPBottleneck / CSPLayer / Focus   yolox/models/network_blocks.py:31-213
  CSPDarknet                                                yolox/models/darknet.py:97-180
  YOLOPAFPN / SpikingYOLOPAFPN                              yolox/models/yolo_pafpn.py:12-116, spiking_yolo_pafpn.py:13-120
  YOLOXHead / SpikingYOLOXHead (+ SimOTA loss)              yolox/models/yolo_head.py:18-560, spiking_yolo_head.py:18-270
  SpikingYOLOX / YOLOX                                      yolox/models/spiking_yolox.py:16-76, yolox.py:10-60
  convert_to_spiking                                        yolox/utils/utils_snn.py:16-58
  EventExp.get_model                                        yolox/exp/event_yolox_base.py:160-217
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import sj_ref as sj
from .embedding_ref import (AdaptiveRSNNEmbeddingRef, LIFEmbeddingRef, SpikeCountEmbeddingRef,
                            SpikingEmbeddingRef)


def _act(name):
    if name == 'silu':
        return nn.SiLU(inplace=True)
    if name == 'relu':
        return nn.ReLU(inplace=True)
    if name == 'lrelu':
        return nn.LeakyReLU(0.1, inplace=True)
    raise AttributeError(name)


class BaseConv(nn.Module):
    def __init__(self, cin, cout, ksize, stride, act='silu'):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, ksize, stride, (ksize - 1) // 2, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.act = _act(act)

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Bottleneck(nn.Module):
    def __init__(self, cin, cout, shortcut=True, expansion=0.5, act='silu'):
        super().__init__()
        hid = int(cout * expansion)
        self.conv1 = BaseConv(cin, hid, 1, 1, act)
        self.conv2 = BaseConv(hid, cout, 3, 1, act)
        self.use_add = shortcut and cin == cout

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + x if self.use_add else y          # SEW add of spikes when converted


class SPPBottleneck(nn.Module):
    def __init__(self, cin, cout, ks=(5, 9, 13), act='silu'):
        super().__init__()
        hid = cin // 2
        self.conv1 = BaseConv(cin, hid, 1, 1, act)
        self.m = nn.ModuleList([nn.MaxPool2d(k, 1, k // 2) for k in ks])
        self.conv2 = BaseConv(hid * (len(ks) + 1), cout, 1, 1, act)

    def forward(self, x):
        x = self.conv1(x)
        return self.conv2(torch.cat([x] + [m(x) for m in self.m], dim=-3))


class CSPLayer(nn.Module):
    def __init__(self, cin, cout, n=1, shortcut=True, expansion=0.5, act='silu'):
        super().__init__()
        hid = int(cout * expansion)
        self.conv1 = BaseConv(cin, hid, 1, 1, act)
        self.conv2 = BaseConv(cin, hid, 1, 1, act)
        self.conv3 = BaseConv(2 * hid, cout, 1, 1, act)
        self.m = nn.Sequential(*[Bottleneck(hid, hid, shortcut, 1.0, act) for _ in range(n)])

    def forward(self, x):
        return self.conv3(torch.cat((self.m(self.conv1(x)), self.conv2(x)), dim=-3))


class Focus(nn.Module):
    def __init__(self, cin, cout, ksize=1, stride=1, act='silu'):
        super().__init__()
        self.conv = BaseConv(cin * 4, cout, ksize, stride, act)

    def forward(self, x):
        tl, tr = x[..., ::2, ::2], x[..., ::2, 1::2]
        bl, br = x[..., 1::2, ::2], x[..., 1::2, 1::2]
        return self.conv(torch.cat((tl, bl, tr, br), dim=1))


class CSPDarknet(nn.Module):
    def __init__(self, dep_mul, wid_mul, out_features=('dark3', 'dark4', 'dark5'), act='silu', in_dim=3):
        super().__init__()
        self.out_features = out_features
        c = int(wid_mul * 64)
        d = max(round(dep_mul * 3), 1)
        self.stem = Focus(in_dim, c, 3, act=act)
        self.dark2 = nn.Sequential(BaseConv(c, c * 2, 3, 2, act), CSPLayer(c * 2, c * 2, d, act=act))
        self.dark3 = nn.Sequential(BaseConv(c * 2, c * 4, 3, 2, act), CSPLayer(c * 4, c * 4, d * 3, act=act))
        self.dark4 = nn.Sequential(BaseConv(c * 4, c * 8, 3, 2, act), CSPLayer(c * 8, c * 8, d * 3, act=act))
        self.dark5 = nn.Sequential(BaseConv(c * 8, c * 16, 3, 2, act), SPPBottleneck(c * 16, c * 16, act=act),
                                   CSPLayer(c * 16, c * 16, d, shortcut=False, act=act))

    def forward(self, x):
        out = {}
        x = self.stem(x)
        out['stem'] = x
        for name in ('dark2', 'dark3', 'dark4', 'dark5'):
            x = getattr(self, name)(x)
            out[name] = x
        return {k: v for k, v in out.items() if k in self.out_features}


class YOLOPAFPN(nn.Module):
    def __init__(self, depth=1.0, width=1.0, in_features=('dark3', 'dark4', 'dark5'),
                 in_channels=(256, 512, 1024), in_dim=3, act='silu', spike_fn=None, spiking_backbone=False):
        super().__init__()
        self.backbone = CSPDarknet(depth, width, act=act, in_dim=in_dim)
        self.spiking_backbone = spiking_backbone
        if spiking_backbone:                       # SpikingYOLOPAFPN (spiking_yolo_pafpn.py:31-32)
            self.backbone = convert_to_spiking(self.backbone, spike_fn)
        self.in_features = in_features
        c0, c1, c2 = (int(c * width) for c in in_channels)
        n = round(3 * depth)
        self.upsample = nn.Upsample(scale_factor=2, mode='nearest')
        self.lateral_conv0 = BaseConv(c2, c1, 1, 1, act)
        self.C3_p4 = CSPLayer(2 * c1, c1, n, False, act=act)
        self.reduce_conv1 = BaseConv(c1, c0, 1, 1, act)
        self.C3_p3 = CSPLayer(2 * c0, c0, n, False, act=act)
        self.bu_conv2 = BaseConv(c0, c0, 3, 2, act)
        self.C3_n3 = CSPLayer(2 * c0, c1, n, False, act=act)
        self.bu_conv1 = BaseConv(c1, c1, 3, 2, act)
        self.C3_n4 = CSPLayer(2 * c1, c2, n, False, act=act)

    def forward(self, x):
        feats = self.backbone(x)
        if self.spiking_backbone:                  # firing-rate readout (spiking_yolo_pafpn.py:98)
            x2, x1, x0 = [feats[f].mean(axis=0) for f in self.in_features]
        else:
            x2, x1, x0 = [feats[f] for f in self.in_features]
        fpn_out0 = self.lateral_conv0(x0)
        f_out0 = self.C3_p4(torch.cat([self.upsample(fpn_out0), x1], -3))
        fpn_out1 = self.reduce_conv1(f_out0)
        pan_out2 = self.C3_p3(torch.cat([self.upsample(fpn_out1), x2], -3))
        pan_out1 = self.C3_n3(torch.cat([self.bu_conv2(pan_out2), fpn_out1], -3))
        pan_out0 = self.C3_n4(torch.cat([self.bu_conv1(pan_out1), fpn_out0], -3))
        return pan_out2, pan_out1, pan_out0


def convert_to_spiking(model, spike_fn):
    """utils_snn.py:16-58 tree rewrite (row a8)."""
    for name, module in model.named_children():
        if isinstance(module, Focus):
            setattr(model, name, sj.SeqToANNContainer(module))          # not recursed: stem stays ANN
        elif isinstance(module, (nn.Conv2d, nn.Upsample)):
            setattr(model, name, sj.SeqToANNContainer(module))
        elif isinstance(module, nn.BatchNorm2d):
            setattr(model, name, sj.BatchNorm2d(module.num_features, module.eps, module.momentum, step_mode='m'))
        elif name.endswith('act') or isinstance(module, (nn.ReLU, nn.SiLU, nn.LeakyReLU)):
            setattr(model, name, sj.ParametricLIFNode(
                init_tau=2.0, decay_input=False, v_threshold=1.0, v_reset=None,
                surrogate_function=copy.deepcopy(spike_fn), detach_reset=False, step_mode='m', backend='torch'))
        elif isinstance(module, nn.MaxPool2d):
            setattr(model, name, sj.SeqToANNContainer(module))
        else:
            convert_to_spiking(module, spike_fn)
    return model


# ----------------------------------------------------------------------------- head + loss
def bboxes_iou_cxcywh(a, b):
    """yolox/utils/boxes.py:80-104 with xyxy=False."""
    tl = torch.max(a[:, None, :2] - a[:, None, 2:] / 2, b[:, :2] - b[:, 2:] / 2)
    br = torch.min(a[:, None, :2] + a[:, None, 2:] / 2, b[:, :2] + b[:, 2:] / 2)
    area_a = torch.prod(a[:, 2:], 1)
    area_b = torch.prod(b[:, 2:], 1)
    en = (tl < br).type(tl.type()).prod(dim=2)
    area_i = torch.prod(br - tl, 2) * en
    return area_i / (area_a[:, None] + area_b - area_i)


def iou_loss(pred, target):
    """yolox/models/losses.py:16-37, loss_type 'iou', reduction 'none'."""
    tl = torch.max(pred[:, :2] - pred[:, 2:] / 2, target[:, :2] - target[:, 2:] / 2)
    br = torch.min(pred[:, :2] + pred[:, 2:] / 2, target[:, :2] + target[:, 2:] / 2)
    area_p = torch.prod(pred[:, 2:], 1)
    area_g = torch.prod(target[:, 2:], 1)
    en = (tl < br).type(tl.type()).prod(dim=1)
    area_i = torch.prod(br - tl, 1) * en
    iou = area_i / (area_p + area_g - area_i + 1e-16)
    return 1 - iou ** 2


class YOLOXHead(nn.Module):
    """mode: 'ann' (YOLOXHead), 'mean_in' (SpikingYOLOXHead, full_spike=False), 'spiking' (full_spike=True)."""

    def __init__(self, num_classes, width=1.0, strides=(8, 16, 32), in_channels=(256, 512, 1024), act='silu',
                 mode='ann', spike_fn=None):
        super().__init__()
        self.num_classes, self.mode = num_classes, mode
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        self.cls_preds, self.reg_preds, self.obj_preds = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.stems = nn.ModuleList()
        hid = int(256 * width)
        for c in in_channels:
            self.stems.append(BaseConv(int(c * width), hid, 1, 1, act))
            self.cls_convs.append(nn.Sequential(BaseConv(hid, hid, 3, 1, act), BaseConv(hid, hid, 3, 1, act)))
            self.reg_convs.append(nn.Sequential(BaseConv(hid, hid, 3, 1, act), BaseConv(hid, hid, 3, 1, act)))
            self.cls_preds.append(nn.Conv2d(hid, num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(hid, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(hid, 1, 1, 1, 0))
        if mode == 'spiking':
            convert_to_spiking(self, spike_fn)
        self.use_l1 = False
        self.strides = strides

    def initialize_biases(self, prior_prob):
        for preds in (self.cls_preds, self.obj_preds):
            for m in preds.modules():
                if isinstance(m, nn.Conv2d):
                    m.bias.data.fill_(-math.log((1 - prior_prob) / prior_prob))

    def forward(self, xin, labels=None, imgs=None):
        outputs, origin_preds, x_shifts, y_shifts, exp_strides = [], [], [], [], []
        for k, (stride, x) in enumerate(zip(self.strides, xin)):
            if self.mode == 'mean_in':
                x = x.mean(axis=0)
            x = self.stems[k](x)
            cls_feat = self.cls_convs[k](x)
            cls_out = self.cls_preds[k](cls_feat)
            reg_feat = self.reg_convs[k](x)
            reg_out = self.reg_preds[k](reg_feat)
            obj_out = self.obj_preds[k](reg_feat)
            if self.mode == 'spiking':             # mean current over T (spiking_yolo_head.py:175-178)
                cls_out, reg_out, obj_out = cls_out.mean(axis=0), reg_out.mean(axis=0), obj_out.mean(axis=0)
            if self.training:
                out = torch.cat([reg_out, obj_out, cls_out], 1)
                B, C, H, W = out.shape
                yv, xv = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
                grid = torch.stack((xv, yv), 2).view(1, -1, 2).to(out)
                out = out.view(B, 1, C, H, W).permute(0, 1, 3, 4, 2).reshape(B, H * W, -1)
                out = torch.cat([(out[..., :2] + grid) * stride, torch.exp(out[..., 2:4]) * stride, out[..., 4:]], -1)
                x_shifts.append(grid[:, :, 0])
                y_shifts.append(grid[:, :, 1])
                exp_strides.append(torch.full((1, grid.shape[1]), float(stride)).to(out))
                if self.use_l1:
                    origin_preds.append(reg_out.view(B, 1, 4, H, W).permute(0, 1, 3, 4, 2).reshape(B, -1, 4))
            else:
                out = torch.cat([reg_out, obj_out.sigmoid(), cls_out.sigmoid()], 1)
            outputs.append(out)
        if self.training:
            return self.get_losses(x_shifts, y_shifts, exp_strides, labels, torch.cat(outputs, 1), origin_preds)
        hw = [o.shape[-2:] for o in outputs]
        out = torch.cat([o.flatten(start_dim=2) for o in outputs], dim=2).permute(0, 2, 1)
        grids, strides = [], []
        for (H, W), s in zip(hw, self.strides):
            yv, xv = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
            grids.append(torch.stack((xv, yv), 2).view(1, -1, 2))
            strides.append(torch.full((1, H * W, 1), float(s)))
        grids = torch.cat(grids, 1).to(out)
        strides = torch.cat(strides, 1).to(out)
        return torch.cat([(out[..., 0:2] + grids) * strides, torch.exp(out[..., 2:4]) * strides, out[..., 4:]], -1)

    # ---- loss (yolo_head.py:262-420)
    def get_losses(self, x_shifts, y_shifts, exp_strides, labels, outputs, origin_preds):
        bbox_preds, obj_preds, cls_preds = outputs[:, :, :4], outputs[:, :, 4:5], outputs[:, :, 5:]
        nlabel = (labels.sum(dim=2) > 0).sum(dim=1)
        A = outputs.shape[1]
        x_shifts, y_shifts, exp_strides = torch.cat(x_shifts, 1), torch.cat(y_shifts, 1), torch.cat(exp_strides, 1)
        if self.use_l1:
            origin_preds = torch.cat(origin_preds, 1)
        cls_t, reg_t, l1_t, obj_t, fg_masks = [], [], [], [], []
        num_fg, num_gts = 0.0, 0.0
        for b in range(outputs.shape[0]):
            num_gt = int(nlabel[b])
            num_gts += num_gt
            if num_gt == 0:
                cls_t.append(outputs.new_zeros((0, self.num_classes)))
                reg_t.append(outputs.new_zeros((0, 4)))
                l1_t.append(outputs.new_zeros((0, 4)))
                obj_t.append(outputs.new_zeros((A, 1)))
                fg_masks.append(outputs.new_zeros(A).bool())
                continue
            gt_boxes = labels[b, :num_gt, 1:5]
            gt_cls = labels[b, :num_gt, 0]
            matched_cls, fg_mask, matched_iou, matched_idx, nfg = self.get_assignments(
                b, num_gt, gt_boxes, gt_cls, bbox_preds[b], exp_strides, x_shifts, y_shifts, cls_preds, obj_preds)
            num_fg += nfg
            cls_t.append(F.one_hot(matched_cls.to(torch.int64), self.num_classes) * matched_iou.unsqueeze(-1))
            obj_t.append(fg_mask.unsqueeze(-1).to(outputs.dtype))
            reg_t.append(gt_boxes[matched_idx])
            fg_masks.append(fg_mask)
            if self.use_l1:
                gt, s = gt_boxes[matched_idx], exp_strides[0][fg_mask]
                l1 = outputs.new_zeros((nfg, 4))
                l1[:, 0] = gt[:, 0] / s - x_shifts[0][fg_mask]
                l1[:, 1] = gt[:, 1] / s - y_shifts[0][fg_mask]
                l1[:, 2] = torch.log(gt[:, 2] / s + 1e-8)
                l1[:, 3] = torch.log(gt[:, 3] / s + 1e-8)
                l1_t.append(l1)
        cls_t, reg_t, obj_t, fg_masks = torch.cat(cls_t, 0), torch.cat(reg_t, 0), torch.cat(obj_t, 0), torch.cat(fg_masks, 0)
        num_fg = max(num_fg, 1)
        loss_iou = iou_loss(bbox_preds.reshape(-1, 4)[fg_masks], reg_t).sum() / num_fg
        loss_obj = F.binary_cross_entropy_with_logits(obj_preds.reshape(-1, 1), obj_t, reduction='none').sum() / num_fg
        loss_cls = F.binary_cross_entropy_with_logits(
            cls_preds.reshape(-1, self.num_classes)[fg_masks], cls_t, reduction='none').sum() / num_fg
        if self.use_l1:
            loss_l1 = (origin_preds.reshape(-1, 4)[fg_masks] - torch.cat(l1_t, 0)).abs().sum() / num_fg
        else:
            loss_l1 = 0.0
        loss = 5.0 * loss_iou + loss_obj + loss_cls + loss_l1
        return loss, 5.0 * loss_iou, loss_obj, loss_cls, loss_l1, num_fg / max(num_gts, 1)

    @torch.no_grad()
    def get_assignments(self, b, num_gt, gt_boxes, gt_cls, boxes_pred, exp_strides, x_shifts, y_shifts,
                        cls_preds, obj_preds):
        s = exp_strides[0]
        xc = ((x_shifts[0] + 0.5) * s).unsqueeze(0)
        yc = ((y_shifts[0] + 0.5) * s).unsqueeze(0)
        dist = s.unsqueeze(0) * 1.5
        deltas = torch.stack([xc - (gt_boxes[:, 0:1] - dist), yc - (gt_boxes[:, 1:2] - dist),
                              (gt_boxes[:, 0:1] + dist) - xc, (gt_boxes[:, 1:2] + dist) - yc], 2)
        in_centers = deltas.min(dim=-1).values > 0.0
        fg_mask = in_centers.sum(dim=0) > 0
        geom = in_centers[:, fg_mask]
        boxes_pred = boxes_pred[fg_mask]
        cls_p = cls_preds[b][fg_mask]
        obj_p = obj_preds[b][fg_mask]
        n_in = boxes_pred.shape[0]
        ious = bboxes_iou_cxcywh(gt_boxes, boxes_pred)
        gt_onehot = F.one_hot(gt_cls.to(torch.int64), self.num_classes).float()
        iou_cost = -torch.log(ious + 1e-8)
        joint = (cls_p.float().sigmoid() * obj_p.float().sigmoid()).sqrt()
        cls_cost = F.binary_cross_entropy(joint.unsqueeze(0).repeat(num_gt, 1, 1),
                                          gt_onehot.unsqueeze(1).repeat(1, n_in, 1), reduction='none').sum(-1)
        cost = cls_cost + 3.0 * iou_cost + float(1e6) * (~geom)
        # SimOTA dynamic-k matching (yolo_head.py:523-553)
        match = torch.zeros_like(cost, dtype=torch.uint8)
        topk_ious, _ = torch.topk(ious, min(10, ious.size(1)), dim=1)
        dyn_k = torch.clamp(topk_ious.sum(1).int(), min=1)
        for g in range(num_gt):
            _, pos = torch.topk(cost[g], k=int(dyn_k[g]), largest=False)
            match[g][pos] = 1
        per_anchor = match.sum(0)
        if per_anchor.max() > 1:
            multi = per_anchor > 1
            _, arg = torch.min(cost[:, multi], dim=0)
            match[:, multi] *= 0
            match[arg, multi] = 1
        fg_in = per_anchor > 0
        nfg = int(fg_in.sum())
        fg_mask[fg_mask.clone()] = fg_in
        matched_idx = match[:, fg_in].argmax(0)
        return gt_cls[matched_idx], fg_mask, (match * ious).sum(0)[fg_in], matched_idx, nfg


class SpikingYOLOX(nn.Module):
    """spiking_yolox.py:16-76 (row a13): sampler -> broadcast to T -> backbone -> head."""

    def __init__(self, backbone, head, embedding, T=4):
        super().__init__()
        self.nb_steps = T
        self.embedding, self.backbone, self.head = embedding, backbone, head

    def forward(self, x, targets=None):
        x = self.embedding(x)
        if x.dim() > 5:
            x = x[0]
        if x.dim() == 4:
            x, _ = torch.broadcast_tensors(x, torch.zeros((self.nb_steps,) + x.shape))
        elif x.shape[0] == 1:
            x, _ = torch.broadcast_tensors(x, torch.zeros((self.nb_steps,) + x.shape[1:]))
        else:
            assert x.shape[0] == self.nb_steps
        fpn = self.backbone(x)
        if self.training:
            assert targets is not None
            loss, iou, conf, cls, l1, nfg = self.head(fpn, targets, x)
            return {'total_loss': loss, 'iou_loss': iou, 'l1_loss': l1, 'conf_loss': conf, 'cls_loss': cls,
                    'num_fg': nfg}
        return self.head(fpn)


class YOLOX(nn.Module):
    """yolox.py:10-60 (use_spike 'False': ANN network behind the sampler)."""

    def __init__(self, backbone, head, embedding):
        super().__init__()
        self.embedding, self.backbone, self.head = embedding, backbone, head

    def forward(self, x, targets=None):
        x = self.embedding(x)
        if x.dim() > 4:
            x = x[0]
        fpn = self.backbone(x)
        if self.training:
            loss, iou, conf, cls, l1, nfg = self.head(fpn, targets, x)
            return {'total_loss': loss, 'iou_loss': iou, 'l1_loss': l1, 'conf_loss': conf, 'cls_loss': cls,
                    'num_fg': nfg}
        return self.head(fpn)


DEFAULT_CFG = dict(depth=0.33, width=0.5, num_classes=2, T=3, Tm=4, Ts=1, embedding='arsnn', embedding_ksize=5,
                   embedding_depth=2, thresh=1.0, reset=0.0, readout='sum', spike_attach=True, write_zero=True,
                   abs=False, use_spike='True', spike_fn='atan', alpha=2.0, act='silu', in_dim=2, decay=0.5)


def build_model(**overrides):
    """EventExp.get_model (event_yolox_base.py:160-217) for the README Gen1 recipe by default."""
    c = dict(DEFAULT_CFG)
    c.update(overrides)
    if c['spike_fn'] == 'atan':
        fn = sj.ATan(c['alpha'])
    elif c['spike_fn'] == 'sigmoid':
        fn = sj.Sigmoid(c['alpha'])
    elif c['spike_fn'] == 'patan':
        fn = sj.PATan(c['alpha'])
    else:
        raise NotImplementedError(c['spike_fn'])
    if c['embedding'] == 'arsnn':
        emb = AdaptiveRSNNEmbeddingRef(c['embedding_ksize'], 2, 2, Ts=c['Ts'], spike_attach=c['spike_attach'],
                                       write_zero=c['write_zero'], abs=c['abs'], depth=c['embedding_depth'],
                                       readout=c['readout'], Tm=c['Tm'], thresh=c['thresh'], vreset=c['reset'])
    elif c['embedding'] == 'count':
        emb = SpikeCountEmbeddingRef(c['Tm'])
    elif c['embedding'] == 'rsnn':
        emb = SpikingEmbeddingRef(c['embedding_ksize'], 2, 2, readout=c['readout'], relu=c['abs'],
                                  depth=c['embedding_depth'], Tm=c['Tm'], thresh=c['thresh'], vreset=c['reset'])
    elif c['embedding'] == 'snn':
        emb = LIFEmbeddingRef(c['embedding_ksize'], 2, 2, readout=c['readout'], depth=c['embedding_depth'],
                              Tm=c['Tm'], thresh=c['thresh'], vreset=c['reset'], decay=c['decay'])
    else:
        raise NotImplementedError(c['embedding'])
    chans = (256, 512, 1024)
    us = c['use_spike']
    if us == 'True' or us is True:
        backbone = YOLOPAFPN(c['depth'], c['width'], in_channels=chans, in_dim=c['in_dim'], act=c['act'],
                             spike_fn=fn, spiking_backbone=True)
        head = YOLOXHead(c['num_classes'], c['width'], in_channels=chans, act=c['act'], mode='ann')
        model = SpikingYOLOX(backbone, head, emb, T=c['T'])
    elif 'full_spike' in str(us):
        backbone = convert_to_spiking(YOLOPAFPN(c['depth'], c['width'], in_channels=chans, in_dim=2, act=c['act']), fn)
        head = YOLOXHead(c['num_classes'], c['width'], in_channels=chans, act=c['act'],
                         mode='spiking' if 'v2' in us else 'mean_in', spike_fn=fn)
        model = SpikingYOLOX(backbone, head, emb, T=c['T'])
    else:
        backbone = YOLOPAFPN(c['depth'], c['width'], in_channels=chans, in_dim=2, act=c['act'])
        head = YOLOXHead(c['num_classes'], c['width'], in_channels=chans, act=c['act'], mode='ann')
        model = YOLOX(backbone, head, emb)
    for m in model.modules():                      # init_yolo (event_yolox_base.py:179-183)
        if isinstance(m, nn.BatchNorm2d):
            m.eps = 1e-3
            m.momentum = 0.03
    model.head.initialize_biases(1e-2)
    model.train()
    return model
