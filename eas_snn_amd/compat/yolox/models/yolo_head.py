"""Decoupled YOLOX head with the SimOTA loss (reference: yolox/models/yolo_head.py:18-560,
spiking_yolo_head.py:18-270).  Module names equal the reference's (checkpoint keys).

The loss is the same computation as the reference's per-image loop, restated batch-wide with masks so the
step issues no device->host synchronisation (the reference syncs several times per image:
``int(nlabel[b])``, ``.item()`` in simota_matching, boolean-mask indexing)."""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from eas_snn_amd import ops
from eas_snn_amd import ops_group as G
from yolox.utils.utils_snn import convert_to_spiking

from .losses import IOUloss
from .network_blocks import BaseConv, DWConv


# EAS_FUSED_EVAL_DECODE=0: the inference output through the tensor operators of the reference (sigmoid, cat, grid arithmetic)
FUSED_EVAL_DECODE = os.environ.get('EAS_FUSED_EVAL_DECODE', '1') != '0'


def _pred(conv, x):
    """prediction conv through ops.conv2d (own kernels where eligible, and visible to the statistics tap of eas_snn_amd.stats)"""
    return ops.conv2d(x, conv) if type(conv) is nn.Conv2d and x.is_cuda else conv(x)


def _inner_conv(block):
    """the nn.Conv2d of a BaseConv: plain, or the single module inside its SeqToANNContainer once converted; else None"""
    c = block.conv
    if isinstance(c, nn.Sequential) and len(c) == 1:
        c = c[0]
    return c if type(c) is nn.Conv2d else None


def _fusable_pair(a, b, x):
    """both converted blocks run the fused BN+LIF kernels on a [T,N,C,H,W] input (no rate output, no hooks, fixed-slope surrogate)"""
    for blk in (a, b):
        c = _inner_conv(blk)
        if (c is None or c.stride != (1, 1) or c.groups != 1 or c._forward_hooks or c._forward_pre_hooks or blk.emit_rate
                or blk._forward_hooks or blk._forward_pre_hooks or not blk.bn.affine or blk.bn.step_mode != 'm'
                or (blk.bn.momentum is None and blk.bn.training) or getattr(blk.act.surrogate_function, 'hip_id', None) == 'patan'):
            return False
    if a.bn.num_features % 8 != 0 or b.bn.num_features % 8 != 0:
        return False
    return (x.shape[-1] * x.shape[-2]) % 4 == 0 and x.shape[0] <= 8 and not torch.nn.modules.module._global_forward_hooks


def _prod2(t):
    """Product over a last axis of length 2 (box width x height).  ``torch.prod``'s backward inspects the input for zeros
    on the host, which stalls the stream and cannot be captured in a HIP graph; the explicit product has neither problem."""
    return t[..., 0] * t[..., 1]



def _grid(h, w, device, dtype):
    yv, xv = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing='ij')
    return torch.stack((xv, yv), 2).view(1, -1, 2).to(dtype)


class YOLOXHead(nn.Module):
    def __init__(self, num_classes, width=1.0, strides=[8, 16, 32], in_channels=[256, 512, 1024], act='silu',
                 depthwise=False):
        super().__init__()
        self.num_classes = num_classes
        self.decode_in_inference = True
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        self.cls_preds, self.reg_preds, self.obj_preds = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.stems = nn.ModuleList()
        Conv = DWConv if depthwise else BaseConv
        hid = int(256 * width)
        for c in in_channels:
            self.stems.append(BaseConv(int(c * width), hid, ksize=1, stride=1, act=act))
            self.cls_convs.append(nn.Sequential(Conv(hid, hid, 3, 1, act=act), Conv(hid, hid, 3, 1, act=act)))
            self.reg_convs.append(nn.Sequential(Conv(hid, hid, 3, 1, act=act), Conv(hid, hid, 3, 1, act=act)))
            self.cls_preds.append(nn.Conv2d(hid, self.num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(hid, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(hid, 1, 1, 1, 0))
        self.use_l1 = False
        self.fused_assign = True       # SimOTA assignment in one HIP launch (ops.simota_assign); False = the tensor-op form (_assign)
        self.fused_loss = True         # decode + assignment + loss terms + gradient on the HIP kernels (ops.det_loss)
        self.l1_loss = nn.L1Loss(reduction='none')
        self.bcewithlog_loss = nn.BCEWithLogitsLoss(reduction='none')
        self.iou_loss = IOUloss(reduction='none')
        self.strides = strides
        self.full_spike = False

    def initialize_biases(self, prior_prob):
        for preds in (self.cls_preds, self.obj_preds):
            for m in preds.modules():
                if isinstance(m, nn.Conv2d):
                    b = m.bias.view(1, -1)
                    b.data.fill_(-math.log((1 - prior_prob) / prior_prob))
                    m.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    # ---- per-level raw predictions
    def eas_dual_convs(self):
        """the first cls / reg tower convolutions of a level read the same stem output: ``ops.packed_weights`` packs their concatenated
        weight for ``ops.conv2d_dual``"""
        out = []
        for k in range(len(self.stems)):
            a, b = self.cls_convs[k][0], self.reg_convs[k][0]
            if not (isinstance(a, BaseConv) and isinstance(b, BaseConv)):
                continue
            ca, cb = _inner_conv(a), _inner_conv(b)          # plain, or inside the SeqToANNContainer of a converted block
            if ca is not None and cb is not None and (ca.bias is None) == (cb.bias is None):      # (biases on both: the pair fuse_model folded)
                out.append((f'tower{k}', ca, cb))
        return out

    def _towers(self, k, x):
        """cls_convs[k](x), reg_convs[k](x); their first convolutions as ONE convolution where the kernels allow it (x is read once, one
        input gradient instead of two and an addition, twice the blocks on the small maps)"""
        a, b = self.cls_convs[k][0], self.reg_convs[k][0]
        ann_pair = (isinstance(a, BaseConv) and isinstance(b, BaseConv) and not a.spiking() and not b.spiking() and x.dim() == 4
                    and type(a.conv) is nn.Conv2d and type(b.conv) is nn.Conv2d and not os.environ.get('EAS_NO_DUAL')
                    and not os.environ.get('EAS_NO_ANN_DUAL'))
        norms = [a.eval_norm(), b.eval_norm()] if (ann_pair and not self.training) else [None]
        if None not in norms and ops.fused_ann_eval_ok(x, (a.conv, b.conv), norms):
            # eval mode (also after fuse_model): the one convolution for both towers AND their BN + SiLU in one kernel
            packs = (getattr(self, '_eas_dual_packs', None) or {}).get(f'tower{k}')
            ca, ra = ops.conv_bn_act_eval(x, (a.conv, b.conv), norms, packs=ops.current_packs(packs))
        elif ann_pair and a.ann_fusable(x[:, :1]) and b.ann_fusable(x[:, :1]) and ops.conv_dual_ok(x, a.conv, b.conv):
            sink = ops.conv_sink()
            if sink is not None:
                sink(a.conv, x, 1)
                sink(b.conv, x, 1)
            with ops.conv_stats_scope(all(bn.training or bn.running_mean is None for bn in (a.bn, b.bn))):
                y12 = ops.conv2d_dual(x, a.conv, b.conv, self, f'tower{k}')
            ca, ra = ops.bn_silu_pair(y12, a.bn, b.bn)
        else:
            ann_pair = False
        if ann_pair:
            for m in list(self.cls_convs[k])[1:]:
                ca = m(ca)
            for m in list(self.reg_convs[k])[1:]:
                ra = m(ra)
            return ca, ra
        if (isinstance(a, BaseConv) and isinstance(b, BaseConv) and a.spiking() and b.spiking() and x.dim() == 5 and x.is_cuda
                and not os.environ.get('EAS_NO_DUAL') and not os.environ.get('EAS_NO_SPIKING_TOWER_DUAL') and _fusable_pair(a, b, x)):
            # converted head (full_spike_v2): the same for the two spiking towers -- one convolution, the two BN+LIF layers on the channel
            # halves of its output (as the 1x1 branches of a CSPLayer)
            from spikingjelly.activation_based import layer as sj_layer
            ca_, cb_ = _inner_conv(a), _inner_conv(b)
            x4 = ops.fold_time(x)
            if ops.conv_dual_ok(x4, ca_, cb_):
                T, N = x.shape[:2]
                sink = ops.conv_sink()
                if sink is not None:
                    sink(ca_, x4, 1)
                    sink(cb_, x4, 1)
                with ops.conv_stats_scope(a.bn._use_batch_stats() and b.bn._use_batch_stats()):
                    y12 = ops.conv2d_dual(x4, ca_, cb_, self, f'tower{k}')
                y12 = y12.view(T, N, *y12.shape[1:])
                ca, ra = sj_layer.fused_pair(a.bn, a.act, b.bn, b.act, y12, planes_a=a.wants_planes(), planes_b=b.wants_planes())
                for m in list(self.cls_convs[k])[1:]:
                    ca = m(ca)
                for m in list(self.reg_convs[k])[1:]:
                    ra = m(ra)
                return ca, ra
        return self.cls_convs[k](x), self.reg_convs[k](x)

    def _level(self, k, x):
        x = self.stems[k](x)
        cls_feat, reg_feat = self._towers(k, x)
        cls_out, reg_out, obj_out = _pred(self.cls_preds[k], cls_feat), _pred(self.reg_preds[k], reg_feat), _pred(self.obj_preds[k], reg_feat)
        if self.full_spike:        # mean input current over T (spiking_yolo_head.py:175-178)
            cls_out, reg_out, obj_out = ops.time_mean(cls_out), ops.time_mean(reg_out), ops.time_mean(obj_out)
        return reg_out, obj_out, cls_out

    def _prepare(self, x):
        return x

    def _levels_grouped(self, xs):
        """The raw predictions of ALL levels, stage by stage: the levels are independent until the loss (the reference walks them one after
        the other, yolo_head.py:149-200), so every stage -- stems, the first tower convolutions (cls | reg as one), the second ones, the
        prediction convolutions, each with its BatchNorm + SiLU -- runs as ONE grouped launch over the levels (``ops_group``): the 16x20 and
        8x10 levels ride in the grid of the 32x40 level instead of leaving most of the chip idle in launches of their own, forward and
        backward.  Training mode of the real-valued head; None = not eligible (the per-level path runs)."""
        n = len(xs)
        if not (G.ENABLED and self.training and not self.full_spike and 1 < n <= 4 and torch.is_grad_enabled()):
            return None
        if os.environ.get('EAS_NO_DUAL') or os.environ.get('EAS_NO_ANN_DUAL'):      # development switches of the per-level path: honoured by running it
            return None
        blocks = []
        for k in range(n):
            cc, rc = self.cls_convs[k], self.reg_convs[k]
            if len(cc) != 2 or len(rc) != 2:
                return None
            blocks.append((self.stems[k], cc[0], rc[0], cc[1], rc[1]))
        for row in blocks:
            for b in row:
                if not (isinstance(b, BaseConv) and not b.spiking() and type(b.conv) is nn.Conv2d and isinstance(b.act, nn.SiLU)
                        and not b._forward_hooks and not b._forward_pre_hooks and not b.bn._forward_hooks and not b.bn._forward_pre_hooks):
                    return None
        stems = [r[0] for r in blocks]
        if not (G.conv_group_ok(xs, [b.conv for b in stems], 1) and G.bn_silu_group_ok(xs, [b.bn for b in stems])):
            return None
        pairs = [(r[1].conv, r[2].conv) for r in blocks]
        if not self._group_geometry_ok(xs, blocks, pairs):
            return None
        # ---- stage 1: stems
        ys, st = G.conv_group(xs, [b.conv for b in stems], 1, True)
        s = G.bn_silu_group(ys, st, [(i, 0, b.bn) for i, b in enumerate(stems)])
        # ---- stage 2: first cls | reg tower convolutions of every level as one convolution each (they read the same stem output)
        ys, st = G.conv_group(s, pairs, 3, True, owners=[(self, f'tower{k}') for k in range(n)])
        layers = []
        for k, r in enumerate(blocks):
            layers += [(k, 0, r[1].bn), (k, r[1].conv.out_channels, r[2].bn)]
        o = G.bn_silu_group(ys, st, layers)
        ca, ra = o[0::2], o[1::2]
        # ---- stage 3: second tower convolutions, 2 n problems
        convs2 = [r[3].conv for r in blocks] + [r[4].conv for r in blocks]
        ys, st = G.conv_group(ca + ra, convs2, 3, True)
        f = G.bn_silu_group(ys, st, [(i, 0, b.bn) for i, b in enumerate([r[3] for r in blocks] + [r[4] for r in blocks])])
        cf, rf = f[:n], f[n:]
        # ---- stage 4: prediction convolutions, 3 n problems
        cls_o, reg_o, obj_o = G.pred_group(cf, rf, list(self.cls_preds), list(self.reg_preds), list(self.obj_preds))
        return [(reg_o[k], obj_o[k], cls_o[k]) for k in range(n)]

    def _levels_grouped_eval(self, xs):
        """Eval mode: the same stage-by-stage walk with every stage ONE grouped launch of the fused conv -> BatchNorm (running statistics) ->
        SiLU kernel (``ops_group.conv_bn_act_eval_group``; also for a head folded by ``fuse_model``) and the nine prediction convolutions one
        launch.  Bit-identical to the per-level forward.  None = not eligible."""
        n = len(xs)
        if not (G.ENABLED and not self.training and not self.full_spike and 1 < n <= 4 and not torch.is_grad_enabled()):
            return None
        blocks = []
        for k in range(n):
            cc, rc = self.cls_convs[k], self.reg_convs[k]
            if len(cc) != 2 or len(rc) != 2:
                return None
            row = (self.stems[k], cc[0], rc[0], cc[1], rc[1])
            for b in row:
                if not (isinstance(b, BaseConv) and not b.spiking() and type(b.conv) is nn.Conv2d and b.eval_norm() is not None
                        and not b._forward_hooks and not b._forward_pre_hooks):
                    return None
            blocks.append(row)
        preds = list(self.cls_preds) + list(self.reg_preds) + list(self.obj_preds)
        if any(type(c) is not nn.Conv2d or c.kernel_size != (1, 1) or c.bias is None or c._forward_hooks or c._forward_pre_hooks for c in preds):
            return None
        stems = [r[0] for r in blocks]
        if not G.conv_bn_act_eval_group_ok(xs, [b.conv for b in stems], [[b.eval_norm()] for b in stems], 1):
            return None
        s = G.conv_bn_act_eval_group(xs, [b.conv for b in stems], [[b.eval_norm()] for b in stems], 1)
        pairs = [(r[1].conv, r[2].conv) for r in blocks]
        pnorms = [[r[1].eval_norm(), r[2].eval_norm()] for r in blocks]
        convs2 = [r[3].conv for r in blocks] + [r[4].conv for r in blocks]
        norms2 = [[r[3].eval_norm()] for r in blocks] + [[r[4].eval_norm()] for r in blocks]
        if not G.conv_bn_act_eval_group_ok(s, pairs, pnorms, 3):
            return [self._level_from_stem(k, s[k]) for k in range(n)]
        o = G.conv_bn_act_eval_group(s, pairs, pnorms, 3, owners=[(self, f'tower{k}') for k in range(n)])
        ca, ra = [p[0] for p in o], [p[1] for p in o]
        if G.conv_bn_act_eval_group_ok(ca + ra, convs2, norms2, 3):
            f = G.conv_bn_act_eval_group(ca + ra, convs2, norms2, 3)
            cf, rf = f[:n], f[n:]
        else:
            cf = [blocks[k][3](ca[k]) for k in range(n)]
            rf = [blocks[k][4](ra[k]) for k in range(n)]
        if G.pred_eval_ok(cf + rf + rf, preds):
            po = G.conv_bias_group(cf + rf + rf, preds)
            cls_o, reg_o, obj_o = po[:n], po[n:2 * n], po[2 * n:]
        else:
            cls_o = [_pred(self.cls_preds[k], cf[k]) for k in range(n)]
            reg_o = [_pred(self.reg_preds[k], rf[k]) for k in range(n)]
            obj_o = [_pred(self.obj_preds[k], rf[k]) for k in range(n)]
        return [(reg_o[k], obj_o[k], cls_o[k]) for k in range(n)]

    def _level_from_stem(self, k, x):
        cls_feat, reg_feat = self._towers(k, x)
        return _pred(self.reg_preds[k], reg_feat), _pred(self.obj_preds[k], reg_feat), _pred(self.cls_preds[k], cls_feat)

    def _group_geometry_ok(self, xs, blocks, pairs):
        """every later stage of ``_levels_grouped`` has a grouped tile plan (cached per input geometry; the stems keep H x W, so the stages'
        inputs are known by shape before they exist)"""
        key = tuple(tuple(x.shape) for x in xs)
        cache = self.__dict__.setdefault('_eas_group_ok', {})
        if key not in cache:
            n = len(xs)
            hs = [G.ShapeProbe((x.shape[0], r[0].conv.out_channels) + tuple(x.shape[2:])) for x, r in zip(xs, blocks)]
            ok = G.conv_group_ok(hs, pairs, 3)
            ok = ok and G.conv_group_ok(hs + hs, [r[3].conv for r in blocks] + [r[4].conv for r in blocks], 3)
            ok = ok and G.pred_group_ok(hs, hs, list(self.cls_preds), list(self.reg_preds), list(self.obj_preds))
            ok = ok and G.bn_silu_group_ok(hs * 4, [b.bn for r in blocks for b in r[1:]])
            cache[key] = bool(ok) and n == len(blocks)
        return cache[key]

    def forward(self, xin, labels=None, imgs=None):
        outputs, origin_preds, grids, strides = [], [], [], []
        if self.training and self.fused_loss and self.fused_assign:
            feats = [self._prepare(x) for x in xin]
            raws = self._levels_grouped(feats)
            if raws is None:
                raws = [self._level(k, x) for k, x in enumerate(feats)]
            regs, objs, clss = [r[0] for r in raws], [r[1] for r in raws], [r[2] for r in raws]
            fused = ops.det_loss_supported(regs, labels, self.iou_loss.loss_type)
            self.__dict__['_eas_fused_loss_used'] = bool(fused)
            if fused:
                # decode + assignment + loss terms + their gradient in five launches (ops.det_loss)
                return ops.det_loss(regs, objs, clss, labels, self.strides, self.num_classes, self.use_l1)
        elif not self.training:
            feats = [self._prepare(x) for x in xin]
            raws = self._levels_grouped_eval(feats)
            if raws is None:
                raws = [self._level(k, x) for k, x in enumerate(feats)]
        else:
            raws = None
        for k, (stride, x) in enumerate(zip(self.strides, xin)):
            reg_out, obj_out, cls_out = raws[k] if raws is not None else self._level(k, self._prepare(x))
            if self.training:
                out = torch.cat([reg_out, obj_out, cls_out], 1)
                B, Cn, H, W = out.shape
                grid = _grid(H, W, out.device, out.dtype)
                out = out.flatten(2).permute(0, 2, 1)                           # [B, H*W, 5+nc]
                out = torch.cat([(out[..., :2] + grid) * stride, torch.exp(out[..., 2:4]) * stride, out[..., 4:]], -1)
                grids.append(grid)
                strides.append(torch.full((1, H * W), float(stride), device=out.device, dtype=out.dtype))
                if self.use_l1:
                    origin_preds.append(reg_out.flatten(2).permute(0, 2, 1))
            else:
                out = (reg_out, obj_out, cls_out)
            outputs.append(out)
        if self.training:
            return self.get_losses(torch.cat(grids, 1), torch.cat(strides, 1), labels, torch.cat(outputs, 1),
                                   torch.cat(origin_preds, 1) if self.use_l1 else None)
        return self.assemble_eval(outputs)

    def assemble_eval(self, raws):
        """inference output [B, A, 5 + num_classes] from the per-level raw maps (reg, obj, cls): sigmoid on objectness / classes,
        levels concatenated along the anchors, boxes decoded (yolo_head.py:187-199 of the reference)"""
        if (self.decode_in_inference and FUSED_EVAL_DECODE and not torch.is_grad_enabled() and ops.det_decode_eval_supported(raws)
                and raws[0][2].shape[1] == self.num_classes):
            # sigmoid + concatenations + grid / stride decode: ~25 tensor operators as one launch (eas_det_decode_eval)
            self.hw = [r.shape[-2:] for r, _, _ in raws]
            return ops.det_decode_eval(raws, self.strides, self.num_classes)
        outputs = [torch.cat([reg_out, obj_out.sigmoid(), cls_out.sigmoid()], 1) for reg_out, obj_out, cls_out in raws]
        self.hw = [o.shape[-2:] for o in outputs]
        out = torch.cat([o.flatten(start_dim=2) for o in outputs], dim=2).permute(0, 2, 1)
        return self.decode_outputs(out, dtype=out.type()) if self.decode_in_inference else out

    def decode_outputs(self, outputs, dtype=None):
        grids, strides = [], []
        for (h, w), s in zip(self.hw, self.strides):
            grids.append(_grid(h, w, outputs.device, outputs.dtype))
            strides.append(torch.full((1, h * w, 1), float(s), device=outputs.device, dtype=outputs.dtype))
        grids, strides = torch.cat(grids, 1), torch.cat(strides, 1)
        return torch.cat([(outputs[..., 0:2] + grids) * strides, torch.exp(outputs[..., 2:4]) * strides, outputs[..., 4:]],
                         dim=-1)

    # ---- SimOTA, batch-wide
    @torch.no_grad()
    def _assign(self, grids, strides, gt_boxes, gt_cls, gt_valid, bbox_preds, obj_preds, cls_preds):
        """gt_*: [B, G, ...] (padded, gt_valid marks real rows); preds: [B, A, ...].
        Returns fg [B,A] bool, matched gt index [B,A], matched IoU [B,A]."""
        B, G = gt_valid.shape
        s = strides[0]                                                          # [A]
        xc = ((grids[0, :, 0] + 0.5) * s)[None, None]                           # [1,1,A]
        yc = ((grids[0, :, 1] + 0.5) * s)[None, None]
        dist = (s * 1.5)[None, None]
        gx, gy = gt_boxes[..., 0:1], gt_boxes[..., 1:2]                         # [B,G,1]
        deltas = torch.stack([xc - (gx - dist), yc - (gy - dist), (gx + dist) - xc, (gy + dist) - yc], -1)
        in_centers = (deltas.min(dim=-1).values > 0.0) & gt_valid[..., None]    # [B,G,A]
        cand = in_centers.any(dim=1)                                            # [B,A] anchors kept by the geometry filter

        # pairwise IoU (cxcywh), yolox/utils/boxes.py:80-104
        a, b = gt_boxes[:, :, None, :], bbox_preds[:, None, :, :]
        tl = torch.max(a[..., :2] - a[..., 2:] / 2, b[..., :2] - b[..., 2:] / 2)
        br = torch.min(a[..., :2] + a[..., 2:] / 2, b[..., :2] + b[..., 2:] / 2)
        en = _prod2((tl < br).to(tl.dtype))
        area_i = _prod2(br - tl) * en
        ious = area_i / (_prod2(a[..., 2:]) + _prod2(b[..., 2:]) - area_i)
        ious = torch.where(cand[:, None, :] & gt_valid[..., None], ious, torch.zeros_like(ious))
        iou_cost = -torch.log(ious + 1e-8)

        joint = (cls_preds.float().sigmoid() * obj_preds.float().sigmoid()).sqrt()            # [B,A,nc]
        onehot = F.one_hot(gt_cls.to(torch.int64).clamp(0, self.num_classes - 1), self.num_classes).float()   # [B,G,nc]
        cls_cost = F.binary_cross_entropy(joint[:, None].expand(B, G, -1, -1), onehot[:, :, None].expand(-1, -1, joint.shape[1], -1),
                                          reduction='none').sum(-1)                          # [B,G,A]
        cost = cls_cost + 3.0 * iou_cost + float(1e6) * (~in_centers)
        big = torch.full_like(cost, 1e12)
        cost = torch.where(cand[:, None, :] & gt_valid[..., None], cost, big)

        kk = min(10, ious.shape[-1])
        topk_ious, _ = torch.topk(ious, kk, dim=-1)
        dyn_k = torch.clamp(topk_ious.sum(-1).int(), min=1)                                   # [B,G]
        _, pos = torch.topk(cost, kk, dim=-1, largest=False)                                  # [B,G,kk]
        take = (torch.arange(kk, device=cost.device)[None, None] < dyn_k[..., None]) & gt_valid[..., None]
        match = torch.zeros_like(cost, dtype=torch.bool)
        match.scatter_(-1, pos, take)
        match &= cand[:, None, :]
        per_anchor = match.sum(1)                                                             # [B,A]
        multi = per_anchor > 1
        best = cost.argmin(dim=1)                                                             # [B,A]
        only_best = F.one_hot(best, G).permute(0, 2, 1).bool()                                # [B,G,A]
        match = torch.where(multi[:, None, :], only_best, match)
        fg = per_anchor > 0
        matched = match.to(torch.int8).argmax(1)
        matched_iou = (match * ious).sum(1)
        return fg, matched, matched_iou

    def get_losses(self, grids, strides, labels, outputs, origin_preds):
        bbox_preds, obj_preds, cls_preds = outputs[:, :, :4], outputs[:, :, 4:5], outputs[:, :, 5:]
        B, A = outputs.shape[:2]
        nlabel = (labels.sum(dim=2) > 0).sum(dim=1)                                           # [B]
        G = labels.shape[1]
        if B * G * A * self.num_classes > (1 << 27):                                          # bound the [B,G,A,nc] cost tensor
            G = max(int(nlabel.max()), 1)
        gt_valid = torch.arange(G, device=labels.device)[None] < nlabel[:, None]
        gt_cls, gt_boxes = labels[:, :G, 0], labels[:, :G, 1:5]
        if self.fused_assign and ops.simota_supported(gt_valid, bbox_preds):
            # one HIP launch over the valid label rows instead of ~100 tensor ops over all G padded rows
            fg, matched, matched_iou = ops.simota_assign(grids, strides, gt_boxes, gt_cls, gt_valid, bbox_preds.detach(),
                                                         obj_preds.detach(), cls_preds.detach())
        else:
            fg, matched, matched_iou = self._assign(grids, strides, gt_boxes, gt_cls, gt_valid, bbox_preds.detach(),
                                                    obj_preds.detach(), cls_preds.detach())
        num_fg = fg.sum().to(outputs.dtype).clamp(min=1)
        num_gts = nlabel.sum().to(outputs.dtype).clamp(min=1)
        fgf = fg.to(outputs.dtype)
        idx = matched[..., None]
        reg_t = torch.gather(gt_boxes, 1, idx.expand(-1, -1, 4))                              # [B,A,4]
        cls_id = torch.gather(gt_cls, 1, matched).to(torch.int64).clamp(0, self.num_classes - 1)
        cls_t = F.one_hot(cls_id, self.num_classes).to(outputs.dtype) * matched_iou[..., None]
        safe_reg_t = torch.where(fg[..., None], reg_t, bbox_preds.detach())                   # keep masked rows finite
        loss_iou = (self.iou_loss(bbox_preds.reshape(-1, 4), safe_reg_t.reshape(-1, 4)).view(B, A) * fgf).sum() / num_fg
        loss_obj = self.bcewithlog_loss(obj_preds.reshape(-1, 1), fgf.reshape(-1, 1)).sum() / num_fg
        loss_cls = (self.bcewithlog_loss(cls_preds, cls_t) * fgf[..., None]).sum() / num_fg
        if self.use_l1:
            s = strides[0][None, :, None]
            l1_xy = reg_t[..., :2] / s - grids
            l1_wh = torch.log(torch.where(fg[..., None], reg_t[..., 2:], torch.ones_like(reg_t[..., 2:])) / s + 1e-8)
            l1_t = torch.cat([l1_xy, l1_wh], -1)
            loss_l1 = (self.l1_loss(origin_preds, l1_t) * fgf[..., None]).sum() / num_fg
        else:
            loss_l1 = 0.0
        reg_weight = 5.0
        loss = reg_weight * loss_iou + loss_obj + loss_cls + loss_l1
        return loss, reg_weight * loss_iou, loss_obj, loss_cls, loss_l1, num_fg / num_gts


class SpikingYOLOXHead(YOLOXHead):
    """reference: yolox/models/spiking_yolo_head.py.  ``full_spike=False``: firing-rate input (mean over T) into the
    ANN head; ``full_spike=True``: the head itself is converted and its output currents are averaged over T."""

    def __init__(self, num_classes, width=1.0, strides=[8, 16, 32], in_channels=[256, 512, 1024], act='silu',
                 depthwise=False, spike_fn=None, full_spike=False):
        super().__init__(num_classes, width, strides, in_channels, act, depthwise)
        self.full_spike = full_spike
        if full_spike:
            convert_to_spiking(self, spike_fn=spike_fn)

    def _prepare(self, x):
        if self.full_spike:
            return x
        return x[1] if isinstance(x, tuple) else ops.time_mean(x)
