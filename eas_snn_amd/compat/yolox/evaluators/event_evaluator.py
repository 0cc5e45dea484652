"""Inference loop of the evaluation side of the boundary (reference: yolox/evaluators/event_evaluator.py:83-263, the part that
``tools/eval_event.py:209-211`` and ``Trainer.evaluate_and_save_model`` call).

What is kept: the constructor and ``evaluate`` signatures, per batch forward -> ``reset_net`` -> (decoder) -> ``postprocess`` with the
reference's two timers (inference, NMS; the last batch is not timed), the ``statistics`` triple reduced to rank 0, detections in the
reference's COCO-style ``data_list`` / image-wise dict shapes gathered to rank 0, the return triple ``(ap50_95, ap50, summary)``.

What is different, and why (MI355X): on the GPU the two timed phases of a batch are two HIP-graph replays (the eval forward + reset_net
is ~350 launches whose enqueue time would otherwise bound a 6 ms forward; the post-processing is ``eas_postprocess``, one launch for
the whole batch instead of per-image torchvision NMS).  The graphs are recorded on static input / output buffers per (model, batch
shape) -- the full batch and the short last batch each keep theirs across evaluations -- and dropped when a parameter or buffer of the
model moved.  With a live RCCL process group a capture waits until the group's watchdog thread holds no event
(``yolox.utils.wait_process_group_idle``: a condition read from the flight recorder, not a delay); where that cannot be established
the evaluator launches eagerly.

mAP itself (COCO API) is outside the hot path: it is computed when ``pycocotools`` is importable, else the AP fields
are ``None`` and the summary says so -- detections, timings and the loop are what this file provides."""
import itertools
import os
import time
from collections import ChainMap, defaultdict

import torch

from yolox.utils import gather, get_rank, is_main_process, postprocess, synchronize, time_synchronized, wait_process_group_idle

__all__ = ['EventEvaluator']


def _reset_net(model):
    from spikingjelly.activation_based.functional import reset_net
    reset_net(model)


class _NoGraph(Exception):
    """recording is not safe right now (a live process group whose watchdog cannot be observed): the caller launches eagerly"""


class _GraphedBatch:
    """forward + reset_net as one HIP graph and the device half of the post-processing as a second one, on static buffers"""

    def __init__(self, model, imgs, snn_reset, num_classes, confthre, nmsthre):
        from eas_snn_amd import ops
        self.static_in = imgs.clone()
        self.key = self.signature(model, imgs)
        from eas_snn_amd import _lib
        side = _lib.private_stream()          # the capture stream: never a pooled stream (the process group's stream is one of those)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.no_state_writeback() if snn_reset else _null():
            for _ in range(2):                           # allocator warm-up on the capture stream
                out = model(self.static_in)
                if snn_reset:
                    _reset_net(model)
                ops.postprocess_device(out, num_classes, confthre, nmsthre)
            torch.cuda.synchronize()
            live_group = torch.distributed.is_initialized()
            # ProcessGroupNCCL's watchdog polls the events of collectives it has not retired yet; it must hold none while a capture is
            # open (trainer.py TrainStep.capture).  Wait for that condition; if it cannot be observed, do not record.
            if live_group and not wait_process_group_idle():
                raise _NoGraph()
            self.g_fwd, self.g_post = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            mode = 'thread_local' if live_group else 'global'
            with torch.cuda.graph(self.g_fwd, stream=side, capture_error_mode=mode):
                self.out = model(self.static_in)
                if snn_reset:
                    _reset_net(model)
            if live_group and not wait_process_group_idle():
                raise _NoGraph()
            with torch.cuda.graph(self.g_post, stream=side, capture_error_mode=mode):
                self.rows, self.counts = ops.postprocess_device(self.out, num_classes, confthre, nmsthre)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    @staticmethod
    def signature(model, imgs):
        ptrs = tuple(t.data_ptr() for t in itertools.chain(model.parameters(), model.buffers()))
        return (id(model), tuple(imgs.shape), imgs.device, hash(ptrs))

    def forward(self, imgs):
        self.static_in.copy_(imgs)
        self.g_fwd.replay()
        return self.out

    def post(self):
        self.g_post.replay()
        counts = self.counts.tolist()                    # the one host synchronisation: the result is a ragged list
        return [self.rows[i, :n].clone() if n else None for i, n in enumerate(counts)]


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class EventEvaluator:
    """reference: yolox/evaluators/event_evaluator.py:83-120 (same arguments)."""

    def __init__(self, dataloader, img_size, confthre, nmsthre, num_classes, testdev=False, per_class_AP=True, per_class_AR=True,
                 snn_reset=False):
        self.dataloader = dataloader
        self.img_size = img_size
        self.confthre = confthre
        self.nmsthre = nmsthre
        self.num_classes = num_classes
        self.testdev = testdev
        self.per_class_AP = per_class_AP
        self.per_class_AR = per_class_AR
        # EventExp passes ``use_spike`` here (event_yolox_base.py:533): the option strings 'True' / 'full_spike*' are truthy and so is
        # 'False' in the reference -- resetting a network without neurons is a no-op, so that quirk is harmless and kept
        self.snn_reset = snn_reset
        self.use_graph = os.environ.get('EAS_EVAL_GRAPH', '1') == '1'
        self._graphed = {}                  # signature -> _GraphedBatch (full batch, short last batch)
        self.graphs_recorded = 0
        self.last_statistics = None

    # ------------------------------------------------------------------ the loop
    def evaluate(self, model, distributed=False, half=False, trt_file=None, decoder=None, test_size=None, return_outputs=False):
        """-> (ap50_95, ap50, summary) [, image-wise outputs].  Puts the model into eval mode (like the reference)."""
        if half:
            raise NotImplementedError('the HIP hot path computes in fp32 (reference parity); --fp16 is not provided')
        if trt_file is not None:
            raise NotImplementedError('TensorRT engines do not exist on MI355X; the eval forward is a HIP-graph replay instead')
        model = model.eval()
        ds = getattr(self.dataloader, 'dataset', None)
        assert ds is None or (getattr(ds, 'map_val', True) and not getattr(ds, 'random_aug', False)), \
            'the dataset must be set as the mode of map val. and not random_aug'
        dev = next(model.parameters()).device
        graph_ok = self.use_graph and dev.type == 'cuda' and decoder is None
        if dev.type == 'cuda':
            # nobody writes weights or BatchNorm statistics between the first and the last batch: pack / derive them once (ops.frozen_weights)
            from eas_snn_amd import ops
            with ops.frozen_weights(model):
                return self._evaluate_loop(model, distributed, decoder, return_outputs, dev, graph_ok)
        return self._evaluate_loop(model, distributed, decoder, return_outputs, dev, graph_ok)

    def _evaluate_loop(self, model, distributed, decoder, return_outputs, dev, graph_ok):
        data_list, output_data, gt_dict = [], {}, {}
        inference_time = nms_time = 0.0
        n_batches = len(self.dataloader)
        n_samples = max(n_batches - 1, 1)
        for cur_iter, (imgs, labels, info_imgs, ids) in enumerate(self.dataloader):
            with torch.no_grad():                       # (the reference: inference_mode; graph replays write ordinary static buffers)
                imgs = imgs.to(dev, torch.float32)
                # the last batch may be short: not timed, as in the reference (:191-193)
                timed = cur_iter < n_batches - 1
                batch = self._graph_for(model, imgs) if (graph_ok and self.use_graph) else None
                if timed:
                    # the device is idle when the clock starts: the loader's event binning and the copy of the batch above are not "forward"
                    start = time_synchronized()
                if batch is not None:
                    outputs = batch.forward(imgs)
                else:
                    outputs = model(imgs)
                    if self.snn_reset:
                        _reset_net(model)
                    if decoder is not None:
                        outputs = decoder(outputs, dtype=outputs.type())
                if timed:
                    infer_end = time_synchronized()
                    inference_time += infer_end - start
                outputs = batch.post() if batch is not None else postprocess(outputs, self.num_classes, self.confthre, self.nmsthre)
                if timed:
                    nms_end = time_synchronized()
                    nms_time += nms_end - infer_end
            elems, image_wise = self.convert_to_coco_format(outputs, info_imgs, ids, return_outputs=True)
            data_list.extend(elems)
            output_data.update(image_wise)
            for label, height, width, img_id in zip(labels, info_imgs[0], info_imgs[1], ids):
                label = torch.as_tensor(label)
                gt_dict[int(img_id)] = {'bboxes': label[:, :4].tolist(), 'width': int(width), 'height': int(height),
                                        'category_ids': [int(c) for c in label[:, 4]]}
        statistics = torch.tensor([inference_time, nms_time, n_samples], dtype=torch.float32, device=dev)
        if distributed:
            # ranks finish at different times: meet first, then collect everything on rank 0 (:236-247)
            synchronize()
            data_list = list(itertools.chain(*gather(data_list, dst=0)))
            gt_dict = dict(ChainMap(*gather(gt_dict, dst=0)))
            output_data = dict(ChainMap(*gather(output_data, dst=0)))
            torch.distributed.reduce(statistics, dst=0)
        self.last_statistics = statistics
        eval_results = self.evaluate_prediction(data_list, gt_dict, statistics)
        synchronize()
        if return_outputs:
            return eval_results, output_data
        return eval_results

    def _graph_for(self, model, imgs):
        """the recorded batch for this (model, parameter addresses, input shape); a loader whose last batch is short alternates between two
        signatures in every evaluation -- both stay recorded"""
        key = _GraphedBatch.signature(model, imgs)
        g = self._graphed.get(key)
        if g is None:
            for k in [k for k in self._graphed if k[0] != key[0] or k[3] != key[3]]:
                del self._graphed[k]                    # another model, or this one's tensors moved: free those graphs' pools first
            if len(self._graphed) >= 4:
                self._graphed.pop(next(iter(self._graphed)))
            try:
                g = self._graphed[key] = _GraphedBatch(model, imgs, bool(self.snn_reset), self.num_classes, self.confthre, self.nmsthre)
                self.graphs_recorded += 1
            except _NoGraph:
                self.use_graph = False                  # eager launches from here on (see the module docstring)
                return None
        return g

    # ------------------------------------------------------------------ detections -> COCO-style records (:265-323)
    def convert_to_coco_format(self, outputs, info_imgs, ids, return_outputs=False):
        """rows (x1, y1, x2, y2, obj, cls_conf, cls) per image -> boxes scaled back to the raw sensor size, records
        {image_id, category_id, bbox xywh, score, segmentation}; image-wise {bboxes xyxy, scores, categories}"""
        data_list, image_wise = [], defaultdict(dict)
        for output, img_h, img_w, img_id in zip(outputs, info_imgs[0], info_imgs[1], ids):
            if output is None:
                continue
            output = output.detach().cpu()
            scale = min(self.img_size[0] / float(img_h), self.img_size[1] / float(img_w))
            bboxes = output[:, 0:4] / scale
            cls = output[:, 6]
            scores = output[:, 4] * output[:, 5]
            image_wise[int(img_id)] = {'bboxes': bboxes.tolist(), 'scores': scores.tolist(), 'categories': [int(c) for c in cls]}
            xywh = bboxes.clone()
            xywh[:, 2:4] -= xywh[:, 0:2]
            for box, c, s in zip(xywh.tolist(), cls.tolist(), scores.tolist()):
                data_list.append({'image_id': int(img_id), 'category_id': int(c), 'bbox': box, 'score': s, 'segmentation': []})
        if return_outputs:
            return data_list, image_wise
        return data_list

    # ------------------------------------------------------------------ summary (:396-462)
    def evaluate_prediction(self, data_dict, gt_dict, statistics):
        if not is_main_process():
            return 0, 0, None
        inference_time, nms_time, n_samples = (float(v) for v in statistics.tolist())
        per = 1000.0 / (n_samples * self.dataloader.batch_size)
        a_infer, a_nms = inference_time * per, nms_time * per
        info = ', '.join('Average {} time: {:.2f} ms'.format(k, v) for k, v in
                         zip(['forward', 'NMS', 'inference'], [a_infer, a_nms, a_infer + a_nms])) + '\n'
        if not data_dict:
            return 0, 0, info
        try:
            from pycocotools.coco import COCO
            from pycocotools.cocoeval import COCOeval
        except ImportError:
            return None, None, info + '{} detections on {} images; pycocotools is not installed: AP not computed\n'.format(
                len(data_dict), len(gt_dict))
        import contextlib
        import io
        names = getattr(getattr(self.dataloader, 'dataset', None), 'class_names', None) or [str(i) for i in range(self.num_classes)]
        gt = COCO()
        gt.dataset = {'images': [{'id': i, 'file_name': str(i), 'width': g['width'], 'height': g['height']} for i, g in gt_dict.items()],
                      'annotations': [], 'categories': [{'id': i, 'name': n, 'supercategory': n} for i, n in enumerate(names)]}
        for i, g in gt_dict.items():
            for box, c in zip(g['bboxes'], g['category_ids']):
                gt.dataset['annotations'].append({'id': len(gt.dataset['annotations']), 'image_id': i, 'category_id': c, 'bbox': box,
                                                  'area': box[2] * box[3], 'iscrowd': 0})
        gt.createIndex()
        ev = COCOeval(gt, gt.loadRes(data_dict), 'bbox')
        ev.evaluate()
        ev.accumulate()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ev.summarize()
        return ev.stats[0], ev.stats[1], info + buf.getvalue()
