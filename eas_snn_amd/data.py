"""Synthetic event streams and their GPU-side preprocessing (no real datasets in scope).

Stream definition = BASELINE.md section 2 / SURVEY.md 8d: per sample ``n_events`` events over a 200 ms window
on the 240x304 Gen1 sensor, ``t`` sorted uniform, ``x``,``y`` uniform, ``p`` Bernoulli(0.5), numpy default_rng.
Raw events (9 B/event) are what crosses PCIe; binning (K1) and canvas padding run on the GPU.
"""
import numpy as np
import torch

from . import ops


def synth_event_batch(batch, n_events=200_000, height=240, width=304, t0=1_000_000, span_us=200_000, seed=0):
    """-> dict of numpy arrays (t u32, x u16, y u16, p u8, offsets i64[B+1])."""
    rng = np.random.default_rng(seed)
    n = batch * n_events
    t = np.sort(rng.integers(t0, t0 + span_us, size=(batch, n_events), dtype=np.int64), axis=1).astype(np.uint32).reshape(-1)
    x = rng.integers(0, width, size=n, dtype=np.int64).astype(np.uint16)
    y = rng.integers(0, height, size=n, dtype=np.int64).astype(np.uint16)
    p = (rng.random(n) < 0.5).astype(np.uint8)
    off = np.arange(batch + 1, dtype=np.int64) * n_events
    return dict(t=t, x=x, y=y, p=p, offsets=off)


def events_to_device(ev, device):
    return {k: torch.from_numpy(v).to(device, non_blocking=True) for k, v in ev.items()}


def events_to_frames(ev_dev, Tm, sensor_hw, canvas_hw):
    """Device events -> model input [B, Tl=1, Tm, 2, Hc, Wc] fp32 via the HIP histogram (K1)."""
    H, W = sensor_hw
    frames = ops.event_frames(ev_dev['t'], ev_dev['x'], ev_dev['y'], ev_dev['p'], ev_dev['offsets'], Tm, H, W, canvas_hw[0], canvas_hw[1])
    return frames.unsqueeze(1)


def synth_targets(batch, canvas_hw, device, n_boxes=2, max_labels=50):
    """[B, 50, 5] rows (cls, cx, cy, w, h), zero padded (yolox/data/.../event_data_augment.py:19-65 contract)."""
    Hc, Wc = canvas_hw
    t = torch.zeros(batch, max_labels, 5)
    boxes = [(0, 0.3, 0.4, 0.25, 0.3), (1, 0.7, 0.6, 0.2, 0.35), (0, 0.5, 0.5, 0.4, 0.4), (1, 0.2, 0.7, 0.15, 0.2)]
    for i in range(min(n_boxes, len(boxes))):
        c, cx, cy, w, h = boxes[i]
        t[:, i] = torch.tensor([c, cx * Wc, cy * Hc, w * Wc, h * Hc])
    return t.to(device)


class SyntheticEventDataset:
    def __init__(self, exp, length=1024, n_events=200_000):
        self.exp, self.length, self.n_events = exp, length, n_events

    def __len__(self):
        return self.length


class SyntheticEventLoader:
    """Iterable of (frames [B,Tl,Tm,2,H,W] fp32 on the GPU, targets [B,50,5]); one 'epoch' = ``iters`` batches."""

    def __init__(self, exp, batch_size, iters=16, n_events=200_000, sensor_hw=(240, 304)):
        self.exp, self.batch_size, self.iters, self.n_events, self.sensor_hw = exp, batch_size, iters, n_events, sensor_hw
        self.dataset = SyntheticEventDataset(exp)

    def __len__(self):
        return self.iters

    def close_mosaic(self):
        pass

    def __iter__(self):
        dev = torch.device('cuda', torch.cuda.current_device())
        for i in range(self.iters):
            ev = events_to_device(synth_event_batch(self.batch_size, self.n_events, *self.sensor_hw, seed=i), dev)
            frames = events_to_frames(ev, self.exp.Tm, self.sensor_hw, self.exp.input_size)
            yield frames, synth_targets(self.batch_size, self.exp.input_size, dev)

class SyntheticEvalDataset(SyntheticEventDataset):
    """what the evaluator asks of its dataset (event_evaluator.py:166-169, :348-356): map_val without random augmentation, class / sample names"""
    map_val, random_aug = True, False

    def __init__(self, exp, length=256, n_events=200_000):
        super().__init__(exp, length, n_events)
        self.class_names = [str(i) for i in range(exp.num_classes)]
        self.sample_names = [f'synthetic_{i:06d}' for i in range(length)]


class SyntheticEvalLoader:
    """Iterable of ``(frames [B,Tl,Tm,2,H,W] fp32 on the GPU, labels [B][n,5] rows (x, y, w, h, cls), (heights, widths), ids)`` -- the tuple
    the reference's evaluation loader yields (gen1_collact_func; event_evaluator.py:183).  Sample ``i`` is the seeded stream ``i`` on every
    rank; ``indices`` are this rank's samples (rank, rank + world, ... like DistributedSampler(shuffle=False), event_yolox_base.py:489-494);
    the last batch may be short."""

    def __init__(self, exp, batch_size, indices, n_events=200_000, sensor_hw=(240, 304), dataset=None):
        self.exp, self.batch_size, self.indices, self.n_events, self.sensor_hw = exp, batch_size, list(indices), n_events, sensor_hw
        self.dataset = dataset if dataset is not None else SyntheticEvalDataset(exp, n_events=n_events)

    def __len__(self):
        return (len(self.indices) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        dev = torch.device('cuda', torch.cuda.current_device())
        H, W = self.sensor_hw
        for b in range(len(self)):
            ids = self.indices[b * self.batch_size:(b + 1) * self.batch_size]
            parts = [synth_event_batch(1, self.n_events, H, W, seed=10_000 + i) for i in ids]
            ev = {k: np.concatenate([p[k] for p in parts]) for k in ('t', 'x', 'y', 'p')}
            ev['offsets'] = np.arange(len(ids) + 1, dtype=np.int64) * self.n_events
            frames = events_to_frames(events_to_device(ev, dev), self.exp.Tm, self.sensor_hw, self.exp.test_size)
            labels = [torch.tensor([[0.3 * W, 0.4 * H, 0.25 * W, 0.3 * H, float(i % self.exp.num_classes)]]) for i in ids]
            yield frames, labels, (torch.full((len(ids),), H), torch.full((len(ids),), W)), torch.tensor(ids)


# ------------------------------------------------------------------------------------------------ augmentation parameters
def letterbox_params(ih, iw, h, w, letterbox=True, center=False):
    """(nw, nh, dx, dy, flip) of the deterministic branch of GEN1Dataset.get_random_data (gen1.py:438-483)."""
    if letterbox:
        scale = min(w / iw, h / ih)
        nw, nh = int(iw * scale), int(ih * scale)
        dx, dy = ((w - nw) // 2, (h - nh) // 2) if center else (0, 0)
    else:
        nw, nh, dx, dy = w, h, 0, 0
    return nw, nh, dx, dy, 0


def jitter_params(ih, iw, h, w, jitter=.3, rng=np.random):
    """(nw, nh, dx, dy, flip) of the random branch of get_random_data (gen1.py:485-504).  Six uniform draws from ``rng`` in the
    reference's order: two for the aspect-ratio distortion (numerator, denominator, each in [1-jitter, 1+jitter)), the scale in
    [0.4, 1), the horizontal and vertical paste offsets, the flip coin."""
    u = lambda lo, hi: lo + (hi - lo) * rng.rand()
    aspect = (iw / ih) * u(1 - jitter, 1 + jitter) / u(1 - jitter, 1 + jitter)
    scale = u(.4, 1)
    if aspect >= 1:                      # wide result: the width is scaled, the height follows the aspect ratio
        nw = int(scale * w)
        nh = int(nw / aspect)
    else:
        nh = int(scale * h)
        nw = int(nh * aspect)
    dx, dy = int(u(0, w - nw)), int(u(0, h - nh))
    return nw, nh, dx, dy, int(u(0, 1) < .5)


def transform_boxes(bboxes, params, ih, iw, h, w, rng=None):
    """Box side of get_random_data (gen1.py:462-473 / :509-520) for given draw results: rows (x1, y1, x2, y2, ...) are
    truncated to int64, optionally shuffled with ``rng``, mapped through the resize / paste / flip (every intermediate result
    truncated to int64 again, as the reference's in-place integer array does), clipped to the canvas, and boxes that are not
    wider and higher than one pixel are dropped.  Returns float32."""
    nw, nh, dx, dy, flip = params
    box = np.array(bboxes, dtype=np.int64)
    if len(box) == 0:
        return box.astype(np.float32)
    if rng is not None:
        rng.shuffle(box)
    xs = (box[:, [0, 2]] * nw / iw + dx).astype(np.int64)        # float64 arithmetic, truncation toward zero
    ys = (box[:, [1, 3]] * nh / ih + dy).astype(np.int64)
    if flip:
        xs = w - xs[:, ::-1]
    box[:, 0], box[:, 2] = np.maximum(xs[:, 0], 0), np.minimum(xs[:, 1], w)
    box[:, 1], box[:, 3] = np.maximum(ys[:, 0], 0), np.minimum(ys[:, 1], h)
    keep = ((box[:, 2] - box[:, 0]) > 1) & ((box[:, 3] - box[:, 1]) > 1)
    return box[keep].astype(np.float32)


def events_to_frames_augmented(ev_dev, Tm, sensor_hw, canvas_hw, params):
    """events -> histogram (K1) -> resize / paste / flip on the device -> [B, 1, Tm, 2, Hc, Wc] fp32; ``params``: per-sample
    (nw, nh, dx, dy, flip) rows."""
    H, W = sensor_hw
    counts = ops.event_histogram(ev_dev['t'], ev_dev['x'], ev_dev['y'], ev_dev['p'], ev_dev['offsets'], Tm, H, W)
    par = torch.as_tensor(np.asarray(params, dtype=np.int32).reshape(-1, 5)).to(counts.device)
    return ops.counts_letterbox(counts, par, canvas_hw[0], canvas_hw[1]).unsqueeze(1)
