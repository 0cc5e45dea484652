"""TEST INFRASTRUCTURE -- numpy restatement of the reference's event -> voxel
preprocessing.  Integer work: results must be
bit-exact against the reference (pinned by tests/golden/events_*.npz, which
were produced by calling the reference's own functions; see gen_golden.py).

Events are passed as four parallel arrays (t sorted ascending, x, y, p), the
struct-of-arrays form of the reference's structured array
(yolox/utils/psee_loader/io/dat_events_tools.py:24-52: t:u4, x:u2, y:u2, p:u1).
"""
import numpy as np


def slice_bounds(t, num_slice):
    """[start, end) index pairs of the ``num_slice`` equal time windows.

    Follows GEN1Dataset.slice_events (yolox/data/datasets/gen1.py:313-328;
    identical copies at ncaltech.py:368-378, rvt_gen4.py:411-426,
    gen4.py:445-460) with overlap = 0:
        window = (t[-1] - t[0]) // num_slice            (integer floor, dtype of t)
        start_k = t[0] + k * window ; end_k = start_k + window
        indices = searchsorted(t, ., side='left')
    Events with t >= t[0] + num_slice*window fall in no slice.  Empty input -> None.
    """
    t = np.asarray(t)
    if t.shape[0] <= 0:
        return None
    window = (t[-1] - t[0]) // num_slice
    starts = np.arange(num_slice, dtype=np.int64) * np.int64(window) + np.int64(t[0])
    ends = starts + np.int64(window)
    lo = np.searchsorted(t, starts, side='left')
    hi = np.searchsorted(t, ends, side='left')
    return list(zip(lo.tolist(), hi.tolist()))


def aggregate_sum(x, y, p, height, width):
    """Per-polarity event-count frame [2, H, W] (float64, like the reference).

    Follows GEN1Dataset.agrregate(method='sum') (gen1.py:333-349): channel 0
    counts p == 0, channel 1 counts p != 0, position = y * W + x.
    ``x is None`` (no events) -> zeros.  N-Caltech's np.add.at form
    (ncaltech.py:231-237 with count_measure) gives the same counts.
    """
    frame = np.zeros((2, height * width), dtype=np.float64)
    if x is None or len(x) == 0:
        return frame.reshape(2, height, width)
    x = np.asarray(x).astype(np.int64)
    y = np.asarray(y).astype(np.int64)
    p = np.asarray(p)
    neg = p == 0
    for c, m in enumerate((neg, ~neg)):
        pos = y[m] * width + x[m]
        counts = np.bincount(pos, minlength=0)
        frame[c, :counts.size] += counts
    return frame.reshape(2, height, width)


def micro_sum(t, x, y, p, num_micro, height, width):
    """[Tm, 2, H, W] float64 count frames (gen1.py:355-360 'micro_sum')."""
    bounds = slice_bounds(t, num_micro)
    if bounds is None:
        return np.zeros((num_micro, 2, height, width), dtype=np.float64)
    return np.stack([aggregate_sum(x[a:b], y[a:b], p[a:b], height, width) for a, b in bounds])


def stacked_hist_event_sum(ev_repr, num_slice, height, width):
    """RVT stacked-histogram slices -> per-polarity counts, restating RVTGEN4Dataset.generate_slices(method='event_sum')
    (yolox/data/datasets/rvt_gen4.py:118-125) from the point where ``ev_repr`` [n <= num_slice, 2*nbins, H, W] (u8) has been
    read: reshape to [n, 2, nbins, H, W], sum over the bins (numpy promotes u8 sums to 64-bit: no wrap-around), zero slices
    in FRONT up to num_slice, leading macro axis of 1.  Returns float64 [1, num_slice, 2, H, W] like the reference.
    Pinned by tests/golden/stacked_hist.npz (generated from the reference method, oracle/gen_golden.py)."""
    ev_repr = np.asarray(ev_repr)
    n = ev_repr.shape[0]
    summed = ev_repr.reshape(n, 2, -1, height, width).sum(axis=2)
    pad = np.zeros([num_slice - n] + list(summed.shape[1:]))
    return np.expand_dims(np.concatenate([pad, summed], axis=0), axis=0)


def pad_to_canvas(frames, canvas_h, canvas_w):
    """zero padding bottom/right of the trailing two axes: the validation letterbox of get_random_data at scale 1, dx = dy = 0
    (gen1.py:447-455, rvt_gen4.py:516-533)"""
    out = np.zeros(frames.shape[:-2] + (canvas_h, canvas_w), dtype=frames.dtype)
    out[..., :frames.shape[-2], :frames.shape[-1]] = frames
    return out


def micro_sum_batch(t, x, y, p, sample_offsets, num_micro, height, width):
    """Batched form used to check the HIP kernel: events of sample b are
    ``[sample_offsets[b], sample_offsets[b+1])``.  Returns int32 [B, Tm, 2, H, W]."""
    B = len(sample_offsets) - 1
    out = np.zeros((B, num_micro, 2, height, width), dtype=np.int32)
    for b in range(B):
        a, e = int(sample_offsets[b]), int(sample_offsets[b + 1])
        out[b] = micro_sum(t[a:e], x[a:e], y[a:e], p[a:e], num_micro, height, width).astype(np.int32)
    return out


def voxel_grid(t, x, y, p_signed, n_time_bins, height, width):
    """Bilinear-in-time voxel grid [nb, 1, H, W] float64.

    Follows to_voxel_grid_numpy (yolox/utils/event_reps.py:30-89):
        ts = nb * (t - t0) / (t_last - t0); ti = int(ts); dt = ts - ti
        grid[ti]   += pol * (1 - dt)   if ti     < nb
        grid[ti+1] += pol * dt         if ti + 1 < nb
    with pol in {-1, +1}.  The reference rewrites ``p == 0 -> -1`` in place
    (event_reps.py:60), which only has defined behaviour for a SIGNED polarity
    dtype (as ncaltech.py:44 uses); that is the semantics restated here:
    ``p_signed`` is an integer array, 0 is mapped to -1.
    A single-timestamp stream divides by zero in the reference (nan/inf ts);
    here it is declared invalid input and returns zeros.
    """
    if len(t) == 0:
        return np.zeros((n_time_bins, 1, height, width), dtype=np.float64)
    t = np.asarray(t)
    grid = np.zeros(n_time_bins * height * width, dtype=np.float64)
    span = float(t[-1]) - float(t[0])
    if span == 0.0:
        return grid.reshape(n_time_bins, 1, height, width)
    ts = n_time_bins * (t.astype(np.float64) - float(t[0])) / span
    xs = np.asarray(x).astype(np.int64)
    ys = np.asarray(y).astype(np.int64)
    pol = np.asarray(p_signed).astype(np.float64).copy()
    pol[pol == 0] = -1.0
    tis = ts.astype(np.int64)
    dts = ts - tis
    left = pol * (1.0 - dts)
    right = pol * dts
    v = tis < n_time_bins
    np.add.at(grid, xs[v] + ys[v] * width + tis[v] * width * height, left[v])
    v = (tis + 1) < n_time_bins
    np.add.at(grid, xs[v] + ys[v] * width + (tis[v] + 1) * width * height, right[v])
    return grid.reshape(n_time_bins, 1, height, width)


def voxel_cube(t, x, y, p, num_slices, tbins, height, width):
    """Voxel cube [num_slices, 2*tbins, H, W] float64 (event counts).

    Follows to_voxel_cube_numpy (yolox/utils/event_reps.py:92-138):
        t -= t[0]; window = (t[-1] - t[0]) // num_slices; keep t < window * num_slices
        slice   = floor(float32(t) / float32(window))         (torch int32 / int64 tensor division is float32)
        tbin    = (t % window) // (window / tbins)             (float64 floor division)
        channel = (p + 1) * (tbin + 1) - 1                     (p in {0, 1}: p=0 -> tbin, p=1 -> 2*tbin + 1;
                                                                 so (p=0, tbin=1) and (p=1, tbin=0) share channel 1 and
                                                                 channel 2 stays empty for tbins = 2 -- kept as is)
        counts accumulated by sparse_coo(...).coalesce()
    The reference needs a SIGNED timestamp dtype here (torch cannot promote a uint32 window); ``t`` is taken as int64.
    Empty input -> zeros (the reference returns the transposed shape [n, 2*tbins, W, H] for it; here always [.., H, W]).
    window == 0 keeps no event -> zeros.
    """
    out = np.zeros((num_slices, 2 * tbins, height, width), dtype=np.float64)
    if len(t) == 0:
        return out
    t = np.asarray(t).astype(np.int64)
    t = t - t[0]
    window = (t[-1] - t[0]) // num_slices
    keep = t < window * num_slices
    if window == 0 or not keep.any():
        return out
    t, xs, ys, ps = t[keep], np.asarray(x)[keep].astype(np.int64), np.asarray(y)[keep].astype(np.int64), np.asarray(p)[keep].astype(np.int64)
    sl = np.floor(t.astype(np.int32).astype(np.float32) / np.float32(window)).astype(np.int64)
    tbin = (t % window) // (window / tbins)
    ch = ((ps + 1) * (tbin + 1) - 1).astype(np.int64)
    np.add.at(out, (sl, ch, ys, xs), 1.0)
    return out


def time_surface(t, x, y, p, num_slices, height, width, tau):
    """Exponential time surfaces [num_slices, 2, H, W] float64 at the end of each micro-slice.

    Follows GEN1Dataset.agrregate(method='timesurface') (gen1.py:362-369) = slice_events (gen1.py:313-328) +
    to_timesurface_numpy (yolox/utils/event_reps.py:141-160):
        memory[p, y, x] = timestamp of the latest event so far (slices are applied in order, later writes win)
        surface_i = exp(-((i + 1) * window + t[0] - memory) / tau)        (memory starts at 0: untouched pixels give
                                                                          exp(-end_i / tau), tiny but not zero)
    Empty input -> zeros.  window == 0 makes the reference index an empty first slice (IndexError): declared invalid
    input here and returns zeros.
    """
    out = np.zeros((num_slices, 2, height, width), dtype=np.float64)
    bounds = slice_bounds(t, num_slices)
    if bounds is None:
        return out
    t = np.asarray(t).astype(np.int64)
    window = (t[-1] - t[0]) // num_slices
    if window == 0:
        return out
    memory = np.zeros((2, height, width), dtype=np.int64)
    xs, ys, ps = np.asarray(x).astype(np.int64), np.asarray(y).astype(np.int64), np.asarray(p).astype(np.int64)
    for i, (a, b) in enumerate(bounds):
        # later events win: timestamps ascend, so the maximum is the last write
        np.maximum.at(memory, (ps[a:b], ys[a:b], xs[a:b]), t[a:b])
        diff = -((i + 1) * window + t[0] - memory)
        out[i] = np.exp(diff / tau)
    return out


def _linear_taps(n_src, n_dst):
    """cv2.resize INTER_LINEAR tap table along one axis (OpenCV imgproc/src/resize.cpp, cv::resize generic path for
    CV_64F: float32 weights, borders clamped with weight 0): returns (s0, s1, w0, w1)."""
    scale = np.float64(n_src) / np.float64(n_dst)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0, 0
    hi = s >= n_src - 1
    f[hi], s[hi] = 0, n_src - 1
    return s, np.minimum(s + 1, n_src - 1), (np.float32(1) - f).astype(np.float64), f.astype(np.float64)


def resize_linear(img, nw, nh):
    """cv2.resize(img [H,W,C] float64, (nw, nh), INTER_LINEAR) restated (parity unpinned: cv2 is not installed here).
    Horizontal pass, then vertical pass, float64 arithmetic with float32 weights; same size -> copy."""
    H, W = img.shape[:2]
    if (nw, nh) == (W, H):
        return img.copy()
    x0, x1, a0, a1 = _linear_taps(W, nw)
    y0, y1, b0, b1 = _linear_taps(H, nh)
    a0, a1 = a0[None, :, None], a1[None, :, None]
    rows = img[:, x0] * a0 + img[:, x1] * a1                   # [H, nw, C]
    return rows[y0] * b0[:, None, None] + rows[y1] * b1[:, None, None]


def letterbox_frames(frames, params, h, w):
    """Image side of GEN1Dataset.get_random_data (gen1.py:433-521) for given draw results ``params`` =
    (nw, nh, dx, dy, flip): frames [nf, nc, ih, iw] float64 -> [nf, nc, h, w] float64."""
    nw, nh, dx, dy, flip = params
    nf, nc, ih, iw = frames.shape
    image = frames.transpose(0, 2, 3, 1)
    image = np.stack([resize_linear(im, nw, nh) for im in image])
    new_image = np.zeros([nf, h, w, nc])
    new_image[:, dy:dy + nh, dx:dx + nw] = image
    if flip:
        new_image = np.ascontiguousarray(new_image[:, :, ::-1, :])
    return np.transpose(new_image, (0, 3, 1, 2))


def synth_events(n_events, height=240, width=304, t0=1_000_000, span_us=200_000, seed=0):
    """Synthetic stream of BASELINE.md section 2 / SURVEY 8d config 1."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.integers(t0, t0 + span_us, size=n_events, dtype=np.int64)).astype(np.uint32)
    x = rng.integers(0, width, size=n_events, dtype=np.int64).astype(np.uint16)
    y = rng.integers(0, height, size=n_events, dtype=np.int64).astype(np.uint16)
    p = (rng.random(n_events) < 0.5).astype(np.uint8)
    return t, x, y, p


# ----------------------------------------------------------------------------------------- .dat records
def parse_dat_header(buf):
    """Offset of the first event record, event type and event size of a Prophesee ``.dat`` file image (bytes).
    Follows parse_header (yolox/utils/psee_loader/io/dat_events_tools.py:121-175): comment lines start with '% ';
    if there was at least one, two bytes follow (event type, event size); else type 0 / size 8."""
    pos, ncomment = 0, 0
    while True:
        end = buf.find(b'\n', pos)
        line = buf[pos:] if end < 0 else buf[pos:end + 1]
        if line[:2] != b'% ':
            break
        ncomment += 1
        pos = end + 1
    if ncomment > 0:
        return pos + 2, int(buf[pos]), int(buf[pos + 1])
    return pos, 0, 8


def decode_dat_records(rec):
    """rec: uint32 [nev, 2] little-endian (t, packed) -> (t u4, x u2, y u2, p u1).
    Follows load_td_data (dat_events_tools.py:44-54): x = bits 0..13, y = bits 14..27, p = bit 28 of the second word."""
    rec = np.asarray(rec, dtype=np.uint32).reshape(-1, 2)
    packed = rec[:, 1].astype(np.int64)
    x = np.bitwise_and(packed, 16383)
    y = np.right_shift(np.bitwise_and(packed, 268419072), 14)
    pp = np.right_shift(np.bitwise_and(packed, 268435456), 28)
    return rec[:, 0].copy(), x.astype(np.uint16), y.astype(np.uint16), pp.astype(np.uint8)


SEEK_TERM_CRITERION = 100000      # PSEELoader.seek_time's default term_criterion (psee_loader.py:196)


def seek_time(t, final_time):
    """(event index the reader stands at, its current_time) after PSEELoader.seek_time(final_time) on a recording with sorted
    timestamps ``t`` (yolox/utils/psee_loader/io/psee_loader.py:196-238), quirks included:
      * final_time past the last event -> end of file;  final_time <= 0 -> reset(): index 0 and current_time 0 (not final_time);
      * the bisection probes ``t[middle]`` while more than 100 000 events remain; a probe that hits final_time EXACTLY returns at
        once with the reader standing one event AFTER the probed one (np.fromfile has consumed it);
      * otherwise the index of the first event with t >= final_time (np.searchsorted, side='left')."""
    n = len(t)
    total = int(t[-1]) if n else 0
    if final_time > total:
        return n, total + 1
    if final_time <= 0:
        return 0, 0
    low, high = 0, n
    while high - low > SEEK_TERM_CRITERION:
        middle = (low + high) // 2
        mid = int(t[middle])
        if mid > final_time:
            high = middle
        elif mid < final_time:
            low = middle + 1
        else:
            return middle + 1, final_time
    return low + int(np.searchsorted(t[low:high], final_time)), final_time


def load_delta_t(t, pos, current_time, delta_t):
    """[pos, end): the events PSEELoader.load_delta_t(delta_t) returns from reader position ``pos`` (psee_loader.py:128-170): all
    events from there with t < current_time + delta_t."""
    if pos >= len(t):
        return pos, pos
    final_time = current_time + delta_t
    return pos, pos + int(np.searchsorted(t[pos:], final_time))


def search_events(t, timestamp, window, num_slice):
    """Event index range [start, end) GEN1Dataset.search_events returns for the label at ``timestamp`` with the 'fix_t' policy
    (yolox/data/datasets/gen1.py:217-232): the window [timestamp + window[0], + (window[1] - window[0])) first; while it is empty
    it is moved back by its own length, for at most num_slice + 2 attempts in all; the last attempt's (possibly empty) range is
    returned."""
    delta = window[1] - window[0]
    cur = timestamp + window[0]
    zero_trigger = 0
    while True:
        pos, now = seek_time(t, cur)
        a, e = load_delta_t(t, pos, now, delta)
        if e > a or zero_trigger > num_slice:
            return a, e
        zero_trigger += 1
        cur -= delta


def encode_dat_file(t, x, y, p, height=240, width=304):
    """A ``.dat`` file image (header + 8-byte records) for synthetic events -- test input only."""
    head = ('% Data file containing Event2D events.\n% Version 2\n% Date 2024-1-1 0:0:0\n'
            f'% Height {height}\n% Width {width}\n').encode('latin-1')
    rec = np.empty((len(t), 2), dtype='<u4')
    rec[:, 0] = t
    rec[:, 1] = (np.asarray(x, np.uint32) & 16383) | ((np.asarray(y, np.uint32) & 16383) << 14) | ((np.asarray(p, np.uint32) & 1) << 28)
    return head + bytes([0, 8]) + rec.tobytes()
