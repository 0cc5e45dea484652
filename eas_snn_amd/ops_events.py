"""K1 and the other event representations: raw events / .dat records / stacked histograms -> count frames, voxel grids, time surfaces,
letterbox resize (reference: yolox/data/datasets/gen1.py:313-360, 433-521, psee_loader.py, event_reps.py, rvt_gen4.py:109-125).

Part of the operator layer of ``eas_snn_amd.ops`` (split by kernel family; ``ops`` re-exports everything here, so ``ops.<name>`` keeps working)."""
import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr, stream
from .ops_core import _call, _dev, _f32c, _timer_add, _timer_mark


# ------------------------------------------------------------------------------------------------ K1
def event_histogram(t, x, y, p, sample_offsets, Tm, H, W, return_oob=False):
    """Per-sample micro-slice count frames: int32 [B, Tm, 2, H, W] (bit-exact 'micro_sum')."""
    _dev(t, x, y, p, sample_offsets)
    assert t.dtype == torch.uint32 or t.dtype == torch.int32, 't must be 32-bit timestamps'
    assert x.dtype in (torch.uint16, torch.int16) and y.dtype in (torch.uint16, torch.int16) and p.dtype in (torch.uint8, torch.int8)
    assert sample_offsets.dtype == torch.int64
    B = sample_offsets.numel() - 1
    out = torch.empty((B, Tm, 2, H, W), dtype=torch.int32, device=t.device)
    oob = torch.empty(1, dtype=torch.int32, device=t.device) if return_oob else None
    _call('eas_event_histogram', 9 * t.numel() + 4 * out.numel(), _lib.lib().eas_event_histogram, ptr(t), ptr(x), ptr(y), ptr(p),
          t.numel(), ptr(sample_offsets), B, Tm, H, W, ptr(out), ptr(oob), stream())
    return (out, oob) if return_oob else out


def event_histogram_dat(records, sample_offsets, Tm, H, W, return_oob=False):
    """Count frames int32 [B, Tm, 2, H, W] straight from .dat records: ``records`` is the raw byte image of the events
    (uint8 [8*nev], or any 8-byte-record view) already on the device; decode + binning happen in one kernel."""
    _dev(records, sample_offsets)
    assert sample_offsets.dtype == torch.int64
    rec = records.contiguous().view(torch.uint8)
    assert rec.numel() % 8 == 0, '.dat event records are 8 bytes'
    nev = rec.numel() // 8
    B = sample_offsets.numel() - 1
    out = torch.empty((B, Tm, 2, H, W), dtype=torch.int32, device=rec.device)
    oob = torch.empty(1, dtype=torch.int32, device=rec.device) if return_oob else None
    _call('eas_event_histogram_dat', 8 * nev + 4 * out.numel(), _lib.lib().eas_event_histogram_dat, ptr(rec), nev, ptr(sample_offsets), B, Tm,
          H, W, ptr(out), ptr(oob), stream())
    return (out, oob) if return_oob else out


def event_window_search(records, label_t, window, num_slice, file_offsets=None, file_id=None):
    """Record ranges int64 [B, 2] of the events GEN1Dataset.search_events returns for every label (gen1.py:217-232), found on the
    device.  records: the record area(s) of .dat recording(s) already in HBM (uint8 [8*nev] or any 8-byte-record view);
    label_t int64 [B] label timestamps (us); window = (lo, hi) us relative to the label; file_offsets int64 [F+1] (record index of
    every recording's first event; default: one recording) and file_id int32 [B]."""
    _dev(records, label_t, file_offsets, file_id)
    rec = records.contiguous().view(torch.uint8)
    assert rec.numel() % 8 == 0 and label_t.dtype == torch.int64
    nev = rec.numel() // 8
    if file_offsets is None:
        file_offsets = torch.tensor([0, nev], dtype=torch.int64, device=rec.device)
    assert file_offsets.dtype == torch.int64 and (file_id is None or file_id.dtype == torch.int32)
    B = label_t.numel()
    ranges = torch.empty((B, 2), dtype=torch.int64, device=rec.device)
    check(_lib.lib().eas_event_window_search(ptr(rec), ptr(file_offsets.contiguous()), file_offsets.numel() - 1,
                                             ptr(file_id.contiguous() if file_id is not None else None), ptr(label_t.contiguous()), B,
                                             int(window[0]), int(window[1]), int(num_slice), ptr(ranges), stream()), 'eas_event_window_search')
    return ranges


def event_histogram_dat_ranges(records, ranges, Tm, H, W, return_oob=False):
    """Count frames int32 [B, Tm, 2, H, W] of the record ranges [B, 2] (``event_window_search``) of a .dat image in HBM: label
    timestamps in, frames out, nothing read back to the host in between."""
    _dev(records, ranges)
    rec = records.contiguous().view(torch.uint8)
    assert ranges.dtype == torch.int64 and ranges.dim() == 2 and ranges.shape[1] == 2
    B = ranges.shape[0]
    out = torch.empty((B, Tm, 2, H, W), dtype=torch.int32, device=rec.device)
    oob = torch.empty(1, dtype=torch.int32, device=rec.device) if return_oob else None
    _call('eas_event_histogram_dat', 8 * (rec.numel() // 8) + 4 * out.numel(), _lib.lib().eas_event_histogram_dat_ranges, ptr(rec),
          ptr(ranges.contiguous()), B, Tm, H, W, ptr(out), ptr(oob), stream())
    return (out, oob) if return_oob else out


def event_frames(t, x, y, p, sample_offsets, Tm, H, W, Hc, Wc):
    """raw events -> fp32 count frames on the zero-padded model canvas [B, Tm, 2, Hc, Wc] in one call (K1 + canvas)."""
    _dev(t, x, y, p, sample_offsets)
    B = sample_offsets.numel() - 1
    out = torch.empty((B, Tm, 2, Hc, Wc), dtype=torch.float32, device=t.device)
    scratch = torch.empty((B, Tm, 2, H, W), dtype=torch.int32, device=t.device)
    _call('eas_event_histogram', 9 * t.numel() + 4 * out.numel(), _lib.lib().eas_event_frames, ptr(t), ptr(x), ptr(y), ptr(p), t.numel(),
          ptr(sample_offsets), B, Tm, H, W, Hc, Wc, ptr(out), ptr(scratch), None, stream())
    return out


def stacked_hist_event_sum(hist, Hc, Wc, nbins=10, n_valid=None):
    """RVT stacked histogram u8 [B, Tm, 2*nbins, H, W] -> fp32 model input [B, 1, Tm, 2, Hc, Wc]: sum over the time bins of each
    polarity, zero padded to the canvas (RVTGEN4Dataset.generate_slices 'event_sum' + validation letterbox, rvt_gen4.py:109-125,
    516-533).  n_valid int32 [B]: samples that supply only their first n_valid[b] slices (zero slices in front)."""
    _dev(hist, n_valid)
    assert hist.dtype == torch.uint8 and hist.dim() == 5 and hist.shape[2] == 2 * nbins
    hist = hist.contiguous()
    B, Tm, _, H, W = hist.shape
    if n_valid is not None:
        assert n_valid.dtype == torch.int32 and n_valid.shape == (B,)
        n_valid = n_valid.contiguous()
    out = torch.empty((B, 1, Tm, 2, Hc, Wc), dtype=torch.float32, device=hist.device)
    _call('eas_stacked_hist_event_sum', hist.numel() + 4 * out.numel(), _lib.lib().eas_stacked_hist_event_sum, ptr(hist), ptr(n_valid), B, Tm,
          int(nbins), H, W, Hc, Wc, ptr(out), stream())
    return out


def counts_to_canvas(counts, Hc, Wc):
    """int32 [..., H, W] -> float32 [..., Hc, Wc], zero padded bottom/right."""
    _dev(counts)
    assert counts.dtype == torch.int32
    counts = counts.contiguous()
    H, W = counts.shape[-2:]
    F = counts.numel() // (H * W)
    out = torch.empty(counts.shape[:-2] + (Hc, Wc), dtype=torch.float32, device=counts.device)
    check(_lib.lib().eas_counts_to_canvas(ptr(counts), F, H, W, Hc, Wc, ptr(out), stream()), 'eas_counts_to_canvas')
    return out


def counts_letterbox(counts, params, Hc, Wc):
    """int32 counts [B, ..., H, W] -> fp32 [B, ..., Hc, Wc]: per-sample resize (cv2 INTER_LINEAR semantics) to (nw, nh), paste at
    (dx, dy), optional left-right flip; ``params`` int32 [B, 5] = (nw, nh, dx, dy, flip) (see data.letterbox_params / jitter_params)."""
    _dev(counts, params)
    assert counts.dtype == torch.int32 and params.dtype == torch.int32 and params.shape == (counts.shape[0], 5)
    counts, params = counts.contiguous(), params.contiguous()
    B, (H, W) = counts.shape[0], counts.shape[-2:]
    F = counts.numel() // (B * H * W)
    out = torch.empty(counts.shape[:-2] + (Hc, Wc), dtype=torch.float32, device=counts.device)
    check(_lib.lib().eas_counts_letterbox(ptr(counts), ptr(params), B, F, H, W, Hc, Wc, ptr(out), stream()), 'eas_counts_letterbox')
    return out


def event_voxel_grid(t, x, y, p, sample_offsets, n_bins, H, W):
    """float64 [B, n_bins, 1, H, W] bilinear-in-time voxel grid."""
    _dev(t, x, y, p, sample_offsets)
    B = sample_offsets.numel() - 1
    out = torch.empty((B, n_bins, 1, H, W), dtype=torch.float64, device=t.device)
    check(_lib.lib().eas_event_voxel_grid(ptr(t), ptr(x), ptr(y), ptr(p), t.numel(), ptr(sample_offsets), B, n_bins, H, W,
                                          ptr(out), stream()), 'eas_event_voxel_grid')
    return out


def event_voxel_cube(t, x, y, p, sample_offsets, num_slices, H, W, tbins=2):
    """int32 [B, num_slices, 2*tbins, H, W] voxel-cube counts (the reference returns them as float64)."""
    _dev(t, x, y, p, sample_offsets)
    B = sample_offsets.numel() - 1
    out = torch.empty((B, num_slices, 2 * tbins, H, W), dtype=torch.int32, device=t.device)
    check(_lib.lib().eas_event_voxel_cube(ptr(t), ptr(x), ptr(y), ptr(p), t.numel(), ptr(sample_offsets), B, num_slices, tbins, H, W,
                                          ptr(out), stream()), 'eas_event_voxel_cube')
    return out


def event_time_surface(t, x, y, p, sample_offsets, num_slices, H, W, tau=50e3):
    """float64 [B, num_slices, 2, H, W] exponential time surfaces at the end of every micro-slice."""
    _dev(t, x, y, p, sample_offsets)
    B = sample_offsets.numel() - 1
    ws = torch.empty((B, num_slices, 2, H, W), dtype=torch.int32, device=t.device)
    out = torch.empty((B, num_slices, 2, H, W), dtype=torch.float64, device=t.device)
    check(_lib.lib().eas_event_time_surface(ptr(t), ptr(x), ptr(y), ptr(p), t.numel(), ptr(sample_offsets), B, num_slices, H, W,
                                            float(tau), ptr(ws), ptr(out), stream()), 'eas_event_time_surface')
    return out
