#!/usr/bin/env python3
"""Summarise the SQ-counter passes of scripts/pmc_conv_step.sh per convolution kernel symbol.

usage: pmc_conv_summary.py <dir with pass*/> <out.json>

Units (MI355X_MICROARCH.md, per-instruction table): SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16 wave
instruction, summed over the chip's 1024 SIMDs); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles;
GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the dispatch was resident.  Derived here:
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
"""
import collections
import csv
import glob
import json
import os
import re
import sys

SIMDS = 1024


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_]+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:80]


def main():
    root, out_path = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(root, 'pass*', '**', '*counter_collection.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                a = acc[short(row['Kernel_Name'])][row['Counter_Name']]
                a[0] += 1
                a[1] += float(row['Counter_Value'])
    dur = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(root, 'pass1', '**', '*kernel_trace.csv'), recursive=True):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                d = dur[short(row['Kernel_Name'])]
                d[0] += 1
                d[1] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
    out = {}
    for k, cs in acc.items():
        n = max(v[0] for v in cs.values())
        e = {'launches': n, 'total_us_under_profiler': round(dur[k][1], 1) if k in dur else None}
        for c, (cn, tot) in cs.items():
            e[c] = tot / max(cn, 1)          # mean per launch
        gui = e.get('GRBM_GUI_ACTIVE')
        if gui:
            cyc = gui / 8.0
            e['kernel_cycles'] = round(cyc)
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
                e['mfma_busy_frac'] = round(e['SQ_VALU_MFMA_BUSY_CYCLES'] / (SIMDS * cyc), 4)
            if 'SQ_ACTIVE_INST_VALU' in e:
                e['valu_active_frac_quad'] = round(4.0 * e['SQ_ACTIVE_INST_VALU'] / (SIMDS * cyc), 4)
            if 'SQ_ACTIVE_INST_LDS' in e:
                e['lds_active_frac_quad'] = round(4.0 * e['SQ_ACTIVE_INST_LDS'] / (SIMDS * cyc), 4)
        if e.get('SQ_WAVE_CYCLES'):
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS'):
                if c in e:
                    e[c.lower() + '_share_of_wave_cycles'] = round(e[c] / e['SQ_WAVE_CYCLES'], 4)
        if e.get('SQ_LDS_IDX_ACTIVE'):
            e['lds_bank_conflict_share'] = round(e.get('SQ_LDS_BANK_CONFLICT', 0.0) / e['SQ_LDS_IDX_ACTIVE'], 4)
        out[k] = e
    order = sorted(out, key=lambda k: -(out[k]['total_us_under_profiler'] or 0))
    import hashlib
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'eas_snn_amd', 'csrc')
    meta = {'csrc_sha16': {f: hashlib.sha256(open(os.path.join(d, f), 'rb').read()).hexdigest()[:16] for f in sorted(os.listdir(d)) if f.endswith(('.hip', '.h'))},
            'note': 'sha256[:16] of every file under eas_snn_amd/csrc at the time of the counter passes (bench.py quotes mfma_busy_measured only while they match)'}
    with open(out_path, 'w') as fh:
        json.dump(dict({k: out[k] for k in order}, _meta=meta), fh, indent=1)
    for k in order:
        e = out[k]
        print(f"{k[:64]:64s} n={e['launches']:3d} {e['total_us_under_profiler'] or 0:9.1f} us  mfma_busy {e.get('mfma_busy_frac', float('nan')):.3f}  "
              f"valu_act {e.get('valu_active_frac_quad', float('nan')):.3f}  lds_act {e.get('lds_active_frac_quad', float('nan')):.3f}  "
              f"wait_any {e.get('sq_wait_any_share_of_wave_cycles', float('nan')):.2f} wait_inst {e.get('sq_wait_inst_any_share_of_wave_cycles', float('nan')):.2f} "
              f"bank_conf {e.get('lds_bank_conflict_share', float('nan')):.2f}")


if __name__ == '__main__':
    main()
