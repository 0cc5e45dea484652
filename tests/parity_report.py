#!/usr/bin/env python3
"""The parity figures of the teacher-forced GPU tests as numbers: runs tests/test_gpu_model.py::_teacher_forced --
every spiking conv -> BN -> PLIF block of SYOLOX-S at the benchmark canvas 256x320 fed the ORACLE's input for that block -- in eval and
train mode and writes, per mode: layers, spike flips / neuron-steps, the worst layer's flip fraction, the worst membrane-potential
relative error (neurons whose spike train did not flip), the worst prediction-convolution element in units of the 1e-4 tolerance.
With --backward also the layer-wise backward (worst gradient element in units of its bound).

Test infrastructure (imports oracle/ through the test module); not part of the product.  bench.py quotes the file
(profiles/parity_teacher_forced_latest.json) in its `parity` object only while the recorded kernel-source hashes match the build's.

usage (GPU box): python tests/parity_report.py gpurun_out/parity_teacher_forced.json [--backward]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch


def csrc_hashes():
    d = os.path.join(ROOT, 'eas_snn_amd', 'csrc')
    return {f: hashlib.sha256(open(os.path.join(d, f), 'rb').read()).hexdigest()[:16] for f in sorted(os.listdir(d)) if f.endswith(('.hip', '.h'))}


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else os.path.join(ROOT, 'gpurun_out', 'parity_teacher_forced.json')
    import eas_snn_amd
    eas_snn_amd.hip_library()
    import test_gpu_model as M
    dev = torch.device('cuda:0')
    shape = (1, 1, 4, 2, 256, 320)
    out = {'model': 'SYOLOX-S use_spike=True T=3 Tm=4', 'canvas': [256, 320], 'batch': 1,
           'what': 'every spiking conv->BN->PLIF block fed the oracle\'s input (tests/test_gpu_model.py::_teacher_forced)'}
    for train in (False, True):
        st = M._teacher_forced(dev, 'e-yolox-s', dict(use_spike='True'), shape, train)
        out['train' if train else 'eval'] = {
            'layers': st['layers'], 'spike_flips': st['flips'], 'neuron_steps': st['steps'],
            'flip_fraction_overall': st['flips'] / max(st['steps'], 1), 'flip_fraction_worst_layer': st['worst_flip'],
            'membrane_rel_err_worst': st['worst_v'], 'membrane_tolerance': M.RTOL,
            'prediction_conv_worst_element_in_units_of_1e-4': st['pred_worst']}
    # one whole training step against the reference's own step (fixtures from the unmodified reference classes): achieved figures
    out['train_step_vs_reference_fixture'] = {name: M._train_step_figures(dev, name) for name in M.TRAIN_STEP_FIXTURES}
    if '--backward' in sys.argv:
        sb = M._teacher_forced_backward(dev, 'e-yolox-s', dict(use_spike='True'), (2, 1, 4, 2, 256, 320), 34)
        out['backward'] = {k: (float(v) if isinstance(v, (int, float)) else v) for k, v in sb.items()} if isinstance(sb, dict) else None
    out['_meta'] = {'csrc_sha16': csrc_hashes(), 'device': torch.cuda.get_device_name(0)}
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, 'w') as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith('_')}))


if __name__ == '__main__':
    main()
