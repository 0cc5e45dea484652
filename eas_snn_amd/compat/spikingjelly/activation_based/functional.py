"""spikingjelly ``functional`` helpers used by EAS-SNN."""
import torch.nn as nn


def reset_net(net: nn.Module):
    """Restore every stateful module (anything with ``reset()``) to its registered initial state
    (call sites: yolox/core/trainer.py:115-117, yolox/evaluators/event_evaluator.py:196-198)."""
    for m in net.modules():
        if hasattr(m, 'reset'):
            m.reset()


def seq_to_ann_forward(x_seq, stateless_module):
    """[T, N, ...] -> fold T into the batch axis -> module(s) -> unfold."""
    y_shape = [x_seq.shape[0], x_seq.shape[1]]
    y = x_seq.flatten(0, 1)
    if isinstance(stateless_module, (list, tuple, nn.Sequential)):
        for m in stateless_module:
            y = m(y)
    else:
        y = stateless_module(y)
    y_shape.extend(y.shape[1:])
    return y.view(y_shape)


def set_step_mode(net: nn.Module, step_mode: str):
    for m in net.modules():
        if hasattr(m, 'step_mode'):
            m.step_mode = step_mode


def set_backend(net: nn.Module, backend: str, instance=nn.Module):
    for m in net.modules():
        if isinstance(m, instance) and hasattr(m, 'backend') and backend in getattr(m, 'supported_backends', ()):
            m.backend = backend


def detach_net(net: nn.Module):
    for m in net.modules():
        if hasattr(m, 'detach') and hasattr(m, '_memories'):
            m.detach()
