"""Stateful-module base (spikingjelly ``base.MemoryModule`` semantics): memories are plain attributes kept
outside ``state_dict``; ``reset()`` restores the value each memory was registered with."""
import copy

import torch.nn as nn


class StepModule:
    def supported_step_mode(self):
        return ('s', 'm')

    @property
    def step_mode(self):
        return self._step_mode

    @step_mode.setter
    def step_mode(self, value):
        if value not in self.supported_step_mode():
            raise ValueError(f'step_mode can only be {self.supported_step_mode()}, but got "{value}"')
        self._step_mode = value


class MemoryModule(nn.Module, StepModule):
    def __init__(self):
        super().__init__()
        self._memories = {}
        self._memories_rv = {}
        self._backend = 'hip'
        self._step_mode = 's'

    @property
    def supported_backends(self):
        return ('torch', 'hip')

    @property
    def backend(self):
        return self._backend

    @backend.setter
    def backend(self, value):
        if value not in self.supported_backends:
            raise NotImplementedError(f'{value} is not a supported backend of {self._get_name()}')
        self._backend = value

    def register_memory(self, name, value):
        assert not hasattr(self, name), f'{name} has been set as a member variable'
        self._memories[name] = value
        self._memories_rv[name] = copy.deepcopy(value)

    def reset(self):
        for key in self._memories:
            self._memories[key] = copy.deepcopy(self._memories_rv[key])

    def set_reset_value(self, name, value):
        self._memories_rv[name] = copy.deepcopy(value)

    def __getattr__(self, name):
        if '_memories' in self.__dict__ and name in self.__dict__['_memories']:
            return self.__dict__['_memories'][name]
        return super().__getattr__(name)

    def __setattr__(self, name, value):
        mem = self.__dict__.get('_memories')
        if mem is not None and name in mem:
            mem[name] = value
        else:
            super().__setattr__(name, value)

    def __delattr__(self, name):
        if name in self._memories:
            del self._memories[name]
            del self._memories_rv[name]
        else:
            super().__delattr__(name)

    def memories(self):
        return self._memories.values()

    def named_memories(self):
        return self._memories.items()

    def detach(self):
        import torch
        for key, value in self._memories.items():
            if isinstance(value, torch.Tensor):
                value.detach_()

    def _apply(self, fn, *args, **kwargs):
        import torch
        for key, value in self._memories.items():
            if isinstance(value, torch.Tensor):
                self._memories[key] = fn(value)
        return super()._apply(fn, *args, **kwargs)

    def single_step_forward(self, x, *args, **kwargs):
        raise NotImplementedError

    def multi_step_forward(self, x_seq, *args, **kwargs):
        raise NotImplementedError

    def forward(self, *args, **kwargs):
        if self.step_mode == 's':
            return self.single_step_forward(*args, **kwargs)
        return self.multi_step_forward(*args, **kwargs)

    def extra_repr(self):
        return f'step_mode={self.step_mode}, backend={self.backend}'
