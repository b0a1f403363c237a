"""Model registry with the reference's interface (mono/model/registry.py:8-41):
`MONO.module_dict[cfg.model['name']](cfg.model)` is how train.py:81 builds the model."""
import torch.nn as nn


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def register_module(self, cls):
        if not issubclass(cls, nn.Module):
            raise TypeError("module must be a child of nn.Module, but got {}".format(cls))
        if cls.__name__ in self._module_dict:
            raise KeyError("{} is already registered in {}".format(cls.__name__, self.name))
        self._module_dict[cls.__name__] = cls
        return cls


MONO = Registry("mono")
