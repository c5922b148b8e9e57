"""Plugin registry + config import with the reference's semantics.

Mirrors ``zeus/common/util/register.py:39-82`` (``RegisterStub`` / ``Registers``: classes
are looked up by ``__name__``) and ``zeus/common/util/common.py:32-44``
(``import_config``: YAML keys override same-named module-level UPPERCASE globals).
"""
import logging


class RegisterStub(object):
    def __init__(self, name):
        self._dict = dict()
        self._name = name

    def __getitem__(self, key):
        try:
            return self._dict[key]
        except KeyError as exc:
            logging.error("module %s not found in registry '%s'", key, self._name)
            raise exc

    def __contains__(self, key):
        return key in self._dict

    def __call__(self, param):
        if not callable(param):
            raise Exception("To Registry must be callable, Got: {}.".format(param))
        register_name = param.__name__
        if register_name in self._dict:
            logging.warning("Key:%s is registered, will replace with %s.", register_name, self._name)
        self._dict[register_name] = param
        return param

    def keys(self):
        return self._dict.keys()


class Registers(object):
    """All module registers (zeus/common/util/register.py:72-82)."""

    def __init__(self):
        raise RuntimeError("Registries prohibit instancing !")

    agent = RegisterStub("agent")
    model = RegisterStub("model")
    algorithm = RegisterStub("algorithm")
    env = RegisterStub("env")
    comm = RegisterStub("comm")


def import_config(global_para, config):
    """zeus/common/util/common.py:32-44."""
    if not config:
        return
    for key in config.keys():
        if key in global_para:
            global_para[key] = config[key]
