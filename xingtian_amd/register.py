"""Name-keyed plugin registries and YAML-over-module-globals configuration.

The host framework selects plugins by class ``__name__`` (``alg_para.alg_name``, ``model_para.actor.model_name``)
through decorator registries and lets YAML sections override UPPERCASE module constants; reference behaviour:
zeus/common/util/register.py:39-82 and zeus/common/util/common.py:32-44.
"""
import logging

_log = logging.getLogger(__name__)


class Registry(object):
    """``@registry`` registers a class under its name; ``registry[name]`` returns it (KeyError if unknown)."""

    def __init__(self, kind):
        self.kind = kind
        self._by_name = {}

    def __call__(self, plugin):
        if not callable(plugin):
            raise Exception("To Registry must be callable, Got: {}.".format(plugin))
        if plugin.__name__ in self._by_name:
            _log.warning("%s plugin %s registered twice: the later one wins", self.kind, plugin.__name__)
        self._by_name[plugin.__name__] = plugin
        return plugin

    def __getitem__(self, name):
        if name not in self._by_name:
            _log.error("no %s plugin named %s (known: %s)", self.kind, name, sorted(self._by_name))
        return self._by_name[name]

    def __contains__(self, name):
        return name in self._by_name

    def keys(self):
        return self._by_name.keys()

    def build(self, name, *args, **kwargs):
        """Instantiate the plugin called ``name``."""
        return self[name](*args, **kwargs)


RegisterStub = Registry      # the reference's class name


class Registers(object):
    """Namespace of the registries (never instantiated)."""

    agent = Registry("agent")
    model = Registry("model")
    algorithm = Registry("algorithm")
    env = Registry("env")
    comm = Registry("comm")

    def __init__(self):
        raise RuntimeError("Registries prohibit instancing !")


def import_config(global_para, config):
    """Override the module-level constants named in ``config`` (keys the module does not define are ignored)."""
    for key in (config or {}):
        if key in global_para:
            global_para[key] = config[key]
