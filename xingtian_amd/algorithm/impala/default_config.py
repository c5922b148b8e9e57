"""Module-level defaults (see xingtian_amd/defaults.py); YAML keys override them via import_config."""
from xingtian_amd.defaults import publish

publish(globals(), "algorithm/impala")
