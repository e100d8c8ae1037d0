"""Import alias: the package directory is `radio-sim_amd/` (a hyphen is not importable), so
`import radio_sim_amd` loads that directory as a regular package under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "radio-sim_amd")
_spec = importlib.util.spec_from_file_location(
    "radio_sim_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["radio_sim_amd"] = _mod
_spec.loader.exec_module(_mod)
