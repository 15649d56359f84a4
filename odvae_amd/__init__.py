"""Importable alias of the package directory `generative-detection_amd/` (a hyphen is not a Python identifier).
`import odvae_amd.ops` loads generative-detection_amd/ops.py; nothing lives here."""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "generative-detection_amd")
__path__ = [_PKG_DIR]
with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
