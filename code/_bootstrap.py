"""Makes `pytorch_tecogan_amd` importable when the reference-style entry points are run with ./code on sys.path."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
import pytorch_tecogan_amd  # noqa: E402,F401
