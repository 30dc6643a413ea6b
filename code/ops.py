"""Reference entry point name (code/ops.py): re-exports the MI355X implementation."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401  (the reference's `from ops import *` leaks these names; main.py relies on np)
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401
import torch.nn.functional as F  # noqa: F401
from pytorch_tecogan_amd.ops import *  # noqa: F401,F403
