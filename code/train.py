"""Reference entry point name (code/train.py): FRVSR_Train / TecoGAN on HIP kernels."""
import _bootstrap  # noqa: F401
from models import *  # noqa: F401,F403
from pytorch_tecogan_amd.train import FRVSR_Train, Network, TecoGAN  # noqa: F401
