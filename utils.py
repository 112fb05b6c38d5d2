"""Import-root shim: `from utils import l1_loss, l2_loss` (reference model.py:4)."""
import vcgan_pkg  # noqa: F401
from voice_converter_cyclegan_amd.utils import l1_loss, l2_loss  # noqa: F401
