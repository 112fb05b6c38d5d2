"""Import-root shim: the reference's drivers do `from module import ...` (model.py:3).
The implementation lives in voice-converter-cyclegan_amd/module.py (HIP through the C-ABI)."""
import vcgan_pkg  # noqa: F401
from voice_converter_cyclegan_amd.module import (  # noqa: F401
    generator_gatedcnn, discriminator, get_variable_store, reset_default_graph)
