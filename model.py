"""Import-root shim: `from model import CycleGAN` (reference train.py:8, convert.py:5)."""
import vcgan_pkg  # noqa: F401
from voice_converter_cyclegan_amd.model import CycleGAN  # noqa: F401

if __name__ == '__main__':
    model = CycleGAN(num_features=24)
    print('Graph Compile Successeded.')
