"""voice-converter-cyclegan_amd: MI355X-native CycleGAN-VC train-step hot path.

Importable as `voice_converter_cyclegan_amd` (the directory name carries a
hyphen; the repo-root shim `vcgan_pkg.py` registers it under that module name).
"""
__version__ = '0.1.0'
