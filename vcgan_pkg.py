"""Registers the in-tree package directory `voice-converter-cyclegan_amd/`
(hyphenated, so not a legal identifier) as module `voice_converter_cyclegan_amd`."""
import importlib.util
import os
import sys

NAME = 'voice_converter_cyclegan_amd'


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'voice-converter-cyclegan_amd')
    spec = importlib.util.spec_from_file_location(NAME, os.path.join(d, '__init__.py'),
                                                  submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod


load()
