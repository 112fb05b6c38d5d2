"""Generates tests/golden/ref_call_trace.json: every call the reference's UNMODIFIED train.py (one epoch + validation)
and convert.py make on model.CycleGAN, recorded by running those files (imported from /root/reference) against a spy
model with synthetic librosa / pyworld stand-ins (tests/refdrive.py).  Needs /root/reference; no GPU.

    python tests/golden/make_ref_trace.py
"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import refdrive  # noqa: E402

if __name__ == '__main__':
    with tempfile.TemporaryDirectory() as d:
        trace = refdrive.record_trace(d)
    out = os.path.join(ROOT, 'tests', 'golden', 'ref_call_trace.json')
    with open(out, 'w') as f:
        json.dump(trace, f, indent=1, sort_keys=True)
    print('%d calls -> %s' % (len(trace), out))
