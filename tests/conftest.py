"""pytest configuration: registers the ``gpu`` marker and puts the product
package dir (``sr-caco-2_amd/``) and ``oracle/`` on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sr-caco-2_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
