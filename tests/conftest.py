import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run via gpurun)')
  config.addinivalue_line('markers', 'slow: tens of seconds on the GPU (long convergence / full-shape oracle runs); deselect with -m "gpu and not slow"')


def pytest_collection_modifyitems(config, items):
  import torch
  if torch.cuda.is_available():
    return
  skip = pytest.mark.skip(reason='no GPU visible')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)
