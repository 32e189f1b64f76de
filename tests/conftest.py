import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
  # A fresh checkout has no built artefacts (they are git-ignored): build the HIP library (hipcc cross-compiles
  # without a GPU) and the C oracle once, exactly as __graft_entry__.build() does.
  lib = os.path.join(ROOT, 'nerf-hugs_amd', 'csrc', 'libhugs_hip.so')
  orc = os.path.join(ROOT, 'oracle', 'liborc_stepfun.so')
  if not os.path.exists(lib) or not os.path.exists(orc):
    import subprocess
    if not os.path.exists(lib) and os.path.exists('/opt/rocm/bin/hipcc'):
      subprocess.run([os.path.join(ROOT, 'nerf-hugs_amd', 'csrc', 'build.sh')], check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(orc):
      subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope='session')
def golden():
  import numpy as np
  return np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_leaves.npz'))
