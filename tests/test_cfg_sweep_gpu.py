"""-m gpu: every explicit tile cfg on every conv flavour (3x3, 1x1, transposed conv forward and its 2x2/s2 dgrad) on
small shapes must either be rejected with an error or agree with the automatic choice (tools/cfg_sweep.py)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_every_explicit_cfg_is_rejected_or_correct():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import cfg_sweep
    cfg_sweep.main()
