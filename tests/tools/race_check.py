"""Shows that tests/test_network_gpu.py::test_free_running_training_equals_step_synchronised_training detects the hazard it guards:
with the pinned staging ring reduced to ONE unfenced slot (what the optimizer once did) the test must fail."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import rumpy_amd.staging as S

_init = S.PinnedRing.__init__


def one_slot(self, shape, dtype, slots=4):
    _init(self, shape, dtype, slots=1)


def no_fence(self, i, stream=None):
    pass


S.PinnedRing.__init__ = one_slot
S.PinnedRing.sent = no_fence
rc = pytest.main(['-q', '-x', 'tests/test_network_gpu.py', '-m', 'gpu', '-k', 'free_running and edsr'])
print('unfenced single slot: pytest exit code %d (expected: 1 = the test catches the race)' % rc)
sys.exit(0 if rc == 1 else 1)
