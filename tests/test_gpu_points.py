"""SURVEY.md 8f rows N1-N4 and the wrapper mirrors through the HIP library on
the MI355X against the oracle (same bodies as the CPU tier runs on the emulator)."""
import pytest
import torch

from cameracalibration_amd import _lib
from tests import points_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    assert not _lib.load().is_emulator


@pytest.mark.parametrize("case", points_cases.ALL, ids=[c.__name__ for c in points_cases.ALL])
def test_points_on_gpu(case):
    case()
