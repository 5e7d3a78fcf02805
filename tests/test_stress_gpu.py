"""The randomized generators that found real bugs (round 3: a linear download that overwrote caller bytes between rows when
ostride > w) as bounded, fixed-seed GPU tests: the same code as tools/stress_*.py (imported from there), a few hundred
cases and a few seconds each.  Longer soaks stay in tools/ (`python tools/stress_vs_generic.py 2500 7`)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stress_column_wave_vs_generic_kernel_random_shapes():
    """Random shapes 1x1 .. 4000x260, BlockSize 1..33 (a third of the draws on the special forms: 9, 14-17), MaxDisparity
    1..400, six image statistics, raw / scaled, regions: the column-wave kernel against the generic kernel."""
    assert _tool("stress_vs_generic").run(320, 404) == 0


def test_stress_device_resident_batches_strided_views():
    """Random batch lengths 1..77, row / frame strides (views into larger tensors), parameters: both kernels through
    sadhip_disparity_dev_u8."""
    assert _tool("stress_dev_batch").run(150, 405) == 0


def test_stress_boundary_left_right_sizes_and_page_locked_layouts():
    """Random left / right picture sizes through sadhip_disparity_lr_u8 against the literal restatement; random
    page-locked batch layouts (column windows of wider images, scattered order, tight / windowed outputs with sentinel
    bytes) against the pageable path."""
    assert _tool("stress_boundary").run(240, 406) == (0, 0)


def test_stress_staged_host_copies_random_sizes_strides_regions():
    """Random frame sizes around the thresholds of the staged host path (copy threads from 512 KB, bands above 8 MB), row
    strides, base alignments, guard pages, regions, batches, decoded pictures: against the device-resident entry point."""
    assert _tool("stress_hostcopy").run(36, 407) == 0
