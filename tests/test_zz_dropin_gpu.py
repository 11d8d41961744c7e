"""GPU tier of tests/test_dropin_cli.py: the reference's console front-end on libwelship.so (the product library) on the
MI355X, compared with the same front-end on the reference encoder.  Collected last on purpose."""
import os

import pytest

import test_dropin_cli as T


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["p_320x192_4slices_c1", "p_152x100_crop_raster20_idc2"])
def test_reference_cli_on_the_gpu(name, hip_lib, tmp_path):
    T.check_case(name, hip_lib, tmp_path)
