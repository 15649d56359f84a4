"""The C ABI library builds for gfx950, loads without a GPU and exports every symbol include/odvae_hip.h declares."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as g
    g.build()
    from odvae_amd import lib
    return lib


def test_library_exports_every_header_symbol(built_lib):
    handle = built_lib.load()
    names = built_lib.header_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(handle, name), name


def test_binding_table_matches_header(built_lib):
    assert sorted(built_lib.PROTOTYPES) == built_lib.header_symbols()


def test_identification_calls(built_lib):
    handle = built_lib.load()
    assert handle.odvae_abi_version() == built_lib.ABI_VERSION == 4
    assert handle.odvae_target_arch() == b"gfx950"
    # pure host queries (no kernel launch)
    assert handle.odvae_conv3x3_pack_reduce_pad(3) == 32 and handle.odvae_conv3x3_pack_out_pad(3) == 32
    assert handle.odvae_conv3x3_pack_out_pad(256) == 256
    assert handle.odvae_conv3x3_pack_floats(128, 128) == 9 * 128 * 128
    assert handle.odvae_gemm_f32_workspace_bytes(128, 128, 1 << 20, 1) > 0
    assert handle.odvae_gemm_f32_workspace_bytes(4096, 4096, 256, 32) == 0


def test_product_path_fails_loudly_without_gpu(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from odvae_amd import ops
    with pytest.raises(built_lib.HipLibraryError):
        ops.group_norm(torch.randn(1, 32, 4, 4), torch.ones(32), torch.zeros(32), 32, 1e-6, True)
    with pytest.raises(built_lib.HipLibraryError):
        ops.conv3x3(torch.randn(1, 4, 4, 4), torch.randn(4, 4, 3, 3))


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "generative-detection_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in text and "from oracle" not in text, fn
