"""CPU: the C-ABI library loads and exports every symbol include/xmaps.h declares (no compute calls)."""
import ctypes
import os
import re

from x_maps_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "xmaps.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(xm_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert len(names) >= 25
    lib = N.load_library()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in xmaps.h but not exported by libxmaps_hip.so"
    assert set(names) == set(N.SYMBOLS), set(names) ^ set(N.SYMBOLS)
    assert lib.xm_api_version() == 5


def test_struct_layouts_match_the_header():
    # xm_config: 14 int32 + double + 2 float + 4 pointers; xm_frame_stats: 4 u64 + 2 double + 4 float
    assert ctypes.sizeof(N.xm_config) == 14 * 4 + 8 + 2 * 4 + 4 * 8
    assert ctypes.sizeof(N.xm_frame_stats) == 4 * 8 + 2 * 8 + 4 * 4 + 8
    assert N.xm_config.p03.offset == 56 and N.xm_config.cam_mapx_i16.offset == 72


def test_no_cpu_fallback_in_product_package():
    """The product package must never import the oracle."""
    pkg = os.path.join(ROOT, "x_maps_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert "xmaps_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, fn


def test_header_is_plain_c_and_layouts_agree(tmp_path):
    """include/xmaps.h must compile as C99 (the drop-in boundary is a C ABI, not C++) and the compiler's struct layout
    must be the one the ctypes binding assumes."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "xmaps.h"\n'
                   'int main(void) {\n'
                   '  printf("%zu %zu %zu %zu %d %d %zu %zu %zu %zu\\n", sizeof(xm_config), sizeof(xm_frame_stats), offsetof(xm_config, p03),\n'
                   '         offsetof(xm_config, cam_mapx_i16), XM_ERR_UNSORTED, XM_MEM_HOST_PINNED, sizeof(xm_ingest_config),\n'
                   '         sizeof(xm_ingest_frame), offsetof(xm_ingest_config, capacity_events), offsetof(xm_ingest_frame, depth));\n'
                   '  return 0;\n}\n')
    exe = tmp_path / "t"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(v) for v in out] == [ctypes.sizeof(N.xm_config), ctypes.sizeof(N.xm_frame_stats), N.xm_config.p03.offset,
                                     N.xm_config.cam_mapx_i16.offset, N.XM_ERR_UNSORTED, N.XM_MEM_HOST_PINNED,
                                     ctypes.sizeof(N.xm_ingest_config), ctypes.sizeof(N.xm_ingest_frame),
                                     N.xm_ingest_config.capacity_events.offset, N.xm_ingest_frame.depth.offset]


def test_graft_entry_build_checks_the_same_api_version():
    """__graft_entry__.build() asserts the library's API version: it must be the header's (the driver's build check runs it)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = int(re.search(r"#define\s+XM_API_VERSION\s+(\d+)", open(os.path.join(root, "include", "xmaps.h")).read()).group(1))
    src = open(os.path.join(root, "__graft_entry__.py")).read()
    assert int(re.search(r"xm_api_version\(\)\s*==\s*(\d+)", src).group(1)) == hdr
