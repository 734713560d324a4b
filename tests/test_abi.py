"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/hip_dsp.h declares (no compute calls here)."""

import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'hip_dsp.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(hipdsp_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from audian_amd import _lib
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(_lib.lib, name), f'{name} declared in hip_dsp.h but not exported'
    # and the Python binding covers the whole header
    assert sorted(_lib._SIGNATURES) == names


def test_no_gpu_calls_fail_loudly():
    """Without a device the library reports an error instead of computing on the CPU."""
    import ctypes
    import pytest
    from audian_amd import _lib
    n = ctypes.c_int(0)
    rc = _lib.lib.hipdsp_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip('a GPU is present')
    h = ctypes.c_void_p()
    rc = _lib.lib.hipdsp_ctx_create(0, None, ctypes.byref(h))
    assert rc != 0 and _lib.last_error()
    with pytest.raises(Exception):
        from audian_amd import hipdsp
        hipdsp.Context(0)


def test_product_does_not_import_oracle():
    """Nothing under audian_amd/ may reference the oracle (it is test infrastructure)."""
    pkg = os.path.join(ROOT, 'audian_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, fn)).read()
                assert 'oracle' not in text.lower(), f'{fn} mentions the oracle'


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 (no C++-isms, no torch or HIP types),
    and a C translation unit can name every entry point."""
    import shutil
    import subprocess
    import pytest
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    src = tmp_path/'use_header.c'
    calls = '\n'.join(f'    (void)&{name};' for name in declared_symbols())
    src.write_text('#include "hip_dsp.h"\nint main(void)\n{\n' + calls + '\n    return HIPDSP_OK;\n}\n')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', '-I',
                           os.path.join(ROOT, 'include'), '-c', str(src), '-o', str(tmp_path/'use_header.o')])
