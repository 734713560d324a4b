"""Several BUILDS of libhip_dsp in one process: a private copy of the audian_amd package per library (the package binds
its library at import).  Used by tools/entry_points_bench.py and tools/spec_sizes_bench.py (LIBS=a.so,b.so,tree)."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_count = [0]


def load_build(libpath):
    """(hipdsp module, design module) bound to `libpath` ("tree" or None: audian_amd/libhip_dsp.so)."""
    if libpath in (None, 'tree'):
        os.environ.pop('AUDIAN_AMD_LIB', None)
    else:
        os.environ['AUDIAN_AMD_LIB'] = libpath if os.path.isabs(libpath) else os.path.join(ROOT, libpath)
    os.environ['AUDIAN_AMD_NO_AUTOBUILD'] = '1'
    alias = f'audian_amd_build{_count[0]}'
    _count[0] += 1
    pkg_dir = os.path.join(ROOT, 'audian_amd')
    spec = importlib.util.spec_from_file_location(alias, os.path.join(pkg_dir, '__init__.py'), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    h = importlib.import_module(alias + '.hipdsp')
    d = importlib.import_module(alias + '.design')
    os.environ.pop('AUDIAN_AMD_LIB', None)
    return h, d


def build_name(libpath):
    return 'tree' if libpath in (None, 'tree') else os.path.splitext(os.path.basename(libpath))[0]
