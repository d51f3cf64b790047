"""ctypes binding of librgda_hip.so (the C ABI declared in include/rgda_hip.h).

The prototypes are read from the header itself so the binding cannot drift from
the ABI.  Fails loudly when the shared library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C regda_amd/csrc`).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'librgda_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'rgda_hip.h')

_CT = {
    'int': ctypes.c_int, 'float': ctypes.c_float, 'int64_t': ctypes.c_int64, 'size_t': ctypes.c_size_t,
    'rgda_stream_t': ctypes.c_void_p, 'rgda_comm_t': ctypes.c_void_p, 'double': ctypes.c_double, 'uint64_t': ctypes.c_uint64,
}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    protos = {}
    for m in re.finditer(r'\b(int|size_t|const char\s*\*)\s+(rgda_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = {'int': ctypes.c_int, 'size_t': ctypes.c_size_t}.get(ret.strip(), ctypes.c_char_p)
        argtypes = []
        args = args.strip()
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.replace('const ', '').split()[0]
                    argtypes.append(_CT[ty])
        protos[name] = (restype, argtypes)
    return protos


class RgdaError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} is missing: the HIP extension has not been built. regda_amd has no CPU '
                f'fallback; run `make -C regda_amd/csrc` (or __graft_entry__.build()).')
        # ONE HIP runtime per process: PyTorch (device memory, streams) ships its own libamdhip64 and this library is
        # linked against /opt/rocm's.  Loaded after torch, the dependency resolves to the copy torch already mapped and the
        # streams / pointers torch hands over belong to the runtime that launches the kernels; loaded BEFORE torch the two
        # copies coexist and the first launch on a torch stream fails (seen with `python __graft_entry__.py smoke`, where
        # build() loads the library for its export check before smoke() imports torch).
        import torch  # noqa: F401
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.missing = []
        for name, (restype, argtypes) in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:             # declared in the header but not exported: a build bug,
                self.missing.append(name)      # reported by tests/test_abi.py and on first use
                continue
            fn.restype = restype
            fn.argtypes = argtypes
        want = int(re.search(r'#define\s+RGDA_ABI_VERSION\s+(\d+)', open(HEADER_PATH).read()).group(1))
        if self._dll.rgda_abi_version() != want:
            raise ImportError('librgda_hip.so ABI version mismatch: rebuild with `make -C regda_amd/csrc`')

    def raw(self, name):
        return getattr(self._dll, name)

    def call(self, name, *args):
        """Call an int-returning entry point; raise on a negative status."""
        if name in self.missing:
            raise RgdaError(f'{name} is declared in rgda_hip.h but not exported by {LIB_PATH}')
        st = getattr(self._dll, name)(*args)
        if st != 0:
            msg = self._dll.rgda_strerror(st).decode()
            raise (ValueError if st in (-1, -4) else RgdaError)(f'{name}: {msg} (status {st})')
        if _plan is not None and _plan._ACTIVE is not None:      # a step is being recorded (regda_amd/plan.py)
            _plan._ACTIVE._call(name, args)

    def size(self, name, *args):
        return int(getattr(self._dll, name)(*args))


_lib = None
_plan = None        # set to the regda_amd.plan module when it is imported (the recorder lives there)


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
