"""The C-ABI library builds, loads without a GPU and exports every symbol include/dmh_hip.h declares."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, "include", "dmh_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from depthmodelhardening_amd import _native
    from depthmodelhardening_amd.build import build
    build(verbose=False)
    lib = _native.lib()
    names = _declared()
    assert len(names) >= 23
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_native.EXPORTS) == names, "ctypes signature table and header disagree"
    assert b"gfx950" in lib.dmh_version()


def test_struct_layouts_match_the_header(tmp_path):
    from depthmodelhardening_amd import _native
    src = tmp_path / "sz.c"
    src.write_text('#include "dmh_hip.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(void){printf("%zu %zu %zu %zu %zu\\n",'
                   'sizeof(dmh_photo_args),sizeof(dmh_smooth_args),sizeof(dmh_paste_args),'
                   'offsetof(dmh_photo_args,seed),offsetof(dmh_paste_args,mode));return 0;}\n')
    exe = tmp_path / "sz"
    assert os.system("gcc -I%s %s -o %s" % (os.path.join(REPO, "include"), src, exe)) == 0
    out = os.popen(str(exe)).read().split()
    assert [int(v) for v in out] == [ctypes.sizeof(_native.PhotoArgs), ctypes.sizeof(_native.SmoothArgs),
                                     ctypes.sizeof(_native.PasteArgs), _native.PhotoArgs.seed.offset,
                                     _native.PasteArgs.mode.offset]


def test_host_side_argument_checks_without_gpu():
    """Shape errors are rejected on the host before any launch (no GPU needed)."""
    from depthmodelhardening_amd import _native as N
    lib = N.lib()
    a = N.PasteArgs()
    assert lib.dmh_eot_paste_fwd(ctypes.byref(a), None, None, None) == 1
    assert b"null input" in lib.dmh_last_error() or b"requirement" in lib.dmh_last_error()
    assert lib.dmh_pgd_linf_step(None, None, None, 0.1, 0.1, None, 0, None) == 1
    p = N.PhotoArgs()
    assert lib.dmh_photo_loss_fwd(ctypes.byref(p), (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)(), None, None) == 1
