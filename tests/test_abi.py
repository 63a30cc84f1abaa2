"""The C-ABI library loads and exports every symbol include/mtseg.h declares (no compute, no GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'mtseg.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(mt_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_header_symbol():
    from multitalent_amd import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 28
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s
    assert set(_lib.SIGNATURES) == set(syms), (set(_lib.SIGNATURES) ^ set(syms))
    assert lib.mt_abi_version() == _lib.MT_ABI_VERSION == 4


def test_struct_sizes_match_c_layout():
    """ctypes mirrors of the ABI structs must have the C compiler's layout."""
    import ctypes as C
    import subprocess
    import tempfile
    from multitalent_amd._lib import mt_conv3d_t, mt_pointwise_t, mt_src_t
    src = '#include <stdio.h>\n#include "mtseg.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(mt_src_t), sizeof(mt_conv3d_t), sizeof(mt_pointwise_t));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 't.c'), 'w').write(src)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 't.c'), '-o', os.path.join(d, 't')])
        out = subprocess.check_output([os.path.join(d, 't')]).decode().split()
    assert [int(x) for x in out] == [C.sizeof(mt_src_t), C.sizeof(mt_conv3d_t), C.sizeof(mt_pointwise_t)]


def test_cpu_call_fails_loudly():
    """No silent fallback: the network refuses CPU tensors."""
    import pytest
    import torch
    from torch import nn
    from multitalent_amd.network_architecture.generic_UNet import Generic_UNet
    net = Generic_UNet(1, 4, 2, 1, 2, 2, nn.Conv3d, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                       {'p': 0, 'inplace': True}, nn.LeakyReLU, {'negative_slope': 1e-2, 'inplace': True}, True, False,
                       lambda x: x, None, [[2, 2, 2]], [[3, 3, 3]] * 2, False, True, True)
    with pytest.raises(RuntimeError, match="HIP device"):
        net(torch.zeros(1, 1, 4, 8, 8))


def test_selection_word_matches_the_header():
    """Host logic of ABI 4's kernel selection (no GPU): the field shifts of multitalent_amd.ops are include/mtseg.h's MT_SEL_* shifts, the
    legacy option values map onto MT_SEL_DEFAULT / OFF / FORCE, offsets of `select` / `max_workgroups` equal the C compiler's, MT_SELECT
    is parsed into the same word, and set-then-reset leaves the process default untouched."""
    import ctypes as C
    import subprocess
    import sys
    import tempfile
    from multitalent_amd import ops
    from multitalent_amd._lib import mt_conv3d_t
    txt = open(os.path.join(ROOT, 'include', 'mtseg.h')).read()
    shifts = {m.group(1): int(m.group(2)) for m in re.finditer(r'#define MT_SEL_([A-Z0-9_]+)\s+(\d+)u?\b', txt)}
    assert (shifts.pop('DEFAULT'), shifts.pop('OFF'), shifts.pop('FORCE')) == (0, 1, 2)
    names = {'WINO': 'conv_wino', 'M16': 'conv_bf16', 'X16': 'conv_x16', 'TAPSPLIT': 'conv_tapsplit', 'BWDW_WINO': 'bwdw_wino',
             'BWDW_TR16': 'bwdw_tr16', 'BWDW_CW': 'bwdw_cw'}
    assert {names[k]: v for k, v in shifts.items()} == ops._SEL_SHIFT
    assert len(set(shifts.values())) == len(shifts) and all(v % 2 == 0 and v + 2 <= 32 for v in shifts.values())
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "mtseg.h"\nint main(){printf("%zu %zu\\n", offsetof(mt_conv3d_t, select), '
           'offsetof(mt_conv3d_t, max_workgroups));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 't.c'), 'w').write(src)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 't.c'), '-o', os.path.join(d, 't')])
        offs = [int(x) for x in subprocess.check_output([os.path.join(d, 't')]).decode().split()]
    assert offs == [mt_conv3d_t.select.offset, mt_conv3d_t.max_workgroups.offset]
    assert ops.options_are_default()
    base = ops._select
    try:
        ops.set_option('conv_wino', 0)
        assert (ops._select >> shifts['WINO']) & 3 == 1                  # never -> MT_SEL_OFF
        ops.set_option('conv_wino', 2)
        assert (ops._select >> shifts['WINO']) & 3 == 2                  # wherever eligible -> MT_SEL_FORCE
        ops.set_option('conv_x16', 4096)
        assert (ops._select >> shifts['X16']) & 3 == 2 and not ops._caps
        ops.set_option('conv_x16', 7)
        p = ops.apply_selection(mt_conv3d_t())
        assert p.max_workgroups == 7 and p.select == ops._select
        ops.set_option('bwdw_cw', 2)
        assert (ops._select >> shifts['BWDW_CW']) & 3 == 2
        assert not ops.options_are_default()
    finally:
        ops.set_option('conv_wino', 1); ops.set_option('conv_x16', 1); ops.set_option('bwdw_cw', 4)
    assert ops._select == base and ops.options_are_default()
    code = ("import multitalent_amd.ops as o; print(o._select, o.options_are_default())")
    env = dict(os.environ, MT_SELECT='x16=off, wino=force,tapsplit=off')
    out = subprocess.check_output([sys.executable, '-c', code], env=env, cwd=ROOT).decode().split()
    want = (1 << shifts['X16']) | (2 << shifts['WINO']) | (1 << shifts['TAPSPLIT'])
    assert int(out[0]) == want and out[1] == 'True'                     # (the environment's word IS that process's default)
