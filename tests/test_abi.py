"""CPU checks of the drop-in boundary: libppn.so (built by __graft_entry__.build()) loads, exports every symbol
include/ppn.h declares, and fails loudly -- no silent CPU fallback -- when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import ROOT, load_env

HEADER = os.path.join(ROOT, 'include', 'ppn.h')


@pytest.fixture(scope='session')
def lib():
    import __graft_entry__ as g
    g.build_hip()
    from pypownet_amd import _lib
    return _lib.load_library()


def declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r'\b(ppn_[a-z_]+)\s*\(', src)))


def test_header_and_binding_agree(lib):
    from pypownet_amd import _lib
    decl = declared_symbols()
    assert set(decl) == set(_lib.EXPORTS), (set(decl) ^ set(_lib.EXPORTS))
    for s in decl:
        assert hasattr(lib, s), 'libppn.so does not export %s' % s


def test_field_enum_matches_header():
    from pypownet_amd import _lib
    src = open(HEADER).read()
    body = src[src.index('typedef enum ppn_field'):src.index('} ppn_field;')]
    names = re.findall(r'PPN_F_([A-Z_]+)', re.sub(r'/\*.*?\*/', '', body, flags=re.S))
    names = [n for n in names if n != 'COUNT']
    assert names == _lib.FIELDS


def test_struct_sizes_match_c_layout():
    """ctypes mirrors of ppn_rules / ppn_case / ppn_chronic must have the C layout (checked with gcc)."""
    import subprocess
    import tempfile
    from pypownet_amd import _lib
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "ppn.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(ppn_case), sizeof(ppn_rules), sizeof(ppn_chronic),
         offsetof(ppn_rules, n_timesteps_consecutive_soft_overflow_breaks), offsetof(ppn_rules, lu_capacity));
  printf("%zu %zu %zu\n", offsetof(ppn_rules, q_plane_auto), sizeof(ppn_mpc_batch), offsetof(ppn_mpc_batch, success));
  printf("%zu %zu %zu\n", sizeof(ppn_async_config), offsetof(ppn_async_config, obs_device), offsetof(ppn_async_config, report_device));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 't.c'), 'w').write(prog)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 't.c'), '-o', os.path.join(d, 't')])
        out = subprocess.check_output([os.path.join(d, 't')]).decode().split()
    got = [C.sizeof(_lib.PpnCase), C.sizeof(_lib.PpnRules), C.sizeof(_lib.PpnChronic),
           _lib.PpnRules.n_timesteps_consecutive_soft_overflow_breaks.offset, _lib.PpnRules.lu_capacity.offset,
           _lib.PpnRules.q_plane_auto.offset, C.sizeof(_lib.PpnMpcBatch), _lib.PpnMpcBatch.success.offset,
           C.sizeof(_lib.PpnAsyncConfig), _lib.PpnAsyncConfig.obs_device.offset, _lib.PpnAsyncConfig.report_device.offset]
    assert [int(v) for v in out] == got


def test_no_gpu_is_a_loud_error(lib):
    """In the build container there is no GPU: ppn_create must refuse (PPN_E_NODEVICE), never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible here')
    from pypownet_amd.engine import Engine, EngineError
    case, cfg, chronics = load_env('default14_for_tests')
    with pytest.raises(EngineError) as ei:
        Engine(case, cfg, 2, chronics=chronics)
    assert 'no HIP device' in str(ei.value) or 'failed' in str(ei.value)


def test_missing_extension_raises(tmp_path, monkeypatch):
    """The product loads ONE fixed library path and raises when it is absent (no fallback of any kind)."""
    from pypownet_amd import _lib
    import inspect
    assert not inspect.signature(_lib.load_library).parameters      # no way to point the product at another library
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'libppn_missing.so'))
    with pytest.raises(ImportError):
        _lib.load_library()
    from pypownet_amd.engine import Engine
    case, cfg, chronics = load_env('default14_for_tests')
    with pytest.raises(ImportError):
        Engine(case, cfg, 1, chronics=chronics)


def test_wave_full_statement_keeps_the_work_loop_in_one_piece():
    """DESIGN 12.9, at compile time (no GPU): the optimised IR of tools/ubench/convergent_threading_repro.hip.  With the statement of
    PPN_WAVE_FULL at the loop head (FIX=3) the work loop keeps ONE `atomicrmw` for the item counter and one for the work; the
    miscompiled form (what this compiler makes of FIX=0) carries a third -- the private loop in which lanes 1..63 replay item 0."""
    import shutil
    import subprocess
    import tempfile
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip('no hipcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, 'tools', 'ubench', 'convergent_threading_repro.hip')

    def atomics(fix):
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, 'k.ll')
            subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-DFIX=%d' % fix, '--cuda-device-only', '-emit-llvm', '-S', src, '-o', out],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            body, on = [], False
            for line in open(out):
                if line.startswith('define') and 'work_loop' in line:
                    on = True
                if on:
                    body.append(line)
                    if line.startswith('}'):
                        break
            return sum('atomicrmw' in ln for ln in body), sum('readfirstlane' in ln for ln in body)
    assert atomics(3) == (2, 1)
    assert atomics(1) == (2, 1)      # (__builtin_amdgcn_wave_barrier(): the same protection)


def test_kernels_ir_has_no_lane_dependent_cycle_around_a_convergent_operation():
    """tools/dev/ir_lint_convergent.py over the optimised IR of all six kernel translation units (hipcc -emit-llvm, ~30 s in
    parallel): no cycle that holds a convergent operation (readfirstlane, readlane, barriers, DPP ...) may be left or re-entered on a
    lane-dependent condition -- the shape simplifycfg gave round 5's rollout kernel.  (The lint reports that kernel in every width
    and solver when the library is built with -DPPN_WAVE_FULL_OFF: profiles/r06_ir_lint_convergent.txt.)"""
    import shutil
    import subprocess
    import sys
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip('no hipcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'dev', 'ir_lint_convergent.py'), '--build'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(' 0 reported') == 6, r.stdout


MIR_JOIN_BLOCK = """--- |
  ; module
...
---
name:            kernel_under_test
body:             |
  bb.0:
    successors: %bb.1, %bb.2
    renamable $sgpr0_sgpr1 = COPY $exec, implicit-def $exec
    $exec = S_MOV_B64_term killed renamable $sgpr6_sgpr7
    S_CBRANCH_EXECZ %bb.2, implicit $exec

  bb.1:
    successors: %bb.2
    renamable $vgpr26 = V_ACCVGPR_READ_B32_e64 $agpr111, implicit $exec

  bb.2:
  HEAD
    $exec = S_OR_B64 $exec, killed renamable $sgpr0_sgpr1, implicit-def $scc
    renamable $vgpr30_vgpr31 = V_MOV_B64_e32 0, implicit $exec
    S_ENDPGM 0
...
"""


def test_mir_lint_reports_a_vector_copy_in_front_of_the_exec_restore(tmp_path):
    """The shape of GPU-only incident (i) in a dozen lines of machine IR: an SGPR copy in front of the join block's S_OR_B64 is
    scalar code and fine, an SGPR <-> VGPR-lane spill ignores EXEC and is fine, a vector copy / spill there runs under the branch's mask."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('mir_lint_exec_restore', os.path.join(root, 'tools', 'dev', 'mir_lint_exec_restore.py'))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    heads = {
        'clean': '  renamable $sgpr78_sgpr79 = COPY killed renamable $sgpr60_sgpr61\n    $sgpr3 = SI_RESTORE_S32_FROM_VGPR $vgpr247, 49',
        'copy': '  renamable $agpr20 = COPY killed renamable $vgpr235\n    renamable $sgpr78_sgpr79 = COPY killed renamable $sgpr60_sgpr61',
        'spill': '  SI_SPILL_V32_SAVE killed $vgpr235, %stack.5, $sgpr32, 0, implicit $exec :: (store (s32) into %stack.5, addrspace 5)',
        'remat': '  renamable $vgpr109 = V_MOV_B32_e32 255, implicit $exec',
    }
    found = {}
    for tag, head in heads.items():
        p = tmp_path / (tag + '.mir')
        p.write_text(MIR_JOIN_BLOCK.replace('HEAD', head))
        reps, n = lint.lint(str(p))
        assert n == 1
        found[tag] = [(r[0], r[1]) for r in reps]
    assert found['clean'] == []
    assert found['copy'] == [('kernel_under_test', 'bb.2')]
    assert found['spill'] == [('kernel_under_test', 'bb.2')]
    assert found['remat'] == [('kernel_under_test', 'bb.2')]


def test_kernels_mir_has_no_vector_code_in_front_of_an_exec_restore():
    """tools/dev/mir_lint_exec_restore.py over the machine IR of all seven translation units, stopped behind the last register
    allocation (hipcc -mllvm -stop-after=amdgpu-mark-last-scratch-load, ~80 s in parallel): no vector copy / spill / rematerialisation may stand
    in front of a join block's `$exec = S_OR_B64 $exec, ...` -- the placement that made round 2's four-word Newton reset kernel save a
    uniform LDS base for the lanes of one branch only (incident (i)).  On the round-2 tree the lint separates the seven known builds without a
    GPU: the three that fail on the MI355X are reported, the four that pass are clean (profiles/r06_incident_i_mir_lint_variants.txt); so are round 3's
    incident (ii) (20 reports) and its fix (none): profiles/r06_incident_ii_mir_lint.txt."""
    import shutil
    import subprocess
    import sys
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip('no hipcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'dev', 'mir_lint_exec_restore.py'), '--build'], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(' 0 vector instruction(s) in front of one') == 7, r.stdout
