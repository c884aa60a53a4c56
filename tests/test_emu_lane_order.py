"""Race-hunting mode of the lane-serial emulation build (VERDICT r04 #2a).

On the GPU the 64 lanes of a phase run in lockstep; the emulation runs a phase lane after lane.  A phase in which one lane reads
or overwrites a location that another lane writes IN THE SAME PHASE has a lane-order-dependent result in the emulation and an
instruction-order-dependent one on the GPU: an ordering assumption that nothing guarantees (and the class of bug the plain
emulation -- lanes in ascending order -- cannot see).  `ppn_emu_set_lane_order` (pypownet_amd/csrc/ppn_device.h, emulation build
only) runs every LANE_LOOP region in descending order or in a fresh pseudo-random permutation per region; the lock-step checks
against the oracles must come out the same in every order.  Floating-point atomic sums are accumulated in another order then (the
GPU does not fix their order either), which the checks' tolerances cover; integers, flags and iteration counts stay bit-exact.
"""
import ctypes as C
import os
import subprocess

import pytest

import engine_checks as ec
from helpers import ROOT
from test_emu_engine import emu_lib      # noqa: F401  (session fixture that compiles build/libppn_emu.so)

ORDERS = [1, 7, 1234567]      # descending, two random seeds (0 = ascending is what every other emulation test runs)


@pytest.fixture
def lane_order(emu_lib):
    lib = C.CDLL(emu_lib)

    def set_order(mode):
        lib.ppn_emu_set_lane_order(C.c_int(int(mode)))
    yield set_order
    set_order(0)


@pytest.mark.parametrize('order', ORDERS)
@pytest.mark.parametrize('env,solver,steps,batch,nb', [('default14_for_tests_alpha', 'newton', 30, 6, 0),
                                                        ('default14_for_tests_alpha', 'fdxb', 30, 6, 0),
                                                        ('default30', 'newton', 20, 6, 0),
                                                        ('default118', 'newton', 10, 3, 0),      # four-word kernels (every busbar may be active)
                                                        ('default118', 'fdxb', 8, 3, 0),
                                                        ('default118', 'newton', 8, 3, 150)])    # an intermediate busbar capacity
def test_emu_lane_order_random_actions_vs_c_oracle(emu_lib, lane_order, order, env, solver, steps, batch, nb):
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    lane_order(order)
    kw = {'max_active_buses': nb} if nb else {}
    st = ec.check_random_actions_vs_c_oracle(emu_lib, env, steps, batch, solver, seed=11 + order % 5, **kw)
    assert st['split_buses'] > 0


@pytest.mark.parametrize('order', [1])
@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_emu_lane_order_cascade_and_restarts_118(emu_lib, lane_order, order, solver):
    """Two-word kernels on the bench limits: cascades, game overs, fused restarts.  (Descending order only: this check compares two
    emulated engines BIT for bit, and under the random mode they draw different permutations -- their atomic sums round apart.)"""
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    lane_order(order)
    assert ec.check_auto_reset_and_cascade_118(emu_lib, steps=12, batch=4, solver=solver) > 0


@pytest.mark.parametrize('order', ORDERS[:2])
def test_emu_lane_order_dc_and_scenarios(emu_lib, lane_order, order):
    lane_order(order)
    ec.check_do_nothing(emu_lib, 'default14_for_tests_beta', 'fdxb', steps=8, batch=1)
    ec.check_hard_overflow_scenario(emu_lib, 'newton')
    ec.check_soft_overflow_scenario(emu_lib, 'fdxb')
    if order == 1:      # (compares two emulated engines bit for bit: deterministic orders only)
        assert ec.check_deferred_restart(emu_lib, 'default14_for_tests_alpha', steps=30, batch=6, bench_limits=False, max_active_buses=0,
                                         random_acts=True) > 0
